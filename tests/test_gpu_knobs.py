"""mv_options (include/melvoc.h) select other kernels for the same arithmetic: layer-by-layer instead of fused,
float32 instead of float16 / bfloat16 storage between kernels, the reference's WaveGlow skip path instead of the
folded one.  Options are per handle (the library keeps no process-global state and reads no environment), so every
setting runs in this one process; the bar is the same as for the defaults (bf16: SNR >= 40 dB against the golden
vectors captured from the reference, BASELINE.json)."""
import copy
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tacotron2_amd import _native, synth  # noqa: E402
from tacotron2_amd.hifigan import AttrDict, Generator  # noqa: E402
from tacotron2_amd.waveglow import WaveGlow  # noqa: E402


def _snr(ref, x):
    ref = np.asarray(ref, np.float64)
    return float(10 * np.log10((ref ** 2).sum() / max(((ref - np.asarray(x, np.float64)) ** 2).sum(), 1e-300)))


HIFIGAN_OPTIONS = [
    None,                       # the defaults
    {"fuse": 0},                # one convolution per launch
    {"fuse": 1},                # fused ResBlock kernels, no whole-stage launches
    {"io16": 0},                # float32 activations between kernels
    {"fuse": 1, "io16": 0},
    {"fuse": 0, "io16": 0},
]


@pytest.mark.parametrize("options", HIFIGAN_OPTIONS, ids=lambda o: str(o))
def test_hifigan_bf16_parity_under_options(options, golden_dir):
    d = np.load(os.path.join(golden_dir, "hifigan_v1_T64.npz"))
    g = Generator(AttrDict(synth.HIFIGAN_V1), precision="bf16", options=options)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
    g.eval().cuda()
    mel = torch.from_numpy(synth.synthetic_mel(64, int(d["mel_seed"]))).cuda()
    assert _snr(d["wav"], g(mel[None])[0, 0].cpu().numpy()) >= 40.0
    # ragged batch: every row equals its own batch-1 result to well below the bf16 error
    lens = [64, 37, 5]
    x = torch.full((3, 80, 64), synth.MEL_PAD_VALUE)
    for b, n in enumerate(lens):
        x[b, :, :n] = torch.from_numpy(synth.synthetic_mel(n, 40 + b))
    y = g(x.cuda(), lengths=lens)[:, 0].cpu().numpy()
    solo = g(x[1:2, :, :37].contiguous().cuda())[0, 0].cpu().numpy()
    assert float(np.abs(solo - y[1, :37 * 256]).max()) <= 5e-3
    assert not y[1, 37 * 256:].any() and not y[2, 5 * 256:].any()


@pytest.mark.parametrize("options", [None, {"wg_fold": 0}, {"io16": 0}, {"wg_fold": 0, "io16": 0}, {"wg_gemm": 2},
                                     {"wg_gemm": 3}, {"wg_gemm": 0}, {"wg_gemm": 2, "wg_fold": 0}], ids=lambda o: str(o))
@pytest.mark.parametrize("name,c", [("waveglow_c512_T16", 512), ("waveglow_c256_T64", 256), ("waveglow_c64_T16", 64)])
def test_waveglow_bf16_parity_under_options(options, name, c, golden_dir):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = c
    B, T = int(d["B"]), int(d["T"])
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(cfg, int(d["weight_seed"])).items()})
    m.cuda().eval()
    m.precision = "bf16"
    m.options = options
    mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, int(d["mel_seed"]) + b) for b in range(B)])).cuda()
    noise = [torch.from_numpy(n) for n in synth.waveglow_noise(cfg, B, T, int(d["noise_seed"]))]
    assert _snr(d["wav"], m.infer(mel, sigma=float(d["sigma"]), noise=noise).cpu().numpy()) >= 40.0


def test_waveglow_gate_gemm_agrees_with_the_conv_kernel():
    """wg_gemm selects gate_gemm_kernel (256 x 256 LDS-DMA GEMM) or conv_cl_kernel for the gated in-layer conv: the same
    bf16 operands, fp32 accumulation in another order (which later bf16 roundings amplify, so the two are not bit-equal).
    Both are held against the library's fp32 mode on a ragged batch whose sequences end inside a 256-row tile, within a
    dilation's reach of a tile edge, and before the first edge; a wrong row at a sequence or tile edge would cost tens
    of dB."""
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    B, T, lens = 3, 41, [41, 23, 7]
    mel = torch.full((B, 80, T), synth.MEL_PAD_VALUE)
    for b, n in enumerate(lens):
        mel[b, :, :n] = torch.from_numpy(synth.synthetic_mel(n, 70 + b))
    noise = [torch.from_numpy(n) for n in synth.waveglow_noise(cfg, B, T, 5)]
    out = {}
    for key, prec, opts in (("fp32", "fp32", None), ("conv", "bf16", {"wg_gemm": 0}), ("gemm", "bf16", {"wg_gemm": 2}),
                            ("gemm256", "bf16", {"wg_gemm": 3})):
        m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(cfg, 3).items()})
        m.cuda().eval()
        m.precision = prec
        m.options = opts
        out[key] = m.infer(mel.cuda(), sigma=0.666, noise=noise, lengths=lens).cpu().numpy()
    for b, n in enumerate(lens):
        ref = out["fp32"][b, :n * 256]
        s_conv, s_gemm = _snr(ref, out["conv"][b, :n * 256]), _snr(ref, out["gemm"][b, :n * 256])
        assert s_gemm >= 40.0 and abs(s_gemm - s_conv) <= 3.0, (b, s_conv, s_gemm)
        assert _snr(out["conv"][b, :n * 256], out["gemm"][b, :n * 256]) >= 45.0, b
        assert not out["gemm"][b, n * 256:].any()
        # the residual conv on 256-row blocks: other tiles, the same contraction order per element
        assert _snr(ref, out["gemm256"][b, :n * 256]) >= 40.0 and not out["gemm256"][b, n * 256:].any()
        assert _snr(out["gemm"][b, :n * 256], out["gemm256"][b, :n * 256]) >= 45.0, b


def test_waveglow_gate_gemm_is_deterministic_at_bench_size():
    """gate_gemm_kernel hands LDS buffers from DMA requests to fragment reads by counted waits and barriers only; a
    read that ran ahead of its data would show as run-to-run differences.  25 inferences of the bench utterance (250
    blocks per launch, one per CU, 96 launches each) must be bit-identical, and so must the ragged batch of 8."""
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(cfg, 0).items()})
    m.cuda().eval()
    m.precision = "bf16"
    m.options = {"wg_gemm": 2}
    for B, T, lens, reps in ((1, 500, None, 25), (8, 300, [300, 287, 255, 200, 129, 64, 33, 1], 5)):
        mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, 1234 + b) for b in range(B)])).cuda()
        noise = [torch.from_numpy(n).cuda() for n in synth.waveglow_noise(cfg, B, T, 9)]
        first = m.infer(mel, sigma=0.666, noise=noise, lengths=lens).clone()
        assert torch.isfinite(first).all()
        for _ in range(reps - 1):
            assert torch.equal(m.infer(mel, sigma=0.666, noise=noise, lengths=lens), first)


def test_bad_options_are_refused():
    with pytest.raises(_native.MelvocError):
        _native.make_options({"no_such_knob": 1})
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    with pytest.raises(_native.MelvocError, match="mv_options"):
        _native.create_hifigan(synth.HIFIGAN_V1, st, 0, _native.PREC_BF16, {"fuse": 7})
