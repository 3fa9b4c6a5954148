"""MV_PREC_BF16X3 (include/melvoc.h): float32-grade results from split bf16 operands (w_hi*x_hi + w_hi*x_lo + w_lo*x_hi
on the bf16 MFMA, fp32 accumulation, float32 activations between kernels).  It is held to the FP32 bars of
BASELINE.json -- max-abs <= 1e-3 against the golden vectors captured from the reference -- not to the bf16 ones."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from tacotron2_amd import _native, synth  # noqa: E402
from tacotron2_amd.hifigan import AttrDict, Generator  # noqa: E402
from tacotron2_amd.waveglow import WaveGlow  # noqa: E402


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.fixture(scope="module")
def gen_x3():
    g = Generator(AttrDict(synth.HIFIGAN_V1), precision="bf16x3")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
    return g.eval().to("cuda")


@pytest.mark.parametrize("case", [(32, 32, 3, 1, 300, None), (32, 32, 11, 5, 700, [700, 333]), (64, 64, 7, 3, 515, [515, 100, 1]),
                                  (128, 128, 11, 5, 260, None), (256, 256, 7, 1, 130, [130, 77]), (96, 512, 7, 1, 70, [70, 64])])
def test_conv1d_vs_oracle(case):
    cin, cout, k, dil, L, lengths = case
    rng = np.random.default_rng(hash((cin, cout, k, dil)) & 0xFFFF)
    B = len(lengths) if lengths else 2
    x = rng.standard_normal((B, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    yd = torch.full((B, L, cout), float("nan"), device="cuda")
    ld = torch.tensor(lengths, dtype=torch.int32, device="cuda") if lengths else None
    rc = _native.lib().mv_test_conv1d(xd.data_ptr(), ld.data_ptr() if ld is not None else None, B, L, cin, _fp(w),
                                      _fp(b), cout, k, dil, 0.1, _native.PREC_BF16X3, yd.data_ptr(), None)
    _native.check(rc, "mv_test_conv1d")
    y = yd.cpu().numpy()
    for bi in range(B):
        n = lengths[bi] if lengths else L
        xi = np.where(x[bi, :, :n] >= 0, x[bi, :, :n], x[bi, :, :n] * np.float32(0.1)).astype(np.float32)
        ref = c_oracle.conv1d(xi, w, b, dil, dil * (k - 1) // 2)  # fp32 operands, double accumulation
        got = y[bi, :n].T
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (case, bi, np.abs(got - ref).max())
        assert np.isnan(y[bi, n:]).all()  # rows past the sequence end are never written


def test_conv_transpose_vs_oracle():
    rng = np.random.default_rng(3)
    cin, cout, k, u, p, L = 128, 64, 16, 8, 4, 90  # an ups.1-shaped layer
    x = rng.standard_normal((2, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    yd = torch.zeros((2, L * u, cout), device="cuda")
    rc = _native.lib().mv_test_conv_transpose1d(xd.data_ptr(), None, 2, L, cin, _fp(w), _fp(b), cout, k, u, p, 1.0,
                                                _native.PREC_BF16X3, yd.data_ptr(), None)
    _native.check(rc, "mv_test_conv_transpose1d")
    y = yd.cpu().numpy()
    for bi in range(2):
        ref = c_oracle.conv_transpose1d(x[bi], w, b, u, p)
        assert np.abs(y[bi].T - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_hifigan_goldens_at_the_fp32_bar(golden_dir, gen_x3):
    for name, T in (("hifigan_v1_T64", 64), ("hifigan_v1_T500", 500)):
        d = np.load(os.path.join(golden_dir, name + ".npz"))
        mel = torch.from_numpy(synth.synthetic_mel(T, int(d["mel_seed"]))).cuda()
        y = gen_x3(mel[None])[0, 0].cpu().numpy()
        err = np.abs(y - d["wav"]).max()
        assert err <= 1e-3, (name, err)
        assert err <= 2e-4, (name, err)  # what the split actually delivers (5e-5 expected): a regression guard
    d = np.load(os.path.join(golden_dir, "hifigan_v1_ragged.npz"))
    lengths = [int(t) for t in d["lengths"]]
    Tmax = max(lengths) + 3
    mel = np.full((len(lengths), 80, Tmax), 3.0, np.float32)
    for b, T in enumerate(lengths):
        mel[b, :, :T] = synth.synthetic_mel(T, int(d["mel_seed"]) + b)
    y = gen_x3(torch.from_numpy(mel).cuda(), lengths=lengths)[:, 0].cpu().numpy()
    for b, T in enumerate(lengths):
        assert np.abs(y[b, :T * 256] - d[f"wav.{b}"]).max() <= 1e-3
        assert (y[b, T * 256:] == 0).all()
    p = gen_x3(torch.from_numpy(mel).cuda(), lengths=lengths, out="pcm16")[:, 0].cpu().numpy()
    assert np.array_equal(p, c_oracle.pcm16(y).reshape(p.shape))


@pytest.mark.parametrize("name,c", [("waveglow_c512_T16", 512), ("waveglow_c64_T16", 64)])
def test_waveglow_goldens_at_the_fp32_bar(name, c, golden_dir):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = c
    B, T = int(d["B"]), int(d["T"])
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(cfg, int(d["weight_seed"])).items()})
    m.cuda().eval()
    m.precision = "bf16x3"
    mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, int(d["mel_seed"]) + b) for b in range(B)])).cuda()
    noise = [torch.from_numpy(n) for n in synth.waveglow_noise(cfg, B, T, int(d["noise_seed"]))]
    y = m.infer(mel, sigma=float(d["sigma"]), noise=noise).cpu().numpy()
    assert np.abs(y - d["wav"]).max() <= 1e-3
