"""The C ABI on its own (ctypes straight into libmelvoc.so, no Python-side checks in between): out-of-range
`lengths` entries are clamped by the kernels instead of walking them out of the buffers, and the
library's own weight-norm fold (`weight_g` / `weight_v` handed to mv_*_create, layers.hip WeightMap::get) gives the
same waveform as the reference's remove_weight_norm done on the host (hifigan/models.py:167-174, glow.py:297-313)."""
import copy
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from tacotron2_amd import _native, synth  # noqa: E402

GUARD = 4096  # elements either side of every buffer the library writes


def _forward_raw(h, mel, lengths, out_pcm=False):
    """mv_hifigan_forward through ctypes with guard bands around wav and the workspace."""
    B, _, T = mel.shape
    hop = h.hop
    n = B * T * hop
    wav = torch.full((n + 2 * GUARD,), 7.0 if not out_pcm else 7, dtype=torch.int16 if out_pcm else torch.float32,
                     device="cuda")
    nws = h.workspace_bytes(B, T)
    ws = torch.full((nws + 2 * 4096,), 0x5A, dtype=torch.uint8, device="cuda")
    esz = 2 if out_pcm else 4
    ld = torch.tensor(lengths, dtype=torch.int32, device="cuda")
    rc = _native.lib().mv_hifigan_forward(h.ptr, mel.data_ptr(), ld.data_ptr(), B, T, wav.data_ptr() + GUARD * esz,
                                          _native.OUT_PCM16 if out_pcm else _native.OUT_F32, ws.data_ptr() + 4096,
                                          nws, None)
    _native.check(rc, "mv_hifigan_forward")
    torch.cuda.synchronize()
    w = wav.cpu().numpy()
    assert (w[:GUARD] == 7).all() and (w[-GUARD:] == 7).all(), "wav guard band overwritten"
    g = ws.cpu().numpy()
    assert (g[:4096] == 0x5A).all() and (g[-4096:] == 0x5A).all(), "workspace guard band overwritten"
    return w[GUARD:-GUARD].reshape(B, T * hop)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_hifigan_out_of_range_lengths_are_clamped(prec):
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    h = _native.create_hifigan(synth.HIFIGAN_V1, st, 0, _native.PRECISIONS[prec])
    T = 21
    mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, 70 + b) for b in range(4)])).cuda()
    bad = _forward_raw(h, mel, [T + 1000, -5, 2 ** 31 - 1, 9])
    good = _forward_raw(h, mel, [T, 0, T, 9])
    assert np.array_equal(bad, good)
    assert not bad[1].any() and not bad[3, 9 * 256:].any()
    ref = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, synth.synthetic_mel(T, 70), bf16_operands=prec == "bf16")
    if prec == "fp32":
        assert np.abs(bad[0] - ref).max() <= 1e-3
    else:
        r = ref.astype(np.float64)
        assert 10 * np.log10((r ** 2).sum() / ((r - bad[0]) ** 2).sum()) >= 40.0
    pcm = _forward_raw(h, mel, [T + 1000, -5, 2 ** 31 - 1, 9], out_pcm=True)
    assert np.array_equal(pcm, c_oracle.pcm16(bad).reshape(pcm.shape))


def test_waveglow_and_postnet_out_of_range_lengths_are_clamped():
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = 64
    st = synth.waveglow_state(cfg, 3)
    for prec in ("fp32", "bf16"):
        h = _native.create_waveglow(cfg, st, 0, _native.PRECISIONS[prec])
        B, T = 3, 6
        mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, 5 + b) for b in range(B)])).cuda()
        noise = [torch.from_numpy(n).cuda() for n in synth.waveglow_noise(cfg, B, T, 17)]
        nptr = (C.c_void_p * len(noise))(*[n.data_ptr() for n in noise])
        outs = []
        for lens in ([T + 99, -1, 4], [T, 0, 4]):
            wav = torch.full((B * T * 256 + 2 * GUARD,), 7.0, device="cuda")
            nws = h.workspace_bytes(B, T)
            ws = torch.full((nws + 8192,), 0x5A, dtype=torch.uint8, device="cuda")
            ld = torch.tensor(lens, dtype=torch.int32, device="cuda")
            rc = _native.lib().mv_waveglow_infer(h.ptr, mel.data_ptr(), ld.data_ptr(), B, T, nptr, len(noise), 0.666,
                                                 wav.data_ptr() + GUARD * 4, _native.OUT_F32, ws.data_ptr() + 4096, nws,
                                                 None)
            _native.check(rc, "mv_waveglow_infer")
            torch.cuda.synchronize()
            w, g = wav.cpu().numpy(), ws.cpu().numpy()
            assert (w[:GUARD] == 7).all() and (w[-GUARD:] == 7).all()
            assert (g[:4096] == 0x5A).all() and (g[-4096:] == 0x5A).all()
            outs.append(w[GUARD:-GUARD].reshape(B, T * 256))
        assert np.array_equal(outs[0], outs[1])
        assert not outs[0][1].any() and not outs[0][2, 4 * 256:].any()

    pst = synth.postnet_state(synth.POSTNET_TC2, 0)
    pcfg = _native.mv_postnet_cfg(80, 512, 5, 5, 1e-5)
    h = _native.create_postnet(pcfg, pst, 0, _native.PREC_FP32)
    B, T = 3, 11
    x = torch.from_numpy(np.stack([synth.synthetic_mel(T, 30 + b) for b in range(B)])).cuda()
    outs = []
    for lens in ([T + 5, -3, 7], [T, 0, 7]):
        y = torch.full((B * 80 * T + 2 * GUARD,), 7.0, device="cuda")
        nws = h.workspace_bytes(B, T)
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        ld = torch.tensor(lens, dtype=torch.int32, device="cuda")
        rc = _native.lib().mv_postnet_forward(h.ptr, x.data_ptr(), ld.data_ptr(), B, T, y.data_ptr() + GUARD * 4, 1,
                                              ws.data_ptr(), nws, None)
        _native.check(rc, "mv_postnet_forward")
        torch.cuda.synchronize()
        w = y.cpu().numpy()
        assert (w[:GUARD] == 7).all() and (w[-GUARD:] == 7).all()
        outs.append(w[GUARD:-GUARD])
    assert np.array_equal(outs[0], outs[1])


def test_library_folds_weight_norm_pairs_itself(golden_dir):
    """weight_g / weight_v straight into mv_hifigan_create: the C fold, incl. ConvTranspose1d's per-INPUT-channel
    norm, against the golden waveform of the reference generator after remove_weight_norm()."""
    import os

    d = np.load(os.path.join(golden_dir, "hifigan_v1_T64.npz"))
    wn = synth.hifigan_state(synth.HIFIGAN_V1, 0, weight_norm=True)
    assert any(k.endswith("weight_g") for k in wn) and not any(k.endswith(".weight") for k in wn)
    h = _native.create_hifigan(synth.HIFIGAN_V1, wn, 0, _native.PREC_FP32)
    mel = torch.from_numpy(synth.synthetic_mel(64, int(d["mel_seed"]))[None]).cuda()
    y = _forward_raw(h, mel, [64])[0]
    assert np.abs(y - d["wav"]).max() <= 1e-3
    # a g vector of the wrong size is an error code, not a crash
    bad = dict(wn)
    k = next(k for k in bad if k.endswith("ups.0.weight_g"))
    bad[k] = bad[k][:-1]
    with pytest.raises(_native.MelvocError):
        _native.create_hifigan(synth.HIFIGAN_V1, bad, 0, _native.PREC_FP32)


def test_library_folds_waveglow_weight_norm_pairs_itself():
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = 64
    st = synth.waveglow_state(cfg, 5)
    # un-fold every weight-normalised tensor of glow.py:297-313 into a (g, v) pair with a non-trivial g
    rng = np.random.default_rng(0)
    wn = {}
    for k, w in st.items():
        normed = k.endswith(".weight") and any(s in k for s in (".start.", ".in_layers.", ".cond_layer.", ".res_skip_layers."))
        if not normed:
            wn[k] = w
            continue
        scale = rng.uniform(0.5, 2.0, size=(w.shape[0],) + (1,) * (w.ndim - 1)).astype(np.float32)
        v = (w * scale).astype(np.float32)
        g = np.sqrt((w.astype(np.float64) ** 2).reshape(w.shape[0], -1).sum(1)).astype(np.float32).reshape(scale.shape)
        wn[k + "_g"], wn[k + "_v"] = g, v
    B, T = 1, 8
    mel = torch.from_numpy(synth.synthetic_mel(T, 2)[None]).cuda()
    noise_np = synth.waveglow_noise(cfg, B, T, 4)
    noise = [torch.from_numpy(n).cuda() for n in noise_np]
    nptr = (C.c_void_p * len(noise))(*[n.data_ptr() for n in noise])
    h = _native.create_waveglow(cfg, wn, 0, _native.PREC_FP32)
    wav = torch.empty((B, T * 256), device="cuda")
    nws = h.workspace_bytes(B, T)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    _native.check(_native.lib().mv_waveglow_infer(h.ptr, mel.data_ptr(), None, B, T, nptr, len(noise), 0.666,
                                                  wav.data_ptr(), _native.OUT_F32, ws.data_ptr(), nws, None),
                  "mv_waveglow_infer")
    ref = c_oracle.waveglow_infer(st, cfg, synth.synthetic_mel(T, 2), [n[0] for n in noise_np], 0.666)
    assert np.abs(wav[0].cpu().numpy() - ref).max() <= 1e-3


def test_pcm16_wraps_like_numpy_and_singular_convinv_is_an_error():
    y = np.array([0.99999, -1.0, 1.0, 1.5, -1.5, 3.0, 1234.5678, 65535.9, 70000.0, -70000.0, 1e10, -1e10, np.inf, -np.inf,
                  np.nan, 0.0, -0.00001], np.float32)
    yd = torch.from_numpy(y).cuda()
    pd = torch.zeros(y.size, dtype=torch.int16, device="cuda")
    _native.check(_native.lib().mv_pcm16(yd.data_ptr(), y.size, pd.data_ptr(), None), "mv_pcm16")
    torch.cuda.synchronize()
    assert np.array_equal(pd.cpu().numpy(), c_oracle.pcm16(y))  # the oracle is pinned to numpy on the CPU side
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = 64
    st = dict(synth.waveglow_state(cfg, 0))
    w = st["convinv.3.conv.weight"].copy()
    w[1] = w[0]  # two equal rows: singular
    st["convinv.3.conv.weight"] = w
    with pytest.raises(_native.MelvocError, match="singular"):
        _native.create_waveglow(cfg, st, 0, _native.PREC_FP32)
