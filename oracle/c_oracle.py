"""ORACLE — test infrastructure, not product code.

ctypes binding of oracle/libvocoder_oracle.so (built from vocoder_oracle.c by
oracle/Makefile; __graft_entry__.build() runs it).  Only tests/, smoke() and
bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SO: another build of the same source (the ASan + UBSan one of tests/test_oracle_asan.py)
_SO = os.environ.get("ORACLE_SO") or os.path.join(_HERE, "libvocoder_oracle.so")
_lib = None


class _HifiganCfg(C.Structure):
    _fields_ = [
        ("num_mels", C.c_int),
        ("upsample_initial_channel", C.c_int),
        ("num_upsamples", C.c_int),
        ("upsample_rates", C.c_int * 8),
        ("upsample_kernel_sizes", C.c_int * 8),
        ("num_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * 8),
        ("resblock_dilations", (C.c_int * 3) * 8),
    ]


class _WaveglowCfg(C.Structure):
    _fields_ = [
        ("n_mel_channels", C.c_int),
        ("n_flows", C.c_int),
        ("n_group", C.c_int),
        ("n_early_every", C.c_int),
        ("n_early_size", C.c_int),
        ("n_layers", C.c_int),
        ("n_channels", C.c_int),
        ("kernel_size", C.c_int),
    ]


class _Weights(C.Structure):
    _fields_ = [
        ("names", C.POINTER(C.c_char_p)),
        ("data", C.POINTER(C.POINTER(C.c_float))),
        ("n", C.c_int),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vocoder_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvocoder_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_hifigan_forward.restype = C.c_int
        _lib.orc_waveglow_infer.restype = C.c_int
        _lib.orc_waveglow_upsample.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class _W:
    def __init__(self, state):
        self.arrays = [(k, _f32(v)) for k, v in state.items()]
        n = len(self.arrays)
        self.names = (C.c_char_p * n)(*[k.encode() for k, _ in self.arrays])
        self.ptrs = (C.POINTER(C.c_float) * n)(*[_fp(a) for _, a in self.arrays])
        self.struct = _Weights(self.names, self.ptrs, n)


def _hcfg(cfg):
    c = _HifiganCfg()
    c.num_mels = int(cfg.get("num_mels", 80))
    c.upsample_initial_channel = int(cfg["upsample_initial_channel"])
    c.num_upsamples = len(cfg["upsample_rates"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        c.upsample_rates[i] = int(u)
        c.upsample_kernel_sizes[i] = int(k)
    c.num_kernels = len(cfg["resblock_kernel_sizes"])
    for j, (k, d) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
        c.resblock_kernel_sizes[j] = int(k)
        for m in range(3):
            c.resblock_dilations[j][m] = int(d[m])
    return c


def _wcfg(cfg):
    wn = cfg["WN_config"]
    return _WaveglowCfg(cfg["n_mel_channels"], cfg["n_flows"], cfg["n_group"], cfg["n_early_every"],
                        cfg["n_early_size"], wn["n_layers"], wn["n_channels"], wn["kernel_size"])


def conv1d(x, w, b, dil=1, pad=0, bf16_operands=False):
    x, w = _f32(x), _f32(w)
    cout, cin, k = w.shape
    L = x.shape[1]
    y = np.empty((cout, L + 2 * pad - dil * (k - 1)), np.float32)
    bb = _f32(b) if b is not None else None
    lib().orc_conv1d(_fp(x), cin, L, _fp(w), _fp(bb) if bb is not None else None, cout, k, dil, pad,
                     int(bf16_operands), _fp(y))
    return y


def conv_transpose1d(x, w, b, stride, pad, bf16_operands=False):
    x, w = _f32(x), _f32(w)
    cin, cout, k = w.shape
    L = x.shape[1]
    y = np.empty((cout, (L - 1) * stride - 2 * pad + k), np.float32)
    bb = _f32(b) if b is not None else None
    lib().orc_conv_transpose1d(_fp(x), cin, L, _fp(w), _fp(bb) if bb is not None else None, cout, k, stride,
                               pad, int(bf16_operands), _fp(y))
    return y


def hifigan_forward(state, cfg, mel, bf16_operands=False, want_stage=False):
    """mel [80,T] (one sequence) -> wav [256T] (and the last MRF stage output [C,L])."""
    mel = _f32(mel)
    assert mel.ndim == 2
    T = mel.shape[1]
    up = int(np.prod(cfg["upsample_rates"]))
    W = _W(state)
    wav = np.empty(T * up, np.float32)
    c_last = int(cfg["upsample_initial_channel"]) >> len(cfg["upsample_rates"])
    stage = np.empty((c_last, T * up), np.float32) if want_stage else None
    hc = _hcfg(cfg)
    rc = lib().orc_hifigan_forward(C.byref(hc), C.byref(W.struct), _fp(mel), T, int(bf16_operands), _fp(wav),
                                   _fp(stage) if want_stage else None)
    if rc:
        raise RuntimeError("orc_hifigan_forward failed")
    return (wav, stage) if want_stage else wav


def hifigan_forward_batch(state, cfg, mels, lengths, bf16_operands=False):
    """Per-utterance batch-1 vocoding of a padded batch (the parity definition)."""
    return [hifigan_forward(state, cfg, np.asarray(mels[b])[:, : int(lengths[b])], bf16_operands)
            for b in range(len(lengths))]


def waveglow_upsample(state, cfg, mel):
    mel = _f32(mel)
    T = mel.shape[1]
    L = T * 256 // cfg["n_group"]
    W = _W({k: v for k, v in state.items() if k.startswith("upsample.")})
    spect = np.empty((cfg["n_mel_channels"] * cfg["n_group"], L), np.float32)
    wc = _wcfg(cfg)
    if lib().orc_waveglow_upsample(C.byref(wc), C.byref(W.struct), _fp(mel), T, _fp(spect)):
        raise RuntimeError("orc_waveglow_upsample failed")
    return spect


def waveglow_infer(state, cfg, mel, noise, sigma=1.0, bf16_operands=False, want_flows=False):
    """mel [80,T]; noise: list of unit-variance [c,L] arrays in draw order -> wav [256T]."""
    mel = _f32(mel)
    T = mel.shape[1]
    ng = cfg["n_group"]
    L = T * 256 // ng
    W = _W(state)
    ns = [_f32(n) for n in noise]
    nptr = (C.POINTER(C.c_float) * len(ns))(*[_fp(n) for n in ns])
    wav = np.empty(T * 256, np.float32)
    flows = np.zeros((cfg["n_flows"], ng, L), np.float32) if want_flows else None
    wc = _wcfg(cfg)
    rc = lib().orc_waveglow_infer(C.byref(wc), C.byref(W.struct), _fp(mel), T, nptr, C.c_float(sigma),
                                  int(bf16_operands), _fp(wav), _fp(flows) if want_flows else None)
    if rc:
        raise RuntimeError("orc_waveglow_infer failed")
    return (wav, flows) if want_flows else wav


def invert(A):
    A = _f32(A)
    n = A.shape[0]
    out = np.empty((n, n), np.float32)
    lib().orc_invert(_fp(A), n, _fp(out))
    return out


def pcm16(wav):
    wav = _f32(wav).ravel()
    out = np.empty(wav.shape, np.int16)
    lib().orc_pcm16(_fp(wav), C.c_size_t(wav.size), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def postnet_forward(state, hps, x, add_input=False, bf16_operands=False, i_decoder=0, eps=1e-5):
    """x [dim,T] (one sequence) -> [dim,T]: model.py:170-174 (+ :717) on orc_conv1d (double accumulation);
    BatchNorm1d (eval) is applied in float64 after each conv, or folded first when modelling bf16 operands."""
    n = hps["postnet_n_convolutions"][i_decoder]
    k = hps["postnet_kernel_size"][i_decoder]
    x0 = _f32(x)
    y = x0
    for i in range(n):
        pre = "convolutions.%d" % i
        w, b = np.asarray(state[pre + ".0.conv.weight"], np.float64), np.asarray(state[pre + ".0.conv.bias"], np.float64)
        g, beta = np.asarray(state[pre + ".1.weight"], np.float64), np.asarray(state[pre + ".1.bias"], np.float64)
        mu, var = np.asarray(state[pre + ".1.running_mean"], np.float64), np.asarray(state[pre + ".1.running_var"], np.float64)
        s = g / np.sqrt(var + eps)
        if bf16_operands:
            y = conv1d(y, (w * s[:, None, None]).astype(np.float32), ((b - mu) * s + beta).astype(np.float32),
                       1, (k - 1) // 2, True)
        else:
            c = conv1d(y, w.astype(np.float32), b.astype(np.float32), 1, (k - 1) // 2, False).astype(np.float64)
            y = ((c - mu[:, None]) * s[:, None] + beta[:, None]).astype(np.float32)
        if i < n - 1:
            y = np.tanh(y.astype(np.float64)).astype(np.float32)
    return (x0 + y) if add_input else y
