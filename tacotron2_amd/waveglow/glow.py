"""`glow` module with the reference's class names (waveglow/glow.py), computed by libmelvoc.so.

The reference loads WaveGlow as a *pickled module* after ``sys.path.append('waveglow')``
(do_syn.py:147-150): ``torch.load(path)['model']`` resolves ``glow.WaveGlow``, ``glow.WN`` and
``glow.Invertible1x1Conv`` by name.  Putting THIS directory on ``sys.path`` instead makes the same
checkpoint unpickle into the classes below (same attribute names: ``upsample``, ``WN[k].start /
in_layers / cond_layer / res_skip_layers / end``, ``convinv[k].conv``), whose ``infer`` runs on the
MI355X kernels.  ``WaveGlow.remove_weightnorm(model)``, ``.cuda()``, ``.eval()`` and
``.infer(spect, sigma)`` keep the reference's signatures (glow.py:251-313).

Noise: ``infer`` draws the same tensors, in the same order and shapes, from torch's device
generator as glow.py:262-271,287-292 do, so a fixed ``torch.manual_seed`` gives the reference's
samples; pass ``noise=[...]`` (unit variance, draw order) to supply them explicitly.
There is no PyTorch compute path and no CPU fallback.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

try:  # imported as tacotron2_amd.waveglow.glow ...
    from .. import _native
except ImportError:  # ... or as top-level `glow` with this directory on sys.path (pickle path)
    import importlib
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    _native = importlib.import_module("tacotron2_amd._native")


def _fold(conv: nn.Module) -> nn.Module:
    """remove_weight_norm that also works on hook-less or parametrized modules."""
    if "weight_g" in conv._parameters and "weight_v" in conv._parameters:
        v, g = conv._parameters["weight_v"].data, conv._parameters["weight_g"].data
        w = v * (g / v.flatten(1).norm(dim=1).view_as(g))
        for k in [k for k, hook in conv._forward_pre_hooks.items() if type(hook).__name__ == "WeightNorm"]:
            del conv._forward_pre_hooks[k]
        del conv._parameters["weight_g"], conv._parameters["weight_v"]
        conv.__dict__.pop("weight", None)
        conv.weight = nn.Parameter(w)
    elif hasattr(conv, "parametrizations") and "weight" in getattr(conv, "parametrizations", {}):
        torch.nn.utils.parametrize.remove_parametrizations(conv, "weight", leave_parametrized=True)
    return conv


class Invertible1x1Conv(nn.Module):
    """Parameter holder of glow.py:62-102; W^-1 is computed once inside mv_waveglow_create."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv1d(c, c, kernel_size=1, bias=False)
        w = torch.linalg.qr(torch.randn(c, c))[0]
        if torch.det(w) < 0:
            w[:, 0] = -w[:, 0]
        self.conv.weight.data = w.view(c, c, 1)


class WN(nn.Module):
    """Parameter holder of glow.py:105-151."""

    def __init__(self, n_in_channels, n_mel_channels, n_layers, n_channels, kernel_size):
        super().__init__()
        assert kernel_size % 2 == 1 and n_channels % 2 == 0
        self.n_layers, self.n_channels = n_layers, n_channels
        wn = torch.nn.utils.weight_norm
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.start = wn(nn.Conv1d(n_in_channels, n_channels, 1), name="weight")
        self.end = nn.Conv1d(n_channels, 2 * n_in_channels, 1)
        self.end.weight.data.zero_()
        self.end.bias.data.zero_()
        self.cond_layer = wn(nn.Conv1d(n_mel_channels, 2 * n_channels * n_layers, 1), name="weight")
        for i in range(n_layers):
            d = 2 ** i
            self.in_layers.append(wn(nn.Conv1d(n_channels, 2 * n_channels, kernel_size, dilation=d,
                                               padding=(kernel_size * d - d) // 2), name="weight"))
            rs = 2 * n_channels if i < n_layers - 1 else n_channels
            self.res_skip_layers.append(wn(nn.Conv1d(n_channels, rs, 1), name="weight"))


class WaveGlow(nn.Module):
    def __init__(self, n_mel_channels, n_flows, n_group, n_early_every, n_early_size, WN_config):
        super().__init__()
        assert n_group % 2 == 0
        self.upsample = nn.ConvTranspose1d(n_mel_channels, n_mel_channels, 1024, stride=256)
        self.n_flows, self.n_group = n_flows, n_group
        self.n_early_every, self.n_early_size = n_early_every, n_early_size
        self.WN = nn.ModuleList()
        self.convinv = nn.ModuleList()
        n_half, n_rem = n_group // 2, n_group
        for k in range(n_flows):
            if k % n_early_every == 0 and k > 0:
                n_half -= n_early_size // 2
                n_rem -= n_early_size
            self.convinv.append(Invertible1x1Conv(n_rem))
            self.WN.append(WN(n_half, n_mel_channels * n_group, **WN_config))
        self.n_remaining_channels = n_rem

    # ---- reference surface -----------------------------------------------------------------
    @staticmethod
    def remove_weightnorm(model):
        for wn in model.WN:
            wn.start = _fold(wn.start)
            wn.in_layers = nn.ModuleList([_fold(c) for c in wn.in_layers])
            wn.cond_layer = _fold(wn.cond_layer)
            wn.res_skip_layers = nn.ModuleList([_fold(c) for c in wn.res_skip_layers])
        model.__dict__.pop("_mv_handles", None)
        return model

    def infer(self, spect, sigma=1.0, lengths=None, noise=None, out="float"):
        if spect.dim() != 3 or spect.size(1) != self.upsample.in_channels:
            raise RuntimeError(f"expected spect [B,{self.upsample.in_channels},T], got {tuple(spect.shape)}")
        if not spect.is_cuda:
            raise RuntimeError("glow.WaveGlow.infer computes on an MI355X only (no CPU fallback); "
                               "move the model and the mel to a ROCm device")
        in_dtype = spect.dtype
        mel = spect.detach().to(torch.float32).contiguous()
        B, _, T = mel.shape
        dev = mel.device
        L = T * 256 // self.n_group
        shapes = [(B, self.n_remaining_channels, L)]
        for k in reversed(range(self.n_flows)):
            if k % self.n_early_every == 0 and k > 0:
                shapes.append((B, self.n_early_size, L))
        if noise is None:
            # same draws, same order as glow.py:267-269 and 291 (the Half branches draw half tensors)
            nd = in_dtype if in_dtype == torch.float16 else torch.float32
            noise = [torch.empty(s, dtype=nd, device=dev).normal_() for s in shapes]
        if len(noise) != len(shapes):
            raise RuntimeError(f"noise must hold {len(shapes)} tensors in draw order")
        noise = [torch.as_tensor(n).to(dev, torch.float32).contiguous() for n in noise]
        for n, s in zip(noise, shapes):
            if tuple(n.shape) != s:
                raise RuntimeError(f"noise tensor of shape {tuple(n.shape)}, expected {s}")
        if lengths is not None:
            lengths = _native.device_lengths(lengths, B, T, dev)
        h = self._handle(dev)
        pcm = out == "pcm16"
        wav = torch.empty((B, T * 256), dtype=torch.int16 if pcm else torch.float32, device=dev)
        import ctypes as C

        nptr = (C.c_void_p * len(noise))(*[n.data_ptr() for n in noise])
        with torch.cuda.device(dev):
            ws = _native.workspace(self.__dict__.setdefault("_mv_workspace", {}), dev, h.workspace_bytes(B, T))
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check(_native.lib().mv_waveglow_infer(
                h.ptr, mel.data_ptr(), lengths.data_ptr() if lengths is not None else None, B, T, nptr, len(noise),
                float(sigma), wav.data_ptr(), _native.OUT_PCM16 if pcm else _native.OUT_F32, ws.data_ptr(),
                ws.numel(), stream), "mv_waveglow_infer")
        if not pcm and in_dtype in (torch.float16, torch.bfloat16):
            wav = wav.to(in_dtype)
        return wav

    def forward(self, forward_input):
        raise NotImplementedError("training forward (glow.py:207-249) is outside the vocoding hot path")

    # ---- native plumbing -------------------------------------------------------------------
    @property
    def precision(self):
        return self.__dict__.get("_mv_precision") or os.environ.get("MELVOC_PRECISION", "fp32")

    @precision.setter
    def precision(self, p):
        if p not in _native.PRECISIONS:
            raise ValueError(f"unknown vocoder precision {p!r}")
        self.__dict__["_mv_precision"] = p
        self.__dict__.pop("_mv_handles", None)

    @property
    def options(self):
        """mv_options fields (include/melvoc.h) the native handle is created with; None = the defaults"""
        return self.__dict__.get("_mv_options")

    @options.setter
    def options(self, o):
        self.__dict__["_mv_options"] = dict(o) if o else None
        self.__dict__.pop("_mv_handles", None)

    def config(self):
        wn0 = self.WN[0]
        return {
            "n_mel_channels": self.upsample.in_channels, "n_flows": self.n_flows, "n_group": self.n_group,
            "n_early_every": self.n_early_every, "n_early_size": self.n_early_size,
            "WN_config": {"n_layers": wn0.n_layers, "n_channels": wn0.n_channels,
                          "kernel_size": wn0.in_layers[0].kernel_size[0]},
        }

    def native_state(self):
        return {k: v.detach().to("cpu", torch.float32).numpy() for k, v in self.state_dict().items()}

    def _handle(self, dev):
        handles = self.__dict__.setdefault("_mv_handles", {})
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), self.precision)
        h = handles.get(key)
        if h is None:
            h = _native.create_waveglow(self.config(), self.native_state(), key[0], _native.PRECISIONS[self.precision],
                                        self.options)
            handles[key] = h
        return h

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.__dict__.pop("_mv_handles", None)
        return out

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_mv_handles", None)
        d.pop("_mv_workspace", None)
        return d

    def native_handle(self, dev):
        return self._handle(torch.device(dev))

    def macs_per_frame(self, dev):
        return self._handle(dev).macs_per_frame


def remove(conv_list):
    """glow.py:308-313."""
    return nn.ModuleList([_fold(c) for c in conv_list])
