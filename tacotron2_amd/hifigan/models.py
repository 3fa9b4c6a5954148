"""`Generator` with the reference's call surface (hifigan/models.py:112-174), computed by
libmelvoc.so on an MI355X.

What the drivers do with it (do_syn.py:155-169,278,391; do_tts.py:149-162,351) all works
unchanged: ``Generator(AttrDict(json))``, ``load_state_dict(ckpt["generator"])`` with
``weight_g``/``weight_v`` pairs or folded ``weight`` keys, ``eval()``, ``remove_weight_norm()``,
iteration over ``parameters()`` / ``buffers()``, ``.to(device)``, and ``__call__`` on
``[B,80,T] -> [B,1,256T]`` or ``[80,T] -> [1,256T]``.

Additions that do not break that surface: ``lengths=`` (int32 frame counts of a right-padded
batch; row b then equals the batch-1 result for ``mel[b,:,:lengths[b]]``), ``out="pcm16"``
(fused do_syn.py:389-394 conversion) and the ``vocoder_dtype`` config key / ``precision=``
argument (``"fp32"`` exact-fp32 MFMA, ``"bf16"`` bf16 operands with fp32 accumulation).

The module holds parameters only; there is no PyTorch compute path and no CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native

LRELU_SLOPE = 0.1  # models.py:7 (applied inside the kernels' staging step)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class _NormedConv(nn.Module):
    """Parameters of one weight-normalised convolution: (weight_g, weight_v, bias) as in a
    checkpoint, or (weight, bias) once folded.  dim-0 of the weight is Cout for Conv1d and Cin for
    ConvTranspose1d, which is the axis torch's weight_norm keeps."""

    def __init__(self, wshape, n_bias, std=0.01):
        super().__init__()
        v = torch.randn(*wshape) * std
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, *([1] * (len(wshape) - 1))))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(n_bias))

    @property
    def folded(self):
        return "weight" in self._parameters

    def fold(self):
        if self.folded:
            return
        v, g = self.weight_v.data, self.weight_g.data
        norm = v.flatten(1).norm(dim=1).view_as(g)
        w = v * (g / norm)
        rg = self.weight_v.requires_grad
        del self._parameters["weight_g"], self._parameters["weight_v"]
        self.weight = nn.Parameter(w, requires_grad=rg)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # accept either form of checkpoint, whatever form the module is in
        has_w, has_gv = prefix + "weight" in state_dict, prefix + "weight_g" in state_dict
        if has_w and not self.folded:
            self.fold()
        elif has_gv and self.folded:
            rg = self.weight.requires_grad
            shape = self.weight.shape
            del self._parameters["weight"]
            self.weight_g = nn.Parameter(torch.ones(shape[0], *([1] * (len(shape) - 1))), requires_grad=rg)
            self.weight_v = nn.Parameter(torch.zeros(shape), requires_grad=rg)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class ResBlock(nn.Module):
    """Parameter container of one ResBlock1 (models.py:20-109): convs1[m] dilated, convs2[m] plain."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.kernel_size = kernel_size
        self.dilation = tuple(dilation)
        mk = lambda: _NormedConv((channels, channels, kernel_size), channels)  # noqa: E731
        self.convs1 = nn.ModuleList([mk() for _ in self.dilation])
        self.convs2 = nn.ModuleList([mk() for _ in self.dilation])

    def remove_weight_norm(self):
        for c in list(self.convs1) + list(self.convs2):
            c.fold()


class Generator(nn.Module):
    def __init__(self, h, precision=None, options=None):
        super().__init__()
        self.h = h
        self.options = dict(options) if options else None  # mv_options fields (include/melvoc.h); None = defaults
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        c0 = int(h["upsample_initial_channel"])
        n_mels = int(h.get("num_mels", 80))
        self.conv_pre = _NormedConv((c0, n_mels, 7), c0)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        ch = c0
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
            self.ups.append(_NormedConv((cin, ch, int(k)), ch))
            for k_rb, d in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(ResBlock(h, ch, int(k_rb), d))
        self.conv_post = _NormedConv((1, ch, 7), 1)
        self.hop = int(np.prod(h["upsample_rates"]))
        self.precision = precision or h.get("vocoder_dtype") or os.environ.get("MELVOC_PRECISION", "fp32")
        if self.precision not in _native.PRECISIONS:
            raise ValueError(f"unknown vocoder precision {self.precision!r}")
        self._handles = {}
        self._workspace = {}

    # ---- reference surface -----------------------------------------------------------------
    def remove_weight_norm(self):
        print("Removing weight norm...")  # the reference prints this too (models.py:168)
        for m in self.modules():
            if isinstance(m, _NormedConv):
                m.fold()
        self._handles.clear()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._handles.clear()
        return out

    def _apply(self, fn, *a, **kw):
        self._handles.clear()
        return super()._apply(fn, *a, **kw)

    def forward(self, x, lengths=None, out="float"):
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        if x.dim() != 3 or x.size(1) != self.conv_pre_in_channels:
            raise RuntimeError(f"expected mel of shape [B,{self.conv_pre_in_channels},T] or "
                               f"[{self.conv_pre_in_channels},T], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("tacotron2_amd.hifigan.Generator computes on an MI355X only (no CPU fallback); "
                               "move the module and the mel to a ROCm device")
        in_dtype = x.dtype
        mel = x.detach().to(torch.float32).contiguous()
        B, _, T = mel.shape
        dev = mel.device
        h = self._handle(dev)
        if lengths is not None:
            lengths = _native.device_lengths(lengths, B, T, dev)
        pcm = out == "pcm16"
        wav = torch.empty((B, T * self.hop), dtype=torch.int16 if pcm else torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _native.workspace(self._workspace, dev, h.workspace_bytes(B, T))
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check(_native.lib().mv_hifigan_forward(
                h.ptr, mel.data_ptr(), lengths.data_ptr() if lengths is not None else None, B, T, wav.data_ptr(),
                _native.OUT_PCM16 if pcm else _native.OUT_F32, ws.data_ptr(), ws.numel(), stream),
                "mv_hifigan_forward")
        if not pcm and in_dtype in (torch.float16, torch.bfloat16):
            wav = wav.to(in_dtype)
        wav = wav.unsqueeze(1)  # [B,1,256T]
        return wav[0] if squeeze else wav

    # ---- native plumbing -------------------------------------------------------------------
    @property
    def conv_pre_in_channels(self):
        w = self.conv_pre.weight if self.conv_pre.folded else self.conv_pre.weight_v
        return w.shape[1]

    def native_state(self):
        """state_dict as float32 numpy arrays (either checkpoint form; the library folds)."""
        return {k: v.detach().to("cpu", torch.float32).numpy() for k, v in self.state_dict().items()}

    def _handle(self, dev):
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), self.precision)
        h = self._handles.get(key)
        if h is None:
            h = _native.create_hifigan(self.h, self.native_state(), key[0], _native.PRECISIONS[self.precision],
                                       self.options)
            self._handles[key] = h
        return h

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_handles"], d["_workspace"] = {}, {}
        return d

    def native_handle(self, dev=None):
        """The _native.Handle serving `dev` (built on first use); bench.py profiles through it."""
        dev = torch.device(dev) if dev is not None else next(self.parameters()).device
        return self._handle(dev)

    def macs_per_frame(self, dev=None):
        dev = dev or next(self.parameters()).device
        return self._handle(dev).macs_per_frame
