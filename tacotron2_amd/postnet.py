"""Drop-in for the reference's `model.Postnet` (model.py:140-174), the step right before the vocoder:
five Conv1d(k=5) + BatchNorm1d with tanh between them, computed on the MI355X by libmelvoc
(`mv_postnet_create/forward`, BatchNorm folded at create, eval mode only -- this is the synthesis path).

    post = Postnet(hps, i_decoder)            # same constructor, same state_dict keys as the reference
    post.load_state_dict(ref_postnet.state_dict()); post.eval().to(device)
    y = post(mel)                             # model.py:170-174
    mel_post = post(mel, add_input=True)      # model.py:717: par_out + self.postnet(par_out)

`lengths=` (non-breaking extra) makes a right-padded batch equal per-utterance batch-1 calls.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _native


class _ConvNorm(nn.Module):
    """parameter holder with the reference's key layout (`<i>.0.conv.weight`); never run on the CPU"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size=k, stride=1, padding=(k - 1) // 2)


class Postnet(nn.Module):
    def __init__(self, hps, i_decoder=0, precision=None):
        super().__init__()
        dim, emb = int(hps["dim_data"][i_decoder]), int(hps["postnet_embedding_dim"][i_decoder])
        k, n = int(hps["postnet_kernel_size"][i_decoder]), int(hps["postnet_n_convolutions"][i_decoder])
        self.p_postnet_dropout = hps.get("p_postnet_dropout", [0.5] * (i_decoder + 1))[i_decoder]
        self.dim, self.emb, self.k, self.n = dim, emb, k, n
        self.convolutions = nn.ModuleList()
        for i in range(n):
            cin, cout = (dim if i == 0 else emb), (dim if i == n - 1 else emb)
            self.convolutions.append(nn.Sequential(_ConvNorm(cin, cout, k), nn.BatchNorm1d(cout)))
        self.precision = precision or os.environ.get("MELVOC_PRECISION", "fp32")
        if self.precision not in _native.PRECISIONS:
            raise ValueError(f"unknown precision {self.precision!r}")
        self._handles, self._workspace = {}, {}

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._handles.clear()
        return out

    def _apply(self, fn, *a, **kw):
        self._handles.clear()
        return super()._apply(fn, *a, **kw)

    def native_state(self):
        return {k: v.detach().to("cpu", torch.float32).numpy() for k, v in self.state_dict().items()
                if not k.endswith("num_batches_tracked")}

    def native_handle(self, dev=None):
        dev = torch.device(dev) if dev is not None else next(self.parameters()).device
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), self.precision)
        h = self._handles.get(key)
        if h is None:
            cfg = _native.mv_postnet_cfg(self.dim, self.emb, self.k, self.n, float(self.convolutions[0][1].eps))
            h = _native.create_postnet(cfg, self.native_state(), key[0], _native.PRECISIONS[self.precision])
            self._handles[key] = h
        return h

    def macs_per_frame(self, dev=None):
        return self.native_handle(dev).macs_per_frame

    def forward(self, x, lengths=None, add_input=False):
        if self.training:
            raise RuntimeError("tacotron2_amd.postnet.Postnet is the synthesis-time (eval) path: call .eval() first")
        if x.dim() != 3 or x.size(1) != self.dim:
            raise RuntimeError(f"expected [B,{self.dim},T], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("tacotron2_amd.postnet.Postnet computes on an MI355X only (no CPU fallback)")
        in_dtype = x.dtype
        xin = x.detach().to(torch.float32).contiguous()
        B, _, T = xin.shape
        dev = xin.device
        h = self.native_handle(dev)
        if lengths is not None:
            lengths = _native.device_lengths(lengths, B, T, dev)
        y = torch.empty_like(xin)
        with torch.cuda.device(dev):
            ws = _native.workspace(self._workspace, dev, h.workspace_bytes(B, T))
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check(_native.lib().mv_postnet_forward(
                h.ptr, xin.data_ptr(), lengths.data_ptr() if lengths is not None else None, B, T, y.data_ptr(),
                1 if add_input else 0, ws.data_ptr(), ws.numel(), stream), "mv_postnet_forward")
        return y.to(in_dtype) if in_dtype != torch.float32 else y

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_handles"], d["_workspace"] = {}, {}
        return d
