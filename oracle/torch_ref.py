"""ORACLE — test infrastructure, not product code.

PyTorch-CPU functional restatement of the two reference vocoders, used only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (it is the
"reference PyTorch CPU vocoder" timed beside the GPU numbers; the reference's
own .py files cannot travel to the GPU box). Nothing under tacotron2_amd/
imports this file.

Pinned: tests/test_oracle_golden.py checks it against tests/golden/*.npz, which
tests/gen_golden.py produced by importing the real reference modules
(/root/reference/hifigan/models.py, /root/reference/waveglow/glow.py) in the
build container.

Follows:
  hifigan_forward   hifigan/models.py:149-165 (Generator.forward)
  _resblock         hifigan/models.py:96-103  (ResBlock.forward)
  waveglow_infer    waveglow/glow.py:251-295  (WaveGlow.infer), noise explicit
  _wn               waveglow/glow.py:153-175  (WN.forward)
  _gate             waveglow/glow.py:33-40    (fused_add_tanh_sigmoid_multiply)
  postnet_forward   model.py:170-174,717      (Postnet.forward in eval mode, + input)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # hifigan/models.py:7


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _Q:
    """Operand rounding model: identity (fp32) or bf16 operands with fp32 accumulate."""

    def __init__(self, bf16: bool):
        self.bf16 = bf16

    def __call__(self, x):
        return _bf16(x) if self.bf16 else x


def hifigan_forward(state, cfg, mel, bf16_operands: bool = False, taps=None):
    """mel [B,80,T] or [80,T] float32 -> [B,1,256T] / [1,256T] (models.py:149-165).

    `taps`, if a dict, receives per-stage tensors (for kernel unit tests).
    """
    q = _Q(bf16_operands)
    w = {k: _t(v) for k, v in state.items()}
    x = _t(mel)
    squeeze = x.dim() == 2
    if squeeze:
        x = x[None]
    nk = len(cfg["resblock_kernel_sizes"])
    with torch.no_grad():
        x = F.conv1d(q(x), q(w["conv_pre.weight"]), w["conv_pre.bias"], padding=3)
        if taps is not None:
            taps["conv_pre"] = x
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            x = F.leaky_relu(x, LRELU_SLOPE)
            x = F.conv_transpose1d(q(x), q(w[f"ups.{i}.weight"]), w[f"ups.{i}.bias"],
                                   stride=u, padding=(k - u) // 2)
            if taps is not None:
                taps[f"ups.{i}"] = x
            xs = None
            for j, (kr, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
                r = _resblock(w, f"resblocks.{i * nk + j}", x, kr, dil, q)
                xs = r if xs is None else xs + r
            x = xs / nk
            if taps is not None:
                taps[f"stage.{i}"] = x
        x = F.leaky_relu(x)  # default slope 0.01 (models.py:161)
        x = F.conv1d(q(x), q(w["conv_post.weight"]), w["conv_post.bias"], padding=3)
        x = torch.tanh(x)
    return x[0] if squeeze else x


def _resblock(w, p, x, k, dil, q):
    for m, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(q(xt), q(w[f"{p}.convs1.{m}.weight"]), w[f"{p}.convs1.{m}.bias"],
                      dilation=d, padding=(k * d - d) // 2)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(q(xt), q(w[f"{p}.convs2.{m}.weight"]), w[f"{p}.convs2.{m}.bias"],
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def _gate(a, b, C):
    s = a + b
    return torch.tanh(s[:, :C]) * torch.sigmoid(s[:, C:])


def _wn(w, p, audio0, spect, cfg, q):
    wn = cfg["WN_config"]
    C, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    audio = F.conv1d(audio0, w[p + "start.weight"], w[p + "start.bias"])
    output = torch.zeros_like(audio)
    cond = F.conv1d(q(spect), q(w[p + "cond_layer.weight"]), w[p + "cond_layer.bias"])
    for i in range(nl):
        d = 2 ** i
        a = F.conv1d(q(audio), q(w[p + f"in_layers.{i}.weight"]), w[p + f"in_layers.{i}.bias"],
                     dilation=d, padding=(ks * d - d) // 2)
        acts = _gate(a, cond[:, i * 2 * C:(i + 1) * 2 * C], C)
        rs = F.conv1d(q(acts), q(w[p + f"res_skip_layers.{i}.weight"]), w[p + f"res_skip_layers.{i}.bias"])
        if i < nl - 1:
            audio = audio + rs[:, :C]
            output = output + rs[:, C:]
        else:
            output = output + rs
    return F.conv1d(output, w[p + "end.weight"], w[p + "end.bias"])


def waveglow_upsample(state, cfg, mel):
    """glow.py:252-260: ConvTranspose1d k1024 s256, trim 768, regroup to [B, 80*n_group, L]."""
    w = {k: _t(v) for k, v in state.items() if k.startswith("upsample.")}
    x = _t(mel)
    ng = cfg["n_group"]
    with torch.no_grad():
        s = F.conv_transpose1d(x, w["upsample.weight"], w["upsample.bias"], stride=256)
        s = s[:, :, :-(1024 - 256)]
        s = s.unfold(2, ng, ng).permute(0, 2, 1, 3)
        s = s.contiguous().view(s.size(0), s.size(1), -1).permute(0, 2, 1)
    return s


def waveglow_infer(state, cfg, mel, noise, sigma: float = 1.0, bf16_operands: bool = False, taps=None):
    """mel [B,80,T] -> [B,256T]. `noise` = unit-variance tensors in the reference's draw
    order (glow.py:262-271, 287-292): [B,n_rem,L], then [B,n_early_size,L] per early flow."""
    q = _Q(bf16_operands)
    w = {k: _t(v) for k, v in state.items()}
    noise = [_t(n) for n in noise]
    with torch.no_grad():
        spect = waveglow_upsample(state, cfg, mel)
        if taps is not None:
            taps["spect"] = spect
        audio = sigma * noise[0]
        ni = 1
        for k in reversed(range(cfg["n_flows"])):
            n_half = audio.size(1) // 2
            a0, a1 = audio[:, :n_half], audio[:, n_half:]
            out = _wn(w, f"WN.{k}.", a0, spect, cfg, q)
            s, b = out[:, n_half:], out[:, :n_half]
            a1 = (a1 - b) / torch.exp(s)
            audio = torch.cat([a0, a1], 1)
            W = w[f"convinv.{k}.conv.weight"].squeeze(-1)
            Winv = W.float().inverse()[..., None]  # glow.py:91-92
            audio = F.conv1d(audio, Winv)
            if taps is not None:
                taps[f"flow.{k}"] = audio
            if k % cfg["n_early_every"] == 0 and k > 0:
                audio = torch.cat((sigma * noise[ni], audio), 1)
                ni += 1
        audio = audio.permute(0, 2, 1).contiguous().view(audio.size(0), -1)
    return audio


def pcm16(wav) -> np.ndarray:
    """do_syn.py:11,389-394: 32768*y -> host -> astype('int16') (truncate toward zero, no clip)."""
    a = (32768.0 * _t(wav)).cpu().numpy()
    return a.astype("int16")


def postnet_forward(state, hps, x, add_input: bool = False, bf16_operands: bool = False, i_decoder: int = 0,
                    eps: float = 1e-5):
    """x [B,dim,T] or [dim,T] -> same shape (model.py:170-174): tanh(BN(conv(x))) n-1 times, then BN(conv(x));
    BatchNorm1d in eval mode (running statistics), dropout inactive; add_input: model.py:717."""
    q = _Q(bf16_operands)
    w = {k: _t(v) for k, v in state.items()}
    x0 = _t(x)
    squeeze = x0.dim() == 2
    if squeeze:
        x0 = x0[None]
    n = hps["postnet_n_convolutions"][i_decoder]
    k = hps["postnet_kernel_size"][i_decoder]
    y = x0
    with torch.no_grad():
        for i in range(n):
            pre = f"convolutions.{i}"
            if bf16_operands:
                # the library folds BatchNorm into the conv before rounding the weights to bf16
                s_ = w[pre + ".1.weight"] / torch.sqrt(w[pre + ".1.running_var"] + eps)
                wf = w[pre + ".0.conv.weight"] * s_[:, None, None]
                bf = (w[pre + ".0.conv.bias"] - w[pre + ".1.running_mean"]) * s_ + w[pre + ".1.bias"]
                y = F.conv1d(q(y), q(wf), bf, padding=(k - 1) // 2)
            else:
                y = F.conv1d(y, w[pre + ".0.conv.weight"], w[pre + ".0.conv.bias"], padding=(k - 1) // 2)
                y = F.batch_norm(y, w[pre + ".1.running_mean"], w[pre + ".1.running_var"], w[pre + ".1.weight"],
                                 w[pre + ".1.bias"], training=False, eps=eps)
            if i < n - 1:
                y = torch.tanh(y)
        if add_input:
            y = x0 + y
    return y[0] if squeeze else y
