"""Deterministic synthetic weights, mels and noise for the vocoder hot path.

No vocoder checkpoint ships with the reference, so parity
fixtures, smoke() and bench.py all run on weights that any machine can
regenerate bit-for-bit from a seed: numpy ``default_rng(seed)``, tensors
visited in sorted-key order, ``std = gain / sqrt(fan_in)``.
The reference's own initialisers would make parity trivially true
(std=0.01 convs, zero `end` layers: hifigan/models.py:10-13,
waveglow/glow.py:127-130), hence the fan-in scaling and non-zero biases.

Shapes follow the reference modules:
  HiFi-GAN  hifigan/models.py:113-147 (Conv1d [Cout,Cin,K]; ConvTranspose1d [Cin,Cout,K])
  WaveGlow  waveglow/glow.py:111-151,179-205
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

HIFIGAN_V1 = {
    "upsample_rates": [8, 8, 2, 2],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_mels": 80,
    "sampling_rate": 22050,
}

WAVEGLOW_512 = {
    "n_mel_channels": 80,
    "n_flows": 12,
    "n_group": 8,
    "n_early_every": 4,
    "n_early_size": 2,
    "WN_config": {"n_layers": 8, "n_channels": 512, "kernel_size": 3},
}

MEL_PAD_VALUE = -12.59  # log(1e-5/10)+1.23 floor of do_process_files.py:94-95


def hifigan_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Folded (weight-norm removed) tensor shapes of Generator (models.py:113-147)."""
    c0 = int(cfg["upsample_initial_channel"])
    shapes = OrderedDict()
    shapes["conv_pre.weight"] = (c0, int(cfg.get("num_mels", 80)), 7)
    shapes["conv_pre.bias"] = (c0,)
    ch = c0
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        shapes[f"ups.{i}.weight"] = (cin, cout, int(k))
        shapes[f"ups.{i}.bias"] = (cout,)
        ch = cout
        for j, k_rb in enumerate(cfg["resblock_kernel_sizes"]):
            n = i * len(cfg["resblock_kernel_sizes"]) + j
            for m in range(3):
                for which in ("convs1", "convs2"):
                    shapes[f"resblocks.{n}.{which}.{m}.weight"] = (cout, cout, int(k_rb))
                    shapes[f"resblocks.{n}.{which}.{m}.bias"] = (cout,)
    shapes["conv_post.weight"] = (1, ch, 7)
    shapes["conv_post.bias"] = (1,)
    return shapes


def _hifigan_gain(name: str, shape, cfg) -> float:
    if name.endswith(".bias"):
        return 0.0
    if name.startswith("conv_pre"):
        return 1.0 / 3.0  # mel values are O(5); bring conv_pre output to O(1)
    if name.startswith("ups."):
        return math.sqrt(2.0)
    if ".convs1." in name:
        return 1.0
    if ".convs2." in name:
        return 0.7
    if name.startswith("conv_post"):
        return 0.6
    return 1.0


def hifigan_state(cfg=HIFIGAN_V1, seed: int = 0, weight_norm: bool = False):
    """Synthetic HiFi-GAN generator state dict as float32 numpy arrays.

    weight_norm=True emits the checkpoint form (``weight_g``/``weight_v``,
    234 keys for config_v1) whose fold ``g*v/||v||`` over dims (1,2) per dim-0
    index reproduces the folded tensors of weight_norm=False.
    """
    shapes = hifigan_shapes(cfg)
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name in sorted(shapes):
        shape = shapes[name]
        if name.endswith(".bias"):
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
            continue
        if name.startswith("ups."):
            cin, cout, k = shape
            stride = cfg["upsample_rates"][int(name.split(".")[1])]
            fan_in = cin * k / stride
        else:
            cout, cin, k = shape
            fan_in = cin * k
        std = _hifigan_gain(name, shape, cfg) / math.sqrt(fan_in)
        out[name] = (std * rng.standard_normal(shape)).astype(np.float32)
    if not weight_norm:
        return out
    wn = OrderedDict()
    rng2 = np.random.default_rng(seed + 7919)
    for name, w in out.items():
        if not name.endswith(".weight"):
            wn[name] = w
            continue
        base = name[: -len("weight")]
        scale = rng2.uniform(0.5, 2.0, size=(w.shape[0], 1, 1)).astype(np.float32)
        norm = np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
        wn[base + "weight_g"] = norm
        wn[base + "weight_v"] = (w * scale).astype(np.float32)
    return wn


def fold_weight_norm(state) -> "OrderedDict[str, np.ndarray]":
    """w = g * v / ||v||, norm over all dims but 0 (torch weight_norm dim=0)."""
    out = OrderedDict()
    for name, t in state.items():
        if name.endswith("weight_g"):
            base = name[: -len("weight_g")]
            v = np.asarray(state[base + "weight_v"], dtype=np.float32)
            g = np.asarray(t, dtype=np.float32)
            axes = tuple(range(1, v.ndim))
            norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=axes, keepdims=True)).astype(np.float32)
            out[base + "weight"] = (v * (g / norm)).astype(np.float32)
        elif name.endswith("weight_v"):
            continue
        else:
            out[name] = np.asarray(t, dtype=np.float32)
    return out


def waveglow_flow_channels(cfg):
    """(n_remaining_channels, n_half) per flow k (glow.py:194-205)."""
    n_half = cfg["n_group"] // 2
    n_rem = cfg["n_group"]
    out = []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def waveglow_shapes(cfg) -> "OrderedDict[str, tuple]":
    nm = cfg["n_mel_channels"]
    wn = cfg["WN_config"]
    C, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    shapes = OrderedDict()
    shapes["upsample.weight"] = (nm, nm, 1024)
    shapes["upsample.bias"] = (nm,)
    for k, (n_rem, n_half) in enumerate(waveglow_flow_channels(cfg)):
        shapes[f"convinv.{k}.conv.weight"] = (n_rem, n_rem, 1)
        p = f"WN.{k}."
        shapes[p + "start.weight"] = (C, n_half, 1)
        shapes[p + "start.bias"] = (C,)
        shapes[p + "end.weight"] = (2 * n_half, C, 1)
        shapes[p + "end.bias"] = (2 * n_half,)
        shapes[p + "cond_layer.weight"] = (2 * C * nl, nm * cfg["n_group"], 1)
        shapes[p + "cond_layer.bias"] = (2 * C * nl,)
        for i in range(nl):
            shapes[p + f"in_layers.{i}.weight"] = (2 * C, C, ks)
            shapes[p + f"in_layers.{i}.bias"] = (2 * C,)
            rs = 2 * C if i < nl - 1 else C
            shapes[p + f"res_skip_layers.{i}.weight"] = (rs, C, 1)
            shapes[p + f"res_skip_layers.{i}.bias"] = (rs,)
    return shapes


def waveglow_state(cfg=WAVEGLOW_512, seed: int = 0):
    """Synthetic WaveGlow state dict (folded form), float32 numpy arrays.

    `end` layers are non-zero (a fresh reference model zero-initialises them,
    glow.py:127-130, which would make every coupling the identity) but small,
    so exp(s) stays O(1) through the 12 inverse flows.
    """
    shapes = waveglow_shapes(cfg)
    rng = np.random.default_rng(seed)
    nl = cfg["WN_config"]["n_layers"]
    out = OrderedDict()
    for name in sorted(shapes):
        shape = shapes[name]
        if name.startswith("convinv."):
            c = shape[0]
            q, _ = np.linalg.qr(rng.standard_normal((c, c)))
            w = q + 0.1 * rng.standard_normal((c, c))
            out[name] = w.reshape(c, c, 1).astype(np.float32)
            continue
        if name.endswith(".bias"):
            s = 0.05 if ".end." in name else 0.1
            out[name] = (s * rng.standard_normal(shape)).astype(np.float32)
            continue
        if name == "upsample.weight":
            cin, cout, k = shape
            std = 1.0 / math.sqrt(cin * k / 256)
        else:
            cout, cin, k = shape
            gain = 1.0
            if ".start." in name:
                gain = 1.0
            elif ".cond_layer." in name:
                gain = 0.2  # upsampled mel is O(5)
            elif ".res_skip_layers." in name:
                gain = 1.5
            elif ".end." in name:
                gain = 0.25 / math.sqrt(nl)
            std = gain / math.sqrt(cin * k)
        out[name] = (std * rng.standard_normal(shape)).astype(np.float32)
    return out


def synthetic_mel(T: int, seed: int = 1234, n_mels: int = 80) -> np.ndarray:
    """[n_mels, T] float32, clip(N(-5, 2^2), -12.59, 2.5): the value range of real
    mels written by do_process_files.py:94-95."""
    rng = np.random.default_rng(seed)
    m = rng.normal(-5.0, 2.0, size=(n_mels, T))
    return np.clip(m, MEL_PAD_VALUE, 2.5).astype(np.float32)


def synthetic_lengths(B: int, seed: int = 1234, lo: int = 86, hi: int = 430) -> np.ndarray:
    """Config 2 lengths: T_i = 2*U{lo..hi} mel frames, i.e. 2-10 s."""
    rng = np.random.default_rng(seed)
    return (2 * rng.integers(lo, hi + 1, size=B)).astype(np.int32)


def waveglow_noise(cfg, B: int, T: int, seed: int = 1234):
    """The noise tensors of WaveGlow.infer in draw order (glow.py:262-271,287-292):
    [B, n_remaining, L] first, then one [B, n_early_size, L] per early-output flow
    (k = 8, then k = 4 for the 12-flow model), L = T*256/n_group. Unit variance;
    sigma is applied by the consumer."""
    L = T * 256 // cfg["n_group"]
    rng = np.random.default_rng(seed)
    n_rem = waveglow_flow_channels(cfg)[-1][0]
    out = [rng.standard_normal((B, n_rem, L)).astype(np.float32)]
    for k in reversed(range(cfg["n_flows"])):
        if k % cfg["n_early_every"] == 0 and k > 0:
            out.append(rng.standard_normal((B, cfg["n_early_size"], L)).astype(np.float32))
    return out


# ---- Tacotron2 Postnet (model.py:140-174; tc2_french.yaml:15,66,87-89) ---------------------------
POSTNET_TC2 = {"dim_data": [80], "postnet_embedding_dim": [512], "postnet_kernel_size": [5],
               "postnet_n_convolutions": [5], "p_postnet_dropout": [0.5]}


def postnet_state(hps=POSTNET_TC2, seed: int = 0, i_decoder: int = 0):
    """Deterministic Postnet state_dict (the reference's key names): fan-in-scaled conv weights, non-trivial
    BatchNorm affine parameters and running statistics so that the eval-mode folding is exercised."""
    rng = np.random.default_rng(seed)
    dim, emb = hps["dim_data"][i_decoder], hps["postnet_embedding_dim"][i_decoder]
    k, n = hps["postnet_kernel_size"][i_decoder], hps["postnet_n_convolutions"][i_decoder]
    out = OrderedDict()
    for i in range(n):
        cin, cout = (dim if i == 0 else emb), (dim if i == n - 1 else emb)
        gain = 1.0 if i == n - 1 else 5.0 / 3.0
        out[f"convolutions.{i}.0.conv.weight"] = (gain / math.sqrt(cin * k) *
                                                  rng.standard_normal((cout, cin, k))).astype(np.float32)
        out[f"convolutions.{i}.0.conv.bias"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        out[f"convolutions.{i}.1.weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        out[f"convolutions.{i}.1.bias"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        out[f"convolutions.{i}.1.running_mean"] = (0.2 * rng.standard_normal(cout)).astype(np.float32)
        out[f"convolutions.{i}.1.running_var"] = rng.uniform(0.5, 2.0, cout).astype(np.float32)
    return out
