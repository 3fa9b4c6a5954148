"""The library's tuning knobs select other kernels for the same arithmetic (layer-by-layer instead of fused, float32
instead of float16 / bfloat16 storage, the unfolded WaveGlow skip path ...).  They are read once per process, so
each setting runs the bf16 parity check against the golden vectors in a process of its own; the bar is the same as
for the defaults (SNR >= 40 dB against the fp32 reference, BASELINE.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import copy, json, os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tacotron2_amd import synth
from tacotron2_amd.hifigan import Generator, AttrDict
from tacotron2_amd.waveglow import WaveGlow

def snr(ref, x):
    ref = np.asarray(ref, np.float64)
    return float(10 * np.log10((ref ** 2).sum() / max(((ref - np.asarray(x, np.float64)) ** 2).sum(), 1e-300)))

gd = os.path.join(sys.argv[1], "tests", "golden")
out = {}
d = np.load(os.path.join(gd, "hifigan_v1_T64.npz"))
g = Generator(AttrDict(synth.HIFIGAN_V1), precision="bf16")
g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
g.eval().cuda()
mel = torch.from_numpy(synth.synthetic_mel(64, int(d["mel_seed"]))).cuda()
out["hifigan"] = snr(d["wav"], g(mel[None])[0, 0].cpu().numpy())
# ragged batch: every row equals its own batch-1 result to well below the bf16 error
lens = [64, 37, 5]
x = torch.full((3, 80, 64), synth.MEL_PAD_VALUE)
for b, n in enumerate(lens):
    x[b, :, :n] = torch.from_numpy(synth.synthetic_mel(n, 40 + b))
y = g(x.cuda(), lengths=lens)[:, 0].cpu().numpy()
solo = g(x[1:2, :, :37].contiguous().cuda())[0, 0].cpu().numpy()
out["ragged"] = float(np.abs(solo - y[1, :37 * 256]).max())
out["tail_zero"] = bool(not y[1, 37 * 256:].any() and not y[2, 5 * 256:].any())

for c, name in ((512, "waveglow_c512_T16"), (64, "waveglow_c64_T16")):
    d = np.load(os.path.join(gd, name + ".npz"))
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = c
    B, T = int(d["B"]), int(d["T"])
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(cfg, int(d["weight_seed"])).items()})
    m.cuda().eval()
    m.precision = "bf16"
    mel = torch.from_numpy(np.stack([synth.synthetic_mel(T, int(d["mel_seed"]) + b) for b in range(B)])).cuda()
    noise = [torch.from_numpy(n) for n in synth.waveglow_noise(cfg, B, T, int(d["noise_seed"]))]
    out[name] = snr(d["wav"], m.infer(mel, sigma=float(d["sigma"]), noise=noise).cpu().numpy())
print("RESULT " + json.dumps(out))
"""

KNOBS = [
    {},  # the defaults, through the same child script
    {"MV_FUSE": "0", "MV_STAGE": "0"},  # HiFi-GAN layer by layer
    {"MV_STAGE": "0"},  # fused ResBlock kernels only
    {"MV_IO16": "0"},  # float32 activations between kernels
    {"MV_STAGE_SPLIT_C64": "0", "MV_WHOLE_MIN_PCT": "60"},  # wider fusion: whole ResBlocks, one C64 stage launch
    {"MV_RB_ONE": "0", "MV_EPI_LDS": "0", "MV_XCD_SWIZZLE": "2"},
    {"MV_WG_FOLD": "0"},  # WaveGlow with the reference's skip path
    {"MV_WG_OPERAND_BF16": "0", "MV_WG_FOLD_BATCH": "0"},  # float32 operand tensors, fold per layer
    {"MV_WG_IO16": "1", "MV_WG_CFG": "0"},
]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) or "defaults")
def test_bf16_parity_under_knobs(knobs):
    env = dict(os.environ)
    env.update(knobs)
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["hifigan"] >= 40.0, r
    assert r["waveglow_c512_T16"] >= 40.0 and r["waveglow_c64_T16"] >= 40.0, r
    assert r["ragged"] <= 5e-3 and r["tail_zero"], r
