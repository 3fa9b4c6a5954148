"""The C oracle under AddressSanitizer + UBSan (host code only): the golden vectors and the edge cases of the GPU
tests (one frame, ragged lengths, every conv shape class) run against `oracle/libvocoder_oracle_asan.so` in a child
process with the sanitizer runtime preloaded.  Round 1's `convT_core` wrote before its buffer at T = 1 and showed up
as a corrupted heap inside a GPU test; here such an access is a one-line report and a failing CPU test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import copy, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import c_oracle
from tacotron2_amd import synth
assert c_oracle._SO.endswith("_asan.so")
gd = os.path.join(sys.argv[1], "tests", "golden")
cfg = copy.deepcopy(synth.HIFIGAN_V1); cfg["upsample_initial_channel"] = 64
d = np.load(os.path.join(gd, "hifigan_c64_T48.npz"))
st = synth.hifigan_state(cfg, int(d["weight_seed"]))
y = c_oracle.hifigan_forward(st, cfg, synth.synthetic_mel(48, int(d["mel_seed"])))
assert np.abs(y - d["wav"]).max() <= 1e-4
for T in (1, 2, 3):  # the shortest sequences: every window, halo and polyphase edge at once
    for bf in (False, True):
        y = c_oracle.hifigan_forward(st, cfg, synth.synthetic_mel(T, 9), bf16_operands=bf)
        assert y.shape == (T * 256,) and np.isfinite(y).all()
rng = np.random.default_rng(0)
for cin, cout, k, dil, L in ((32, 32, 11, 5, 7), (64, 32, 3, 1, 1), (32, 64, 7, 3, 40)):
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = rng.standard_normal((cout, cin, k)).astype(np.float32)
    c_oracle.conv1d(x, w, None, dil, dil * (k - 1) // 2)
for cin, cout, k, u, p, L in ((32, 16, 16, 8, 4, 1), (32, 16, 4, 2, 1, 1), (16, 16, 16, 8, 4, 9)):
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = rng.standard_normal((cin, cout, k)).astype(np.float32)
    y = c_oracle.conv_transpose1d(x, w, np.zeros(cout, np.float32), u, p)
    assert y.shape == (cout, L * u)
wcfg = copy.deepcopy(synth.WAVEGLOW_512); wcfg["WN_config"]["n_channels"] = 64
d = np.load(os.path.join(gd, "waveglow_c64_T16.npz"))
wst = synth.waveglow_state(wcfg, int(d["weight_seed"]))
B, T = int(d["B"]), int(d["T"])
noise = synth.waveglow_noise(wcfg, B, T, int(d["noise_seed"]))
for b in range(B):
    y = c_oracle.waveglow_infer(wst, wcfg, synth.synthetic_mel(T, int(d["mel_seed"]) + b), [n[b] for n in noise], float(d["sigma"]))
    assert np.abs(y - d["wav"][b]).max() <= 1e-4
y = c_oracle.waveglow_infer(wst, wcfg, synth.synthetic_mel(1, 3), [n[0][:, :32] for n in noise], 0.666)
assert y.shape == (256,)
p = c_oracle.postnet_forward(synth.postnet_state(synth.POSTNET_TC2, 0), synth.POSTNET_TC2, synth.synthetic_mel(1, 2), add_input=True)
assert p.shape == (80, 1)
print("ASAN-ORACLE-OK")
"""


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libvocoder_oracle_asan.so")
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no libasan runtime next to gcc")
    env = dict(os.environ, ORACLE_SO=so, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the sanitizer runtime has to be in the process before the interpreter starts; anything already preloaded stays
    env["LD_PRELOAD"] = ":".join([rt] + [p for p in os.environ.get("LD_PRELOAD", "").split(":") if p])
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ASAN-ORACLE-OK" in p.stdout, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
