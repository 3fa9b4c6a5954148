#!/usr/bin/env python3
"""Per-phase time inside the fused ResBlock / stage kernels, from a MV_TRACE build (tools/trace.sh).
Runs one HiFi-GAN V1 bf16 forward of the bench batch and prints, per kernel variant, the mean shader-clock
cycles a block spends in each phase (thread 0's view) and the share of the block's lifetime."""
import ctypes as C
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacotron2_amd import _native, synth  # noqa: E402
from tacotron2_amd.hifigan import Generator, AttrDict  # noqa: E402

CATS = ["setup", "loadx", "barrier", "kloop", "epi", "store", "post"]


def main():
    dev = torch.device("cuda:0")
    cfg = synth.HIFIGAN_V1
    g = Generator(AttrDict(cfg), precision="bf16")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(cfg, 0).items()})
    g.eval().to(dev)
    lens = np.sort(synth.synthetic_lengths(32, 1234))[::-1].copy()
    T = int(lens.max())
    mel = torch.full((32, 80, T), synth.MEL_PAD_VALUE)
    for i, n in enumerate(lens):
        mel[i, :, :n] = torch.from_numpy(np.asarray(synth.synthetic_mel(int(n), 100 + i)))
    mel = mel.to(dev)
    ln = torch.from_numpy(lens.astype(np.int32)).to(dev)
    lib = _native.lib()
    lib.mv_trace_read.argtypes = [C.c_void_p, C.c_longlong]
    lib.mv_trace_read.restype = C.c_longlong
    for _ in range(2):
        g(mel, lengths=ln)
    torch.cuda.synchronize()
    lib.mv_trace_read(None, 0)  # reset
    g(mel, lengths=ln)
    torch.cuda.synchronize()
    cap = 1 << 19
    buf = np.zeros((cap, 16), dtype=np.uint64)
    n = lib.mv_trace_read(buf.ctypes.data, cap)
    print("records", n)
    rec = buf[:n]
    by = collections.defaultdict(list)
    for tag in np.unique(rec[:, 0]):
        by[int(tag)] = rec[rec[:, 0] == tag]
    print("%-22s %7s %9s | " % ("kernel", "blocks", "cyc/blk") + " ".join("%8s" % c for c in CATS))
    for tag, r in sorted(by.items()):
        Cc, K, npair = tag >> 16, (tag >> 8) & 0xff, tag & 0xff
        name = "C%d %s n%d" % (Cc, {0xff: "stage", 0xfe: "stage+post"}.get(K, "k%d" % K), npair)
        life = (r[:, 3] - r[:, 2]).astype(np.float64)
        cats = r[:, 4:4 + len(CATS)].astype(np.float64)
        print("%-22s %7d %9.0f | " % (name, len(r), life.mean()) +
              " ".join("%7.1f%%" % (100 * cats[:, i].sum() / life.sum()) for i in range(len(CATS))))


if __name__ == "__main__":
    main()
