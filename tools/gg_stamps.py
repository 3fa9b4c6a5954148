#!/usr/bin/env python3
"""Block-phase timing of gate_gemm_kernel from a diagnostic build (-DMV_GG_STAMP, loaded through MELVOC_LIB):
shader-clock stamps at block start / prologue done / K loop done / gate tile in LDS / stored, of the LAST launch.
usage (GPU box): MELVOC_LIB=abtest/libmelvoc_stamp.so python tools/gg_stamps.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from tacotron2_amd import _native  # noqa: E402


def main():
    _m, run, *_ = bench.build_model(torch, "waveglow", "bf16", int(sys.argv[1]) if len(sys.argv) > 1 else 0, 0,
                                     torch.device("cuda:0"))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), np.uint64)
    rc = _native.lib().mv_debug_gate_gemm_stamps(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    st = buf[buf[:, 0] > 0].astype(np.int64)
    # (the buffer holds the LAST launch of either kernel: the final flow's tail follows its last gated conv, so this is
    # gate_gemm_kernel; build with -DMV_GG_STAMP_RES to stamp res_gemm_kernel instead)
    d = np.diff(st[:, :5], axis=1)
    names = ["prologue", "K loop", "gate -> LDS", "store"]
    print("blocks", len(st))
    for i, n in enumerate(names):
        print("%-12s median %7.0f  p10 %7.0f  p90 %7.0f cycles" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    tot = st[:, 4] - st[:, 0]
    print("block total  median %7.0f  max %7.0f cycles" % (np.median(tot), tot.max()))
    real = (st[:, 5] - st[:, 6]) * 10.0  # ns (the constant 100 MHz clock)
    print("block wall   median %7.0f  max %7.0f ns -> %.2f GHz shader clock" % (np.median(real), real.max(), np.median(tot) / np.median(real)))
    print("launch wall  %.0f ns (first block start -> last block end)" % ((st[:, 5].max() - st[:, 6].min()) * 10.0))
    print("first start -> last end: %.0f cycles (shader clocks are per XCD: indicative only)" % (st[:, 4].max() - st[:, 0].min()))


if __name__ == "__main__":
    main()
