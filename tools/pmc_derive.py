#!/usr/bin/env python3
"""Per-kernel clock, MFMA-pipe utilisation and hardware FLOP rate from one rocprofv3 --pmc pass that
holds SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16 (or _F16 for MV_PREC_F16 runs) and GRBM_GUI_ACTIVE.
  clock     = GRBM_GUI_ACTIVE / 8 XCDs / kernel time          (MI355X_MICROARCH.md, DVFS give-back)
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
  hw FLOP/s = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / kernel time (executed, halo recompute included)
usage: pmc_derive.py <counter_collection.csv>"""
import collections
import csv
import re
import sys


def main(path):
    dur = collections.defaultdict(float)
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mv::", "")[:72]
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    print("kernel,dispatches,total_ms,clock_GHz,mfma_busy_pct,hw_TFLOPs")
    n = collections.Counter(k for k, _ in seen)
    for k in sorted(dur, key=lambda k: -dur[k]):
        c, d = cnt[k], dur[k]
        g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
        if g <= 0 or d <= 0:
            continue
        print("%s,%d,%.3f,%.2f,%.1f,%.0f" % (k, n[k], d * 1e3, g / d / 1e9,
                                             100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * g),
                                             (c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) +
                                              c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)) * 512 / d / 1e12))


if __name__ == "__main__":
    main(sys.argv[1])
