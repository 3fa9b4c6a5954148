#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: sum of each counter, dispatches.
usage: pmc_summary.py <counter_collection.csv> [...]"""
import csv
import collections
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void mv::", "").replace("mv::", "")[:72]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add((p, r["Dispatch_Id"]))
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel,dispatches," + ",".join(counters))
    for k in sorted(agg, key=lambda k: -max(agg[k].values())):
        print(",".join([k, str(len(disp[k]))] + ["%.6g" % agg[k].get(c, 0.0) for c in counters]))


if __name__ == "__main__":
    main(sys.argv[1:])
