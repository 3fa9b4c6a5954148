#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the raw PMC passes of tools/profile_round.sh.

usage: tools/traffic_json.py <tag> <workload> <fetch counter_collection.csv> <write counter_collection.csv>

HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB / launches, FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950 (wide reads count half; Infinity-Cache hits are included).  Keys are the kernels' symbols as
rocprofv3 prints them -- what bench.py's `roofline.kernel` carries -- averaged over ALL launches of the symbol, like
`roofline.achieved`; `_workload` ("model/dtype/B<batch>") says which bench run the passes were taken on.
"""
import collections
import csv
import json
import sys


def per_symbol(path, counter):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    out = collections.defaultdict(list)
    for k in sorted(rows):
        out[rows[k][0]].append(rows[k][1])
    return out


def main():
    tag, workload, fetch_csv, write_csv = sys.argv[1:5]
    fetch, write = per_symbol(fetch_csv, "FETCH_SIZE"), per_symbol(write_csv, "WRITE_SIZE")
    out = {
        "_note": f"HBM-side bytes per launch from the PMC passes of profiles/{tag}_* (tools/traffic_json.py): "
        "(2*FETCH_SIZE + WRITE_SIZE) KB averaged over all launches of each kernel symbol (FETCH_SIZE doubled per "
        "MI355X_MICROARCH.md, the gfx950 wide-read correction; Infinity-Cache hits are included).",
        "_workload": workload,
    }
    for name in fetch:
        if "mv::" not in name or name not in write or len(write[name]) != len(fetch[name]):
            continue
        n = len(fetch[name])
        kb = (2 * sum(fetch[name]) + sum(write[name])) / n
        out[name] = {"bytes_per_launch": int(kb * 1024), "dispatches": n}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
