#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the raw PMC passes of tools/profile_round.sh.

usage: tools/traffic_json.py <tag> <fetch counter_collection.csv> <write counter_collection.csv> > profiles/<tag>_traffic.json

HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB / launches, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 (wide reads count half).  Keys are the library's own kernel names
(what bench.py's `roofline.kernel` carries, e.g. "fused_rb_kernel<C128,WM4,WN2,TN7,xreg> k=11"), so bench.py
can attach the measured traffic to the dominant kernel.

rocprofv3 sees one symbol for all ResBlock kernel sizes of `fused_rb_kernel<C...>`; they are told apart by the
position of the dispatch within a step, which for HiFi-GAN V1 (resblock_kernel_sizes 3/7/11, hifigan.hip
launch order) is fixed: C256 runs 3 + 3 + 3 single-pair launches, C128 one whole k=3 launch, then 3 + 3
single-pair ones (a different symbol: the ONE variant), C64 three k=11 single-pair launches after its stage launch.
"""
import collections
import csv
import json
import re
import sys

# launches per step of each ResBlock kernel size, in launch order, per kernel symbol: (C, single-pair variant)
RB_PATTERN = {
    (256, True): [(3, 3), (7, 3), (11, 3)],
    (128, False): [(3, 1)],
    (128, True): [(7, 3), (11, 3)],
    (64, True): [(11, 3)],
}


def lib_name(rocprof_name, nth):
    """library name of the nth dispatch (0-based, in order) of a rocprof kernel symbol; None = not a fused kernel"""
    m = re.search(r"fused_rb_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>", rocprof_name)
    if m:
        c, wm, wn, tn, _, one = m.groups()
        base = f"fused_rb_kernel<C{c},WM{wm},WN{wn},TN{tn},xreg>"
        pat = RB_PATTERN.get((int(c), one == "true")) or []
        per_step = sum(n for _, n in pat)
        if per_step:
            i = nth % per_step
            for k, n in pat:
                if i < n:
                    return f"{base} k={k}"
                i -= n
        return base
    m = re.search(r"fused_stage_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>", rocprof_name)
    if m:
        c, wm, wn, tn, _, post = m.groups()
        base = f"fused_stage_kernel<C{c},WM{wm},WN{wn},TN{tn}>"
        return base + ("+conv_post" if post == "true" else " first ResBlocks" if c == "64" else "")
    return None


def per_dispatch(path, counter):
    """[(kernel symbol, value)] in dispatch order"""
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def by_name(path, counter):
    seen = collections.Counter()
    out = collections.defaultdict(list)
    for sym, val in per_dispatch(path, counter):
        name = lib_name(sym, seen[sym])
        seen[sym] += 1
        if name:
            out[name].append(val)
    return out


def main():
    tag, fetch_csv, write_csv = sys.argv[1:4]
    fetch, write = by_name(fetch_csv, "FETCH_SIZE"), by_name(write_csv, "WRITE_SIZE")
    out = {
        "_note": f"HBM-side bytes per launch from the PMC passes of profiles/{tag}_* (tools/traffic_json.py): "
        "(2*FETCH_SIZE + WRITE_SIZE) KB averaged over the launches of each kernel (FETCH_SIZE doubled per "
        "MI355X_MICROARCH.md, the gfx950 wide-read correction; Infinity-Cache hits are included). Measured on "
        "bench.py's default batch; bench.py reports it for the dominant kernel when the name matches."
    }
    for name in fetch:
        if name not in write or len(write[name]) != len(fetch[name]):
            continue
        n = len(fetch[name])
        kb = (2 * sum(fetch[name]) + sum(write[name])) / n
        out[name] = {"bytes_per_launch": int(kb * 1024), "dispatches": n}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
