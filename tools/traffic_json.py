#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from a tools/profile_round.sh PMC summary.

usage: tools/traffic_json.py gpurun_out/<tag>/pmc_summary.csv <tag> [passes=3] > profiles/<tag>_traffic.json

HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB / launches, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 (wide reads count half); the summary adds the dispatches of all PMC
passes, each counter comes from one of them.  Keys are the library's own kernel names (what bench.py's
`roofline.kernel` carries), so bench.py can attach the measured traffic to the dominant kernel.
"""
import json
import re
import sys


def lib_name(rocprof_name):
    m = re.match(r"fused_rb_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)", rocprof_name)
    if m:
        c, wm, wn, tn, xreg = m.groups()
        return f"fused_rb_kernel<C{c},WM{wm},WN{wn},TN{tn},{'xreg' if xreg == 'true' else 'pair'}>"
    m = re.match(r"fused_stage_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>", rocprof_name)
    if m:
        c, wm, wn, tn, _, post = m.groups()
        return f"fused_stage_kernel<C{c},WM{wm},WN{wn},TN{tn}>" + ("+conv_post" if post == "true" else "")
    return None


def main():
    path, tag = sys.argv[1], sys.argv[2]
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lines = open(path).read().strip().split("\n")
    cols = lines[0].split(",")
    n = len(cols) - 1
    out = {
        "_note": f"HBM-side bytes per launch from profiles/{tag}_pmc_summary.csv: (2*FETCH_SIZE + WRITE_SIZE) KB / "
        "launches (FETCH_SIZE doubled per MI355X_MICROARCH.md, the gfx950 wide-read correction; Infinity-Cache hits "
        "are included). Measured on bench.py's default batch; bench.py reports it for the dominant kernel when the "
        "name matches."
    }
    for ln in lines[1:]:
        parts = ln.rsplit(",", n)  # kernel names contain commas
        row = dict(zip(cols[1:], parts[1:]))
        name = lib_name(parts[0])
        if not name:
            continue
        launches = int(row["dispatches"]) // passes
        kb = 2 * float(row["FETCH_SIZE"]) + float(row["WRITE_SIZE"])
        out[name] = {"bytes_per_launch": int(kb * 1024 / launches), "dispatches": launches}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
