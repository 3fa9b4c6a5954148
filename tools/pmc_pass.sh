#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>   (run on the GPU box from the repo root; one pass per call)
# Writes gpurun_out/pmc_<tag>/ and prints the per-kernel sums.
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out" -o pmc --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/log.txt" 2>&1
cd "$root"
python3 tools/pmc_summary.py $(find "$out" -name '*counter_collection.csv') > "$out/summary.csv"
cut -c1-400 "$out/summary.csv" | head -12
