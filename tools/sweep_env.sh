#!/bin/bash
# usage (GPU box): tools/sweep_env.sh "VAR=1 VAR2=3" "VAR=2" ...   -- one bench run per environment set, per-kernel rows
cd "$(dirname "$0")/.."
for e in "" "$@"; do
    echo "=== [$e]"
    env $e MELVOC_BENCH_ROWS=1 timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>&1 | grep -E "^#|value" | cut -c1-130 || exit 1
done
