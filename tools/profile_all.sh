#!/bin/bash
# usage (GPU box, repo root): tools/profile_all.sh <round tag, e.g. r03_a>
# every record of the driver's bench line gets its rocprofv3 kernel stats + PMC passes (tools/profile_round.sh);
# stops at the first failing step.
set -e
t=$1
run() {  # tag workload bench-args [mfma-counter]
    echo "== $1"
    BENCH_ARGS="$3" WORKLOAD="$2" MFMA_MOPS="${4:-SQ_INSTS_VALU_MFMA_MOPS_BF16}" timeout -k 10 900 tools/profile_round.sh "$1" > "gpurun_out/$1.log" 2>&1 || { tail -20 "gpurun_out/$1.log"; exit 1; }
    tail -12 "gpurun_out/$1.log" | cut -c1-150
}
mkdir -p gpurun_out
run ${t}_hifigan_bf16 hifigan/bf16/B32 ""
run ${t}_hifigan_f16 hifigan/f16/B32 "--dtype f16" SQ_INSTS_VALU_MFMA_MOPS_F16
run ${t}_hifigan_fp32 hifigan/fp32/B32 "--dtype fp32"
run ${t}_hifigan_bf16x3 hifigan/bf16x3/B32 "--dtype bf16x3"
run ${t}_waveglow_bf16 waveglow/bf16/B1 "--model waveglow"
run ${t}_waveglow_bf16_B8 waveglow/bf16/B8 "--model waveglow --batch 8"
run ${t}_waveglow_fp32 waveglow/fp32/B1 "--model waveglow --dtype fp32"
run ${t}_waveglow_bf16x3 waveglow/bf16x3/B1 "--model waveglow --dtype bf16x3"
