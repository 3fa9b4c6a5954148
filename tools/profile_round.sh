#!/bin/bash
# usage (on the GPU box, repo root): tools/profile_round.sh <tag>
# rocprofv3 kernel stats + three PMC passes (SQ / FETCH / WRITE+LDS) of the default bf16 HiFi-GAN bench;
# summaries land in gpurun_out/<tag>/ for copying into profiles/.
set -e
tag=$1
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/stats" -o st --output-format csv -- $BENCH > "$out/stats.log" 2>&1
cp $(find "$out/stats" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats.csv"
pass() {  # name counters...
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out/pmc_$name" -o pmc --output-format csv -- $BENCH > "$out/pmc_$name.log" 2>&1
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ${MFMA_MOPS:-SQ_INSTS_VALU_MFMA_MOPS_BF16} SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
cd "$root"
python3 tools/pmc_summary.py $(find "$out"/pmc_* -name '*counter_collection.csv') > "$out/pmc_summary.csv"
python3 tools/pmc_derive.py $(find "$out/pmc_sq" -name '*counter_collection.csv') > "$out/mfma_util.csv"
python3 tools/traffic_json.py "$tag" "${WORKLOAD:-hifigan/bf16/B32}" $(find "$out/pmc_fetch" -name '*counter_collection.csv') $(find "$out/pmc_write" -name '*counter_collection.csv') > "$out/traffic.json"
head -12 "$out/kernel_stats.csv" | cut -c1-160
cat "$out/mfma_util.csv"
