#!/bin/bash
# Timing-only diagnostic builds of libmelvoc (outputs are WRONG by construction; never shipped).
# MV_ABLATE is a bit mask: 1 = no MFMA issued, 2 = no B-fragment LDS reads, 4 = no weight (A) loads,
# 8 = no K loop at all, 16 = half the B reads, 128 = conflict-free (wrong) fragment addresses in the epilogues, 256 = the weight stream re-reads two L1-resident fragments, 512 = one weight load per K step.  usage: tools/ablate.sh [mask ...]   (default: 1 3 7 8 2 4 6)
# Run bench.py with MELVOC_LIB=diag/libmelvoc_ab<mask>.so MELVOC_BENCH_ROWS=1 and compare per-kernel times.
set -e
cd "$(dirname "$0")/../tacotron2_amd/csrc"
mkdir -p ../../diag
masks="$@"
[ -z "$masks" ] && masks="1 3 7 8 2 4 6"
for m in $masks; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 \
     -DMV_ABLATE=$m -shared -o ../../diag/libmelvoc_ab$m.so layers.hip small_kernels.hip hifigan.hip waveglow.hip gate_gemm.hip postnet.hip api.hip &
done
wait
