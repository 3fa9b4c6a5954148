#!/bin/bash
# Timing-only diagnostic builds of libmelvoc (outputs are WRONG by construction; never shipped):
#   diag/libmelvoc_ab1.so  no MFMA issued      diag/libmelvoc_ab2.so  no MFMA, no B-fragment LDS reads
# Run bench.py with MELVOC_LIB=<path> MELVOC_BENCH_ROWS=1 and compare per-kernel times with the real build.
set -e
cd "$(dirname "$0")/../tacotron2_amd/csrc"
mkdir -p ../../diag
for lvl in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 \
     -DMV_ABLATE=$lvl -shared -o ../../diag/libmelvoc_ab$lvl.so layers.hip small_kernels.hip hifigan.hip waveglow.hip api.hip
done
