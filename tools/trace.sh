#!/bin/bash
# Diagnostic build with per-block phase timers (fused_rb.h MV_TRACE) -> diag/libmelvoc_trace.so.
# On the GPU box: MELVOC_LIB=diag/libmelvoc_trace.so python3 tools/trace_report.py
set -e
cd "$(dirname "$0")/../tacotron2_amd/csrc"
mkdir -p ../../diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 \
   -DMV_TRACE=1 -shared -o ../../diag/libmelvoc_trace.so layers.hip small_kernels.hip hifigan.hip waveglow.hip postnet.hip api.hip
