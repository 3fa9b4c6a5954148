#!/bin/bash
# A/B of the working tree against the last commit, because box-to-box variance (3-5 %) exceeds most single changes:
#   tools/ab.sh build      (here: compiles HEAD's csrc into diag/libmelvoc_prev.so)
#   tools/ab.sh run [N]    (on the GPU box: alternates N rounds of bench.py between the two libraries)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    rm -rf /tmp/ab_prev && mkdir -p /tmp/ab_prev/include diag && git archive HEAD tacotron2_amd/csrc include | tar -x -C /tmp/ab_prev
    (cd /tmp/ab_prev/tacotron2_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
        -mllvm -amdgpu-mfma-vgpr-form=1 -shared -o "$OLDPWD/diag/libmelvoc_prev.so" $(ls *.hip))
    ls -la diag/libmelvoc_prev.so
else
    n=${2:-3}
    for i in $(seq $n); do
        for l in diag/libmelvoc_prev.so tacotron2_amd/libmelvoc.so; do
            echo -n "$l "
            MELVOC_LIB=$l timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile $BENCH_ARGS 2>&1 | tail -1 | cut -c75-92
        done
    done
fi
