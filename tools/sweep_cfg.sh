#!/bin/bash
# Per-layer tile-configuration sweep of HiFi-GAN's conv_cl_kernel layers (conv_pre, the four upsamplers), all with
# kernels the product library already ships: tools/sweep_cfg.sh build   (here: diag/libmelvoc_sweep.so)
#                                            tools/sweep_cfg.sh run     (GPU box: ms/step of those layers per setting)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    mkdir -p diag
    (cd tacotron2_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
        -mllvm -amdgpu-mfma-vgpr-form=1 -DMV_SWEEP_CFG=1 -shared -o ../../diag/libmelvoc_sweep.so \
        layers.hip small_kernels.hip hifigan.hip waveglow.hip gate_gemm.hip postnet.hip api.hip)
    ls -la diag/libmelvoc_sweep.so
    exit 0
fi
mkdir -p gpurun_out/sweep
one() {  # setting -> "value  ms/step  [conv_cl rows]"
    MELVOC_LIB=diag/libmelvoc_sweep.so MV_SWEEP_CFG=$1 timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 \
        --no-cpu-baseline --no-extras --rows > gpurun_out/sweep/b.json 2> gpurun_out/sweep/r.txt || return 1
    echo "$1 $(python -c "import json;d=json.load(open('gpurun_out/sweep/b.json'));print(d['value'],d['ms_per_step'])") :: $(grep conv_cl gpurun_out/sweep/r.txt | awk '{print $4 $5 $6 $7, $(NF-2)}' | tr '\n' ' ')"
}
one -1,-1,-1,-1,-1
for layer in 0 1 2 3 4; do
  for cfg in 0 1 2 3 4 5 6 7 8; do
    s="-1,-1,-1,-1,-1"
    s=$(echo $s | awk -F, -v l=$layer -v c=$cfg 'BEGIN{OFS=","}{$(l+1)=c; print}')
    one $s || echo "$s unsupported"
  done
done
one -1,-1,-1,-1,-1
