#!/bin/bash
# timing-only ablation libraries against the real one, same box: per-kernel rows of the fused kernels
mkdir -p gpurun_out/ab
for f in ${FUSES:-2 3}; do
for l in tacotron2_amd/libmelvoc.so $(ls diag/libmelvoc_ab*.so); do
  tag=$(basename $l .so)_f$f
  MELVOC_LIB=$l timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --rows --options "{\"fuse\": $f}" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.txt || exit 1
  echo "== $tag $(python -c "import json;d=json.load(open('gpurun_out/ab/$tag.json'));print(d['value'],d['ms_per_step'])")"
  grep -E "fused_(ip|rb)_kernel<(128|256)" gpurun_out/ab/$tag.txt | sort | awk '{print $3,$4,$5,$6,$7,$8,$9, $(NF-5), $(NF-2), $NF}' | cut -c1-150
done; done
