#!/bin/bash
set -e
mkdir -p gpurun_out/f16
timeout -k 10 600 python -u -m pytest tests/test_gpu_f16.py -x -q -s --timeout 200 --timeout-method thread > gpurun_out/f16/test.log 2>&1 || { tail -40 gpurun_out/f16/test.log; exit 1; }
grep -E "worst-case|passed|failed" gpurun_out/f16/test.log
for d in bf16 f16 bf16 f16; do
  timeout -k 10 200 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --dtype $d > gpurun_out/f16/bench_$d.json 2>gpurun_out/f16/bench_$d.err
  python -c "import json;d=json.load(open('gpurun_out/f16/bench_$d.json'));print('$d',d['value'],d['ms_per_step'],d['roofline']['frac'])"
done
timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --model waveglow --dtype f16 > gpurun_out/f16/wg_f16.json 2>gpurun_out/f16/wg_f16.err
python -c "import json;d=json.load(open('gpurun_out/f16/wg_f16.json'));print('waveglow f16',d['value'],d['ms_per_step'])"
