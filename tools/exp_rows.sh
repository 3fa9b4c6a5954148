#!/bin/bash
# per-kernel rows of two precisions / option sets, back to back on one box: tools/exp_rows.sh "<bench args A>" "<bench args B>"
mkdir -p gpurun_out/rows
i=0
for a in "$@" "$@"; do
  i=$((i+1))
  timeout -k 10 200 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --rows $a > gpurun_out/rows/b$i.json 2> gpurun_out/rows/r$i.txt || exit 1
  echo "== $a: $(python -c "import json;d=json.load(open('gpurun_out/rows/b$i.json'));print(d['value'],d['ms_per_step'])")"
  grep "^#" gpurun_out/rows/r$i.txt | sed -e 's/void mv:://' -e 's/(mv::[A-Za-z]*)//' | awk '{printf "%-62s %-14s %s %s %s %s\n", $2" "$3" "$4" "$5" "$6" "$7" "$8" "$9, "", $(NF-5), $(NF-3), $(NF-2), $NF}' | cut -c1-140
done
