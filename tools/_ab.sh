#!/bin/bash
# A/B helper: for each value V of env var $1 run the bf16 tile bench and print step ms and conv27 totals
VAR=$1; shift
for V in "$@"; do
  export $VAR=$V
  timeout -k 10 200 python bench.py --tile --dtype bf16 --no-cpu-baseline 2>/dev/null > /tmp/ab.json || exit 1
  python - <<PY
import json
d=json.load(open('/tmp/ab.json')); r=d['roofline']
print("$VAR=$V", d['ms_per_step'], round(r['avg_launch_ms']*r['launches_per_step'],2), r['executed_mfma_tflops'])
PY
done
