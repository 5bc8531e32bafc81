#!/bin/bash
# A/B of one module-level knob (tools/run_with.py) inside ONE gpurun call: tools/gpu_ab_knob.sh kernels._NORM_FUSED False True [bench args]
# three rounds each, interleaved; prints img/s and ms per step of the bench line
knob=$1; a=$2; b=$3; shift 3
mkdir -p gpurun_out/ab_knob
export PYTHONDONTWRITEBYTECODE=1
for r in 1 2 3; do
  for v in "$a" "$b"; do
    python tools/run_with.py $knob=$v -- bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 > gpurun_out/ab_knob/cur.json
    python3 -c "import json;d=json.load(open('gpurun_out/ab_knob/cur.json'));print('$knob=$v', $r, round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
  done
done
