#!/bin/bash
# A/B of one environment switch on the same box: tools/gpu_ab_env.sh VAR valueA valueB [bench args]; three rounds each, interleaved
var=$1; a=$2; b=$3; shift 3
mkdir -p gpurun_out/ab_env
for r in 1 2 3; do
  for v in "$a" "$b"; do
    env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 > gpurun_out/ab_env/cur.json
    python3 -c "import json;d=json.load(open('gpurun_out/ab_env/cur.json'));print('$var=$v', $r, round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
  done
done
