#!/bin/bash
# A/B of environment settings inside ONE gpurun call (boxes differ by a few percent):
#   tools/gpu_ab.sh "UD_FUSED_MBCONV=0" "UD_FUSED_MBCONV=1" ...   -> ms_per_step of the default bench per setting, twice
mkdir -p gpurun_out/ab
export PYTHONDONTWRITEBYTECODE=1
for round in 1 2; do
  for setting in "$@"; do
    r=$(env $setting python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.1f img/s  gemm %.2f ms' % (d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step']))")
    echo "[$round] $setting : $r"
  done
done
