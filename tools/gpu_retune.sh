#!/bin/bash
# re-measure the GEMM plans of the BASELINE configs from scratch (shipped defaults ignored); the cache is merged into the shipped file
# with tools/merge_plans.py.  usage: tools/gpu_retune.sh <cache name>
mkdir -p gpurun_out
export UD_GEMM_TUNE_DEFAULTS=0
export UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/${1:-plans_retune}.json
common="--steps 5 --warmup 2 --no-cpu-baseline --no-extra"
for args in "" "--dtype f16 --batch 64" "--model UDR50 --size 320 --batch 16" "--model UDR18 --size 128 --batch 8" "--train-step"; do
  t0=$(date +%s)
  python bench.py $args $common 2>/dev/null | tail -1 | cut -c1-160
  echo "  [$args] $(( $(date +%s) - t0 )) s"
done
python3 -c "import json;print(len(json.load(open('$UD_GEMM_TUNE_CACHE'))), 'plans')"
