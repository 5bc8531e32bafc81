#!/bin/bash
# do freshly measured GEMM plans beat the shipped ones with the current kernel?  (first run of each pair tunes, second measures)
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/retune
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms %.1f img/s gemm %.2f' % (d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step']))"; }
for r in 1 2; do
echo "fp32 shipped: $(b)"
echo "fp32 fresh  : $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/p32.json b)"
echo "f16 shipped : $(b --dtype f16 --batch 64)"
echo "f16 fresh   : $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/p64h.json b --dtype f16 --batch 64)"
done
