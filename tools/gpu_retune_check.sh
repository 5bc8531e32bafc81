#!/bin/bash
# do freshly measured GEMM plans beat the shipped ones with the current kernel?  (first run of each pair tunes, second measures)
# usage: tools/gpu_retune_check.sh            -> plan files in gpurun_out/retune/
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/retune
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms %.1f img/s gemm %.2f' % (d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step']))"; }
for r in 1 2; do
echo "UDR18 shipped: $(b --model UDR18 --size 128 --batch 8)"
echo "UDR18 fresh  : $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/r18.json b --model UDR18 --size 128 --batch 8)"
echo "UDR50 shipped: $(b --model UDR50 --size 320 --batch 16)"
echo "UDR50 fresh  : $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/r50.json b --model UDR50 --size 320 --batch 16)"
done
echo "train step shipped: $(python tools/bench_train_step.py 2>/dev/null | tail -1)"
echo "train step fresh  : $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/train.json python tools/bench_train_step.py 2>/dev/null | tail -1)"
echo "train step fresh 2: $(UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/retune/train.json python tools/bench_train_step.py 2>/dev/null | tail -1)"
