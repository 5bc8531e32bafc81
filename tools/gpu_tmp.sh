#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/sk
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/sk/plans32.json python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/sk/plans32.json python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
python bench.py --no-cpu-baseline --dtype f16 --batch 64 2>/dev/null | tail -1 | cut -c1-260
UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/sk/plans64h.json python bench.py --no-cpu-baseline --dtype f16 --batch 64 2>/dev/null | tail -1 | cut -c1-260
UD_GEMM_TUNE_DEFAULTS=0 UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/sk/plans64h.json python bench.py --no-cpu-baseline --dtype f16 --batch 64 2>/dev/null | tail -1 | cut -c1-260
