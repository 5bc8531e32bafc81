#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
export UD_BENCH_ARGS=""
bash tools/gpu_lib_ab.sh base
export UD_BENCH_ARGS="--model UDR18 --size 128 --batch 8"
bash tools/gpu_lib_ab.sh base
export UD_BENCH_ARGS="--model UDR50 --size 320 --batch 16"
bash tools/gpu_lib_ab.sh base
export UD_BENCH_ARGS="--dtype f16 --batch 64"
bash tools/gpu_lib_ab.sh base
