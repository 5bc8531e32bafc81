#!/bin/bash
bash tools/gpu_lib_ab.sh base
export UD_BENCH_ARGS="--dtype f16 --batch 64"
bash tools/gpu_lib_ab.sh base
