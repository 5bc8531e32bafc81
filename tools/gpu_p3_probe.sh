#!/bin/bash
# p3 kernel: PMC passes on two shapes
mkdir -p gpurun_out/p3probe
export UD_ONE_GEMM_P3=1
bash tools/gpu_pmc.sh 4096 4096 4096 nn p3probe/pmc4096 > gpurun_out/p3probe/pmc4096.txt 2>&1
bash tools/gpu_pmc.sh 4352 1920 1920 nn p3probe/pmc4352 > gpurun_out/p3probe/pmc4352.txt 2>&1
tail -32 gpurun_out/p3probe/pmc4096.txt
