#!/bin/bash
out=gpurun_out/r06pj
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "project_conv or expand_conv_backward" > $out/pytest_op.txt 2>&1 || { tail -40 $out/pytest_op.txt; exit 1; }
tail -3 $out/pytest_op.txt
timeout -k 10 200 python tools/bench_pjbwd.py > $out/bench_pjbwd.txt 2>&1 || { tail -20 $out/bench_pjbwd.txt; exit 1; }
cat $out/bench_pjbwd.txt
