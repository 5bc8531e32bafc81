#!/bin/bash
out=gpurun_out/r06pj
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "project_conv or expand_conv" > $out/pytest_op.txt 2>&1 || { tail -40 $out/pytest_op.txt; exit 1; }
tail -2 $out/pytest_op.txt
timeout -k 10 900 python -m pytest tests/test_c_model_gpu.py tests/test_d_engine_gpu.py tests/test_z_fused_selfcheck_gpu.py -x -q -m gpu > $out/pytest_model.txt 2>&1 || { tail -40 $out/pytest_model.txt; exit 1; }
tail -2 $out/pytest_model.txt
bash tools/gpu_ab_env.sh UD_PROJECT_FUSED_NARROW 0 1 > $out/step_ab_narrow.txt 2>&1
cat $out/step_ab_narrow.txt
