#!/bin/bash
out=gpurun_out/r06pj
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests/test_c_model_gpu.py tests/test_d_engine_gpu.py tests/test_z_fused_selfcheck_gpu.py -x -q -m gpu > $out/pytest_model.txt 2>&1 || { tail -40 $out/pytest_model.txt; exit 1; }
tail -3 $out/pytest_model.txt
bash tools/gpu_ab_env.sh UD_PROJECT_BWD_FUSED_WIDE 0 1 > $out/step_ab_wide.txt 2>&1
cat $out/step_ab_wide.txt
