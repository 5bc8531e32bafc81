#!/bin/bash
# round 6, second session: the one-pass expand-conv backward (csrc/pwbwd.hip) — operator test, micro-bench, model / engine goldens, step A/B
out=gpurun_out/r06pw
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "expand_conv_backward" > $out/pytest_op.txt 2>&1 || { tail -30 $out/pytest_op.txt; exit 1; }
tail -3 $out/pytest_op.txt
timeout -k 10 200 python tools/bench_pwbwd.py 0 > $out/bench_pwbwd.txt 2>&1 || { tail -20 $out/bench_pwbwd.txt; exit 1; }
cat $out/bench_pwbwd.txt
timeout -k 10 900 python -m pytest tests/test_c_model_gpu.py tests/test_d_engine_gpu.py tests/test_z_fused_selfcheck_gpu.py tests/test_f_dp2_gpu.py -x -q -m gpu > $out/pytest_model.txt 2>&1 || { tail -40 $out/pytest_model.txt; exit 1; }
tail -3 $out/pytest_model.txt
bash tools/gpu_ab_env.sh UD_EXPAND_BWD_FUSED 0 1 > $out/step_ab.txt 2>&1
cat $out/step_ab.txt
