export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=s2 UD_MARGIN_DIR=$PWD/gpurun_out/margins
mkdir -p gpurun_out/s2
timeout 1500 python -m pytest tests/test_y_fullsize_gpu.py tests/test_y_atomics_mode_gpu.py tests/test_d_optim_gpu.py tests/test_c_perturb.py tests/test_f_dp2_gpu.py tests/test_d_train_engine.py tests/test_z_fused_selfcheck_gpu.py -q -m gpu -rA --timeout 900 > gpurun_out/s2/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/s2/pytest.log | tail -15
timeout 600 python tools/bench_pair.py > gpurun_out/s2/pair.log 2>&1; cat gpurun_out/s2/pair.log | tail -20
