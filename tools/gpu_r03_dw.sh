export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=dw UD_MARGIN_DIR=$PWD/gpurun_out/margins
mkdir -p gpurun_out/dw
timeout 900 python -m pytest tests/test_b_fused_kernels_gpu.py::test_tiled_depthwise_kernels tests/test_d_optim_gpu.py -q -m gpu -x --timeout 600 > gpurun_out/dw/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/dw/pytest.log | tail -15
timeout 600 python tools/bench_dwtile.py > gpurun_out/dw/bench_f32.log 2>&1; tail -12 gpurun_out/dw/bench_f32.log
timeout 600 python tools/bench_dwtile.py --half > gpurun_out/dw/bench_f16.log 2>&1; tail -12 gpurun_out/dw/bench_f16.log
