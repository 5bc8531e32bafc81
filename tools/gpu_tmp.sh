export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=t7 UD_MARGIN_DIR=$PWD/gpurun_out/margins
timeout 900 python -m pytest tests/test_b_fused_kernels_gpu.py::test_tiled_depthwise_kernels tests/test_z_fused_selfcheck_gpu.py -q -m gpu --timeout 600 2>&1 | grep -E "passed|failed|^E " | head
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
done
