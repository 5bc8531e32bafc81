export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_b_fused_kernels_gpu.py -k "tiled_depthwise" -q -m gpu -x --timeout 600 2>&1 | grep -E "passed|failed|^E " | head
timeout 600 python tools/bench_dwtile.py --stride2 2>&1 | tail -6
timeout 600 python tools/bench_dwtile.py --stride2 --half 2>&1 | tail -6
