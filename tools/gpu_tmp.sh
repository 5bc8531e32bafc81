export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=t6 UD_MARGIN_DIR=$PWD/gpurun_out/margins
timeout 900 python -m pytest tests/test_e_mixed_precision_gpu.py::test_half_step_backward_scales_exactly -q -m gpu -rA --timeout 600 2>&1 | grep -E "passed|failed|^E |worst:|median" | head -20
