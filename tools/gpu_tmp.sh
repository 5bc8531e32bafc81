export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=t8 UD_MARGIN_DIR=$PWD/gpurun_out/margins
timeout 1200 python -m pytest tests/test_b_fused_kernels_gpu.py tests/test_e_mixed_precision_gpu.py "tests/test_a_kernels_gpu.py::test_gemm_nt_nn_tn" "tests/test_a_kernels_gpu.py::test_gemm_every_tile_configuration_and_split" tests/test_d_train_engine.py -q -m gpu --timeout 900 2>&1 | grep -E "passed|failed|^E |^FAILED" | head
for i in 1 2; do
timeout 600 python bench.py --dtype f16 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c120-260
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
