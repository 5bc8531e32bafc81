export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=dw2 UD_MARGIN_DIR=$PWD/gpurun_out/margins
mkdir -p gpurun_out/dw2
timeout 900 python -m pytest tests/test_z_fused_selfcheck_gpu.py tests/test_b_fused_kernels_gpu.py::test_tiled_depthwise_kernels tests/test_e_mixed_precision_gpu.py "tests/test_c_model_gpu.py::test_train_fwd_bwd_vs_reference_golden" tests/test_d_optim_gpu.py -q -m gpu --timeout 600 > gpurun_out/dw2/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/dw2/pytest.log | tail -15
for i in 1 2; do
echo "== f32 strip"; timeout 600 python tools/run_with.py tape._DW_TILED=False -- bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
echo "== f32 tiled"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
done
echo "== f16 bs64 strip"; timeout 600 python tools/run_with.py tape._DW_TILED=False -- bench.py --dtype f16 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-190
echo "== f16 bs64 tiled"; timeout 600 python bench.py --dtype f16 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-190
