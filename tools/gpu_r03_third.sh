export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=s3 UD_MARGIN_DIR=$PWD/gpurun_out/margins
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_y_atomics_mode_gpu.py tests/test_d_optim_gpu.py "tests/test_f_dp2_gpu.py::test_two_ranks_equal_one_process_full_batch" tests/test_a_kernels_gpu.py::test_gemm_nt_nn_tn -q -m gpu -rA --timeout 900 > gpurun_out/s3/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|worst:" gpurun_out/s3/pytest.log | tail -25
for i in 1 2; do
echo "== bench plain"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
echo "== bench xcd-contiguous"; timeout 600 python tools/run_with.py kernels._XCD_CONTIGUOUS=True -- bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
done
