export PYTHONDONTWRITEBYTECODE=1
python3 -m pytest tests/test_a_gemm_planes_gpu.py tests/test_b_fused_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
python3 -m pytest tests/test_c_model_gpu.py tests/test_z_fused_selfcheck_gpu.py tests/test_d_engine_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('planes from rfft2 (fwd + bwd)', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['grad_l1'])"
python3 tools/run_with.py kernels._RFFT_PLANES=False -- bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('split passes                 ', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['grad_l1'])"
done
