export PYTHONDONTWRITEBYTECODE=1
python3 -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "adjoint_transform" 2>&1 | tail -5
python3 -m pytest tests/test_c_model_gpu.py -x -q -m gpu -k "golden or elementwise" 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('bwd in irfft2 8+16', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['grad_l1'])"
python3 tools/run_with.py 'kernels._IRFFT_DWBWD_SIZES=(8,)' -- bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('8 only            ', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['grad_l1'])"
done
