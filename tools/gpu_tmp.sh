#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=t14
python -m pytest tests/test_b_fused_kernels_gpu.py tests/test_a_kernels_gpu.py -q -m gpu -k "fft or half" 2>&1 | tail -2
python tools/bench_fft2p.py --half 2>&1 | grep -E "^ 16|^  8|storage"
for r in 1 2; do
  echo "f16 : $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype f16 --batch 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms %.1f img/s'%(d['ms_per_step'],d['value']))")"
  echo "f16 old lib: $(UD_LIB_PATH=$PWD/unidefense_amd/libud_exp_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype f16 --batch 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms %.1f img/s'%(d['ms_per_step'],d['value']))")"
done
