#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=t13
python -m pytest tests/test_b_fused_kernels_gpu.py -q -m gpu -k "fft or half" 2>&1 | tail -2
python tools/bench_fft2p.py 2>&1 | grep -v amdgpu.ids
python tools/bench_fft2p.py --half 2>&1 | grep -v amdgpu.ids
