#!/bin/bash
out=gpurun_out/r06pj
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
for v in 0 1; do
  UD_PROJECT_BWD_FUSED=$v timeout -k 10 600 python -m pytest tests/test_c_model_gpu.py -q -m gpu -s -k "n8" > $out/n8_fused$v.txt 2>&1
  echo "== UD_PROJECT_BWD_FUSED=$v"; grep -n "sf_coef\|passed\|failed\|outside the plain\|worst" $out/n8_fused$v.txt | head -40
done
