#!/bin/bash
# Run on the GPU box via gpurun: operator tests, model tests, short bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== rocminfo" > gpurun_out/env.log
(rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; nproc; free -g | head -2) >> gpurun_out/env.log 2>&1
echo "== kernels"
timeout 1500 python -m pytest tests/test_a_kernels_gpu.py -m gpu -n 3 -rA -q --timeout 600 > gpurun_out/kernels.log 2>&1
echo "kernels exit $?"
grep -E "passed|failed|error" gpurun_out/kernels.log | tail -3
echo "== model"
timeout 1500 python -m pytest tests/test_c_model_gpu.py -m gpu -rA -q --timeout 900 > gpurun_out/model.log 2>&1
echo "model exit $?"
grep -E "passed|failed|error" gpurun_out/model.log | tail -3
if [ "$1" == "bench" ]; then
  echo "== bench"
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
  echo "bench exit $?"; tail -2 gpurun_out/bench.log
fi
