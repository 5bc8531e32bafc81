#!/bin/bash
# fused-path check on the GPU box: fused-vs-operator test, [model goldens], kernel trace of the bench.  Logs -> gpurun_out/$1
tag=${1:-fused}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_z_fused_selfcheck_gpu.py -m gpu -q -rA --timeout 900 > $out/fused.log 2>&1
echo "fused exit $?"; grep -E "passed|failed|worst|Error" $out/fused.log | tail -12
if [ "$2" == "model" ]; then
  timeout 1500 python -m pytest tests/test_c_model_gpu.py tests/test_d_engine_gpu.py -m gpu -q --timeout 900 > $out/model.log 2>&1
  echo "model exit $?"; tail -8 $out/model.log
fi
bash tools/gpu_trace.sh $tag
