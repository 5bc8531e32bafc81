#!/bin/bash
# Time the model's GEMM shapes with experiment builds of the split-bf16 kernel (unidefense_amd/libud_exp_<tag>.so,
# `make -C unidefense_amd/csrc exp_<tag> EXPFLAGS=...`) next to the product library, inside one gpurun call.
export PYTHONDONTWRITEBYTECODE=1
for lib in "" "$@"; do
  if [ -z "$lib" ]; then echo "== product library"; unset UD_LIB_PATH; else echo "== $lib"; export UD_LIB_PATH=$PWD/unidefense_amd/libud_exp_$lib.so; fi
  _UD_WORKER=1 python tools/bench_gemm.py 2>/dev/null | awk '{printf "%s %s %s %s  %s ms %s TF\n", $3,$4,$5,$6,$7,$9}' | head -12
done
