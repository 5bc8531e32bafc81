#!/bin/bash
# A/B of experiment builds of the split-bf16 GEMM (unidefense_amd/libud_exp_<tag>.so, `make -C unidefense_amd/csrc exp_<tag>
# EXPFLAGS=...`) against the product library inside ONE gpurun call: the default bench step, twice per library.
#   tools/gpu_lib_ab.sh pd333 pd468
export PYTHONDONTWRITEBYTECODE=1
for round in 1 2; do
  for lib in "" "$@"; do
    if [ -z "$lib" ]; then unset UD_LIB_PATH; name=product; else export UD_LIB_PATH=$PWD/unidefense_amd/libud_exp_$lib.so; name=$lib; fi
    r=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline $UD_BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.1f img/s  gemm %.2f ms' % (d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step']))")
    echo "[$round] $name : $r"
  done
done
