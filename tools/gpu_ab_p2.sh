#!/bin/bash
# A/B in one call (same box): spectral convs on the in-kernel split (off) vs the fp16 x 2 planes path (auto); two rounds each
mkdir -p gpurun_out/ab_p2
export UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/ab_p2/plans.json
for r in 1 2; do
  for m in off auto; do
    UD_SPECTRAL_P2=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_p2/${m}_$r.json
    python3 -c "import json;d=json.load(open('gpurun_out/ab_p2/${m}_$r.json'));print('$m', $r, round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
  done
done
