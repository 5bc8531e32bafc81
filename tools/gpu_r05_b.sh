#!/bin/bash
# round-5 call b: the fused depthwise backward — kernel test, microbenchmark against the separate kernels, step A/B
out=gpurun_out/r05b
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "tiled_depthwise or se_" > $out/pytest_dw.txt 2>&1; echo "pytest exit $?"
tail -5 $out/pytest_dw.txt
timeout 300 python3 tools/bench_dwbwd.py > $out/dwbwd_f32_bs32.txt 2>&1; echo "bench_dwbwd exit $?"
cat $out/dwbwd_f32_bs32.txt
timeout 300 python3 tools/bench_dwbwd.py --half > $out/dwbwd_f16_bs64.txt 2>&1
cat $out/dwbwd_f16_bs64.txt
for i in 1 2; do
  for v in False True; do
    timeout 300 python3 tools/run_with.py tape._DW_BWD_FUSED=$v -- bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fused=$v', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'img/s', d['config']['final_loss'])"
  done
done | tee $out/ab_step.txt
