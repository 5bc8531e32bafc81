#!/bin/bash
out=gpurun_out/r05x
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "kernels._P3_PAIR_TN_SCALE=1.0" "kernels._P3_PAIR_TN_SCALE=1.5" "kernels._P3_PAIR_TN_SCALE=2.0" "kernels._P3_PAIR_TN_SCALE=3.0"; do
      timeout 300 python3 tools/run_with.py $v -- bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'img/s')"
  done
done | tee $out/ab.txt
