#!/bin/bash
# rocprofv3 kernel trace of a short bench run; summary -> gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps ${1:-3} --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_stdout.log 2>&1
echo "rocprof exit $?"
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"; head -40 "$f"
# drop the big per-dispatch trace, keep stats
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
