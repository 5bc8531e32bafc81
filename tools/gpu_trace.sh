#!/bin/bash
# Per-dispatch kernel trace of a few graph-replayed bench steps (rocprofv3 --kernel-trace): tools/trace_timeline.py
# turns one replay into a timeline (start, gap, duration, workgroups, kernel).  Output: gpurun_out/$1/
tag=${1:-trace}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $out/bench.log 2>&1
echo "rocprof exit $?"; tail -2 $out/bench.log | cut -c1-400
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && mv "$f" $out/kernel_trace.csv && gzip -f $out/kernel_trace.csv
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv
ls -la $out
