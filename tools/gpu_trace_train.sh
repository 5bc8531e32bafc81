#!/bin/bash
# kernel trace of the engine's two-pass train step (tools/bench_train_step.py) -> gpurun_out/$1/kernel_trace.csv.gz
tag=${1:-trace_train}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o ts -- python3 tools/bench_train_step.py > $out/run.log 2>&1
echo "rocprof exit $?"; tail -1 $out/run.log
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && mv "$f" $out/kernel_trace.csv && gzip -f $out/kernel_trace.csv
