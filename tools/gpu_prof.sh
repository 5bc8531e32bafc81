#!/bin/bash
# bench + rocprofv3 kernel stats of the same command (+ per-shape GEMM table).  usage: tools/gpu_prof.sh <tag> [bench args]
tag=${1:-p1}; shift
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
export UD_GEMM_TUNE_CACHE=$PWD/$out/gemm_plans.json
timeout 1200 python bench.py --gemm-table $out/gemm_table.txt --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
echo "bench exit $?"; tail -1 $out/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
find $out/prof -name "*kernel_trace.csv" -delete
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv && python3 tools/roofline_from_rocprof.py $out/kernel_stats.csv $out/gemm_table.txt 18
rm -rf $out/prof
