#!/bin/bash
# Round-end evidence on the GPU box (via gpurun): full GPU test suite, default bench line, rocprofv3 kernel stats of
# the same bench command, per-shape GEMM table.  Everything lands in gpurun_out/$1 (default r01); copy the
# summaries to profiles/ afterwards.
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
(rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; echo "host cores: $(nproc)"; free -g | head -2) > $out/env.log 2>&1
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -rA --timeout 1200 > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $out/pytest_gpu.log | tail -3
echo "== bench (default flags)"
timeout 1200 python bench.py --gemm-table $out/gemm_table.txt > $out/bench.json 2> $out/bench.err
echo "bench exit $?"; tail -1 $out/bench.json
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
find $out/prof -name "*kernel_trace.csv" -delete
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv && head -25 $out/kernel_stats.csv
