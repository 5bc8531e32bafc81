#!/bin/bash
# round-5 first call: baseline of the tree (suite subset, bench, replayed-step launch sequence, aten census, event probe)
out=gpurun_out/r05a
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 120 python3 tools/probe_graph_events.py > $out/graph_events.txt 2>&1; echo "probe exit $?"
timeout 300 python3 tools/aten_census.py > $out/aten_census.txt 2>&1; echo "census exit $?"
timeout 900 python3 -m pytest tests/test_f_dist_gpu.py tests/test_d_train_engine.py tests/test_d_engine_gpu.py -x -q -m gpu > $out/pytest_subset.txt 2>&1; echo "pytest exit $?"
tail -3 $out/pytest_subset.txt
timeout 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench exit $?"
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_trace.log 2>&1
echo "rocprof exit $?"
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py $f $out/step_sequence.txt 8
ms=$(python3 -c "import json;print([json.loads(l) for l in open('$out/bench_trace.log') if l.startswith('{')][-1]['ms_per_step'])")
python3 tools/trace_overlap.py $f 10 $ms > $out/overlap.txt
rm -rf $out/prof
head -c 600 $out/bench_default.json; echo
cat $out/graph_events.txt
