#!/bin/bash
# kernel timeline of the bench (hipGraph replays): union / overlap / idle accounting.  usage: tools/gpu_trace.sh <tag> [bench args]
tag=${1:-tr}; shift
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra "$@" > $out/bench.log 2>&1
echo "rocprof exit $?"
ms=$(python3 -c "import json;print([json.loads(l) for l in open('$out/bench.log') if l.startswith('{')][-1]['ms_per_step'])")
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
head -2 $f | cut -c1-400
python3 tools/trace_overlap.py $f 10 $ms | tee $out/overlap.txt
rm -rf $out/prof
