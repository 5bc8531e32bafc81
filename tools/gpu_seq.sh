#!/bin/bash
# ordered launch list of the replayed bs-32 step (tools/step_sequence.py) into gpurun_out/$1
out=gpurun_out/${1:-seq}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py $f $out/step_sequence.txt 8 split_h2_multi_kernel
rm -rf $out/prof
head -1 $out/step_sequence.txt
