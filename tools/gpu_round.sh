#!/bin/bash
# Round-end evidence on the GPU box (via gpurun): full GPU test suite, default bench line, rocprofv3 kernel stats of
# the same bench command, per-shape GEMM table.  Everything lands in gpurun_out/$1 (default r02); copy the
# summaries to profiles/ afterwards.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
(rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; echo "host cores: $(nproc)"; free -g | head -2) > $out/env.log 2>&1
if [ "$2" != "nopytest" ]; then
  echo "== pytest -m gpu"
  timeout 2700 python -m pytest tests -m gpu -q -rA --timeout 1200 > $out/pytest_gpu.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|error" $out/pytest_gpu.log | tail -3
fi
export UD_GEMM_TUNE_CACHE=$PWD/$out/gemm_plans.json      # the profiled runs below repeat the benchmarked plans
echo "== bench (default flags)"
timeout 1200 python bench.py --gemm-table $out/gemm_table.txt > $out/bench.json 2> $out/bench.err
echo "bench exit $?"; tail -1 $out/bench.json | cut -c1-400
echo "== rocprofv3 kernel stats (STEPS executed: 2 eager warm-ups + 3 + 10 replays + 3 eager instrumented = 18)"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
find $out/prof -name "*kernel_trace.csv" -delete
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv && python3 tools/roofline_from_rocprof.py $out/kernel_stats.csv $out/gemm_table.txt 18 $out/roofline_replayed.json | tee $out/roofline_from_rocprof.txt
echo "== informational configs"
timeout 600 python bench.py --dtype f16 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_f16_bs64.json
timeout 600 python bench.py --model UDR50 --size 320 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_udr50_320.json
timeout 600 python bench.py --model UDR18 --size 128 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_udr18_128.json
timeout 300 python tools/bench_train_step.py > $out/train_step.log 2>&1
timeout 300 python tools/bench_optim.py > $out/optim.log 2>&1
timeout 300 python tools/bench_fft_planes.py > $out/fft_planes.log 2>&1
for f in bench_f16_bs64 bench_udr50_320 bench_udr18_128; do python3 -c "import json;d=json.load(open('$out/$f.json'));print('$f', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"; done
tail -3 $out/train_step.log; tail -3 $out/optim.log
