#!/bin/bash
# round-6 closing evidence, second call: PMC traffic passes (stamped with the GEMM sources' hash), per-kernel bandwidth table + the
# per-CLASS table as JSON (stamped with the hash of csrc/: bench.py prints it as roofline.classes), PMC counters of the matrix-pipe
# kernels inside the bench step, the replayed step's timeline and ordered launch lists.  Everything lands in gpurun_out/r06f2.
out=gpurun_out/r06f2
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
BENCH_ARGS=--no-extra bash tools/gpu_traffic.sh r06_traffic          # (6 eager steps per pass: 1 + 1 warm-ups, 2 timed, 2 instrumented)
python3 tools/hbm_bw_table.py gpurun_out/r06_traffic --json $out/hbm_classes.json 6 > $out/hbm_bw_by_kernel.txt 2>&1
cp gpurun_out/r06_traffic/hbm_traffic_gemm.json $out/
find gpurun_out/r06_traffic -name "*.csv" -delete
bash tools/gpu_pmc_bench.sh r06_pmc > $out/gemm_pmc_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
seq() {   # tag, marker kernel, bench args...
  tag=$1; marker=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$tag -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra "$@" > $out/bench_$tag.log 2>&1
  f=$(find $out/prof_$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/step_sequence.py $f $out/step_sequence_$tag.txt 8 $marker
  if [ "$tag" == "udeb4_256" ]; then
    ms=$(python3 -c "import json;print([json.loads(l) for l in open('$out/bench_$tag.log') if l.startswith('{')][-1]['ms_per_step'])")
    python3 tools/trace_overlap.py $f 10 $ms > $out/timeline.txt
  fi
  rm -rf $out/prof_$tag
  head -1 $out/step_sequence_$tag.txt
}
seq udeb4_256 split_h2_multi_kernel
seq udr50_320 split_h2_multi_kernel --model UDR50 --size 320 --batch 16
seq f16_bs64 split_h2_multi_kernel --dtype f16 --batch 64
head -6 $out/timeline.txt
tail -25 $out/gemm_pmc_bench.txt
