bash tools/gpu_suite.sh s1 all
echo "== bench deterministic (default)"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s1/bench_det.json 2> gpurun_out/s1/bench_det.err; tail -1 gpurun_out/s1/bench_det.json | cut -c1-200
echo "== bench atomics"
UD_DETERMINISTIC=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s1/bench_atom.json 2> gpurun_out/s1/bench_atom.err; tail -1 gpurun_out/s1/bench_atom.json | cut -c1-200
echo "== bench deterministic again"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s1/bench_det2.json 2> gpurun_out/s1/bench_det2.err; tail -1 gpurun_out/s1/bench_det2.json | cut -c1-200
