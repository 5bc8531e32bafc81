export PYTHONDONTWRITEBYTECODE=1
python3 -m pytest tests/test_b_fused_kernels_gpu.py -x -q -m gpu -k "project_planes" 2>&1 | tail -5
python3 -m pytest tests/test_c_model_gpu.py tests/test_z_fused_selfcheck_gpu.py -x -q -m gpu -k "golden or elementwise or fused_mbconv_equals" 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('all planes direct', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['grad_l1'])"
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05f/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r05f_bench_trace.log 2>&1
f=$(find gpurun_out/r05f/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py $f gpurun_out/r05f/step_sequence.txt 8
rm -rf gpurun_out/r05f/prof
tail -1 gpurun_out/r05f/step_sequence.txt; head -1 gpurun_out/r05f/step_sequence.txt
