# A/B of an environment switch inside one gpurun call:  bash tools/ab_env.sh VAR=off VAR=on
for i in 1 2; do
for kv in "$@"; do
env $kv timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$kv', round(d['value'],1), round(d['ms_per_step'],3), 'gemm TF', round(d['roofline']['achieved'],1))"
done
done
