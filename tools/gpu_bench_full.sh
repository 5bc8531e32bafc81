#!/bin/bash
# the default bench command the driver runs (extras + cpu baseline), timed; plans measured on the way are kept
mkdir -p gpurun_out
export UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/plans_r04.json
t0=$(date +%s)
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench exit $? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")})
for e in d.get("extra",[]): print(e)
r=d["roofline"]; print({k:r[k] for k in ("achieved","peak","frac","launches_per_step","traffic")}); print(r["kernel"][:40]); print(r["families"])
print(d.get("cpu_baseline"))
PY
