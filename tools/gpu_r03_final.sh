#!/bin/bash
# Round-3 evidence in one gpurun call (after the full-suite runs of tools/gpu_suite.sh): default bench line incl. cpu_baseline,
# rocprofv3 kernel stats of the same command, per-shape GEMM table, informational configs, f16 bs-64 kernel stats, PMC traffic.
bash tools/gpu_round.sh r03 nopytest
echo "== f16 bs 64 kernel stats"
bash tools/gpu_prof.sh r03_f16 --dtype f16 --batch 64 2>&1 | tail -4
echo "== PMC traffic passes (fp32 bs 32)"
UD_GEMM_TUNE_CACHE=$PWD/gpurun_out/r03/gemm_plans.json bash tools/gpu_traffic.sh r03_traffic 2>&1 | tail -6
python3 tools/hbm_bw_table.py gpurun_out/r03_traffic > gpurun_out/r03_traffic/hbm_bw_by_kernel.txt 2>/dev/null; head -12 gpurun_out/r03_traffic/hbm_bw_by_kernel.txt
find gpurun_out/r03_traffic -name "*counter_collection.csv" -size +20M -delete
