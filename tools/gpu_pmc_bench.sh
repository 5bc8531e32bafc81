#!/bin/bash
# PMC counters of the planes GEMM (pair + plain launches) INSIDE the bs-32 bench step (eager: PMC serialises kernels anyway):
# matrix-pipe busy fraction, clock under load, LDS bank conflicts, MFMA instruction count, L2 hit rate — per kernel family.
# Separate --pmc passes with --kernel-trace only (the pool refuses mixed trace domains).   usage: tools/gpu_pmc_bench.sh <tag>
tag=${1:-pmc_bench}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UD_GEMM_TUNE_CACHE=${UD_GEMM_TUNE_CACHE:-$PWD/unidefense_amd/gemm_plans_gfx950.json}
run() {   # name, counters...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o $n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --eager > $out/$n.log 2>&1
  echo "$n pass exit $?"
}
run p1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS
run p2 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES
run p3 TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
fam = lambda n: ("gemm_p3_pair_kernel" if "gemm_p3_pair_kernel" in n else "gemm_p3_kernel" if "gemm_p3_kernel" in n else
                 "gemm_x3_kernel" if "gemm_x3_kernel" in n else None)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], f)
        if "p1" in f and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# PMC counters of the matrix-pipe kernels summed over the launches of 3 eager bs-32 steps (tools/gpu_pmc_bench.sh);")
print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs: clock = GRBM / 8 / time, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM / 8 * 1024)")
for k in ("gemm_p3_pair_kernel", "gemm_p3_kernel", "gemm_x3_kernel"):
    a = agg[k]
    if not cnt[k]:
        continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    print(f"{k}: {cnt[k]} launches, {dur[k] / 1e3 / cnt[k]:.1f} us average (serialised under PMC)")
    print(f"   clock under load            {cyc / dur[k]:.2f} GHz")
    print(f"   matrix pipe busy            {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} %   (busy x clock = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / dur[k]:.2f} GHz-equivalents)")
    print(f"   SQ_BUSY_CYCLES / GRBM       {a['SQ_BUSY_CYCLES'] / max(a['GRBM_GUI_ACTIVE'], 1):.3f}")
    print(f"   SQ_LDS_BANK_CONFLICT        {a['SQ_LDS_BANK_CONFLICT']:.4g}  (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = {a['SQ_WAIT_INST_LDS'] / max(a['SQ_WAVE_CYCLES'], 1):.4f})")
    print(f"   SQ_INSTS_MFMA               {a['SQ_INSTS_MFMA']:.4g}   VALU per MFMA {a['SQ_INSTS_VALU'] / max(a['SQ_INSTS_MFMA'], 1):.2f}   LDS per MFMA {a['SQ_INSTS_LDS'] / max(a['SQ_INSTS_MFMA'], 1):.2f}")
    print(f"   L2 hit rate                 {100 * a['TCC_HIT_sum'] / max(a['TCC_HIT_sum'] + a['TCC_MISS_sum'], 1):.1f} %")
PY
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
