#!/bin/bash
# PMC counters of pw_bwd_kernel in the micro-bench (form $1, default 1)
form=${1:-1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=r06pw/pmc
mkdir -p gpurun_out/$tag
export PYTHONDONTWRITEBYTECODE=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o p1 -- python3 ${BENCH:-tools/bench_pwbwd.py} $form > gpurun_out/$tag/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/$tag -o p3 -- python3 ${BENCH:-tools/bench_pwbwd.py} $form > gpurun_out/$tag/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$tag -o p4 -- python3 ${BENCH:-tools/bench_pwbwd.py} $form > gpurun_out/$tag/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/$tag/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if ("pw_bwd_kernel" in r["Kernel_Name"] or "pj_bwd_" in r["Kernel_Name"]):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for kn, d in agg.items():
        print(" ", kn)
        for k, v in d.items():
            print("    %-28s n=%d  last=%.5g" % (k, len(v), v[-1]))
for f in sorted(glob.glob("gpurun_out/$tag/*kernel_trace.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        if ("pw_bwd_kernel" in r["Kernel_Name"] or "pj_bwd_" in r["Kernel_Name"]) and r["Kernel_Name"] not in seen:
            seen.add(r["Kernel_Name"])
            print("dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size"), r["Kernel_Name"][:70])
    break
PY
find gpurun_out/$tag -name "*.csv" -delete
