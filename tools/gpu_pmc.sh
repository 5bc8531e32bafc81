#!/bin/bash
# PMC counters for one GEMM shape: $1 M $2 N $3 K $4 kind ; optional UD_GEMM_CFG via env
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${5:-pmc}
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o p1 -- python3 tools/one_gemm.py $1 $2 $3 $4 > gpurun_out/$tag/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/$tag -o p2 -- python3 tools/one_gemm.py $1 $2 $3 $4 > gpurun_out/$tag/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/$tag -o p3 -- python3 tools/one_gemm.py $1 $2 $3 $4 > gpurun_out/$tag/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$tag -o p4 -- python3 tools/one_gemm.py $1 $2 $3 $4 > gpurun_out/$tag/p4.log 2>&1
ls gpurun_out/$tag | head -3
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/$tag/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in agg.items():
        print("  %-28s n=%d  last=%.4g" % (k, len(v), v[-1]))
for f in sorted(glob.glob("gpurun_out/$tag/*kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"]]
    if rows:
        r = rows[-1]
        print("dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size"), r["Kernel_Name"][:60])
        break
PY
