#!/bin/bash
# HBM traffic of the GEMM family from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), with --kernel-trace only; both are in
# KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled.
# Output: gpurun_out/$1/hbm_traffic_gemm.json  (copy to profiles/<round>/; bench.py reports it as roofline.traffic)
tag=${1:-traffic}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UD_GEMM_TUNE_CACHE=${UD_GEMM_TUNE_CACHE:-$PWD/unidefense_amd/gemm_plans_gfx950.json}     # the shipped plans: no tuner launches among the counted ones
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o $c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager $BENCH_ARGS > $out/$c.log 2>&1
  echo "$c pass exit $?"
done
python3 - <<PY
import csv, glob, json, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$out/**/%s_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = r["Kernel_Name"]
            pair = "gemm_p3_pair_kernel" in name          # two products (data + weight gradient) in one grid: counted as two launches
            fam = ("gemm_p3_kernel" if ("gemm_p3_kernel" in name or pair) else "gemm_x3_kernel" if "gemm_x3_kernel" in name else
                   "gemm_kernel" if "gemm_kernel" in name else "other")
            a = agg[fam]; a[0] += 2 if pair else 1; a[1] += float(r["Counter_Value"])
    tot[c] = {k: v for k, v in agg.items()}
import hashlib
def gemm_src_sha():          # bench.py:_gemm_src_sha — the summary belongs to these kernel sources and no others
    h = hashlib.sha256()
    for f in sorted(glob.glob("unidefense_amd/csrc/gemm*")):
        if f.endswith((".hip", ".h")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
res = {"gemm_src_sha": gemm_src_sha(), "recipe": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                 "--warmup 1 --no-cpu-baseline --eager; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE "
                 "half-count correction of MI355X_MICROARCH.md)", "families": {}}
for fam in ("gemm_p3_kernel", "gemm_x3_kernel", "gemm_kernel", "other"):
    f, w = tot["FETCH_SIZE"].get(fam, [0, 0.0]), tot["WRITE_SIZE"].get(fam, [0, 0.0])
    if not f[0] or not w[0]:
        continue
    res["families"][fam] = {"launches": f[0], "fetch_kib_raw_per_launch": f[1] / f[0], "write_kib_per_launch": w[1] / w[0],
                            "hbm_bytes_per_launch": (2 * f[1] / f[0] + w[1] / w[0]) * 1024}
g = [res["families"][k] for k in ("gemm_p3_kernel", "gemm_x3_kernel") if k in res["families"]]          # the matrix-pipe family
if g:
    n = sum(x["launches"] for x in g)
    res["gemm_family"] = {"launches": n, "hbm_bytes_per_launch": sum(x["hbm_bytes_per_launch"] * x["launches"] for x in g) / n}
json.dump(res, open("$out/hbm_traffic_gemm.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -size +8M -delete
