"""Per-kernel HBM bandwidth from the PMC passes of tools/gpu_traffic.sh (FETCH_SIZE / WRITE_SIZE counter_collection
CSVs, one pass each): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (gfx950 half-count of wide
reads), time = the dispatch durations recorded in the same CSVs (kernels run eagerly and serialised under PMC).
    python tools/hbm_bw_table.py gpurun_out/traffic [--json profiles/r06/hbm_classes.json STEPS] > profiles/r01/hbm_bw_by_kernel.txt
--json: also the per-CLASS table SURVEY 8(d) asks for (BatchNorm / transform / squeeze-excite / depthwise / GEMM / splits /
other: launches, ms, GB moved per step, GB/s), stamped with the hash of csrc/ so that bench.py prints it only for this tree.
"""
import collections
import csv
import json
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/traffic"
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
json_steps = float(sys.argv[sys.argv.index("--json") + 2]) if "--json" in sys.argv else 1.0

CLASSES = (("gemm", ("gemm_",)),
           ("thin_conv_passes", ("pw_bwd_", "pj_bwd_", "pj_fwd_", "pj_fold")),          # round 6: BatchNorm / SE element-wise work + a thin 1x1 conv in one pass
           ("batchnorm", ("normbwd_", "colstats", "partials_to_acc", "residual_bn", "colreduce_partial", "bn_apply", "stats_finalize",
                          "norm_apply", "group_sum", "partial_sum_finalize", "stat_slots")),
           ("transform", ("rfft2", "irfft2", "rows_fwd", "cols_fwd", "rows_inv", "cols_inv", "cols_pass", "rows_adj", "sfmix")),
           ("squeeze_excite", ("colsum_bn", "se_scale", "se_bwd", "fc_fwd", "fc_bwd", "se_fc")),
           ("depthwise", ("dw_tile", "dw_fwd", "dw_bwd", "dw_wt")),
           ("plane_splits", ("split_h2", "absmax", "planes_from", "im2col", "col2im", "weight_layouts")),
           ("decoder_direct_conv", ("conv_small",)))


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return "other"
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])          # name -> launches, fetch KiB, write KiB, ns


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:70]


for c, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[col] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            a[0] += 1
            a[3] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for name, (n, f, w, ns) in agg.items():
    if n == 0 or ns == 0:
        continue
    byt = (2 * f + w) * 1024
    rows.append((ns, name, n, byt / n / 1e6, ns / n / 1e3, byt / ns))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("HBM traffic per kernel (PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB) over 6 eager steps of the bs-32 bench; peak 8000 GB/s")
print("%-70s %7s %9s %9s %8s %6s %6s" % ("kernel", "calls", "MB/call", "us/call", "GB/s", "%peak", "%time"))
for ns, name, n, mb, us, gbs in rows[:45]:
    print("%-70s %7d %9.2f %9.1f %8.0f %5.1f%% %5.1f%%" % (name, n, mb, us, gbs, 100 * gbs / 8000, 100 * ns / tot))

if json_out:
    cl = collections.defaultdict(lambda: [0, 0.0, 0.0])          # class -> launches, bytes, ns
    for name, (n, f, w, ns) in agg.items():
        c = cl[klass(name)]
        c[0] += n
        c[1] += (2 * f + w) * 1024
        c[2] += ns
    import hashlib
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "unidefense_amd", "csrc", "*"))):
        if fn.endswith((".hip", ".h")):
            h.update(open(fn, "rb").read())
    out = {"csrc_sha": h.hexdigest()[:16], "steps": json_steps,
           "recipe": "PMC passes of tools/gpu_traffic.sh over eager steps of the bs-32 bench: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, "
                     "time = the dispatch durations of the same (serialised) run",
           "classes": {k: {"launches_per_step": v[0] / json_steps, "ms_per_step": v[2] / json_steps / 1e6,
                           "gb_per_step": v[1] / json_steps / 1e9, "gbs": v[1] / v[2] if v[2] else 0.0,
                           "frac_of_hbm_peak": v[1] / v[2] / 8000 if v[2] else 0.0} for k, v in sorted(cl.items())}}
    with open(json_out, "w") as fh:
        json.dump(out, fh, indent=1)
