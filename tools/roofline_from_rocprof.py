"""Recompute bench.py's `roofline` numbers from the committed profile of the same command:

  python tools/roofline_from_rocprof.py profiles/r02/rocprofv3_kernel_stats_bench_bs32.csv profiles/r02/gemm_table_bs32.txt STEPS

kernel_stats.csv: `rocprofv3 --kernel-trace --stats` summary (graph-replayed + eager steps of one bench run, STEPS executed
steps in total: tools/gpu_round.sh prints the count); gemm_table: bench.py --gemm-table (algorithmic GFLOP per step and shape,
pipe = 2 for the split-bf16 kernel).  Prints the per-step time of gemm_x3_kernel from the profiler, its algorithmic
fp32 TFLOP/s, the executed bf16 MFMA TFLOP/s (x 6) and the fraction of the 2500 TFLOP/s dense BF16 pipe peak."""
import csv
import json
import sys


def main(stats_csv, table_txt, steps, json_out=None):
    steps = float(steps)
    x3_ns = f32_ns = p3_ns = total_ns = 0.0
    with open(stats_csv) as fh:
        for row in csv.DictReader(fh):
            ns = float(row["TotalDurationNs"])
            total_ns += ns
            if "gemm_p3_kernel" in row["Name"] or "gemm_p3_pair_kernel" in row["Name"]:          # (pair: two products in one grid)
                p3_ns += ns
            elif "gemm_x3_kernel" in row["Name"]:
                x3_ns += ns
            elif "gemm_kernel" in row["Name"]:
                f32_ns += ns
    x3_gflop = f32_gflop = p3_gflop = 0.0
    with open(table_txt) as fh:
        for ln in fh:
            if ln.startswith("#") or ln.startswith("M "):
                continue
            left, right = ln.split("|")
            pipe = int(left.split()[7])
            gflop = float(right.split()[4])
            if pipe == 4:
                p3_gflop += gflop
            elif pipe == 2:
                x3_gflop += gflop
            else:
                f32_gflop += gflop
    x3_ms, f32_ms = x3_ns / steps / 1e6, f32_ns / steps / 1e6
    alg = x3_gflop / x3_ms                      # GFLOP / ms = TFLOP/s
    print(f"all kernels           : {total_ns / steps / 1e6:8.3f} ms/step")
    print(f"gemm_x3_kernel        : {x3_ms:8.3f} ms/step, {x3_gflop:8.1f} GFLOP/step algorithmic -> {alg:6.1f} TFLOP/s fp32-equivalent")
    print(f"  executed bf16 MFMA  : {6 * alg:8.1f} TFLOP/s = {6 * alg / 2500:.3f} of the 2500 TFLOP/s dense BF16 peak (roofline.frac)")
    print(f"  as fp32 work        : {alg / 157.3:.3f} of the 157.3 TFLOP/s fp32 matrix peak (roofline.frac_fp32_equiv)")
    if f32_ms > 0:
        print(f"gemm_kernel (fp32 pipe): {f32_ms:8.3f} ms/step, {f32_gflop:8.1f} GFLOP/step -> {f32_gflop / f32_ms:6.1f} TFLOP/s "
              f"= {f32_gflop / f32_ms / 157.3:.3f} of the fp32 matrix peak")
    if p3_ns > 0:
        p3_ms = p3_ns / steps / 1e6
        print(f"gemm_p3_kernel<prec 2>: {p3_ms:8.3f} ms/step, {p3_gflop:8.1f} GFLOP/step algorithmic -> {p3_gflop / p3_ms:6.1f} TFLOP/s "
              f"fp32-equivalent; executed fp16 MFMA (x 3): {3 * p3_gflop / p3_ms:8.1f} TFLOP/s = {3 * p3_gflop / p3_ms / 2500:.3f} of the pipe")
        fam_ms, fam_gf, fam_ex = p3_ms + x3_ms, p3_gflop + x3_gflop, 3 * p3_gflop + 6 * x3_gflop
        print(f"matrix-pipe family    : {fam_ms:8.3f} ms/step, {fam_gf:8.1f} GFLOP/step -> {fam_gf / fam_ms:6.1f} TFLOP/s fp32-equivalent; "
              f"executed MFMA {fam_ex / fam_ms:8.1f} TFLOP/s = {fam_ex / fam_ms / 2500:.3f} of the pipe (roofline.frac); priced at six "
              f"MFMAs per product like rounds 1-3: {6 * fam_gf / fam_ms / 2500:.3f}")
    print(f"non-GEMM kernels      : {(total_ns - x3_ns - f32_ns - p3_ns) / steps / 1e6:8.3f} ms/step")
    if json_out and p3_ns > 0:
        # machine-readable, stamped with the hash of the GEMM sources (bench.py prints it as roofline.replayed only for this tree)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        with open(json_out, "w") as fh:
            json.dump({"gemm_src_sha": bench._gemm_src_sha(), "source": stats_csv, "executed_steps": steps,
                       "what": "rocprofv3 --kernel-trace --stats of the bench command (graph-replayed + eager steps): kernel time "
                               "per executed step; GFLOP per step from bench.py --gemm-table",
                       "family_ms_per_step": fam_ms, "family_gflop_per_step": fam_gf, "family_tflops_fp32_equiv": fam_gf / fam_ms,
                       "family_frac_of_pipe": fam_ex / fam_ms / 2500,
                       "planes_kernel_ms_per_step": p3_ms, "planes_kernel_tflops_fp32_equiv": p3_gflop / p3_ms,
                       "planes_kernel_frac_of_pipe": 3 * p3_gflop / p3_ms / 2500,
                       "x3_ms_per_step": x3_ms, "x3_tflops_fp32_equiv": alg, "x3_frac_of_pipe": 6 * alg / 2500,
                       "non_gemm_ms_per_step": (total_ns - x3_ns - f32_ns - p3_ns) / steps / 1e6,
                       "all_kernels_ms_per_step": total_ns / steps / 1e6}, fh, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
