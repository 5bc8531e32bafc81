"""duration of one plain 128x128-tile GEMM launch as a function of K (fixed tile count): intercept = launch ramp + tail.
usage: python tools/bench_gemm_k.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
for kind in ("nn", "nt"):
    for M, N in ((4352, 1920), (2048, 2048), (4096, 2048), (4096, 4096), (1024, 1024)):
        row = []
        for Kd in (256, 512, 1024, 2048, 4096, 8192):
            a = torch.randn(M, Kd, device=dev)
            b = torch.randn((N, Kd) if kind == "nt" else (Kd, N), device=dev)
            out = torch.empty(M, N, device=dev)
            t = K._time_launches(lambda: K._gemm(a, b, out, M, N, Kd, Kd, Kd if kind == "nt" else N, N, 0, 0 if kind == "nt" else 1, 0, 1, cfg=1), n=8)
            row.append((Kd, t * 1e3))
        (k1, t1), (k2, t2) = row[-3], row[-1]
        slope = (t2 - t1) / (k2 - k1)
        print(kind, M, N, "tiles", -(-M // 128) * -(-N // 128), " ".join("K%d %.1fus" % r for r in row),
              "| slope %.4f us/k -> %.1f TF asymptotic, intercept %.1f us" % (slope, 2.0 * M * N / slope / 1e6, t2 - slope * k2), flush=True)
