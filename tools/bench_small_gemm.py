"""Thin-GEMM study: in-graph time of ud_gemm for the expand / project shapes as a function of split-K.
usage: bench_small_gemm.py   (prints one line per shape and split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
SHAPES = [("nt", 2048, 272, 1632), ("nn", 2048, 1632, 272), ("nt", 2048, 1632, 272), ("nt", 8192, 160, 960),
          ("nt", 8192, 960, 160), ("nt", 131072, 32, 192), ("nt", 131072, 192, 32), ("nt", 32768, 336, 56),
          ("nt", 1152, 3264, 3264), ("tn", 1632, 272, 2048), ("tn", 960, 160, 8192)]


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3          # us per launch


for kind, M, N, Kd in (SHAPES if __name__ == "__main__" else []):
    if kind == "nt":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
        args = lambda out, split: (a, b, out, M, N, Kd, Kd, Kd, N, 0, 0, 2 if split > 1 else 0, split)
    elif kind == "nn":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev)
        args = lambda out, split: (a, b, out, M, N, Kd, Kd, N, N, 0, 1, 2 if split > 1 else 0, split)
    else:
        a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
        args = lambda out, split: (a, b, out, M, N, Kd, M, N, N, 1, 1, 2 if split > 1 else 0, split)
    out = torch.zeros(M, N, device=dev)
    res = []
    for split in (1, 2, 4, 8, 16, 32):
        if Kd // split < 32:
            continue
        t = timed(lambda: K._gemm(*args(out, split)))
        res.append(f"s{split}: {t:6.1f}us {2.0 * M * N * Kd / t * 1e-6:6.1f}TF")
    plan = timed(lambda: (K.gemm_nt(a, b) if kind == "nt" else K.gemm_nn(a, b) if kind == "nn" else K.gemm_tn(a, b)))
    print(f"{kind} {M:6d}x{N:5d}x{Kd:5d}  " + "  ".join(res) + f"   | planned: {plan:6.1f}us", flush=True)
