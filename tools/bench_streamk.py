"""stream-K against the best split-K of the big GEMM shapes of the bs-32 step (device time from replayed graphs).
usage: python tools/bench_streamk.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

SHAPES = [("nn", 1280, 3264, 3264), ("nt", 1280, 3264, 3264), ("nt", 4608, 1920, 1920), ("nn", 4608, 1920, 1920),
          ("tn", 3264, 3264, 1280), ("tn", 1920, 1920, 4608), ("nn", 4352, 1920, 1920), ("nt", 4608, 1344, 1344),
          ("nn", 4608, 1344, 1344), ("tn", 1344, 1344, 4608), ("tn", 672, 672, 17408), ("nn", 17408, 672, 672),
          ("nt", 17408, 672, 672), ("tn", 1632, 272, 2048), ("nn", 2048, 272, 1632), ("nt", 2048, 1632, 272),
          ("tn", 960, 160, 8192), ("nn", 8192, 160, 960), ("nt", 8192, 960, 160), ("nn", 256, 1920, 1920),
          ("nt", 67584, 384, 384), ("tn", 384, 384, 67584)]
dev = torch.device("cuda:0")
for kind, M, N, Kd in SHAPES:
    sa, sb = ((Kd, M) if kind == "tn" else (M, Kd)), ((N, Kd) if kind == "nt" else (Kd, N))
    a, b = torch.randn(sa, device=dev), torch.randn(sb, device=dev)
    out = torch.zeros(M, N, device=dev)
    a_mode, b_mode = (1, 1) if kind == "tn" else (0, 0 if kind == "nt" else 1)
    lda, ldb = (M if kind == "tn" else Kd), (Kd if kind == "nt" else N)
    res = {}
    for cfg, split in K._tune_candidates(M, N, Kd):
        t = K._time_launches(lambda: K._gemm(a, b, out, M, N, Kd, lda, ldb, N, a_mode, b_mode, 2 if split != 1 else 0, split, cfg=cfg))
        res[(cfg, split)] = t
    best_s = min((t, k) for k, t in res.items() if k[1] >= 1)
    best_k = min((t, k) for k, t in res.items() if k[1] < 1)
    fl = 2.0 * M * N * Kd / 1e9
    print("%s %6d %5d %6d  split-K best %s %.3f ms %6.1f TF | stream-K best %s %.3f ms %6.1f TF  (%+.1f %%)" % (
        kind, M, N, Kd, best_s[1], best_s[0], fl / best_s[0], best_k[1], best_k[0], fl / best_k[0],
        100 * (best_s[0] / best_k[0] - 1)), flush=True)
