"""fixed cost of a planes-GEMM launch: tiny problems, graph-replayed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, Kd in ((128, 128, 32), (128, 128, 1024), (2048, 256, 32), (2048, 256, 256), (2048, 256, 1024), (2048, 256, 1632), (2048, 1664, 256)):
    x, w = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
    xp, wp = K.split_planes(x, prec=2), K.split_planes(w, prec=2)
    out = torch.empty(M, N, device=dev)
    t2 = K._time_launches(lambda: K._gemm_p3(xp, wp, out, M, N, Kd, 0, 0), n=16)
    t3 = K._time_launches(lambda: K._gemm(x, w, out, M, N, Kd, Kd, Kd, N, 0, 0, 0, 1, cfg=1), n=16)
    t4 = K._time_launches(lambda: K._gemm(x, w, out, M, N, Kd, Kd, Kd, N, 0, 0, 0, 1, cfg=4), n=16)
    print(f"{M}x{N}x{Kd}: p2 plain {t2*1e3:.1f} us   x3 128x128 {t3*1e3:.1f}  x3 64x64 {t4*1e3:.1f}", flush=True)
