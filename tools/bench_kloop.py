"""k-loop latency of the split-bf16 GEMM: time per 16-deep K-tile of a workgroup, alone on the chip and with 1 ... 4
workgroups per CU, per tile configuration (slope of the launch time over K) and the fixed cost of a launch (intercept)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
from tools.bench_small_gemm import timed

dev = torch.device("cuda:0")
for cfg, (bm, bn) in ((1, (128, 128)), (2, (128, 64)), (4, (64, 64))):
    for ntiles in (1, 256, 512, 1024, 2048):
        M, N = bm * ntiles, bn
        ts = []
        for Kd in (256, 1024, 4096):
            a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
            out = torch.empty(M, N, device=dev)
            ts.append(timed(lambda: K._gemm(a, b, out, M, N, Kd, Kd, Kd, N, 0, 0, 0, 1, cfg=cfg), n=10))
        slope = (ts[2] - ts[1]) / ((4096 - 1024) / 16)
        icpt = ts[0] - slope * 256 / 16
        print(f"tile {bm}x{bn}  {ntiles:5d} workgroups ({ntiles / 256:.1f}/CU): K=256 {ts[0]:7.1f} us  K=1024 {ts[1]:7.1f}  "
              f"K=4096 {ts[2]:8.1f}   per K-tile {slope:.3f} us   fixed {icpt:5.1f} us", flush=True)
