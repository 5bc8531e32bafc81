"""GPU: same-sign operands (running sum grows linearly): relative error of both ud_gemm paths vs float64.
Shows whether the matrix pipe's fp32 accumulation rounds to nearest (unbiased, ~sqrt(n) ulp) or truncates (~n ulp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K, lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for Kd in (64, 256, 1024, 4096, 16384):
    a = torch.rand(512, Kd, device=dev) + 0.5
    b = torch.rand(512, Kd, device=dev) + 0.5
    ref = a.double() @ b.double().t()
    row = []
    for path in (1, 2):
        lib.call("ud_gemm_set_path", path)
        y = K.gemm_nt(a, b).double()
        rel = (y - ref) / ref
        row.append((rel.mean().item(), rel.abs().max().item()))
    yt = (a @ b.t()).double()
    relt = (yt - ref) / ref
    print("K %6d | fp32-mfma mean %+.3e max %.3e | x3 mean %+.3e max %.3e | torch mean %+.3e max %.3e" %
          (Kd, row[0][0], row[0][1], row[1][0], row[1][1], relt.mean().item(), relt.abs().max().item()), flush=True)
