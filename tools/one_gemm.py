"""Run a few ud_gemm launches of chosen shapes (for rocprofv3 --pmc). usage: one_gemm.py M N K kind [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
M, N, Kd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
if kind == "nt":
    a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev); fn = lambda: K.gemm_nt(a, b)
elif kind == "nn":
    a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev); fn = lambda: K.gemm_nn(a, b)
else:
    a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev); fn = lambda: K.gemm_tn(a, b)
if os.environ.get("UD_ONE_GEMM_P3"):
    am, bm = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}[kind]
    ap, bp = K.split_planes(a), K.split_planes(b)
    out = torch.empty(M, N, device=dev)
    cfg = int(os.environ.get("UD_ONE_GEMM_CFG", "0"), 0)
    fn = lambda: K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, cfg=cfg)
for _ in range(it):
    fn()
torch.cuda.synchronize()
