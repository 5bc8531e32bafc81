"""thin 1x1-conv GEMMs of the early stages (expand / project, M >= 32768, K or N <= 192) on the in-kernel-split kernel with
the shipped plans: time per launch, algorithmic HBM bytes / time against 8 TB/s, fp32-equivalent TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
from unidefense_amd.config import override
dev = torch.device("cuda:0")
SHAPES = ((524288, 144, 24), (524288, 24, 144), (131072, 192, 32), (131072, 32, 192), (131072, 144, 32), (32768, 336, 56),
          (32768, 56, 336), (8192, 672, 112), (8192, 112, 672), (8192, 960, 160), (8192, 160, 960))
with override(spectral_p2="off"):
    for M, N, Kd in SHAPES:
        x, w, dy = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev), torch.randn(M, N, device=dev)
        by = 4.0 * (M * Kd + N * Kd + M * N)
        fl = 2.0 * M * N * Kd
        acc = K.zeros64(2 * N, x)
        t = {"nt": K._time_launches(lambda: K.gemm_nt(x, w)), "nt+stats": K._time_launches(lambda: K.gemm_nt(x, w, stats=acc)),
             "nn": K._time_launches(lambda: K.gemm_nn(dy, w)), "tn": K._time_launches(lambda: K.gemm_tn(dy, x))}
        print("%7d %4d %4d | " % (M, N, Kd) + "  ".join("%s %6.1f us %4.0f%% hbm %5.1f TF" % (k, v * 1e3, 100 * by / (v * 1e-3) / 8e12, fl / (v * 1e-3) / 1e12)
                                                      for k, v in t.items()), flush=True)
