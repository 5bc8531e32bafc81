"""thin 1x1 convs (expand / project of the 8^2 and 16^2 stages): in-kernel-split products vs planes products vs the split passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, Kd in ((2048, 272, 1632), (2048, 1632, 272), (8192, 160, 960), (8192, 960, 160), (8192, 672, 112), (32768, 336, 56)):
    x, w, dy = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev), torch.randn(M, N, device=dev)
    t = {"x3 nt": K._time_launches(lambda: K.gemm_nt(x, w)), "x3 nn": K._time_launches(lambda: K.gemm_nn(dy, w)),
         "x3 tn": K._time_launches(lambda: K.gemm_tn(dy, x))}
    xp, wp, dp = K.split_planes(x, prec=2), K.split_planes(w, prec=2), K.split_planes(dy, prec=2)
    t["split x"] = K._time_launches(lambda: K.split_planes(x, xp))
    t["split w"] = K._time_launches(lambda: K.split_planes(w, wp))
    t["split dy"] = K._time_launches(lambda: K.split_planes(dy, dp))
    for kind, (a, b, m, n, k) in (("nt", (xp, wp, M, N, Kd)), ("nn", (dp, wp, M, Kd, N)), ("tn", (dp, xp, N, Kd, M))):
        best = min((K._time_launches(lambda: K._p2_run(kind, plan, a, b, m, n, k, x)), plan) for plan in K._p2_plans(kind, m, n, k))
        t["p2 " + kind] = best[0]
        t["p2 %s plan" % kind] = best[1]
    print(M, N, Kd, " ".join(f"{k} {v * 1e3:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in t.items()), flush=True)
