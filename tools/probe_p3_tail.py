"""planes GEMM launch forms on the forward shapes whose tile count sits just above whole rounds of the CUs: plain / split-K / row
tail (two launches) / TAIL PAIR (ud_gemm_p3_pair: plain leading row tiles + split-K last row tiles in one grid, round 6) / stream-K.
Times are per call inside a replayed hipGraph (kernels._time_launches), the tail forms INCLUDING their zero fill."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
import ast
shapes = tuple(("nt",) + tuple(ast.literal_eval(a)) for a in sys.argv[1:]) or (("nt", 1280, 3264, 3264), ("nt", 1280, 1920, 1920), ("nt", 4608, 1920, 1920), ("nt", 4608, 1344, 1344),
          ("nt", 17408, 672, 672), ("nt", 67584, 384, 384), ("nt", 8192, 960, 160), ("nt", 2048, 1632, 288))
for kind, M, N, Kd in shapes:
    a = torch.randn(M, Kd, device=dev)
    b = torch.randn(N, Kd, device=dev) if kind == "nt" else torch.randn(Kd, N, device=dev)
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    ref = (a.double() @ (b.double().t() if kind == "nt" else b.double())).float()
    row = []
    for plan in K._p2_plans(kind, M, N, Kd):
        out = K._p2_run(kind, plan, ap, bp, M, N, Kd, a)
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        assert err < 5e-6, (plan, err)
        t = K._time_launches(lambda: K._p2_run(kind, plan, ap, bp, M, N, Kd, a))
        row.append("%s %.1f" % ("/".join(str(x) for x in plan), t * 1e3))
    print(kind, M, N, Kd, "tiles", -(-M // 128) * -(-N // 128), " | ".join(row), flush=True)
