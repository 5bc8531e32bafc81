"""ud_gemm_p3 prec 2 on the step's spectral-conv shapes: round-robin tile deal vs the XCD-aware grouped raster (GM tile rows)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
SH = (("nt", 4608, 1920, 1920), ("nn", 4608, 1920, 1920), ("tn", 1920, 1920, 4608), ("nt", 1280, 3264, 3264), ("tn", 3264, 3264, 1280),
      ("nt", 4608, 1344, 1344), ("tn", 1344, 1344, 4608), ("nt", 17408, 672, 672), ("tn", 672, 672, 17408), ("nt", 67584, 384, 384),
      ("nt", 2048, 1632, 288), ("nn", 2048, 272, 1632), ("nt", 8192, 960, 160))
for kind, M, N, Kd in SH:
    if kind == "nt":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
    elif kind == "nn":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev)
    else:
        a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    plans = [p for p in K._p2_plans(kind, M, N, Kd) if p[0] in ("plain", "split")][:3]
    row = []
    ref = None
    for plan in plans:
        for ras in (0, 0x200 | 2 << 12, 0x200 | 4 << 12, 0x200 | 6 << 12, 0x200 | 8 << 12):
            K._P3_RASTER = ras
            out = K._p2_run(kind, plan, ap, bp, M, N, Kd, a)
            if ref is None:
                ref = out.clone()
            else:
                assert (out - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), (plan, ras)
            row.append("%s/%s %.1f" % ("".join(str(x)[0] if i == 0 else str(x) for i, x in enumerate(plan)), (ras >> 12) & 15, 1e3 * K._time_launches(lambda: K._p2_run(kind, plan, ap, bp, M, N, Kd, a))))
    K._P3_RASTER = 0
    print(kind, M, N, Kd, " ".join(row), flush=True)
