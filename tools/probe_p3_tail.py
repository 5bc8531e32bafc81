"""GPU: the planes GEMM on the tile-quantised spectral shapes (540 / 396 / 816 tiles of 128 x 128 on 256 CUs): plain launch,
the tail plan (whole rounds plain + the last row tiles split-K), stream-K, split-K 2 — replayed from a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
for M, N, Kd in [(4608, 1920, 1920), (4608, 1344, 1344), (17408, 672, 672), (1280, 3264, 3264)]:
    x, w, dy = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev), torch.randn(M, N, device=dev)
    xp, wp, dp = K.split_planes(x, prec=2), K.split_planes(w, prec=2), K.split_planes(dy, prec=2)
    for kind, (a, b, m, n, k) in (("nt", (xp, wp, M, N, Kd)), ("nn", (dp, wp, M, Kd, N))):
        row = []
        for plan in K._p2_plans(kind, m, n, k):
            t = K._time_launches(lambda: K._p2_run(kind, plan, a, b, m, n, k, x)) * 1e3
            row.append("%s %.1f" % ("/".join(str(v) for v in plan), t))
        tiles = -(-m // 128) * -(-n // 128)
        print("%s %d x %d x %d (%d tiles): " % (kind, m, n, k, tiles) + "   ".join(row), flush=True)
