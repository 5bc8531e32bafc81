"""GPU: the thin weight-gradient GEMMs (tiny M x N, reduction over 32k-524k pixels) against their split-K factor — the tuner's
list ends at 384 and these shapes picked its last entries.  usage: probe_thin_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
shapes = [(24, 24, 524288), (24, 48, 524288), (144, 24, 524288), (192, 32, 131072), (32, 192, 131072), (32, 144, 131072),
          (336, 56, 32768), (56, 336, 32768), (672, 112, 8192), (112, 672, 8192), (960, 160, 8192)]
splits = (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096)
for M, N, Kd in shapes:
    a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
    out = torch.zeros(M, N, device=dev)
    mb = Kd * (M + N) * 4 / 1e6
    print("tn %d x %d x %d  (%.0f MB: %.1f us at 8 TB/s)" % (M, N, Kd, mb, mb / 8e3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e3 if False else mb / 8.0))
    best = None
    for cfg in (4, 2, 3):
        bm, bn = K._X3_TILES[cfg]
        tiles = -(-M // bm) * -(-N // bn)
        row = []
        for s in splits:
            if Kd // s < 32 or tiles * s > 16384:
                continue
            t = K._time_launches(lambda: K._gemm(a, b, out, M, N, Kd, M, N, N, 1, 1, 2, s, cfg=cfg)) * 1e3
            row.append("%d:%.1f" % (s, t))
            if best is None or t < best[0]:
                best = (t, cfg, s)
        print("   cfg %d (%dx%d, %d tiles)  " % (cfg, bm, bn, tiles) + "  ".join(row))
    print("   best %.1f us: cfg %d split %d  (%.0f GB/s)" % (best[0], best[1], best[2], mb / best[0] * 1e3))
