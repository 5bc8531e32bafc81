"""For every plain GEMM shape of the bs-32 step (profiles/r02/gemm_table_bs32.txt): in-graph time of the planned launch
vs the best (tile config x split-K) found by exhaustive measurement.  Prints the per-step saving a per-shape tuner could make."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
table = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02/gemm_table_bs32.txt"
shapes = []
for l in open(table):
    if "|" not in l or l.startswith(("M", "#")):
        continue
    a, b = l.split("|")
    M, N, Kd, am, bm, split, batch, pipe = map(int, a.split())
    calls = int(b.split()[0]) // 3
    if (am, bm) in ((0, 0), (0, 1), (1, 1)) and batch == 1 and pipe == 2:
        shapes.append((am, bm, M, N, Kd, calls))
# the table lists a tail-split GEMM as two rows; keep every row as its own launch


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


tot_plan = tot_best = 0.0
for am, bm, M, N, Kd, calls in shapes:
    if am == 0:
        a = torch.randn(M, Kd, device=dev)
        lda = Kd
    else:
        a = torch.randn(Kd, M, device=dev)
        lda = M
    if bm == 0:
        b = torch.randn(N, Kd, device=dev)
        ldb = Kd
    else:
        b = torch.randn(Kd, N, device=dev)
        ldb = N
    out = torch.zeros(M, N, device=dev)
    plan = timed(lambda: (K.gemm_nt(a, b) if (am, bm) == (0, 0) else K.gemm_nn(a, b) if (am, bm) == (0, 1) else K.gemm_tn(a, b)))
    best = (1e9, None)
    for cfg in (1, 2, 3, 4):
        bm_, bn_ = ((128, 128), (128, 64), (64, 128), (64, 64))[cfg - 1]
        tiles = -(-M // bm_) * -(-N // bn_)
        for split in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256, 384):
            if split > 1 and (Kd // split < 64 or tiles * split > 4096):
                continue
            if tiles * split < 64 and Kd >= 1024:
                continue
            t = timed(lambda: K._gemm(a, b, out, M, N, Kd, lda, ldb, N, am, bm, 2 if split > 1 else 0, split, cfg=cfg), n=6)
            if split > 1:
                t += M * N * 4 / 4.0e6 * 1e-0 / 1e0 * 0      # (zero fill of the output not charged: pooled)
            if t < best[0]:
                best = (t, (cfg, split))
    tot_plan += plan * calls
    tot_best += min(plan, best[0]) * calls
    print(f"({am},{bm}) {M:7d}x{N:5d}x{Kd:7d} x{calls:2d}: planned {plan:7.1f}us  best {best[0]:7.1f}us {best[1]}  "
          f"{'<<' if best[0] < 0.9 * plan else ''}", flush=True)
print(f"per step: planned {tot_plan / 1e3:.2f} ms, best {tot_best / 1e3:.2f} ms")
