"""GPU microbenchmark of the column-reduction / normalisation kernels on the model's BatchNorm shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
SHAPES = [(524288, 24), (524288, 144), (131072, 32), (131072, 192), (32768, 56), (32768, 336), (8192, 112), (8192, 672),
          (8192, 160), (8192, 960), (2048, 272), (2048, 1632), (2048, 448), (2048, 2688), (2048, 1792)]


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


print("%8s %5s | %9s %9s | %9s %9s | %9s %9s" % ("R", "C", "stats us", "GB/s", "apply us", "GB/s", "bwd us", "GB/s"))
for R, C in SHAPES:
    x = torch.randn(R, C, device=dev)
    dy = torch.randn(R, C, device=dev)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = K.norm_stats(x, 1, R, 1e-3, 0.01, rm, rv)
    t0 = timeit(lambda: K.norm_stats(x, 1, R, 1e-3, 0.01, rm, rv))
    t1 = timeit(lambda: K.norm_apply(x, 1, R, mean, invstd, g, b, 1))
    t2 = timeit(lambda: K.norm_bwd(x, dy, 1, R, mean, invstd, g, b, 1))
    by = R * C * 4
    print("%8d %5d | %9.1f %9.0f | %9.1f %9.0f | %9.1f %9.0f" % (R, C, t0, by / t0 / 1e3, t1, 2 * by / t1 / 1e3, t2,
                                                                 5 * by / t2 / 1e3), flush=True)
