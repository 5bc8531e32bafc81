"""GPU microbenchmark of the depthwise-conv kernels on the UDEB4 layer shapes (bs 32, 256x256 input)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
from unidefense_amd.model.arch import build_arch

dev = torch.device("cuda:0")
arch = build_arch("efficientnet-b4", "ortho", None)
seen = {}
H = 128
for b in arch["blocks"]:
    key = (H, b.cexp, b.k, b.stride, tuple(b.pad))
    seen[key] = seen.get(key, 0) + 1
    H = -(-H // b.stride)


def timeit(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


tf = tb = tw = 0.0
print("%4s %5s k s cnt | %8s %6s | %8s %6s | %8s" % ("H", "C", "fwd us", "GB/s", "bwdD us", "GB/s", "bwdW us"))
for (H, C, k, s, pad), cnt in seen.items():
    N = 32
    pl, pr, pt, pb = pad
    Ho = (H + pt + pb - k) // s + 1
    Wo = (H + pl + pr - k) // s + 1
    x = torch.randn(N, H, H, C, device=dev)
    wt = torch.randn(k * k, C, device=dev)
    y = K.dwconv_fwd(x, wt, k, s, pt, pl, Ho, Wo)
    dy = torch.randn_like(y)
    t0 = timeit(lambda: K.dwconv_fwd(x, wt, k, s, pt, pl, Ho, Wo))
    t1 = timeit(lambda: K.dwconv_bwd_data(dy, wt, k, s, pt, pl, H, H))
    t2 = timeit(lambda: K.dwconv_bwd_weight(x, dy, k, s, pt, pl))
    by = (x.numel() + y.numel()) * 4
    print("%4d %5d %d %d %3d | %8.1f %6.0f | %8.1f %6.0f | %8.1f" % (H, C, k, s, cnt, t0, by / t0 / 1e3, t1,
                                                                   by / t1 / 1e3, t2), flush=True)
    tf += t0 * cnt
    tb += t1 * cnt
    tw += t2 * cnt
print("per step: fwd %.2f ms  bwd_data %.2f ms  bwd_weight %.2f ms" % (tf / 1e3, tb / 1e3, tw / 1e3))
