"""Two-pass FFT forms (row kernel + column kernel) against the LDS-resident kernels on the 64 x 64 / 32 x 32 shapes of the step.
usage: python tools/bench_fft2p.py [--half]   (fp32 storage at bs 32, half storage at bs 64)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

half = "--half" in sys.argv
dev = torch.device("cuda:0")
dt = torch.float16 if half else torch.float32
N = 64 if half else 32
print("storage", dt, "batch", N)
print(" S    C | rfft2 plain | rfft2_ex (bn+act out) | irfft2 plain | irfft2_mix     one-kernel / two-pass (us), MB moved by the one-kernel form")
for S, Cc in ((64, 144), (64, 192), (32, 192), (32, 336), (16, 672)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    Yin = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev).to(dt)
    spat = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    alpha = torch.tensor(-0.3, device=dev)
    acc0 = K.zeros64(2 * Cc, x)
    K.colstats(x.view(N * S * S, Cc), acc0)
    bn = K.DeferredBN(acc0, Cc, N * S * S, gamma, beta, 1e-3, 1)
    acc = K.zeros64(2 * Cc, x)
    fns = [lambda: K.rfft2(x, 1.0 / S, 1.0), lambda: K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True),
           lambda: K.irfft2(Yin, 1.0 / S, 1.0), lambda: K.irfft2_mix(Yin, 1.0 / S, spat, alpha, acc)]
    es = x.element_size()
    mb = [x.numel() * es + Yin.numel() * es, 2 * x.numel() * es + Yin.numel() * es, x.numel() * es + Yin.numel() * es,
          4 * x.numel() * es + Yin.numel() * es]
    cells = []
    for fn, m in zip(fns, mb):
        ts = []
        for tp in (False, True):
            if tp and S not in (32, 64):
                ts.append(float("nan"))
                continue
            K._FFT_TWO_PASS = tp
            ts.append(K._time_launches(fn, n=4) * 1e3)
        K._FFT_TWO_PASS = None
        cells.append("%7.1f / %7.1f (%4.0f MB)" % (ts[0], ts[1], m / 1e6))
    print("%3d %4d | %s" % (S, Cc, " | ".join(cells)), flush=True)
