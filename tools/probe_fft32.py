"""S = 32 FFT kernels at the step's shape (N 32, C 336 / 192): us per launch and algorithmic bytes / time; run with UD_FFT32_WAVE=0 / 1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
print("UD_FFT32_WAVE =", os.environ.get("UD_FFT32_WAVE", "(default)"))
for N, S, Cc in ((32, 32, 336), (32, 32, 192), (64, 32, 336)):
    x = torch.randn(N, S, S, Cc, device=dev)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * S * S, gamma, beta, 1e-3, 1)
    Yin = torch.randn(N, S, S // 2 + 1, 2 * Cc, device=dev)
    spat = torch.randn(N, S, S, Cc, device=dev)
    alpha = torch.tensor(-0.3, device=dev)
    nx, ny = x.numel() * 4, Yin.numel() * 4
    for two in (False, True):
        K._FFT_TWO_PASS = two
        t = {}
        t["rfft2"] = (K._time_launches(lambda: K.rfft2(x, 1.0 / S, 1.0)), nx + ny)
        t["rfft2_ex bn+act"] = (K._time_launches(lambda: K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True, want_absmax=True)), 2 * nx + ny)
        t["irfft2"] = (K._time_launches(lambda: K.irfft2(Yin, 1.0 / S, 0.5)), nx + ny)
        a2 = K.zeros64(2 * Cc, x)
        t["irfft2_mix"] = (K._time_launches(lambda: K.irfft2_mix(Yin, 1.0 / S, spat, alpha, a2)), ny + 3 * nx)
        print(N, S, Cc, "two-pass" if two else "one-kernel", "  ".join("%s %.1f us %.0f%%" % (k, v * 1e3, 100 * b / (v * 1e-3) / 8e12) for k, (v, b) in t.items()), flush=True)
