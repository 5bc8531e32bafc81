"""Timing of the pass-2 perturbations at the bench shape (bs 32, 3x256x256) — HIP kernels of csrc/perturb.hip vs the
same arithmetic as stock torch device ops (what the reference executes on a GPU), in one process on one box.
    python tools/bench_perturb.py [bs] [size]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unidefense_amd.model import perturb as P


def t_freq(c, s, l):
    l = l.view(-1, 1, 1, 1)
    fa = torch.fft.rfft2(c, norm="ortho")
    fb = torch.fft.rfft2(s, norm="ortho")
    return torch.fft.irfft2((l * fa.abs() + (1 - l) * fb.abs()) * torch.exp(1j * fa.angle()), s=c.shape[-2:], norm="ortho")


def t_spat(c, s, l):
    b, ch, h, w = c.shape
    l = l.view(-1, 1, 1)
    cv = c.view(b, ch, -1)
    _, idx = torch.sort(cv, dim=-1)
    sv, _ = torch.sort(s.view(b, ch, -1), dim=-1)
    return (cv + (1 - l) * sv.gather(-1, idx.argsort(-1)) - (1 - l) * cv).view(b, ch, h, w)


def t_down(x):
    import torch.nn.functional as F
    return F.interpolate(F.interpolate(x, scale_factor=0.75, mode="nearest"), size=x.shape[-2:], mode="nearest")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(bs, 3, size, size, generator=g) * 2 - 1).cuda()
    s = x[torch.randperm(bs, generator=g)].contiguous()
    lm = (torch.rand(bs, generator=g) / 2 + 0.5).cuda()
    mb = x.numel() * 4 / 1e6
    rows = [("freq amplitude transfer", lambda: P.freq_transfer_with(x, s, lm), lambda: t_freq(x, s, lm)),
            ("EFDM rank matching", lambda: P.spatial_transfer_with(x, s, lm), lambda: t_spat(x, s, lm)),
            ("downscale 0.75", lambda: P.downscale(x), lambda: t_down(x)),
            ("gaussian blur 5x5", lambda: P.random_blur(x), None),
            ("CORAL colour transfer", lambda: P.coral(s, x), None)]
    print(f"bs {bs}, 3x{size}x{size} ({mb:.1f} MB per batch)")
    for name, hip, ref in rows:
        a = timeit(hip)
        b = timeit(ref) if ref else float("nan")
        print(f"  {name:26s} HIP {a:8.3f} ms   torch ops {b:8.3f} ms")


if __name__ == "__main__":
    main()
