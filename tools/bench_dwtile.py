"""LDS-tiled depthwise kernels (csrc/dwtile.hip) against the strip kernels they would replace, on the stride-1 depthwise
shapes of UDEB4 at bs 32 (and with --half: half storage at bs 64).  Graph-replayed launches, us per call."""
import sys
import torch

sys.path.insert(0, ".")
from unidefense_amd import kernels as K          # noqa: E402

half = "--half" in sys.argv
N = 64 if half else 32
dev = torch.device("cuda:0")
st = torch.float16 if half else torch.float32
# (blocks, H, C, k, plain?)  stride-1 blocks of the b4 trunk
SHAPES = [("b0", 128, 48, 3, True), ("b1", 128, 24, 3, True), ("b3-5", 64, 192, 3, True), ("b7-9", 32, 336, 5, False),
          ("b11-15", 16, 672, 3, False), ("b16-21", 16, 960, 5, False), ("b23-29", 8, 1632, 5, False),
          ("b30-31", 8, 2688, 3, True)]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                K.reset_zero_pool()
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        g.replay()
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


S2 = [("b2", 128, 144, 3, (0, 1, 0, 1), True), ("b6", 64, 192, 5, (2, 2, 2, 2), False), ("b10", 32, 336, 3, (0, 1, 0, 1), False),
      ("b22", 16, 960, 5, (1, 2, 1, 2), False)]


def stride2():
    print("stride 2: %-6s %4s %5s %2s | fwd old / new | bwd-data (+ sums pass) old / new | wgrad old / new  (us)" % ("block", "H", "C", "k"))
    for name, H, Cc, k, (pl, pr, pt, pb), plain in S2:
        Ho = (H + pt + pb - k) // 2 + 1
        x = torch.randn(N, H, H, Cc, device=dev).to(st)
        dy = torch.randn(N, Ho, Ho, Cc, device=dev).to(st)
        add = torch.randn(N, H, H, Cc, device=dev).to(st)
        wt = torch.randn(k * k, Cc, device=dev) * 0.2
        gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        M = N * H * H
        acc = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
        K.colstats(x.view(M, Cc), acc)
        bn = K.DeferredBN(acc, Cc, M, gamma, beta, 1e-3, 1)
        a = K.bn_apply(x, bn, 1, M)

        def fwd_old():
            a_ = K.bn_apply(x, bn, 1, M) if plain else a
            d = K.dwconv_fwd(a_, wt, k, 2, pt, pl, Ho, Ho)
            if plain:
                K.colstats(d.view(-1, Cc), K.zeros64(2 * Cc, x))

        def fwd_new():
            K.dwtile_fwd(x, wt, k, pt, pl, Ho, Ho, bn=bn, stats=K.zeros64(2 * Cc, x) if plain else None, stride=2)

        def bwd_old():
            dz = K.dwconv_bwd_data(dy, wt, k, 2, pt, pl, H, H, add=add)
            K.normbwd_sums(x, dz, None, 1.0, bn, False, 1, M, K.zeros64(2 * Cc, x))

        def bwd_new():
            K.dwtile_bwd_data(dy, wt, k, pt, pl, H, H, None, 0, add, x, bn, K.zeros64(2 * Cc, x), stride=2)

        def wg_old():
            K.dwconv_bwd_weight_ex(a, dy, None, 0, k, 2, pt, pl)

        def wg_new():
            K.dwtile_bwd_weight(x, dy, k, pt, pl, bn=bn, stride=2)
        r = [timed(f) for f in (fwd_old, fwd_new, bwd_old, bwd_new, wg_old, wg_new)]
        print("          %-6s %4d %5d %2d | %7.1f / %7.1f | %9.1f / %9.1f | %8.1f / %8.1f   (%.0f MB input)" % (
            name, H, Cc, k, r[0], r[1], r[2], r[3], r[4], r[5], x.numel() * x.element_size() / 1e6))


def main():
    print("storage", st, "batch", N)
    if "--stride2" in sys.argv:
        stride2()
        sys.exit(0)
    print("%-8s %4s %5s %2s | %-26s | %-26s | %-20s" % ("blocks", "H", "C", "k", "fwd old / new (us)", "bwd-data+bn old / new", "wgrad old / new"))
    for name, H, Cc, k, plain in SHAPES:
        x = torch.randn(N, H, H, Cc, device=dev).to(st)
        dy = torch.randn(N, H, H, Cc, device=dev).to(st)
        add = torch.randn(N, H, H, Cc, device=dev).to(st)
        wt = torch.randn(k * k, Cc, device=dev) * 0.2
        gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        alpha = torch.tensor([0.3], device=dev)
        pad = (k - 1) // 2
        M = N * H * H
        acc = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
        K.colstats(x.view(M, Cc), acc)
        bn = K.DeferredBN(acc, Cc, M, gamma, beta, 1e-3, 1)
        a = K.bn_apply(x, bn, 1, M)

        def fwd_old():
            a_ = K.bn_apply(x, bn, 1, M)                       # plain blocks materialise swish(bn0(e)); SF blocks get it from rfft2_ex
            d = K.dwconv_fwd(a_, wt, k, 1, pad, pad, H, H)
            if plain:
                K.colstats(d.view(M, Cc), K.zeros64(2 * Cc, x))

        def fwd_old_nobn():
            d = K.dwconv_fwd(a, wt, k, 1, pad, pad, H, H)
            if plain:
                K.colstats(d.view(M, Cc), K.zeros64(2 * Cc, x))

        def fwd_new():
            K.dwtile_fwd(x, wt, k, pad, pad, H, H, bn=bn, stats=K.zeros64(2 * Cc, x) if plain else None)

        def bwd_old():
            K.dwconv_bwd_data_bn(dy, alpha, 2, wt, add, x, bn, k, 1, pad, pad, K.zeros64(2 * Cc, x))

        def bwd_new():
            K.dwtile_bwd_data(dy, wt, k, pad, pad, H, H, alpha, 2, add, x, bn, K.zeros64(2 * Cc, x))

        def wg_old():
            K.dwconv_bwd_weight_ex(a, dy, alpha, 2, k, 1, pad, pad)

        def wg_new():
            K.dwtile_bwd_weight(x, dy, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=2)
        r = [timed(f) for f in (fwd_old, fwd_old_nobn, fwd_new, bwd_old, bwd_new, wg_old, wg_new)]
        mb = x.numel() * x.element_size() / 1e6
        print("%-8s %4d %5d %2d | %7.1f (%6.1f) / %7.1f | %9.1f / %9.1f     | %8.1f / %8.1f   (%.0f MB per tensor)" % (
            name, H, Cc, k, r[0], r[1], r[2], r[3], r[4], r[5], r[6], mb))


if __name__ == "__main__":
    main()
