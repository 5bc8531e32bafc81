"""Round 5: the depthwise conv's data + weight gradient as ONE kernel (kernels.dwtile_bwd, csrc/dwtile.hip: dw_tile_bwd_kernel)
against the pair of kernels the step launched until round 4 (the choice of tape._dw_tile_policy per shape: tiled or strip), on
the stride-1 depthwise shapes of UDEB4 at bs 32 (--half: half storage, bs 64).  Graph-replayed launches, us per call."""
import sys
import torch

sys.path.insert(0, ".")
from unidefense_amd import kernels as K          # noqa: E402
from unidefense_amd import tape as T             # noqa: E402
from tools.bench_dwtile import timed             # noqa: E402  (prints its own table when imported as a script only)

half = "--half" in sys.argv
N = 64 if half else 32
dev = torch.device("cuda:0")
st = torch.float16 if half else torch.float32
SHAPES = [("b0", 128, 48, 3, True, False), ("b1", 128, 24, 3, True, True), ("b3-5", 64, 192, 3, True, False),
          ("b7-9", 32, 336, 5, False, False), ("b11-15", 16, 672, 3, False, False), ("b16-21", 16, 960, 5, False, False),
          ("b23-29", 8, 1632, 5, False, False), ("b30-31", 8, 2688, 3, True, False)]
print("storage", st, "batch", N)
print("%-8s %4s %5s %2s | separate: wgrad + data = sum | fused | (us)" % ("blocks", "H", "C", "k"))
for name, H, Cc, k, plain, nobn in SHAPES:
    x = torch.randn(N, H, H, Cc, device=dev).to(st)
    dy = torch.randn(N, H, H, Cc, device=dev).to(st)
    add = None if plain else torch.randn(N, H, H, Cc, device=dev).to(st)
    wt = torch.randn(k * k, Cc, device=dev) * 0.2
    gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    alpha = None if plain else torch.tensor([0.3], device=dev)
    gm = 0 if plain else 2
    pad = (k - 1) // 2
    M = N * H * H
    acc = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
    K.colstats(x.view(M, Cc), acc)
    bn = None if nobn else K.DeferredBN(acc, Cc, M, gamma, beta, 1e-3, 1)
    a = x if nobn else K.bn_apply(x, bn, 1, M)
    t_fwd, t_wg, t_bwd = T._dw_tile_policy(not plain, k, 1, H, half)

    def wg():
        if t_wg:
            K.dwtile_bwd_weight(x, dy, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=gm)
        else:
            K.dwconv_bwd_weight_ex(a, dy, alpha, gm, k, 1, pad, pad)

    def data():
        if bn is not None:
            sb = K.zeros64(2 * Cc, x)
            if t_bwd:
                K.dwtile_bwd_data(dy, wt, k, pad, pad, H, H, alpha, gm, add, x, bn, sb)
            else:
                K.dwconv_bwd_data_bn(dy, alpha, gm, wt, add, x, bn, k, 1, pad, pad, sb)
        elif t_bwd:
            K.dwtile_bwd_data(dy, wt, k, pad, pad, H, H, alpha, gm, add)
        else:
            K.dwconv_bwd_data_ex(dy, alpha, gm, wt, add, k, 1, pad, pad, H, H)

    def fused():
        K.dwtile_bwd(dy, x, wt, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=gm, add=add,
                     sacc=K.zeros64(2 * Cc, x) if bn is not None else None)
    r = [timed(f) for f in (wg, data, fused)]
    print("%-8s %4d %5d %2d | %7.1f + %7.1f = %7.1f | %7.1f |  policy wg %s data %s  (%.0f MB per tensor)" % (
        name, H, Cc, k, r[0], r[1], r[0] + r[1], r[2], "tile" if t_wg else "strip", "tile" if t_bwd else "strip",
        x.numel() * x.element_size() / 1e6))
