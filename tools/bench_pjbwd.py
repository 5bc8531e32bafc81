"""ud_pj_bwd_fused_a / _b (csrc/pjbwd.hip) against the four launches they replace (gemm_tn + gemm_nn + ud_coldot_bn +
ud_se_scale_bwd_bn) on the thin project convs of UDEB4 at bs 32: graph-replayed launches, us per call."""
import sys
import torch

sys.path.insert(0, ".")
from unidefense_amd import kernels as K          # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("b0 128x128", 32, 128 * 128, 48, 24), ("b1 128x128", 32, 128 * 128, 24, 24), ("b2 64x64", 32, 64 * 64, 144, 32), ("b3-5 64x64", 32, 64 * 64, 192, 32), ("b6 32x32", 32, 32 * 32, 192, 56),
          ("b7-9 32x32", 32, 32 * 32, 336, 56)]


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


for name, N, HW, Ce, Co in SHAPES:
    M = N * HW
    d = torch.randn(N, HW, Ce, device=dev)
    w = torch.randn(Co, Ce, device=dev) / Ce ** 0.5
    dp = torch.randn(M, Co, device=dev)
    s = torch.randn(N, Ce, device=dev)
    dpool = torch.randn(N, Ce, device=dev)
    gamma, beta = torch.rand(Ce, device=dev) + 0.5, torch.randn(Ce, device=dev) * 0.1
    acc = K.zeros64(2 * Ce, d)
    K.colstats(d.view(M, Ce), acc)
    bn = K.DeferredBN(acc, Ce, M, gamma, beta, 1e-3, 1)
    c = K.se_scale_bn(d, bn, s, N, HW)
    mb = 4.0 * M * Ce / 1e6

    def old_a():
        K.gemm_tn(dp, c.view(M, Ce))
        dc = K.gemm_nn(dp, w).view(N, HW, Ce)
        K.coldot_bn(dc, d, bn, N, HW, K.zeros64(N * Ce, d))
        return dc

    dc0 = old_a()

    def old_b():
        K.se_scale_bwd_bn(dc0, d, bn, s, dpool, 1.0 / HW, N, HW, K.zeros64(2 * Ce, d))

    def new_a():
        K.project_bwd_fused_a(d, bn, s, dp, w, N, HW, K.zeros64(N * Ce, d))

    def new_b():
        K.project_bwd_fused_b(d, bn, s, dpool, 1.0 / HW, dp, w, N, HW, K.zeros64(2 * Ce, d))

    def old_f():
        cc = K.se_scale_bn(d, bn, s, N, HW, want_absmax=True)
        st = K.zeros64(2 * Co, d)
        r = K.gemm_nt(cc.view(M, Ce), w, stats=st)
        if not r[1]:
            K.colstats(r[0], st)

    def new_f():
        K.project_fwd_fused(d, bn, s, w, N, HW, stats=K.zeros64(2 * Co, d))

    if K.project_fwd_fused_ok(d, w, HW):
        tf0, tf1 = timed(old_f), timed(new_f)
        print("%-12s forward: se_scale_bn + gemm_nt (+ statistics) %6.1f us | one pass %6.1f us (%.2f TB/s)" % (name, tf0, tf1, mb / tf1), flush=True)
    ta0, tb0, ta1, tb1 = timed(old_a), timed(old_b), timed(new_a), timed(new_b)
    print("%-12s M %7d %3d -> %2d (%3.0f MB per tensor) | wgrad + dgrad + coldot %6.1f us, se_scale_bwd %6.1f us | pass a %6.1f us (%.2f TB/s), pass b %6.1f us (%.2f TB/s)" %
          (name, M, Ce, Co, mb, ta0, tb0, ta1, mb / ta1, tb1, 2 * mb / tb1), flush=True)
