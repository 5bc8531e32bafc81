"""ud_pw_bwd_fused (csrc/pwbwd.hip) against the three launches it replaces (ud_normbwd_apply + gemm_tn + gemm_nn) on the thin
expand convs of UDEB4 at bs 32: graph-replayed launches, us per call and the bytes each form moves."""
import sys
import torch

sys.path.insert(0, ".")
from unidefense_amd import kernels as K          # noqa: E402

dev = torch.device("cuda:0")
FORMS = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]
SHAPES = [("b2 128x128", 32 * 128 * 128, 144, 24), ("b3-6 64x64", 32 * 64 * 64, 192, 32)]


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


for name, M, Ce, Cin in SHAPES:
    x = torch.randn(M, Cin, device=dev)
    w = torch.randn(Ce, Cin, device=dev) / Cin ** 0.5
    e = (x @ w.t()).contiguous()
    dz = torch.randn(M, Ce, device=dev)
    skip = torch.randn(M, Cin, device=dev)
    gamma, beta = torch.rand(Ce, device=dev) + 0.5, torch.randn(Ce, device=dev) * 0.1
    acc = K.zeros64(2 * Ce, x)
    K.colstats(e, acc)
    bn = K.DeferredBN(acc, Ce, M, gamma, beta, 1e-3, 1)
    sb = K.zeros64(2 * Ce, x)
    K.normbwd_sums(e.view(1, M, Ce), dz.view(1, M, Ce), None, 1.0, bn, True, 1, M, sb)

    def old():
        de, _, _ = K.normbwd_apply(e.view(1, M, Ce), dz.view(1, M, Ce), None, 1.0, bn, True, 1, M, sb, want_absmax=True)
        K.gemm_tn(de.view(M, Ce), x)
        K.gemm_nn(de.view(M, Ce), w, out=skip, accumulate=True)

    def new():
        K.expand_bwd_fused(e, dz, bn, sb, None, x, w, add=skip)

    t_old = timed(old)
    b_new = 4.0 * M * (2 * Ce + 3 * Cin)
    b_old = 4.0 * M * (3 * Ce + (Ce + Cin) + (Ce + 2 * Cin))
    print("%-12s M %7d %3d <- %2d | three launches %7.1f us (%.2f TB/s of %4.0f MB)" % (name, M, Ce, Cin, t_old, b_old / t_old * 1e-6, b_old / 1e6),
          flush=True)
    for form in FORMS:
        K._call("ud_pw_bwd_set_form", form)
        t_new = timed(new)
        print("    one pass, form %d (0 = shipped; loads in flight, workgroups per CU = (1,2) (2,2) (3,1) (4,1)): %7.1f us (%.2f TB/s of %4.0f MB)" %
              (form, t_new, b_new / t_new * 1e-6, b_new / 1e6), flush=True)
    K._call("ud_pw_bwd_set_form", 0)
