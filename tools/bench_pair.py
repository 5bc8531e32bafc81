"""How much would launching a 1x1 conv's data gradient (dY W) and weight gradient (dY^T X) TOGETHER buy?  Times the two
ud_gemm launches back to back on one stream against the same two launches on two forked streams (both replayed from a
hipGraph), for the (M, Cin, Cout) of the model's MBConv stages at bs 32.  Upper bound for a paired-launch kernel."""
import sys
import torch

sys.path.insert(0, ".")
from unidefense_amd import kernels as K          # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("project 64^2", 131072, 192, 32), ("project 32^2", 32768, 336, 56), ("project 16^2a", 8192, 672, 112),
          ("project 16^2b", 8192, 960, 160), ("project 8^2", 2048, 1632, 272), ("project 8^2b", 2048, 2688, 448),
          ("expand 64^2", 131072, 32, 192), ("expand 16^2", 8192, 160, 960), ("expand 8^2", 2048, 272, 1632),
          ("freq 32^2", 17408, 672, 672), ("freq 16^2a", 4608, 1344, 1344), ("freq 16^2b", 4608, 1920, 1920),
          ("freq 8^2", 1280, 3264, 3264)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
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


s2 = torch.cuda.Stream()
print("%-14s %8s %6s %6s | %8s %8s %8s %8s  (us)" % ("layer", "M", "Cin", "Cout", "dgrad", "wgrad", "seq", "forked"))
for name, M, Ci, Co in SHAPES:
    dy = torch.randn(M, Co, device=dev)
    x = torch.randn(M, Ci, device=dev)
    w = torch.randn(Co, Ci, device=dev)
    K.gemm_nn(dy, w)
    K.gemm_tn(dy, x)         # tune both first (eager)
    t_d = timed(lambda: K.gemm_nn(dy, w))
    t_w = timed(lambda: K.gemm_tn(dy, x))

    def seq():
        K.gemm_nn(dy, w)
        K.gemm_tn(dy, x)

    def forked():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            K.gemm_tn(dy, x)
        K.gemm_nn(dy, w)
        cur.wait_stream(s2)
    t_s, t_f = timed(seq), timed(forked)
    print("%-14s %8d %6d %6d | %8.1f %8.1f %8.1f %8.1f" % (name, M, Ci, Co, t_d, t_w, t_s, t_f))
