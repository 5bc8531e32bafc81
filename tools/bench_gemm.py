"""GPU microbenchmark of ud_gemm over the model's shapes and the tile configurations (tuning aid).
usage: python tools/bench_gemm.py [cfg ...]   (each cfg runs in a fresh subprocess with UD_GEMM_CFG set)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (M, N, K, kind)   kind: nt = forward, nn = dgrad, tn = wgrad (M,N = output, K = pixels)
    (4096, 4096, 4096, "nt"),
    (1280, 3264, 3264, "nt"), (1280, 3264, 3264, "nn"), (3264, 3264, 1280, "tn"),
    (4608, 1920, 1920, "nt"), (1920, 1920, 4608, "tn"),
    (4608, 1344, 1344, "nt"), (1344, 1344, 4608, "tn"),
    (17408, 672, 672, "nt"), (672, 672, 17408, "tn"),
    (67584, 384, 384, "nt"), (384, 384, 67584, "tn"),
    (524288, 144, 24, "nt"), (524288, 24, 144, "nt"), (144, 24, 524288, "tn"), (24, 144, 524288, "tn"),
    (131072, 192, 32, "nt"), (131072, 32, 192, "nt"), (32, 192, 131072, "tn"), (192, 32, 131072, "tn"),
    (8192, 960, 160, "nt"), (8192, 160, 960, "nt"), (160, 960, 8192, "tn"),
    (2048, 1632, 272, "nt"), (2048, 272, 1632, "nt"), (272, 1632, 2048, "tn"),
]


def worker():
    import torch
    from unidefense_amd import kernels as K
    dev = torch.device("cuda:0")
    for M, N, Kd, kind in SHAPES:
        if kind == "nt":
            a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
            fn = lambda: K.gemm_nt(a, b)
        elif kind == "nn":
            a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev)
            fn = lambda: K.gemm_nn(a, b)
        else:
            a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
            fn = lambda: K.gemm_tn(a, b)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        print("cfg %s %s %7d %5d %7d  %8.3f ms  %6.1f TF" % (os.environ.get("UD_GEMM_CFG", "auto"), kind, M, N, Kd, ms,
                                                          2.0 * M * N * Kd / ms / 1e9), flush=True)


if __name__ == "__main__":
    if os.environ.get("_UD_WORKER"):
        worker()
    else:
        cfgs = sys.argv[1:] or ["auto", "0", "1", "4"]
        for c in cfgs:
            env = dict(os.environ, _UD_WORKER="1")
            if c != "auto":
                env["UD_GEMM_CFG"] = c
            else:
                env.pop("UD_GEMM_CFG", None)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
