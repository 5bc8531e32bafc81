"""GPU: accuracy (vs float64) and speed of the two ud_gemm arithmetic paths: 1 = v_mfma_f32_32x32x2_f32,
2 = split-bf16 (three exact bf16 pieces per operand, six products) on the model's shapes and ragged ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K, lib

SHAPES = [
    (4096, 4096, 4096, "nt"), (1280, 3264, 3264, "nt"), (1280, 3264, 3264, "nn"), (3264, 3264, 1280, "tn"),
    (4608, 1920, 1920, "nt"), (4608, 1920, 1920, "nn"), (1920, 1920, 4608, "tn"),
    (4608, 1344, 1344, "nt"), (1344, 1344, 4608, "tn"), (17408, 672, 672, "nt"), (672, 672, 17408, "tn"),
    (67584, 384, 384, "nt"), (384, 384, 67584, "tn"), (8192, 960, 160, "nt"), (8192, 160, 960, "nn"),
    (160, 960, 8192, "tn"), (2048, 1632, 272, "nt"), (2048, 272, 1632, "nn"), (272, 1632, 2048, "tn"),
    (200, 72, 100, "nt"), (132, 68, 36, "nn"), (68, 76, 1000, "tn"), (64, 64, 64, "nt"),
]


def run(kind, a, b):
    return {"nt": K.gemm_nt, "nn": K.gemm_nn, "tn": K.gemm_tn}[kind](a, b)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("%-4s %7s %5s %7s | %10s %10s | %8s %8s" % ("kind", "M", "N", "K", "err fp32", "err x3", "TF fp32", "TF x3"))
    for M, N, Kd, kind in SHAPES:
        if kind == "nt":
            a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
            ref = a.double() @ b.double().t()
        elif kind == "nn":
            a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev)
            ref = a.double() @ b.double()
        else:
            a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
            ref = a.double().t() @ b.double()
        # wide dynamic range inside the operands (exercises all three pieces)
        a *= torch.exp(2.0 * torch.randn_like(a)); b *= torch.exp(2.0 * torch.randn_like(b))
        ref = {"nt": lambda: a.double() @ b.double().t(), "nn": lambda: a.double() @ b.double(),
               "tn": lambda: a.double().t() @ b.double()}[kind]()
        scale = {"nt": lambda: a.double().abs() @ b.double().abs().t(), "nn": lambda: a.double().abs() @ b.double().abs(),
                 "tn": lambda: a.double().abs().t() @ b.double().abs()}[kind]()
        out = []
        for path in (1, 2):
            lib.call("ud_gemm_set_path", path)
            y = run(kind, a, b)
            err = ((y.double() - ref).abs() / scale).max().item()      # relative to sum |a||b|
            for _ in range(3):
                run(kind, a, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 10
            e0.record()
            for _ in range(it):
                run(kind, a, b)
            e1.record()
            torch.cuda.synchronize()
            out.append((err, 2.0 * M * N * Kd / (e0.elapsed_time(e1) / it) / 1e9))
        print("%-4s %7d %5d %7d | %10.3e %10.3e | %8.1f %8.1f" % (kind, M, N, Kd, out[0][0], out[1][0], out[0][1], out[1][1]),
              flush=True)
    lib.call("ud_gemm_set_path", 0)


if __name__ == "__main__":
    main()
