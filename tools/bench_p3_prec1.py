"""GPU: the planes kernel reading ONE fp16 plane per operand (ud_gemm_p3 prec 1: mixed precision, BASELINE configs[4]) against the
kernel that mode uses today (gemm_x3_kernel, one fp16 piece split in the k-loop, half-stored activations) and against prec 2, on
the fat spectral-conv shapes of the bs-64 step; with the error of each against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K, lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [("nt", 2560, 3264, 3264), ("nt", 9216, 1920, 1920), ("nt", 9216, 1344, 1344), ("nt", 34816, 672, 672),
          ("nn", 9216, 1920, 1920), ("tn", 1920, 1920, 9216), ("tn", 3264, 3264, 2560), ("nt", 4096, 272, 1632)]
print("kind M N K | x3 half-stored (ms, TFLOP/s, err) | p3 prec 1 (ms, TFLOP/s, err) | p3 prec 2 (ms, TFLOP/s)")
for kind, M, N, Kd in shapes:
    am, bm = K._P2_MODES[kind]
    if kind == "nt":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev) * 0.05
        ref = a.double() @ b.double().t()
    elif kind == "nn":
        a, b = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev) * 0.05
        ref = a.double() @ b.double()
    else:
        a, b = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
        ref = a.double().t() @ b.double()
    flop = 2.0 * M * N * Kd
    scale = float(ref.abs().max())
    # today's f16 path: half-stored activation operand(s), fp32 weight, path 3
    lib.call("ud_gemm_set_path", 3)
    try:
        if kind == "nt":
            f = lambda: K.gemm_nt(a.half(), b)
            ah = a.half()
            f = lambda: K.gemm_nt(ah, b)
        elif kind == "nn":
            ah = a.half()
            f = lambda: K.gemm_nn(ah, b)
        else:
            ah, bh = a.half(), b.half()
            f = lambda: K.gemm_tn(ah, bh)
        out = f()
        torch.cuda.synchronize()
        e_x3 = float((out.double() - ref).abs().max()) / scale
        t_x3 = K._time_launches(f)
    finally:
        lib.call("ud_gemm_set_path", 0)
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    o = torch.empty(M, N, device=dev)
    f2 = lambda: K._gemm_p3(ap, bp, o, M, N, Kd, am, bm)
    t_p2 = K._time_launches(f2)
    ap.prec = bp.prec = 1
    f1 = lambda: K._gemm_p3(ap, bp, o, M, N, Kd, am, bm)
    f1()
    torch.cuda.synchronize()
    e_p1 = float((o.double() - ref).abs().max()) / scale
    t_p1 = K._time_launches(f1)
    print(f"{kind} {M} {N} {Kd} | {t_x3:.4f} {flop / t_x3 / 1e9:.0f} {e_x3:.1e} | {t_p1:.4f} {flop / t_p1 / 1e9:.0f} {e_p1:.1e} | "
          f"{t_p2:.4f} {flop / t_p2 / 1e9:.0f}", flush=True)
