"""GPU: randomized ud_gemm configurations (modes 0/1, ragged sizes, leading dimensions, out modes, split-K, batch)
on the split-bf16 path against float64."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K, lib

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
lib.call("ud_gemm_set_path", 2)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    am, bm = random.choice([(0, 0), (0, 1), (1, 1)])
    M = random.choice([4, 64, 68, 128, 132, 256, 320, 1152, 516]) * (1 if am == 1 else random.choice([1, 1, 3])) // (1 if am == 1 else 1)
    M = M - (M % 4) if am == 1 else M + random.choice([0, 1, 2, 3])
    N = random.choice([4, 64, 72, 128, 136, 256, 512, 388])
    N = N - (N % 4) if bm == 1 else N + random.choice([0, 1, 2, 3])
    Kd = random.choice([4, 16, 36, 64, 100, 256, 272, 1000, 2048])
    if am == 1 and bm == 1:
        Kd += random.choice([0, 1, 2, 3])
    batch = random.choice([1, 1, 1, 3])
    pad_a, pad_b, pad_c = (random.choice([0, 4, 8]) for _ in range(3))
    split = random.choice([1, 1, 2, 5]) if Kd >= 64 else 1
    out_mode = 2 if split > 1 else random.choice([0, 1])
    # operand storage
    if am == 0:
        A = torch.randn(batch, M, Kd + pad_a, device=dev); lda = Kd + pad_a; Av = A[:, :, :Kd]
    else:
        A = torch.randn(batch, Kd, M + pad_a, device=dev); lda = M + pad_a; Av = A[:, :, :M].transpose(1, 2)
    if bm == 0:
        B = torch.randn(batch, N, Kd + pad_b, device=dev); ldb = Kd + pad_b; Bv = B[:, :, :Kd].transpose(1, 2)
    else:
        B = torch.randn(batch, Kd, N + pad_b, device=dev); ldb = N + pad_b; Bv = B[:, :, :N]
    Cb = torch.randn(batch, M, N + pad_c, device=dev) if out_mode == 1 else torch.zeros(batch, M, N + pad_c, device=dev)
    C0 = Cb.clone()
    ref = Av.double() @ Bv.double()
    if out_mode == 1:
        ref = ref + C0[:, :, :N].double()
    K._gemm(A, B, Cb, M, N, Kd, lda, ldb, N + pad_c, am, bm, out_mode, split, None, batch, A.stride(0), B.stride(0), Cb.stride(0))
    scale = (Av.double().abs() @ Bv.double().abs()) + 1e-30
    err = ((Cb[:, :, :N].double() - ref).abs() / scale).max().item()
    untouched = (Cb[:, :, N:] == C0[:, :, N:]).all().item()
    worst = max(worst, err)
    if err > 2e-6 or not untouched:
        print("BAD", dict(am=am, bm=bm, M=M, N=N, K=Kd, batch=batch, lda=lda, ldb=ldb, ldc=N + pad_c, split=split, out=out_mode), "err %.3e" % err, "pad untouched", untouched, flush=True)
print("worst err", worst)
