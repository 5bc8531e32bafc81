"""GPU: 256 x 256 rfft2 of 96 image planes (the bs-32 loss tail): row/column FFT kernels vs the DFT-GEMM stand-in vs torch.fft."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
for S, P in ((256, 96), (320, 48), (128, 24)):
    x = torch.randn(P, S, S, device=dev)
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    t_new = timeit(lambda: K.dft_rfft2_planes(x))
    os.environ["UD_FFT_PLANES_GEMM"] = "1"
    t_gemm = timeit(lambda: K.dft_rfft2_planes(x))
    os.environ["UD_FFT_PLANES_GEMM"] = "0"
    t_torch = timeit(lambda: torch.fft.rfft2(x, norm="ortho"))
    Y = K.dft_rfft2_planes(x)
    t_adj = timeit(lambda: K.dft_rfft2_planes_adjoint(Y, S))
    nbytes = 4 * P * S * S + 8 * P * S * (S // 2 + 1)
    print(f"S={S} P={P}: fft kernels {t_new:.1f} us ({nbytes / t_new / 1e3:.0f} GB/s of in+out), adjoint {t_adj:.1f} us, "
          f"DFT-GEMM {t_gemm:.1f} us, torch.fft (hipFFT) {t_torch:.1f} us")
