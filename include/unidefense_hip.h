/*
 * unidefense_hip.h — C ABI of libunidefense_hip.so (gfx950 / MI355X).
 *
 * The reference (VISION-SJTU/UniDefense) has no native code and no FFI: its hot path reaches the
 * GPU through torch.nn / torch.fft (ATen -> vendor libraries).  Each entry point below replaces
 * one of those implicit kernels; the comment on each names the reference call site it serves
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every tensor is fp32, contiguous, "pixel-major" (NHWC): [N][H][W][C], i.e. a row-major
 *     matrix [M = N*H*W][C]; image-domain tensors with C = 3 are planes [N*C][H][W] (NCHW).
 *   - plain pointers + sizes; the caller owns every buffer (no allocation, no sync inside).
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); functions are
 *     graph-capture safe.  Return 0 on success, a negative hipError_t on launch failure,
 *     UD_EINVAL (-1000) on invalid arguments.
 */
#ifndef UNIDEFENSE_HIP_H
#define UNIDEFENSE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UD_EINVAL (-1000)

typedef void* ud_stream_t;

/* ---- conv geometry used by the gather modes of ud_gemm ------------------------------------
 * rows r = (n, oh, ow) over [N][Hout][Wout]; taps (kh, kw); source pixel of (r, tap):
 *   transposed == 0 :  ih = oh*stride - pad_t + kh                       (F.conv2d)
 *   transposed == 1 :  t = oh + pad_t - kh; valid iff t % stride == 0;  ih = t / stride
 *                                                                          (F.conv_transpose2d)
 * out-of-range sources read as 0. */
typedef struct {
    int N, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_t, pad_l, transposed;
} ud_conv_geom;

/* ---- ud_gemm: C[M][N] (+)= A . B on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) ---------
 * a_mode 0: A[m][k] at A + m*lda + k          (activations [pixels][Cin])
 *        1: A[k][m] at A + k*lda + m          (dY as the A operand of a weight gradient)
 *        2: conv gather: m = (n,oh,ow), k = (kh*KW+kw)*Cin + ci -> In[n][ih][iw][ci]  (geom g)
 * b_mode 0: B[n][k] at B + n*ldb + k          (weights [Cout][K])
 *        1: B[k][n] at B + k*ldb + n          (weights for the data gradient, X for the weight gradient)
 *        2: conv gather with k = (n,oh,ow) and n = (kh*KW+kw)*Cin + ci        (weight gradient of a conv)
 * out_mode 0: store, 1: C += result, 2: atomicAdd (used when split_k > 1; C must be pre-zeroed)
 * Serves: F.conv2d 1x1 in model/efficientnet/model.py:108,125 and exp.py:57 (freq_conv),
 *         nn.Conv2d 3x3 / nn.ConvTranspose2d in model/unidefense.py:59-102, model/modules.py:82,111,
 *         nn.Linear in model/modules.py:27, and their autograd backward (convolution_backward). */
typedef struct {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    int a_mode, b_mode, out_mode, split_k;
    int batch; long strideA, strideB, strideC;   /* batch >= 1; element strides between batches */
    ud_conv_geom g;
} ud_gemm_desc;
int ud_gemm(const ud_gemm_desc* d, ud_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
