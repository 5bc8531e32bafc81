// Batched small real 2-D FFT (S x S, S = 8, 16, 32, 64; 10, 20, 40, 80; 12, 24, 48) on pixel-major [N][S][S][C] fp32, and its inverse.
//
// Serves torch.fft.rfft2 / irfft2 of SFConv2dStaticSamePadding.forward (model/efficientnet/exp.py:55,60),
// of UniDefenseModelEb4.attention (model/unidefense.py:130-145), and both autograd adjoints:
//     d(rfft2)  = irfft2-kernel with interior columns weighted 1/2
//     d(irfft2) = rfft2-kernel  with interior columns weighted 2
// (half-spectrum conventions of pocketfft/cuFFT c2r: the imaginary parts of the kx = 0 and kx = S/2
// columns — after the column transform — are ignored).
//
// Layout choice: the channel is the LANE.  Each lane transforms its own (n, c) plane, so there are no
// cross-lane butterflies at all, every global access of a wave is one contiguous run of channels, and
// the output is written directly in the [pixels][Re(0..C) | Im(0..C)] operand layout of the spectral
// 1x1 GEMM (the torch.cat / tensor_split / torch.complex shuffles of exp.py:56,59 cost nothing).
// A workgroup of 512 threads owns CB channels of one image (S * CB = 512):
//   pass 1: thread (h, c) does the S-point row transform in registers      -> LDS [kx][h][c]
//   pass 2: thread (kx, c) does the S-point column transform in registers  -> global
// (inverse: columns first, then Hermitian-extended rows).  HBM traffic is the algorithmic minimum:
// one read of the input and one write of the output.
#include "ud_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int NT = 512;

__device__ __forceinline__ float gate_factor_f(const float* alpha, int mode) {
    if (mode == 0 || !alpha) return 1.f;
    const float a = ud_sigmoid(alpha[0]);
    return mode == 1 ? a : 1.f - a;
}

// exp(-2*pi*i*j/64), j = 0..31
__device__ constexpr float TW_RE[32] = {
    1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f,
    8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f,
    3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 0.0f, -9.801714033e-02f,
    -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f,
    -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f,
    -9.807852804e-01f, -9.951847267e-01f};
__device__ constexpr float TW_IM[32] = {
    0.0f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f,
    -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f,
    -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f,
    -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f,
    -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f,
    -1.950903220e-01f, -9.801714033e-02f};

template <int S>
__device__ __forceinline__ constexpr int brev2(int i) {
    int r = 0;
    for (int b = 1; b < S; b <<= 1) {
        r = (r << 1) | (i & 1);
        i >>= 1;
    }
    return r;
}

// exp(-2*pi*i*j/80), j = 0..79  (mixed-radix sizes 10, 20, 40, 80)
__device__ constexpr float TW80_RE[80] = {
    1.000000000e+00f, 9.969173337e-01f, 9.876883406e-01f, 9.723699204e-01f, 9.510565163e-01f, 9.238795325e-01f,
    8.910065242e-01f, 8.526401644e-01f, 8.090169944e-01f, 7.604059656e-01f, 7.071067812e-01f, 6.494480483e-01f,
    5.877852523e-01f, 5.224985647e-01f, 4.539904997e-01f, 3.826834324e-01f, 3.090169944e-01f, 2.334453639e-01f,
    1.564344650e-01f, 7.845909573e-02f, 6.123233996e-17f, -7.845909573e-02f, -1.564344650e-01f, -2.334453639e-01f,
    -3.090169944e-01f, -3.826834324e-01f, -4.539904997e-01f, -5.224985647e-01f, -5.877852523e-01f, -6.494480483e-01f,
    -7.071067812e-01f, -7.604059656e-01f, -8.090169944e-01f, -8.526401644e-01f, -8.910065242e-01f, -9.238795325e-01f,
    -9.510565163e-01f, -9.723699204e-01f, -9.876883406e-01f, -9.969173337e-01f, -1.000000000e+00f, -9.969173337e-01f,
    -9.876883406e-01f, -9.723699204e-01f, -9.510565163e-01f, -9.238795325e-01f, -8.910065242e-01f, -8.526401644e-01f,
    -8.090169944e-01f, -7.604059656e-01f, -7.071067812e-01f, -6.494480483e-01f, -5.877852523e-01f, -5.224985647e-01f,
    -4.539904997e-01f, -3.826834324e-01f, -3.090169944e-01f, -2.334453639e-01f, -1.564344650e-01f, -7.845909573e-02f,
    -1.836970199e-16f, 7.845909573e-02f, 1.564344650e-01f, 2.334453639e-01f, 3.090169944e-01f, 3.826834324e-01f,
    4.539904997e-01f, 5.224985647e-01f, 5.877852523e-01f, 6.494480483e-01f, 7.071067812e-01f, 7.604059656e-01f,
    8.090169944e-01f, 8.526401644e-01f, 8.910065242e-01f, 9.238795325e-01f, 9.510565163e-01f, 9.723699204e-01f,
    9.876883406e-01f, 9.969173337e-01f};
__device__ constexpr float TW80_IM[80] = {
    -0.000000000e+00f, -7.845909573e-02f, -1.564344650e-01f, -2.334453639e-01f, -3.090169944e-01f, -3.826834324e-01f,
    -4.539904997e-01f, -5.224985647e-01f, -5.877852523e-01f, -6.494480483e-01f, -7.071067812e-01f, -7.604059656e-01f,
    -8.090169944e-01f, -8.526401644e-01f, -8.910065242e-01f, -9.238795325e-01f, -9.510565163e-01f, -9.723699204e-01f,
    -9.876883406e-01f, -9.969173337e-01f, -1.000000000e+00f, -9.969173337e-01f, -9.876883406e-01f, -9.723699204e-01f,
    -9.510565163e-01f, -9.238795325e-01f, -8.910065242e-01f, -8.526401644e-01f, -8.090169944e-01f, -7.604059656e-01f,
    -7.071067812e-01f, -6.494480483e-01f, -5.877852523e-01f, -5.224985647e-01f, -4.539904997e-01f, -3.826834324e-01f,
    -3.090169944e-01f, -2.334453639e-01f, -1.564344650e-01f, -7.845909573e-02f, -1.224646799e-16f, 7.845909573e-02f,
    1.564344650e-01f, 2.334453639e-01f, 3.090169944e-01f, 3.826834324e-01f, 4.539904997e-01f, 5.224985647e-01f,
    5.877852523e-01f, 6.494480483e-01f, 7.071067812e-01f, 7.604059656e-01f, 8.090169944e-01f, 8.526401644e-01f,
    8.910065242e-01f, 9.238795325e-01f, 9.510565163e-01f, 9.723699204e-01f, 9.876883406e-01f, 9.969173337e-01f,
    1.000000000e+00f, 9.969173337e-01f, 9.876883406e-01f, 9.723699204e-01f, 9.510565163e-01f, 9.238795325e-01f,
    8.910065242e-01f, 8.526401644e-01f, 8.090169944e-01f, 7.604059656e-01f, 7.071067812e-01f, 6.494480483e-01f,
    5.877852523e-01f, 5.224985647e-01f, 4.539904997e-01f, 3.826834324e-01f, 3.090169944e-01f, 2.334453639e-01f,
    1.564344650e-01f, 7.845909573e-02f};

// exp(-2*pi*i*j/48), j = 0..47  (mixed-radix sizes 12, 24, 48: the feature maps of the EfficientNet-b4 trunk at its native 380 x 380)
__device__ constexpr float TW48_RE[48] = {
    1.000000000e+00f, 9.914448614e-01f, 9.659258263e-01f, 9.238795325e-01f, 8.660254038e-01f, 7.933533403e-01f,
    7.071067812e-01f, 6.087614290e-01f, 5.000000000e-01f, 3.826834324e-01f, 2.588190451e-01f, 1.305261922e-01f,
    6.123233996e-17f, -1.305261922e-01f, -2.588190451e-01f, -3.826834324e-01f, -5.000000000e-01f, -6.087614290e-01f,
    -7.071067812e-01f, -7.933533403e-01f, -8.660254038e-01f, -9.238795325e-01f, -9.659258263e-01f, -9.914448614e-01f,
    -1.000000000e+00f, -9.914448614e-01f, -9.659258263e-01f, -9.238795325e-01f, -8.660254038e-01f, -7.933533403e-01f,
    -7.071067812e-01f, -6.087614290e-01f, -5.000000000e-01f, -3.826834324e-01f, -2.588190451e-01f, -1.305261922e-01f,
    -1.836970199e-16f, 1.305261922e-01f, 2.588190451e-01f, 3.826834324e-01f, 5.000000000e-01f, 6.087614290e-01f,
    7.071067812e-01f, 7.933533403e-01f, 8.660254038e-01f, 9.238795325e-01f, 9.659258263e-01f, 9.914448614e-01f};
__device__ constexpr float TW48_IM[48] = {
    -0.000000000e+00f, -1.305261922e-01f, -2.588190451e-01f, -3.826834324e-01f, -5.000000000e-01f, -6.087614290e-01f,
    -7.071067812e-01f, -7.933533403e-01f, -8.660254038e-01f, -9.238795325e-01f, -9.659258263e-01f, -9.914448614e-01f,
    -1.000000000e+00f, -9.914448614e-01f, -9.659258263e-01f, -9.238795325e-01f, -8.660254038e-01f, -7.933533403e-01f,
    -7.071067812e-01f, -6.087614290e-01f, -5.000000000e-01f, -3.826834324e-01f, -2.588190451e-01f, -1.305261922e-01f,
    -1.224646799e-16f, 1.305261922e-01f, 2.588190451e-01f, 3.826834324e-01f, 5.000000000e-01f, 6.087614290e-01f,
    7.071067812e-01f, 7.933533403e-01f, 8.660254038e-01f, 9.238795325e-01f, 9.659258263e-01f, 9.914448614e-01f,
    1.000000000e+00f, 9.914448614e-01f, 9.659258263e-01f, 9.238795325e-01f, 8.660254038e-01f, 7.933533403e-01f,
    7.071067812e-01f, 6.087614290e-01f, 5.000000000e-01f, 3.826834324e-01f, 2.588190451e-01f, 1.305261922e-01f};

// S = 2^k, 5 * 2^k (10, 20, 40, 80: the feature maps of the ResNet50 variant at 320 x 320 inputs) or 3 * 2^k (12, 24, 48: the
// EfficientNet-b4 trunk at 380 x 380)
template <int S>
struct Radix {
    static constexpr int R = (S % 5 == 0) ? 5 : (S % 3 == 0) ? 3 : 1;      // the odd factor
    static constexpr bool MIXED = R > 1;
    static constexpr int P = S / R;                         // length of the radix-2 part
    static constexpr int L = R == 5 ? 80 : 48;              // length of the twiddle table W_L^j the size draws from
};
template <int S> __device__ __forceinline__ constexpr float tw_re(int m) { return Radix<S>::R == 5 ? TW80_RE[m] : TW48_RE[m]; }
template <int S> __device__ __forceinline__ constexpr float tw_im(int m) { return Radix<S>::R == 5 ? TW80_IM[m] : TW48_IM[m]; }

// register slot of input element i: bit reversal for 2^k; for 5*P the decimated sequence r = i % 5 occupies slots
// [r*P, (r+1)*P) in bit-reversed order of n2 = i / 5
template <int S>
__device__ __forceinline__ constexpr int brev(int i) {
    if (Radix<S>::MIXED) return (i % Radix<S>::R) * Radix<S>::P + brev2<Radix<S>::P>(i / Radix<S>::R);
    return brev2<S>(i);
}

// In-register radix-2 DIT on slots [OFF, OFF + P).  Input bit-reversed, output natural.  Fully unrolled: every
// index and twiddle is a compile-time constant, so re[]/im[] live in VGPRs.
template <int S, int P, int OFF, bool INV>
__device__ __forceinline__ void fft_pow2(float (&re)[S], float (&im)[S]) {
#pragma unroll
    for (int len = 2; len <= P; len <<= 1) {
        const int hl = len >> 1;
        const int tstep = 64 / len;
#pragma unroll
        for (int i = 0; i < P; i += len) {
#pragma unroll
            for (int j = 0; j < hl; ++j) {
                const float wr = TW_RE[j * tstep];
                const float wi = INV ? -TW_IM[j * tstep] : TW_IM[j * tstep];
                const float xr = re[OFF + i + j + hl], xi = im[OFF + i + j + hl];
                const float tr = wr * xr - wi * xi;
                const float ti = wr * xi + wi * xr;
                const float ur = re[OFF + i + j], ui = im[OFF + i + j];
                re[OFF + i + j] = ur + tr;
                im[OFF + i + j] = ui + ti;
                re[OFF + i + j + hl] = ur - tr;
                im[OFF + i + j + hl] = ui - ti;
            }
        }
    }
}

// S-point DFT in registers, input in brev<S> slots, output in natural order.  R*P sizes (R = 5 or 3): Cooley-Tukey with N1 = R:
// R P-point FFTs of the decimated sequences, twiddles W_S^(r*k2), then an R-point DFT across r for every k2 —
// X[k2 + P*k1] = sum_r W_R^(r*k1) W_S^(r*k2) Y_r[k2] lands in the slots its inputs came from.
template <int S, int P, int RR, bool INV>
struct SubFfts {
    static __device__ __forceinline__ void run(float (&re)[S], float (&im)[S]) {
        SubFfts<S, P, RR - 1, INV>::run(re, im);
        fft_pow2<S, P, (RR - 1) * P, INV>(re, im);
    }
};
template <int S, int P, bool INV>
struct SubFfts<S, P, 0, INV> {
    static __device__ __forceinline__ void run(float (&)[S], float (&)[S]) {}
};

template <int S, bool INV>
__device__ __forceinline__ void fft_inreg(float (&re)[S], float (&im)[S]) {
    constexpr int P = Radix<S>::P, R = Radix<S>::R, L = Radix<S>::L;
    if constexpr (!Radix<S>::MIXED) {
        fft_pow2<S, P, 0, INV>(re, im);
    } else {
        SubFfts<S, P, R, INV>::run(re, im);
        constexpr int TS = L / S;                          // W_S^m = W_L^(m*TS)
#pragma unroll
        for (int k2 = 0; k2 < P; ++k2) {
            float yr[R], yi[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = (r * k2 * TS) % L;
                const float wr = tw_re<S>(m), wi = INV ? -tw_im<S>(m) : tw_im<S>(m);
                const float xr = re[r * P + k2], xi = im[r * P + k2];
                yr[r] = wr * xr - wi * xi;
                yi[r] = wr * xi + wi * xr;
            }
#pragma unroll
            for (int k1 = 0; k1 < R; ++k1) {
                float ar = yr[0], ai = yi[0];
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    const int m = ((L / R) * r * k1) % L;  // W_R^(r*k1)
                    const float wr = tw_re<S>(m), wi = INV ? -tw_im<S>(m) : tw_im<S>(m);
                    ar += wr * yr[r] - wi * yi[r];
                    ai += wr * yi[r] + wi * yr[r];
                }
                re[k1 * P + k2] = ar;
                im[k1 * P + k2] = ai;
            }
        }
    }
}

// XCD-aware work order.  Workgroups are dealt round-robin to the 8 XCDs by linear id, so with the plain (channel group,
// sample) = blockIdx map the channel groups that share a 128-byte line of a pixel (CB = 8 / 16 channels: 32 / 64-byte
// runs) land on different XCDs and every one of them pulls the whole line into its own L2.  Here XCD j works through a
// CONTIGUOUS range of the (sample, channel group) items: neighbouring channel groups of a sample meet in one L2.
// (applied only where a workgroup's run of channels is shorter than a 128-byte line; passed as xcd_remap.)
__device__ __forceinline__ void work_item(int xcd_remap, int& cgroup, int& n) {
    const int ngroups = gridDim.x;
    if (!xcd_remap) {
        cgroup = blockIdx.x;
        n = blockIdx.y;
        return;
    }
    const int W = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = W >> 3, r = W & 7, xcd = L & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);      // bijective for any W
    cgroup = w % ngroups;
    n = w / ngroups;
}

template <int S, int CB>
struct Lds {
    static constexpr int WH = S / 2 + 1;
    static constexpr int KSTRIDE = S * CB + (CB < 32 ? CB : 0);   // skew so that a 32-lane half never aliases banks
    static constexpr int PLANE = WH * KSTRIDE;                     // floats per (re|im) plane
    static constexpr size_t BYTES = 2ull * PLANE * sizeof(float);
};

// Y[n][ky][kx][c] = f(kx) * sum_{h,w} x[n][h][w][c] e^{-2 pi i (ky h + kx w)/S};  Re at channel c, Im at C + c.
// f(kx) = scale for kx in {0, S/2}, scale * w_int otherwise.
// EX: the input is act(bn(x)) of a deferred BatchNorm (coefficients from the fp64 sums, one channel per thread),
// optionally also written out (act_out), and the result carries the gate factor (include/unidefense_hip.h).
// Round 5: the result written DIRECTLY as the fp16 x 2 planes the spectral GEMM reads (ud_gemm_p3 prec 2, P32 panel layout over the
// matrix [N S (S/2+1)] x [Re 0..C | Im 0..C]) instead of fp32 + a split pass (ud_split_planes_h2t: one more read and write of the
// tensor, 48 launches per step).  The planes' power-of-two scale needs the tensor's |Y|max BEFORE the first element is written;
// an UPPER BOUND serves as well — it costs log2(bound / max) of the 18 binades in which an element keeps its full 22 bits, nothing
// of the precision of the large elements — and one is known a priori: |Y[k]| <= f_max S sqrt(sum_hw a^2) (Cauchy-Schwarz), and
// the energy of ONE plane is at most the channel's over the whole batch:
//   input a = act(bn(x)), |act(z)| <= |z|:   sum a_c^2 <= count (gamma_c^2 + beta_c^2)              (energy == NULL)
//   any input with a per-channel energy bound handed in:   sum a_c^2 <= energy[c]                  (energy != NULL)
// -> bound = pre * sqrt(max_c ...), pre = f_max * S [* sqrt(count)] from the host.  Every workgroup derives the same bound from
// the same numbers (a scan of C values), so all write with one scale; block (0, 0) stores 1 / scale.
struct PlanesOut {
    uint16_t* buf;          // NULL: fp32 result in Y
    long panel, plane;      // strides of the P32 layout (16-bit elements)
    float* inv_scale;
    float pre;
    const double* energy;
};

// DW = 3 / 5 (round 5): the stride-1 depthwise conv of the SAME activated plane (SFConv's spatial branch, exp.py:49-51, pads
// (DW - 1) / 2) computed by this kernel too: the (n, c) plane act(bn(x)) is in this workgroup's registers anyway, so after the
// transform it is laid into the (now free) LDS planes and every row-thread slides the DW x DW window over it — the separate conv
// kernel (one more read of x with the BatchNorm + activation re-evaluated, one launch) is not needed.
struct DwOut {
    const float* wt;          // tap-major [DW*DW][C]
    void* out;                // [N][S][S][C], storage type T
};

template <typename T, int S, int CB, bool EX, int DW = 0>
__global__ __launch_bounds__(NT) void rfft2_kernel(const T* __restrict__ x, T* __restrict__ Y, int C,
                                                   float scale, float w_int, ud_bn_ref bn, int has_bn,
                                                   T* __restrict__ act_out, const float* __restrict__ gate_alpha,
                                                   int gate_mode, const double* __restrict__ gate_acc,
                                                   float* __restrict__ gate_grad, int xcd_remap,
                                                   uint32_t* __restrict__ amax, PlanesOut po, DwOut dwo) {
    using L = Lds<S, CB>;
    static_assert(DW == 0 || (size_t)S * S * CB * sizeof(float) <= L::BYTES, "the activated plane fits the transform's LDS planes");
    __shared__ float po_red[NT / 64];
    // po.pre < 0 (half storage, the mixed-precision mode): the half result itself laid into ONE plane, scale 1 — no bound needed
    const bool po_raw = po.pre < 0.f;
    if (EX && po.buf && !po_raw) {
        // max over ALL channels of the per-channel energy bound: this thread's share, folded per wave; the workgroup's fold
        // happens behind the barrier between the two passes
        float gm = 0.f;
        for (int i = threadIdx.x; i < C; i += NT) {
            if (po.energy) gm = fmaxf(gm, (float)po.energy[i]);
            else gm = fmaxf(gm, bn.gamma[i] * bn.gamma[i] + bn.beta[i] * bn.beta[i]);
        }
        gm = ud_wave_max(gm);
        if ((threadIdx.x & 63) == 0) po_red[threadIdx.x >> 6] = gm;
    }
    // EX, backward of the SF mix: the 64 slots a preceding kernel (ud_normbwd_apply_mix) filled with
    // sum dd * (freq - spat) become the gate's gradient here, by one wave, instead of a launch of their own
    if (EX && gate_grad && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        const double tot = ud_wave_sum_d(gate_acc[threadIdx.x]);
        if (threadIdx.x == 0) {
            const double a = 1.0 / (1.0 + exp(-(double)gate_alpha[0]));
            gate_grad[0] = (float)(tot * a * (1.0 - a));
        }
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Lre = lds;
    float* Lim = lds + L::PLANE;
    const int t = threadIdx.x;
    const int c = t % CB, q = t / CB;          // q = h in pass 1, kx in pass 2   (q in [0, S))
    int cgroup, n;
    work_item(xcd_remap, cgroup, n);
    const int ch = cgroup * CB + c;
    const bool cok = ch < C;
    float re[S], im[S];
    float va[DW ? S : 1];          // DW: this thread's activated row, kept for the conv
    // ---- pass 1: rows   (S * CB may be < 512 for the 5*2^k sizes: q >= S idles)
    if (q < S) {
        const int chl = cok ? ch : C - 1;          // loads unconditional and inside the tensor; stores predicated
        const T* src = x + (((long)n * S + q) * S) * C + chl;
        // all S loads first: the BatchNorm coefficients (fp64 division and square root) are computed while they are in flight
#pragma unroll
        for (int w = 0; w < S; ++w) re[brev<S>(w)] = (float)src[(long)w * C];
        float mu = 0.f, is = 1.f, ga = 1.f, be = 0.f;
        if (EX && has_bn) {
            const double m = bn.sum[chl] * bn.inv_count;
            double vv = bn.sumsq[chl] * bn.inv_count - m * m;
            if (vv < 0.0) vv = 0.0;
            mu = (float)m;
            is = (float)(1.0 / sqrt(vv + (double)bn.eps));
            ga = bn.gamma[chl];
            be = bn.beta[chl];
            if (n == 0 && q == 0 && cok && bn.running_mean) {          // one thread per channel
                bn.running_mean[ch] = (1.f - bn.momentum) * bn.running_mean[ch] + bn.momentum * (float)m;
                bn.running_var[ch] = (1.f - bn.momentum) * bn.running_var[ch] + bn.momentum * (float)(vv * bn.unbias);
            }
        }
        T* aout = (EX && act_out) ? act_out + (((long)n * S + q) * S) * C + ch : nullptr;
#pragma unroll
        for (int w = 0; w < S; ++w) {
            float v = re[brev<S>(w)];
            if (EX && has_bn) {
                v = ud_rounded<T>(ud_act(ga * ((v - mu) * is) + be, bn.act));     // transform what the other branch reads
                if (aout && cok) aout[(long)w * C] = (T)v;
            }
            re[brev<S>(w)] = cok ? v : 0.f;
            im[brev<S>(w)] = 0.f;
            if (DW) va[w] = cok ? v : 0.f;
        }
        fft_inreg<S, false>(re, im);
#pragma unroll
        for (int kx = 0; kx <= S / 2; ++kx) {
            Lre[kx * L::KSTRIDE + q * CB + c] = re[kx];
            Lim[kx * L::KSTRIDE + q * CB + c] = im[kx];
        }
    }
    __syncthreads();
    // ---- pass 2: columns
    float mabs = 0.f;
    if (q <= S / 2 && cok) {
#pragma unroll
        for (int h = 0; h < S; ++h) {
            re[brev<S>(h)] = Lre[q * L::KSTRIDE + h * CB + c];
            im[brev<S>(h)] = Lim[q * L::KSTRIDE + h * CB + c];
        }
        fft_inreg<S, false>(re, im);
        float f = (q == 0 || q == S / 2) ? scale : scale * w_int;
        float gf = 1.f;
        if (EX && gate_mode != 0) {
            const float a = ud_sigmoid(gate_alpha[0]);
            gf = (gate_mode == 1) ? a : 1.f - a;
            f *= gf;
        }
        if (EX && po.buf) {
            float ps = 1.f, pinv = 1.f;
            if (!po_raw) {
                float gm = po_red[0];
#pragma unroll
                for (int i = 1; i < NT / 64; ++i) gm = fmaxf(gm, po_red[i]);
                ud_h2_scale(__float_as_uint(po.pre * gf * sqrtf(gm) * 1.002f), ps, pinv);      // (the 0.2 %: fp32 rounding of the transform)
            }
            if (blockIdx.x == 0 && blockIdx.y == 0 && q == 0 && c == 0) *po.inv_scale = pinv;
            f *= ps;
            const long row0 = ((long)n * S) * L::WH + q;
            const int cim = C + ch;
            uint16_t* pre_ = po.buf + (long)(ch >> 5) * po.panel + row0 * 32 + (ch & 31);
            uint16_t* pim_ = po.buf + (long)(cim >> 5) * po.panel + row0 * 32 + (cim & 31);
#pragma unroll
            for (int ky = 0; ky < S; ++ky) {
                const long o = (long)ky * L::WH * 32;
                if (po_raw) {
                    pre_[o] = __builtin_bit_cast(uint16_t, (_Float16)(re[ky] * f));
                    pim_[o] = __builtin_bit_cast(uint16_t, (_Float16)(im[ky] * f));
                    continue;
                }
                uint16_t a0, a1, b0, b1;
                ud_split_h2(re[ky] * f, a0, a1);
                ud_split_h2(im[ky] * f, b0, b1);
                pre_[o] = a0;
                pre_[o + po.plane] = a1;
                pim_[o] = b0;
                pim_[o + po.plane] = b1;
            }
        } else {
            T* dst = Y + (((long)n * S) * L::WH + q) * (2L * C) + ch;
#pragma unroll
            for (int ky = 0; ky < S; ++ky) {
                dst[(long)ky * L::WH * 2 * C] = (T)(re[ky] * f);
                dst[(long)ky * L::WH * 2 * C + C] = (T)(im[ky] * f);
                if (EX) mabs = fmaxf(mabs, fmaxf(fabsf(re[ky] * f), fabsf(im[ky] * f)));
            }
        }
    }
    if (EX) ud_absmax_commit(mabs, amax);
    if constexpr (DW != 0) {
        constexpr int P = (DW - 1) / 2;
        __syncthreads();                                   // every column transform has read the planes
        float* A = lds;                                    // [h][w][c]
        if (q < S) {
#pragma unroll
            for (int w = 0; w < S; ++w) A[(q * S + w) * CB + c] = va[w];
        }
        __syncthreads();
        if (q < S && cok) {
            float tp[DW * DW];
#pragma unroll
            for (int i = 0; i < DW * DW; ++i) tp[i] = dwo.wt[(long)i * C + ch];
            float acc[S];
#pragma unroll
            for (int w = 0; w < S; ++w) acc[w] = 0.f;
#pragma unroll
            for (int i = 0; i < DW; ++i) {
                const int ih = q + i - P;
                if (ih < 0 || ih >= S) continue;
                float in[S + DW - 1];
#pragma unroll
                for (int j = 0; j < S + DW - 1; ++j) in[j] = (j >= P && j < S + P) ? A[(ih * S + j - P) * CB + c] : 0.f;
#pragma unroll
                for (int w = 0; w < S; ++w)
#pragma unroll
                    for (int j = 0; j < DW; ++j) acc[w] += in[w + j] * tp[i * DW + j];
            }
            T* o = reinterpret_cast<T*>(dwo.out) + (((long)n * S + q) * S) * C + ch;
#pragma unroll
            for (int w = 0; w < S; ++w) o[(long)w * C] = (T)acc[w];
        }
    }
}

// x[n][h][w][c] = scale * C2R( f(kx) * Y[n][ky][kx][c] )   with the Hermitian extension along kx
// MIX: freq = irfft2(Y) * scale;  x = (1 - a) spat + a freq, a = sigmoid(alpha[0]);  freq_out = freq - spat (all the
// backward needs of the two branches: the gate's gradient is sum dx * (freq - spat));  per-channel sums of x and
// x^2 folded over the workgroup's rows through LDS and added (fp64 atomics) to sum / sumsq  (exp.py:60-65 + BN1 stats)
template <typename T, int S, int CB, bool MIX>
__global__ __launch_bounds__(NT) void irfft2_kernel(const T* __restrict__ Y, T* __restrict__ x, int C,
                                                    float scale, float w_int, const T* __restrict__ spat,
                                                    const float* __restrict__ alpha, T* __restrict__ freq_out,
                                                    double* __restrict__ sum, double* __restrict__ sumsq, int xcd_remap) {
    using L = Lds<S, CB>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Lre = lds;
    float* Lim = lds + L::PLANE;
    const int t = threadIdx.x;
    const int c = t % CB, q = t / CB;          // q = kx in pass 1, h in pass 2
    int cgroup, n;
    work_item(xcd_remap, cgroup, n);
    const int ch = cgroup * CB + c;
    const bool cok = ch < C;
    float re[S], im[S];
    // ---- pass 1: inverse transform along ky for each kept column kx.  Only S/2+1 of the S thread-rows own a column;
    // the others would idle while the column owners issue 2S dependent-latency loads each.  So thread-row
    // q > S/2 fetches the IMAGINARY halves for column q - (S/2+1) and hands them over through LDS (the slot the
    // owner overwrites with its own result afterwards): every thread issues S loads, twice the bytes in flight.
    constexpr int NHELP = S - (S / 2 + 1);               // columns [0, NHELP) have a helper row
    if (q > S / 2 && q < S) {
        const int kx = q - (S / 2 + 1);
        const T* src = Y + (((long)n * S) * L::WH + kx) * (2L * C) + ch + C;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) re[ky] = cok ? (float)src[(long)ky * L::WH * 2 * C] : 0.f;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) Lim[kx * L::KSTRIDE + ky * CB + c] = re[ky];
    } else if (q <= S / 2) {
        const T* src = Y + (((long)n * S) * L::WH + q) * (2L * C) + ch;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) re[brev<S>(ky)] = cok ? (float)src[(long)ky * L::WH * 2 * C] : 0.f;
        if (q >= NHELP) {
#pragma unroll
            for (int ky = 0; ky < S; ++ky) im[brev<S>(ky)] = cok ? (float)src[(long)ky * L::WH * 2 * C + C] : 0.f;
        }
    }
    __syncthreads();
    if (q <= S / 2) {
        const float f = (q == 0 || q == S / 2) ? 1.f : w_int;
        if (q < NHELP) {
#pragma unroll
            for (int ky = 0; ky < S; ++ky) im[brev<S>(ky)] = Lim[q * L::KSTRIDE + ky * CB + c];
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            re[k] *= f;
            im[k] *= f;
        }
        fft_inreg<S, true>(re, im);
#pragma unroll
        for (int h = 0; h < S; ++h) {
            Lre[q * L::KSTRIDE + h * CB + c] = re[h];
            Lim[q * L::KSTRIDE + h * CB + c] = im[h];
        }
    }
    __syncthreads();
    // ---- pass 2: Hermitian-extended inverse transform along kx, real part only
    double tot1 = 0.0, tot2 = 0.0;          // MIX: this thread's row totals of y and y^2
    if (cok && q < S) {
#pragma unroll
        for (int kx = 0; kx <= S / 2; ++kx) {
            const float zr = Lre[kx * L::KSTRIDE + q * CB + c];
            const float zi = Lim[kx * L::KSTRIDE + q * CB + c];
            re[brev<S>(kx)] = zr;
            im[brev<S>(kx)] = (kx == 0 || kx == S / 2) ? 0.f : zi;   // c2r ignores these imaginary parts
            if (kx > 0 && kx < S / 2) {
                re[brev<S>(S - kx)] = zr;
                im[brev<S>(S - kx)] = -zi;
            }
        }
        fft_inreg<S, true>(re, im);
        const long o0 = (((long)n * S + q) * S) * C + ch;
        T* dst = x + o0;
        if (!MIX) {
#pragma unroll
            for (int w = 0; w < S; ++w) dst[(long)w * C] = (T)(re[w] * scale);
        } else {
            const float a = ud_sigmoid(alpha[0]);
            const T* sp = spat + o0;
            T* fo = freq_out + o0;
#pragma unroll
            for (int w = 0; w < S; ++w) {
                const float fr = re[w] * scale, spv = (float)sp[(long)w * C];
                const float y = ud_rounded<T>(spv * (1.f - a) + fr * a);
                fo[(long)w * C] = (T)(fr - spv);
                dst[(long)w * C] = (T)y;
                tot1 += (double)y;
                tot2 += (double)y * (double)y;
            }
        }
    }
    if (MIX) {
        // fold the S row-threads of every channel: the FFT's LDS planes are free now
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);
        const bool own = cok && q < S;
        if (own) {
            red[(q * CB + c) * 2] = tot1;
            red[(q * CB + c) * 2 + 1] = tot2;
        }
        __syncthreads();
        if (cok && q == 0) {
            double t1 = 0.0, t2 = 0.0;
            for (int r = 0; r < S; ++r) {
                t1 += red[(r * CB + c) * 2];
                t2 += red[(r * CB + c) * 2 + 1];
            }
            unsafeAtomicAdd(sum + ch, t1);
            unsafeAtomicAdd(sumsq + ch, t2);
        }
    }
}

// remap only where a workgroup's run of channels is shorter than a 128-byte line
inline int xcd_remap_on(int run_bytes) {
    return run_bytes < 128;
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the backward of an SF block's spatial branch inside the ADJOINT transform.  After the spectral conv's data gradient the
// step ran: irfft2 (adjoint of rfft2: da_f) -> depthwise weight gradient -> its finalize -> depthwise data gradient (+ da_f,
// x act'(bn0(e)), BatchNorm backward sums): four launches over (n, c) planes of 8 x 8 pixels — 57 us for 13 MB tensors.  Here the
// workgroup that produces the da_f plane of (n, 64 channels) also stages the planes dd (the conv's output gradient) and a =
// act(bn0(e)) in LDS and finishes the job: dz = (gate * conv_flipped(dd) + da_f) * act'(bn0(e)), its BatchNorm sums (fp64 atomics,
// N workgroups per channel), and the weight-gradient partial [K*K][64] of its image (summed over N by dw_tile_wgrad_finalize).
// Thread (h, c): row h of channel c throughout.  LDS: the transform's planes, then (aliased) the two staged planes, then the folds.
// ---------------------------------------------------------------------------------------------------------
template <int S, int CB, int K>
struct LdsBwd {
    using L = Lds<S, CB>;
    static constexpr size_t PLANES = 2ull * S * S * CB * sizeof(float);
    static constexpr size_t FOLD = (size_t)S * K * K * CB * sizeof(float);
    static constexpr size_t BYTES = L::BYTES > PLANES ? (L::BYTES > FOLD ? L::BYTES : FOLD) : (PLANES > FOLD ? PLANES : FOLD);
};

template <typename T, int S, int CB, int K>
__global__ __launch_bounds__(NT, 4) void irfft2_dwbwd_kernel(const T* __restrict__ Y, int C, float scale, float w_int,
                                                         const T* __restrict__ dd, const T* __restrict__ x,
                                                         ud_bn_ref bn, const float* __restrict__ wt,
                                                         const float* __restrict__ gate_alpha, int gate_mode,
                                                         T* __restrict__ dz, double* __restrict__ s1,
                                                         double* __restrict__ s2, double* __restrict__ s3,
                                                         float* __restrict__ wpart, float* __restrict__ wacc, int xcd_remap) {
    using L = Lds<S, CB>;
    static_assert(S * CB == NT, "one row-thread per (h, c)");
    const float gsw = gate_factor_f(gate_alpha, gate_mode);
    constexpr int P = (K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Lre = lds;
    float* Lim = lds + L::PLANE;
    const int t = threadIdx.x;
    const int c = t % CB, q = t / CB;
    int cgroup, n;
    work_item(xcd_remap, cgroup, n);
    const int ch = cgroup * CB + c;
    const bool cok = ch < C;
    const int chl = cok ? ch : C - 1;
    float re[S], im[S];
    // ---- the adjoint transform, exactly irfft2_kernel<float, S, CB, false>
    constexpr int NHELP = S - (S / 2 + 1);
    if (q > S / 2) {
        const int kx = q - (S / 2 + 1);
        const T* src = Y + (((long)n * S) * L::WH + kx) * (2L * C) + chl + C;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) re[ky] = cok ? (float)src[(long)ky * L::WH * 2 * C] : 0.f;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) Lim[kx * L::KSTRIDE + ky * CB + c] = re[ky];
    } else {
        const T* src = Y + (((long)n * S) * L::WH + q) * (2L * C) + chl;
#pragma unroll
        for (int ky = 0; ky < S; ++ky) re[brev<S>(ky)] = cok ? (float)src[(long)ky * L::WH * 2 * C] : 0.f;
        if (q >= NHELP) {
#pragma unroll
            for (int ky = 0; ky < S; ++ky) im[brev<S>(ky)] = cok ? (float)src[(long)ky * L::WH * 2 * C + C] : 0.f;
        }
    }
    // this thread's rows of the conv's output gradient and of the conv's raw input: in flight behind the transform
    float ddr[S], er[S];
    {
        const long o = (((long)n * S + q) * S) * C + chl;
#pragma unroll
        for (int w = 0; w < S; ++w) {
            ddr[w] = (float)dd[o + (long)w * C];
            er[w] = (float)x[o + (long)w * C];
        }
    }
    __syncthreads();
    if (q <= S / 2) {
        const float f = (q == 0 || q == S / 2) ? 1.f : w_int;
        if (q < NHELP) {
#pragma unroll
            for (int ky = 0; ky < S; ++ky) im[brev<S>(ky)] = Lim[q * L::KSTRIDE + ky * CB + c];
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            re[k] *= f;
            im[k] *= f;
        }
        fft_inreg<S, true>(re, im);
#pragma unroll
        for (int h = 0; h < S; ++h) {
            Lre[q * L::KSTRIDE + h * CB + c] = re[h];
            Lim[q * L::KSTRIDE + h * CB + c] = im[h];
        }
    }
    __syncthreads();
    {
#pragma unroll
        for (int kx = 0; kx <= S / 2; ++kx) {
            const float zr = Lre[kx * L::KSTRIDE + q * CB + c];
            const float zi = Lim[kx * L::KSTRIDE + q * CB + c];
            re[brev<S>(kx)] = zr;
            im[brev<S>(kx)] = (kx == 0 || kx == S / 2) ? 0.f : zi;
            if (kx > 0 && kx < S / 2) {
                re[brev<S>(S - kx)] = zr;
                im[brev<S>(S - kx)] = -zi;
            }
        }
        fft_inreg<S, true>(re, im);          // re[w] * scale = da_f(h = q, w)
    }
    // ---- BatchNorm in front of the conv: coefficients of this channel
    float mu, is, ga, be;
    {
        const double m = bn.sum[chl] * bn.inv_count;
        double vv = bn.sumsq[chl] * bn.inv_count - m * m;
        if (vv < 0.0) vv = 0.0;
        mu = (float)m;
        is = (float)(1.0 / sqrt(vv + (double)bn.eps));
        ga = bn.gamma[chl];
        be = bn.beta[chl];
    }
    __syncthreads();                                    // the transform's planes are free: stage a and dd, [h][w][c]
    float* A = lds;
    float* D = lds + S * S * CB;
#pragma unroll
    for (int w = 0; w < S; ++w) {
        A[(q * S + w) * CB + c] = cok ? ud_act(ga * ((er[w] - mu) * is) + be, bn.act) : 0.f;
        D[(q * S + w) * CB + c] = cok ? ddr[w] : 0.f;
    }
    __syncthreads();
    float tp[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) tp[i] = wt[(long)i * C + chl];
    // ---- data gradient of row q: flipped taps over dd, + da_f, through act'(bn(x)); BatchNorm backward sums
    double v1 = 0.0, v2 = 0.0;
    float v3 = 0.f;          // s3 != NULL: sum dz^2, the energy bound ud_normbwd_apply_planes scales its planes by
    {
        float acc[S];
#pragma unroll
        for (int w = 0; w < S; ++w) acc[w] = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int ih = q + i - P;
            if (ih < 0 || ih >= S) continue;
            float in[S + K - 1];
#pragma unroll
            for (int j = 0; j < S + K - 1; ++j) in[j] = (j >= P && j < S + P) ? D[(ih * S + j - P) * CB + c] : 0.f;
#pragma unroll
            for (int w = 0; w < S; ++w)
#pragma unroll
                for (int j = 0; j < K; ++j) acc[w] += in[w + j] * tp[K * K - 1 - (i * K + j)];
        }
        const float gs = gate_factor_f(gate_alpha, gate_mode);
        if (cok) {
            T* o = dz + (((long)n * S + q) * S) * C + ch;
#pragma unroll
            for (int w = 0; w < S; ++w) {
                const float xh = (er[w] - mu) * is;
                float d = acc[w] * gs + re[w] * scale;
                if (bn.act) d *= ud_act_grad_fast(ga * xh + be, bn.act);
                d = ud_rounded<T>(d);          // the sums are taken of what consumers will read back
                o[(long)w * C] = (T)d;
                v1 += (double)d;
                v2 += (double)d * (double)xh;
                v3 += d * d;
            }
        }
    }
    // ---- weight gradient of this image: accw[i][j] = sum_w a(q + i - P, w + j - P) * dd(q, w)
    float accw[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) accw[i] = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int ih = q + i - P;
        if (ih < 0 || ih >= S) continue;
        float in[S + K - 1];
#pragma unroll
        for (int j = 0; j < S + K - 1; ++j) in[j] = (j >= P && j < S + P) ? A[(ih * S + j - P) * CB + c] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int w = 0; w < S; ++w) accw[i * K + j] += in[w + j] * ddr[w];
    }
    __syncthreads();                                    // A and D are consumed: fold the S row-threads of every channel
    float* F = lds;                                     // [q][tap][c]
#pragma unroll
    for (int i = 0; i < K * K; ++i) F[(q * K * K + i) * CB + c] = cok ? accw[i] : 0.f;
    __syncthreads();
    for (int i = t; i < K * K * CB; i += NT) {
        const int tap = i / CB, cc = i % CB;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < S; ++r) sum += F[(r * K * K + tap) * CB + cc];
        if (cgroup * CB + cc < C) {
            // wacc: fp32 atomics straight onto the parameter-layout gradient [C][K*K] (zeroed by the caller; N adds per address,
            // the gate applied here) instead of a partial row + the fold launch — the default; cfg.deterministic keeps the fold
            if (wacc) atomicAdd(wacc + (long)(cgroup * CB + cc) * K * K + tap, sum * gsw);
            else wpart[((long)n * K * K + tap) * C + cgroup * CB + cc] = sum;
        }
    }
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds);
    red[(q * CB + c) * 3] = v1;
    red[(q * CB + c) * 3 + 1] = v2;
    red[(q * CB + c) * 3 + 2] = (double)v3 * 1.0001;          // rounded up: an upper bound
    __syncthreads();
    if (cok && q == 0) {
        double t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int r = 0; r < S; ++r) {
            t1 += red[(r * CB + c) * 3];
            t2 += red[(r * CB + c) * 3 + 1];
            t3 += red[(r * CB + c) * 3 + 2];
        }
        unsafeAtomicAdd(s1 + ch, t1);
        unsafeAtomicAdd(s2 + ch, t2);
        if (s3) unsafeAtomicAdd(s3 + ch, t3);
    }
}

template <typename T, int S, int CB, int K>
int launch_irfft2_dwbwd(const T* Y, int N, int C, float scale, float w_int, const T* dd, const T* x,
                        const ud_bn_ref& bn, const float* wt, const float* gate_alpha, int gate_mode, T* dz, double* s1,
                        double* s2, double* s3, float* wpart, float* wacc, hipStream_t s) {
    using LB = LdsBwd<S, CB, K>;
    static bool attr_set = false;
    if (LB::BYTES > 65536 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&irfft2_dwbwd_kernel<T, S, CB, K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB::BYTES);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)ud_cdiv(C, CB), (unsigned)N);
    hipLaunchKernelGGL((irfft2_dwbwd_kernel<T, S, CB, K>), grid, dim3(NT), LB::BYTES, s, Y, C, scale, w_int, dd, x, bn, wt,
                       gate_alpha, gate_mode, dz, s1, s2, s3, wpart, wacc, xcd_remap_on(CB * (int)sizeof(float)));
    UD_LAUNCH_CHECK();
    return 0;
}

struct RfftEx {
    const ud_bn_ref* bn;
    void* act_out;
    const float* gate_alpha;
    int gate_mode;
    const double* gate_acc;
    float* gate_grad;
    uint32_t* absmax;          // 256 slots: |Y|max as a side output (ud_absmax_commit), or NULL
    PlanesOut planes = PlanesOut{nullptr, 0, 0, nullptr, 0.f, nullptr};
    DwOut dw = DwOut{nullptr, nullptr};
};

template <typename T, int S, int CB, bool EX, int DW = 0>
int launch_rfft2_t(const T* x, T* Y, int N, int C, float scale, float w_int, const RfftEx& ex, hipStream_t s) {
    using L = Lds<S, CB>;
    static bool attr_set = false;
    if (L::BYTES > 65536 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rfft2_kernel<T, S, CB, EX, DW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)ud_cdiv(C, CB), (unsigned)N);
    ud_bn_ref none{};
    hipLaunchKernelGGL((rfft2_kernel<T, S, CB, EX, DW>), grid, dim3(NT), L::BYTES, s, x, Y, C, scale, w_int,
                       ex.bn ? *ex.bn : none, ex.bn ? 1 : 0, (T*)ex.act_out, ex.gate_alpha, ex.gate_mode, ex.gate_acc,
                       ex.gate_grad, xcd_remap_on(CB * (int)sizeof(T)), ex.absmax, ex.planes, ex.dw);
    UD_LAUNCH_CHECK();
    return 0;
}

template <typename T, int S, int CB>
int launch_rfft2(const T* x, T* Y, int N, int C, float scale, float w_int, const RfftEx* ex, hipStream_t s) {
    if (ex) return launch_rfft2_t<T, S, CB, true>(x, Y, N, C, scale, w_int, *ex, s);
    return launch_rfft2_t<T, S, CB, false>(x, Y, N, C, scale, w_int, RfftEx{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, PlanesOut{nullptr, 0, 0, nullptr, 0.f, nullptr}, DwOut{nullptr, nullptr}}, s);
}

struct IrfftMix {
    const void* spat;
    const float* alpha;
    void* freq_out;
    double* sum;
    double* sumsq;
};

template <typename T, int S, int CB, bool MIX>
int launch_irfft2_t(const T* Y, T* x, int N, int C, float scale, float w_int, const IrfftMix& m, hipStream_t s) {
    using L = Lds<S, CB>;
    static_assert(!MIX || L::BYTES >= (size_t)S * CB * 2 * sizeof(double), "LDS planes hold the row totals");
    static bool attr_set = false;
    if (L::BYTES > 65536 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&irfft2_kernel<T, S, CB, MIX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)ud_cdiv(C, CB), (unsigned)N);
    hipLaunchKernelGGL((irfft2_kernel<T, S, CB, MIX>), grid, dim3(NT), L::BYTES, s, Y, x, C, scale, w_int,
                       (const T*)m.spat, m.alpha, (T*)m.freq_out, m.sum, m.sumsq, xcd_remap_on(CB * (int)sizeof(T)));
    UD_LAUNCH_CHECK();
    return 0;
}

template <typename T, int S, int CB>
int launch_irfft2(const T* Y, T* x, int N, int C, float scale, float w_int, const IrfftMix* m, hipStream_t s) {
    if (m) return launch_irfft2_t<T, S, CB, true>(Y, x, N, C, scale, w_int, *m, s);
    return launch_irfft2_t<T, S, CB, false>(Y, x, N, C, scale, w_int, IrfftMix{nullptr, nullptr, nullptr, nullptr, nullptr}, s);
}

// ---------------------------------------------------------------------------------------------------------
// TWO-PASS form for the large planes (S = 32, 64).  The LDS-resident kernels above hold S x (S/2+1) complex fp32 per
// channel (16.9 KB at 64 x 64), so a workgroup takes 8 / 16 channels: 32 / 64-byte runs per pixel in fp32 storage,
// 16 / 32 in half storage — every 128-byte line is pulled by 4 ... 8 workgroups and a wave-load touches 8 lines for 128
// useful bytes.  Here the transform is a ROW kernel and a COLUMN kernel with the half-spectrum in between kept in HBM
// (Z[n][h][kx][Re 0..C | Im 0..C], fp32): the channel is still the lane, but a wave owns 64 CONSECUTIVE channels of one
// image row (or one kx column), so every access is a whole run of lines; no LDS, no barrier.  Same butterflies on the
// same values in the same order as the one-kernel form: the results agree to the last bit or two (hipcc contracts the scale
// factor into the first butterfly differently in the two forms).  Extra traffic: Z written and read once.
// ---------------------------------------------------------------------------------------------------------
constexpr int NT2 = 256;          // 64 channels x 4 rows (or kx columns)
constexpr int RPT = 4;            // image rows one thread of the inverse row kernel walks (fewer atomics per channel)

template <typename T, int S, bool EX>
__global__ __launch_bounds__(NT2) void rows_fwd_kernel(const T* __restrict__ x, float* __restrict__ Z, int C, ud_bn_ref bn,
                                                       int has_bn, T* __restrict__ act_out,
                                                       const float* __restrict__ gate_alpha,
                                                       const double* __restrict__ gate_acc, float* __restrict__ gate_grad) {
    constexpr int WH = S / 2 + 1;
    if (EX && gate_grad && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) {
        const double tot = ud_wave_sum_d(gate_acc[threadIdx.x]);
        if (threadIdx.x == 0) {
            const double a = 1.0 / (1.0 + exp(-(double)gate_alpha[0]));
            gate_grad[0] = (float)(tot * a * (1.0 - a));
        }
    }
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
    const int h = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (ch >= C || h >= S) return;
    float mu = 0.f, is = 1.f, ga = 1.f, be = 0.f;
    if (EX && has_bn) {
        const double m = bn.sum[ch] * bn.inv_count;
        double vv = bn.sumsq[ch] * bn.inv_count - m * m;
        if (vv < 0.0) vv = 0.0;
        mu = (float)m;
        is = (float)(1.0 / sqrt(vv + (double)bn.eps));
        ga = bn.gamma[ch];
        be = bn.beta[ch];
        if (n == 0 && h == 0 && bn.running_mean) {          // one thread per channel
            bn.running_mean[ch] = (1.f - bn.momentum) * bn.running_mean[ch] + bn.momentum * (float)m;
            bn.running_var[ch] = (1.f - bn.momentum) * bn.running_var[ch] + bn.momentum * (float)(vv * bn.unbias);
        }
    }
    const long o0 = (((long)n * S + h) * S) * C + ch;
    const T* src = x + o0;
    T* aout = (EX && act_out) ? act_out + o0 : nullptr;
    float re[S], im[S];
    // all S loads first (in flight together), then the transform of the values and the activated copy
#pragma unroll
    for (int w = 0; w < S; ++w) re[brev<S>(w)] = (float)src[(long)w * C];
#pragma unroll
    for (int w = 0; w < S; ++w) {
        if (EX && has_bn) {
            const float v = ud_rounded<T>(ud_act(ga * ((re[brev<S>(w)] - mu) * is) + be, bn.act));
            if (aout) aout[(long)w * C] = (T)v;
            re[brev<S>(w)] = v;
        }
        im[brev<S>(w)] = 0.f;
    }
    fft_inreg<S, false>(re, im);
    float* dst = Z + (((long)n * S + h) * WH) * (2L * C) + ch;
#pragma unroll
    for (int kx = 0; kx <= S / 2; ++kx) {
        dst[(long)kx * 2 * C] = re[kx];
        dst[(long)kx * 2 * C + C] = im[kx];
    }
}

template <typename T, int S, bool EX>
__global__ __launch_bounds__(NT2) void cols_fwd_kernel(const float* __restrict__ Z, T* __restrict__ Y, int C, float scale,
                                                       float w_int, const float* __restrict__ gate_alpha, int gate_mode,
                                                       uint32_t* __restrict__ amax) {
    constexpr int WH = S / 2 + 1;
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
    const int kx = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (ch >= C || kx >= WH) {
        ud_absmax_commit(0.f, amax);
        return;
    }
    float mabs = 0.f;
    const float* src = Z + (((long)n * S) * WH + kx) * (2L * C) + ch;
    float re[S], im[S];
#pragma unroll
    for (int h = 0; h < S; ++h) {
        re[brev<S>(h)] = src[(long)h * WH * 2 * C];
        im[brev<S>(h)] = src[(long)h * WH * 2 * C + C];
    }
    fft_inreg<S, false>(re, im);
    float f = (kx == 0 || kx == S / 2) ? scale : scale * w_int;
    if (EX && gate_mode != 0) {
        const float a = ud_sigmoid(gate_alpha[0]);
        f *= (gate_mode == 1) ? a : 1.f - a;
    }
    T* dst = Y + (((long)n * S) * WH + kx) * (2L * C) + ch;
#pragma unroll
    for (int ky = 0; ky < S; ++ky) {
        dst[(long)ky * WH * 2 * C] = (T)(re[ky] * f);
        dst[(long)ky * WH * 2 * C + C] = (T)(im[ky] * f);
        mabs = fmaxf(mabs, fmaxf(fabsf(re[ky] * f), fabsf(im[ky] * f)));
    }
    ud_absmax_commit(mabs, amax);
}

template <typename T, int S>
__global__ __launch_bounds__(NT2) void cols_inv_kernel(const T* __restrict__ Y, float* __restrict__ Z, int C, float w_int) {
    constexpr int WH = S / 2 + 1;
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
    const int kx = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (ch >= C || kx >= WH) return;
    const T* src = Y + (((long)n * S) * WH + kx) * (2L * C) + ch;
    const float f = (kx == 0 || kx == S / 2) ? 1.f : w_int;
    float re[S], im[S];
#pragma unroll
    for (int ky = 0; ky < S; ++ky) {
        re[brev<S>(ky)] = (float)src[(long)ky * WH * 2 * C];
        im[brev<S>(ky)] = (float)src[(long)ky * WH * 2 * C + C];
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        re[k] *= f;
        im[k] *= f;
    }
    fft_inreg<S, true>(re, im);
    float* dst = Z + (((long)n * S) * WH + kx) * (2L * C) + ch;
#pragma unroll
    for (int h = 0; h < S; ++h) {
        dst[(long)h * WH * 2 * C] = re[h];
        dst[(long)h * WH * 2 * C + C] = im[h];
    }
}

template <typename T, int S, bool MIX>
__global__ __launch_bounds__(NT2) void rows_inv_kernel(const float* __restrict__ Z, T* __restrict__ x, int C, float scale,
                                                       const T* __restrict__ spat, const float* __restrict__ alpha,
                                                       T* __restrict__ freq_out, double* __restrict__ sum,
                                                       double* __restrict__ sumsq) {
    constexpr int WH = S / 2 + 1;
    __shared__ double red[MIX ? 2 * NT2 : 1];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;
    const int n = blockIdx.z;
    const bool cok = ch < C;
    double tot1 = 0.0, tot2 = 0.0;
    float a = 0.f;
    if (MIX) a = ud_sigmoid(alpha[0]);
    if (cok) {
#pragma unroll 1
        for (int r = 0; r < RPT; ++r) {
            const int h = (blockIdx.y * 4 + rg) * RPT + r;
            if (h >= S) break;
            const float* src = Z + (((long)n * S + h) * WH) * (2L * C) + ch;
            float re[S], im[S];
#pragma unroll
            for (int kx = 0; kx <= S / 2; ++kx) {
                const float zr = src[(long)kx * 2 * C];
                const float zi = src[(long)kx * 2 * C + C];
                re[brev<S>(kx)] = zr;
                im[brev<S>(kx)] = (kx == 0 || kx == S / 2) ? 0.f : zi;   // c2r ignores these imaginary parts
                if (kx > 0 && kx < S / 2) {
                    re[brev<S>(S - kx)] = zr;
                    im[brev<S>(S - kx)] = -zi;
                }
            }
            fft_inreg<S, true>(re, im);
            const long o0 = (((long)n * S + h) * S) * C + ch;
            T* dst = x + o0;
            if (!MIX) {
#pragma unroll
                for (int w = 0; w < S; ++w) dst[(long)w * C] = (T)(re[w] * scale);
            } else {
                const T* sp = spat + o0;
                T* fo = freq_out + o0;
#pragma unroll
                for (int w = 0; w < S; ++w) {
                    const float fr = re[w] * scale, spv = (float)sp[(long)w * C];
                    const float y = ud_rounded<T>(spv * (1.f - a) + fr * a);
                    fo[(long)w * C] = (T)(fr - spv);
                    dst[(long)w * C] = (T)y;
                    tot1 += (double)y;
                    tot2 += (double)y * (double)y;
                }
            }
        }
    }
    if (MIX) {
        red[threadIdx.x * 2] = tot1;
        red[threadIdx.x * 2 + 1] = tot2;
        __syncthreads();
        if (rg == 0 && cok) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                t1 += red[(g * 64 + lane) * 2];
                t2 += red[(g * 64 + lane) * 2 + 1];
            }
            unsafeAtomicAdd(sum + ch, t1);
            unsafeAtomicAdd(sumsq + ch, t2);
        }
    }
}

template <typename T, int S>
int rfft2_two_pass(const T* x, T* Y, float* Z, int N, int C, float scale, float w_int, const RfftEx* ex, hipStream_t s) {
    constexpr int WH = S / 2 + 1;
    const dim3 gr((unsigned)ud_cdiv(C, 64), (unsigned)(S / 4), (unsigned)N);
    const dim3 gc((unsigned)ud_cdiv(C, 64), (unsigned)ud_cdiv(WH, 4), (unsigned)N);
    ud_bn_ref none{};
    if (ex) {
        hipLaunchKernelGGL((rows_fwd_kernel<T, S, true>), gr, dim3(NT2), 0, s, x, Z, C, ex->bn ? *ex->bn : none,
                           ex->bn ? 1 : 0, (T*)ex->act_out, ex->gate_alpha, ex->gate_acc, ex->gate_grad);
        hipLaunchKernelGGL((cols_fwd_kernel<T, S, true>), gc, dim3(NT2), 0, s, Z, Y, C, scale, w_int, ex->gate_alpha,
                           ex->gate_mode, ex->absmax);
    } else {
        hipLaunchKernelGGL((rows_fwd_kernel<T, S, false>), gr, dim3(NT2), 0, s, x, Z, C, none, 0, (T*)nullptr,
                           (const float*)nullptr, (const double*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL((cols_fwd_kernel<T, S, false>), gc, dim3(NT2), 0, s, Z, Y, C, scale, w_int,
                           (const float*)nullptr, 0, (uint32_t*)nullptr);
    }
    UD_LAUNCH_CHECK();
    return 0;
}

template <typename T, int S>
int irfft2_two_pass(const T* Y, T* x, float* Z, int N, int C, float scale, float w_int, const IrfftMix* m, hipStream_t s) {
    constexpr int WH = S / 2 + 1;
    const dim3 gc((unsigned)ud_cdiv(C, 64), (unsigned)ud_cdiv(WH, 4), (unsigned)N);
    const dim3 gr((unsigned)ud_cdiv(C, 64), (unsigned)ud_cdiv(S, 4 * RPT), (unsigned)N);
    hipLaunchKernelGGL((cols_inv_kernel<T, S>), gc, dim3(NT2), 0, s, Y, Z, C, w_int);
    if (m)
        hipLaunchKernelGGL((rows_inv_kernel<T, S, true>), gr, dim3(NT2), 0, s, Z, x, C, scale, (const T*)m->spat, m->alpha,
                           (T*)m->freq_out, m->sum, m->sumsq);
    else
        hipLaunchKernelGGL((rows_inv_kernel<T, S, false>), gr, dim3(NT2), 0, s, Z, x, C, scale, (const T*)nullptr,
                           (const float*)nullptr, (T*)nullptr, (double*)nullptr, (double*)nullptr);
    UD_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// S = 32, the transform SHARED BY A LANE PAIR (the plane laid across the wave).
// With one lane per (row, channel) a 32 x 32 plane group of 512 threads takes 16 channels: 64-byte runs per pixel, every
// 128-byte line pulled by two workgroups (33-39 % of HBM peak, profiles/r03/hbm_bw_by_kernel.txt).  Here lanes l and l ^ 32
// share one 32-point transform: lane half hh = l >> 5 holds the decimated sequence x[2j + hh] (16 values), runs the
// 16-point radix-2 network in registers, and the LAST butterfly stage  X[k] = E[k] + W^k O[k],  X[k + 16] = E[k] - W^k O[k]
// crosses the two lane halves with wavefront shuffles (__shfl_xor 32).  Half the registers per lane, so a workgroup is
// 1024 threads = 16 waves, a wave = 32 CHANNELS (one whole 128-byte line per pixel) x the two halves: every global access
// of a wave is two full lines.  LDS as in the one-lane form ([kx][h][c], 17 x 32 x 32 complex fp32 = 136 KB: one workgroup
// per CU, 16 waves).  Same butterflies on the same values in the same order as fft_inreg<32>: results equal to the last bit
// or two (the compiler contracts different multiply-adds).
// ---------------------------------------------------------------------------------------------------------
constexpr int NTW = 1024;
constexpr int CBW = 32;
struct LdsW {
    static constexpr int KSTRIDE = 32 * CBW;
    static constexpr int PLANE = 17 * KSTRIDE;
    static constexpr size_t BYTES = 2ull * PLANE * sizeof(float);
};

// on entry lane half hh holds x[2j + hh] in slot brev<16>(j); on exit half 0 holds X[0..15], half 1 holds X[16..31]
template <bool INV>
__device__ __forceinline__ void fft32_pair(float (&re)[16], float (&im)[16], int hh) {
    fft_inreg<16, INV>(re, im);          // E (even samples) on half 0, O (odd samples) on half 1
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float wr = TW_RE[2 * k], wi = INV ? -TW_IM[2 * k] : TW_IM[2 * k];          // W_32^k
        const float tr = wr * re[k] - wi * im[k], ti = wr * im[k] + wi * re[k];
        const float ar = hh ? tr : re[k], ai = hh ? ti : im[k];          // half 0 contributes E[k], half 1 contributes W^k O[k]
        const float br = __shfl_xor(ar, 32, 64), bi = __shfl_xor(ai, 32, 64);
        re[k] = hh ? br - ar : ar + br;
        im[k] = hh ? bi - ai : ai + bi;
    }
}

template <typename T, bool EX>
__global__ __launch_bounds__(NTW) void rfft2_wave_kernel(const T* __restrict__ x, T* __restrict__ Y, int C, float scale,
                                                         float w_int, ud_bn_ref bn, int has_bn, T* __restrict__ act_out,
                                                         const float* __restrict__ gate_alpha, int gate_mode,
                                                         const double* __restrict__ gate_acc, float* __restrict__ gate_grad,
                                                         uint32_t* __restrict__ amax) {
    constexpr int S = 32, WH = 17;
    using L = LdsW;
    if (EX && gate_grad && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {          // see rfft2_kernel
        const double tot = ud_wave_sum_d(gate_acc[threadIdx.x]);
        if (threadIdx.x == 0) {
            const double a = 1.0 / (1.0 + exp(-(double)gate_alpha[0]));
            gate_grad[0] = (float)(tot * a * (1.0 - a));
        }
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Lre = lds;
    float* Lim = lds + L::PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int n = blockIdx.y;
    const int ch = blockIdx.x * CBW + c;
    const bool cok = ch < C;
    const int chl = cok ? ch : C - 1;          // loads stay inside the tensor and unconditional; stores are predicated
    float re[16], im[16];
    // ---- pass 1: rows q = wave, wave + 16 — the loads of both rows are issued before anything waits for them
    float v[2][16];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const T* src = x + (((long)n * S + wave + 16 * it) * S + hh) * C + chl;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[it][j] = (float)src[(long)(2 * j) * C];
    }
    float mu = 0.f, is = 1.f, ga = 1.f, be = 0.f;
    if (EX && has_bn) {
        const double m = bn.sum[chl] * bn.inv_count;
        double vv = bn.sumsq[chl] * bn.inv_count - m * m;
        if (vv < 0.0) vv = 0.0;
        mu = (float)m;
        is = (float)(1.0 / sqrt(vv + (double)bn.eps));
        ga = bn.gamma[chl];
        be = bn.beta[chl];
        if (n == 0 && wave == 0 && hh == 0 && cok && bn.running_mean) {          // one thread per channel
            bn.running_mean[ch] = (1.f - bn.momentum) * bn.running_mean[ch] + bn.momentum * (float)m;
            bn.running_var[ch] = (1.f - bn.momentum) * bn.running_var[ch] + bn.momentum * (float)(vv * bn.unbias);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = wave + 16 * it;
        T* aout = (EX && act_out) ? act_out + (((long)n * S + q) * S + hh) * C + ch : nullptr;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float u = v[it][j];
            if (EX && has_bn) {
                u = ud_rounded<T>(ud_act(ga * ((u - mu) * is) + be, bn.act));          // transform what the other branch reads
                if (aout && cok) {
                    *aout = (T)u;
                    aout += 2L * C;
                }
            }
            re[brev<16>(j)] = cok ? u : 0.f;
            im[brev<16>(j)] = 0.f;
        }
        fft32_pair<false>(re, im, hh);
        // half 0 holds kx = 0..15, half 1 holds kx = 16 (its slot 0) and the mirrored rest
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (hh == 0 || k == 0) {
                Lre[(k + 16 * hh) * L::KSTRIDE + q * CBW + c] = re[k];
                Lim[(k + 16 * hh) * L::KSTRIDE + q * CBW + c] = im[k];
            }
        }
    }
    __syncthreads();
    // ---- pass 2: columns kx = wave (wave 0 also takes kx = 16)
    float mabs = 0.f;
    float gf = 1.f;
    if (EX && gate_mode != 0) {
        const float a = ud_sigmoid(gate_alpha[0]);
        gf = (gate_mode == 1) ? a : 1.f - a;
    }
#pragma unroll 1
    for (int kx = wave; kx < WH; kx += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            re[brev<16>(j)] = Lre[kx * L::KSTRIDE + (2 * j + hh) * CBW + c];
            im[brev<16>(j)] = Lim[kx * L::KSTRIDE + (2 * j + hh) * CBW + c];
        }
        fft32_pair<false>(re, im, hh);
        const float f = ((kx == 0 || kx == S / 2) ? scale : scale * w_int) * gf;
        if (cok) {
            T* dst = Y + (((long)n * S + 16 * hh) * WH + kx) * (2L * C) + ch;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                dst[0] = (T)(re[k] * f);
                dst[C] = (T)(im[k] * f);
                dst += (long)WH * 2 * C;
                if (EX) mabs = fmaxf(mabs, fmaxf(fabsf(re[k] * f), fabsf(im[k] * f)));
            }
        }
    }
    if (EX) ud_absmax_commit(mabs, amax);
}

template <typename T, bool MIX>
__global__ __launch_bounds__(NTW) void irfft2_wave_kernel(const T* __restrict__ Y, T* __restrict__ x, int C, float scale,
                                                          float w_int, const T* __restrict__ spat,
                                                          const float* __restrict__ alpha, T* __restrict__ freq_out,
                                                          double* __restrict__ sum, double* __restrict__ sumsq) {
    constexpr int S = 32, WH = 17;
    using L = LdsW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Lre = lds;
    float* Lim = lds + L::PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int n = blockIdx.y;
    const int ch = blockIdx.x * CBW + c;
    const bool cok = ch < C;
    const int chl = cok ? ch : C - 1;
    float re[16], im[16];
    // ---- pass 1: inverse transform along ky of the kept columns kx = wave (wave 0 also kx = 16)
#pragma unroll 1
    for (int kx = wave; kx < WH; kx += 16) {
        const T* src = Y + (((long)n * S + hh) * WH + kx) * (2L * C) + chl;
        const float f = (kx == 0 || kx == S / 2) ? 1.f : w_int;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            re[brev<16>(j)] = (float)src[(long)(2 * j) * WH * 2 * C];
            im[brev<16>(j)] = (float)src[(long)(2 * j) * WH * 2 * C + C];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            re[k] = cok ? re[k] * f : 0.f;
            im[k] = cok ? im[k] * f : 0.f;
        }
        fft32_pair<true>(re, im, hh);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            Lre[kx * L::KSTRIDE + (k + 16 * hh) * CBW + c] = re[k];
            Lim[kx * L::KSTRIDE + (k + 16 * hh) * CBW + c] = im[k];
        }
    }
    __syncthreads();
    // ---- pass 2: Hermitian-extended inverse transform along kx of rows h = wave, wave + 16; real part only
    double tot1 = 0.0, tot2 = 0.0;
    float a = 0.f;
    if (MIX) a = ud_sigmoid(alpha[0]);
#pragma unroll 1
    for (int h = wave; h < S; h += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kx = 2 * j + hh;                      // this lane's decimated sample
            const int kk = kx <= S / 2 ? kx : S - kx;       // X[S - kx] = conj(X[kx])
            const float zr = Lre[kk * L::KSTRIDE + h * CBW + c];
            float zi = Lim[kk * L::KSTRIDE + h * CBW + c];
            zi = (kk == 0 || kk == S / 2) ? 0.f : (kx <= S / 2 ? zi : -zi);          // c2r ignores the imaginary parts of kx = 0, S/2
            re[brev<16>(j)] = zr;
            im[brev<16>(j)] = zi;
        }
        fft32_pair<true>(re, im, hh);
        if (cok) {
            // uniform plane base + 32-bit lane offsets (global_load saddr + voffset): 64-bit lane addresses, hoisted out of the
            // row loop for 3 tensors x 16 pixels, spilled a hundred registers at 1024 threads
            const long plane = (long)n * S * S * C;
            unsigned o = (unsigned)((h * S + 16 * hh) * C + ch);
            T* dst = x + plane;
            if (!MIX) {
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[o + (unsigned)(k * C)] = (T)(re[k] * scale);
            } else {
                const T* sp = spat + plane;
                T* fo = freq_out + plane;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const unsigned ok = o + (unsigned)(k * C);
                    const float fr = re[k] * scale, spv = (float)sp[ok];
                    const float y = ud_rounded<T>(spv * (1.f - a) + fr * a);
                    fo[ok] = (T)(fr - spv);
                    dst[ok] = (T)y;
                    tot1 += (double)y;
                    tot2 += (double)y * (double)y;
                }
            }
        }
    }
    if (MIX) {
        // per-channel totals: the two lane halves by a wavefront shuffle, the 16 waves through LDS (the planes are free now)
        tot1 += __shfl_xor(tot1, 32, 64);
        tot2 += __shfl_xor(tot2, 32, 64);
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);
        if (hh == 0) {
            red[(wave * CBW + c) * 2] = tot1;
            red[(wave * CBW + c) * 2 + 1] = tot2;
        }
        __syncthreads();
        if (wave == 0 && hh == 0 && cok) {
            double t1 = 0.0, t2 = 0.0;
            for (int r = 0; r < 16; ++r) {
                t1 += red[(r * CBW + c) * 2];
                t2 += red[(r * CBW + c) * 2 + 1];
            }
            unsafeAtomicAdd(sum + ch, t1);
            unsafeAtomicAdd(sumsq + ch, t2);
        }
    }
}

template <typename K>
int wave_lds_attr(K kernel) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LdsW::BYTES);
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T, bool EX>
int launch_rfft2_wave_t(const T* x, T* Y, int N, int C, float scale, float w_int, const RfftEx& ex, hipStream_t s) {
    static int attr = wave_lds_attr(&rfft2_wave_kernel<T, EX>);
    if (attr) return attr;
    dim3 grid((unsigned)ud_cdiv(C, CBW), (unsigned)N);
    ud_bn_ref none{};
    hipLaunchKernelGGL((rfft2_wave_kernel<T, EX>), grid, dim3(NTW), LdsW::BYTES, s, x, Y, C, scale, w_int, ex.bn ? *ex.bn : none,
                       ex.bn ? 1 : 0, (T*)ex.act_out, ex.gate_alpha, ex.gate_mode, ex.gate_acc, ex.gate_grad, ex.absmax);
    UD_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int launch_rfft2_wave(const T* x, T* Y, int N, int C, float scale, float w_int, const RfftEx* ex, hipStream_t s) {
    if (ex) return launch_rfft2_wave_t<T, true>(x, Y, N, C, scale, w_int, *ex, s);
    return launch_rfft2_wave_t<T, false>(x, Y, N, C, scale, w_int, RfftEx{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, PlanesOut{nullptr, 0, 0, nullptr, 0.f, nullptr}, DwOut{nullptr, nullptr}}, s);
}
template <typename T, bool MIX>
int launch_irfft2_wave_t(const T* Y, T* x, int N, int C, float scale, float w_int, const IrfftMix& m, hipStream_t s) {
    static int attr = wave_lds_attr(&irfft2_wave_kernel<T, MIX>);
    if (attr) return attr;
    dim3 grid((unsigned)ud_cdiv(C, CBW), (unsigned)N);
    hipLaunchKernelGGL((irfft2_wave_kernel<T, MIX>), grid, dim3(NTW), LdsW::BYTES, s, Y, x, C, scale, w_int, (const T*)m.spat,
                       m.alpha, (T*)m.freq_out, m.sum, m.sumsq);
    UD_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int launch_irfft2_wave(const T* Y, T* x, int N, int C, float scale, float w_int, const IrfftMix* m, hipStream_t s) {
    if (m) return launch_irfft2_wave_t<T, true>(Y, x, N, C, scale, w_int, *m, s);
    return launch_irfft2_wave_t<T, false>(Y, x, N, C, scale, w_int, IrfftMix{nullptr, nullptr, nullptr, nullptr, nullptr}, s);
}

// Which S = 32 form runs.  Measured on an MI355X (profiles/r04/fft32_wave.txt): the lane-pair form wins where its 32-channel
// groups fill whole lines AND the launch is a single round of workgroups (one 136 KB workgroup per CU) — N 32 x C 192:
// irfft2_mix 28.0 -> 24.2 us, irfft2 15.9 -> 14.9 — and loses on the step's own 32 x 32 shape, C = 336 = 10.5 groups: 352
// workgroups are 1.4 rounds of 256 CUs with a half-empty last group (rfft2 19.6 -> 28.0 us).  So: auto = lane pairs only for
// C % 32 == 0 and N * C / 32 <= 256.  UD_FFT32_WAVE = 0 / 1 (or ud_fft32_set_wave) forces a form.
int g_fft32_wave = [] {
    const char* e = getenv("UD_FFT32_WAVE");
    return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1;
}();
inline bool fft32_wave_on(int N, int C) {
    if (g_fft32_wave >= 0) return g_fft32_wave != 0;
    return C % CBW == 0 && (long)N * (C / CBW) <= 256;
}

// half storage: the power-of-two sizes of the EfficientNet trunk only (the ResNet models' 5*2^k maps stay fp32)
template <typename T>
int rfft2_dispatch(const T* x, T* Y, int N, int S, int C, float scale, float w_interior, const RfftEx* ex,
                   hipStream_t s) {
    switch (S) {
        case 8: return launch_rfft2<T, 8, 64>(x, Y, N, C, scale, w_interior, ex, s);
        case 16: return launch_rfft2<T, 16, 32>(x, Y, N, C, scale, w_interior, ex, s);
        case 32:
            if (fft32_wave_on(N, C) && !(ex && ex->planes.buf)) return launch_rfft2_wave<T>(x, Y, N, C, scale, w_interior, ex, s);
            return launch_rfft2<T, 32, 16>(x, Y, N, C, scale, w_interior, ex, s);
        case 64: return launch_rfft2<T, 64, 8>(x, Y, N, C, scale, w_interior, ex, s);
        case 12: return launch_rfft2<T, 12, 42>(x, Y, N, C, scale, w_interior, ex, s);          // 3 * 2^k: both storage types (the 380 x 380 trunk)
        case 24: return launch_rfft2<T, 24, 21>(x, Y, N, C, scale, w_interior, ex, s);
        case 48: return launch_rfft2<T, 48, 10>(x, Y, N, C, scale, w_interior, ex, s);
        default: break;
    }
    if constexpr (std::is_same<T, float>::value) {
        switch (S) {
            case 10: return launch_rfft2<T, 10, 51>(x, Y, N, C, scale, w_interior, ex, s);
            case 20: return launch_rfft2<T, 20, 25>(x, Y, N, C, scale, w_interior, ex, s);
            case 40: return launch_rfft2<T, 40, 12>(x, Y, N, C, scale, w_interior, ex, s);
            case 80: return launch_rfft2<T, 80, 5>(x, Y, N, C, scale, w_interior, ex, s);
            default: break;
        }
    }
    return UD_EINVAL;
}

template <typename T>
int irfft2_dispatch(const T* Y, T* x, int N, int S, int C, float scale, float w_interior, const IrfftMix* m,
                    hipStream_t s) {
    switch (S) {
        case 8: return launch_irfft2<T, 8, 64>(Y, x, N, C, scale, w_interior, m, s);
        case 16: return launch_irfft2<T, 16, 32>(Y, x, N, C, scale, w_interior, m, s);
        case 32:
            if (fft32_wave_on(N, C)) return launch_irfft2_wave<T>(Y, x, N, C, scale, w_interior, m, s);
            return launch_irfft2<T, 32, 16>(Y, x, N, C, scale, w_interior, m, s);
        case 64: return launch_irfft2<T, 64, 8>(Y, x, N, C, scale, w_interior, m, s);
        case 12: return launch_irfft2<T, 12, 42>(Y, x, N, C, scale, w_interior, m, s);          // 3 * 2^k: both storage types (the 380 x 380 trunk)
        case 24: return launch_irfft2<T, 24, 21>(Y, x, N, C, scale, w_interior, m, s);
        case 48: return launch_irfft2<T, 48, 10>(Y, x, N, C, scale, w_interior, m, s);
        default: break;
    }
    if constexpr (std::is_same<T, float>::value) {
        switch (S) {
            case 10: return launch_irfft2<T, 10, 51>(Y, x, N, C, scale, w_interior, m, s);
            case 20: return launch_irfft2<T, 20, 25>(Y, x, N, C, scale, w_interior, m, s);
            case 40: return launch_irfft2<T, 40, 12>(Y, x, N, C, scale, w_interior, m, s);
            case 80: return launch_irfft2<T, 80, 5>(Y, x, N, C, scale, w_interior, m, s);
            default: break;
        }
    }
    return UD_EINVAL;
}

}  // namespace

extern "C" {

// form of the S = 32 transforms: 0 auto (default), 1 one lane per image row, 2 lane pairs (rfft2_wave_kernel); returns the
// previous setting
int ud_fft32_set_wave(int mode) {
    const int prev = g_fft32_wave < 0 ? 0 : g_fft32_wave == 0 ? 1 : 2;
    if (mode < 0 || mode > 2) return UD_EINVAL;
    g_fft32_wave = mode - 1 < 0 ? -1 : mode - 1;
    return prev;
}

int ud_rfft2(const void* x, void* Y, int N, int S, int C, float scale, float w_interior, int f16, ud_stream_t stream) {
    if (N < 1 || C < 1) return UD_EINVAL;
    UD_STORAGE_DISPATCH(f16, return rfft2_dispatch<T>((const T*)x, (T*)Y, N, S, C, scale, w_interior, nullptr,
                                                      (hipStream_t)stream));
}

int ud_irfft2(const void* Y, void* x, int N, int S, int C, float scale, float w_interior, int f16, ud_stream_t stream) {
    if (N < 1 || C < 1) return UD_EINVAL;
    UD_STORAGE_DISPATCH(f16, return irfft2_dispatch<T>((const T*)Y, (T*)x, N, S, C, scale, w_interior, nullptr,
                                                       (hipStream_t)stream));
}

int ud_rfft2_ex(const void* x, void* Y, int N, int S, int C, float scale, float w_interior, const ud_bn_ref* bn,
                void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc, float* gate_grad,
                int f16, uint32_t* absmax, ud_stream_t stream) {
    if (N < 1 || C < 1 || !x || !Y) return UD_EINVAL;
    if (gate_mode < 0 || gate_mode > 2 || (gate_mode != 0 && !gate_alpha)) return UD_EINVAL;
    if (bn && bn->G != 1) return UD_EINVAL;
    if (act_out && !bn) return UD_EINVAL;
    if (gate_grad && (!gate_acc || !gate_alpha)) return UD_EINVAL;
    RfftEx ex{bn, act_out, gate_alpha, gate_mode, gate_acc, gate_grad, absmax};
    UD_STORAGE_DISPATCH(f16, return rfft2_dispatch<T>((const T*)x, (T*)Y, N, S, C, scale, w_interior, &ex,
                                                      (hipStream_t)stream));
}

int ud_rfft2_ex_planes(const void* x, uint16_t* planes, long panel_stride, long plane_stride, float* inv_scale, float bound_pre,
                       const double* energy, int N, int S, int C, float scale, float w_interior, const ud_bn_ref* bn,
                       void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc, float* gate_grad,
                       const float* dw_wt, void* dw_out, int dw_k, ud_stream_t stream) {
    if (dw_k != 0 && (!dw_wt || !dw_out || (dw_k != 3 && dw_k != 5) || (S != 8 && S != 16 && S != 32))) return UD_EINVAL;
    if (N < 1 || C < 4 || (2 * C) % 32 || !x || !planes || !inv_scale || !(bound_pre > 0.f)) return UD_EINVAL;
    if (S != 8 && S != 16 && S != 32 && S != 12 && S != 24 && S != 48) return UD_EINVAL;          // the one-kernel forms
    if (gate_mode < 0 || gate_mode > 2 || (gate_mode != 0 && !gate_alpha)) return UD_EINVAL;
    if (bn && bn->G != 1) return UD_EINVAL;
    if (!bn && !energy) return UD_EINVAL;          // a bound needs one of the two
    if (act_out && !bn) return UD_EINVAL;
    if (gate_grad && (!gate_acc || !gate_alpha)) return UD_EINVAL;
    const long rows = (long)N * S * (S / 2 + 1);
    if (panel_stride < rows * 32 || panel_stride % 8 || plane_stride % 8 || plane_stride < (long)(2 * C / 32) * panel_stride)
        return UD_EINVAL;
    RfftEx ex{bn, act_out, gate_alpha, gate_mode, gate_acc, gate_grad, nullptr,
              PlanesOut{planes, panel_stride, plane_stride, inv_scale, bound_pre, energy}, DwOut{dw_wt, dw_out}};
    hipStream_t st = (hipStream_t)stream;
    const float* xf = (const float*)x;
    if (dw_k == 3) {
        if (S == 8) return launch_rfft2_t<float, 8, 64, true, 3>(xf, nullptr, N, C, scale, w_interior, ex, st);
        if (S == 16) return launch_rfft2_t<float, 16, 32, true, 3>(xf, nullptr, N, C, scale, w_interior, ex, st);
        return launch_rfft2_t<float, 32, 16, true, 3>(xf, nullptr, N, C, scale, w_interior, ex, st);
    }
    if (dw_k == 5) {
        if (S == 8) return launch_rfft2_t<float, 8, 64, true, 5>(xf, nullptr, N, C, scale, w_interior, ex, st);
        if (S == 16) return launch_rfft2_t<float, 16, 32, true, 5>(xf, nullptr, N, C, scale, w_interior, ex, st);
        return launch_rfft2_t<float, 32, 16, true, 5>(xf, nullptr, N, C, scale, w_interior, ex, st);
    }
    return rfft2_dispatch<float>(xf, (float*)nullptr, N, S, C, scale, w_interior, &ex, st);
}

int ud_rfft2_ex_plane_half(const void* x, uint16_t* plane, long panel_stride, float* inv_scale, int N, int S, int C, float scale,
                           float w_interior, const ud_bn_ref* bn, void* act_out, const float* gate_alpha, int gate_mode,
                           const double* gate_acc, float* gate_grad, const float* dw_wt, void* dw_out, int dw_k,
                           ud_stream_t stream) {
    if (dw_k != 0 && (!dw_wt || !dw_out || (dw_k != 3 && dw_k != 5) || (S != 8 && S != 16 && S != 32))) return UD_EINVAL;
    if (N < 1 || C < 4 || (2 * C) % 32 || !x || !plane || !inv_scale) return UD_EINVAL;
    if (S != 8 && S != 16 && S != 32 && S != 12 && S != 24 && S != 48) return UD_EINVAL;          // the one-kernel forms
    if (gate_mode < 0 || gate_mode > 2 || (gate_mode != 0 && !gate_alpha)) return UD_EINVAL;
    if (bn && bn->G != 1) return UD_EINVAL;
    if (act_out && !bn) return UD_EINVAL;
    if (gate_grad && (!gate_acc || !gate_alpha)) return UD_EINVAL;
    const long rows = (long)N * S * (S / 2 + 1);
    if (panel_stride < rows * 32 || panel_stride % 8) return UD_EINVAL;
    RfftEx ex{bn, act_out, gate_alpha, gate_mode, gate_acc, gate_grad, nullptr,
              PlanesOut{plane, panel_stride, 0, inv_scale, -1.f, nullptr}, DwOut{dw_wt, dw_out}};
    hipStream_t st = (hipStream_t)stream;
    const _Float16* xh = (const _Float16*)x;
    if (dw_k == 3) {
        if (S == 8) return launch_rfft2_t<_Float16, 8, 64, true, 3>(xh, nullptr, N, C, scale, w_interior, ex, st);
        if (S == 16) return launch_rfft2_t<_Float16, 16, 32, true, 3>(xh, nullptr, N, C, scale, w_interior, ex, st);
        return launch_rfft2_t<_Float16, 32, 16, true, 3>(xh, nullptr, N, C, scale, w_interior, ex, st);
    }
    if (dw_k == 5) {
        if (S == 8) return launch_rfft2_t<_Float16, 8, 64, true, 5>(xh, nullptr, N, C, scale, w_interior, ex, st);
        if (S == 16) return launch_rfft2_t<_Float16, 16, 32, true, 5>(xh, nullptr, N, C, scale, w_interior, ex, st);
        return launch_rfft2_t<_Float16, 32, 16, true, 5>(xh, nullptr, N, C, scale, w_interior, ex, st);
    }
    return rfft2_dispatch<_Float16>(xh, (_Float16*)nullptr, N, S, C, scale, w_interior, &ex, st);
}

int ud_irfft2_dwbwd(const void* Y, int N, int S, int C, float scale, float w_interior, const void* dd, const void* x,
                    const ud_bn_ref* bn, const float* wt, int K, const float* gate_alpha, int gate_mode, void* dz, double* s1,
                    double* s2, double* s3, float* wpart, float* wacc, int f16, ud_stream_t stream) {
    if (N < 1 || C < 1 || !Y || !dd || !x || !bn || bn->G != 1 || !wt || !dz || !s1 || !s2 || (!wpart && !wacc)) return UD_EINVAL;
    if (gate_mode < 0 || gate_mode > 2 || (gate_mode != 0 && !gate_alpha)) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define UD_IDW(SS, CC, KK)                                                                                                       \
    UD_STORAGE_DISPATCH(f16, return (launch_irfft2_dwbwd<T, SS, CC, KK>((const T*)Y, N, C, scale, w_interior, (const T*)dd,     \
                                                                        (const T*)x, *bn, wt, gate_alpha, gate_mode, (T*)dz, s1, \
                                                                        s2, s3, wpart, wacc, st)))
    if (S == 8 && K == 5) UD_IDW(8, 64, 5);
    if (S == 8 && K == 3) UD_IDW(8, 64, 3);
    if (S == 16 && K == 5) UD_IDW(16, 32, 5);
    if (S == 16 && K == 3) UD_IDW(16, 32, 3);
#undef UD_IDW
    return UD_EINVAL;
}

int ud_irfft2_mix(const void* Y, void* y, int N, int S, int C, float scale, float w_interior, const void* spat,
                  const float* alpha, void* freq_out, double* sum, double* sumsq, int f16, ud_stream_t stream) {
    if (N < 1 || C < 1 || !Y || !y || !spat || !alpha || !freq_out || !sum || !sumsq) return UD_EINVAL;
    IrfftMix m{spat, alpha, freq_out, sum, sumsq};
    UD_STORAGE_DISPATCH(f16, return irfft2_dispatch<T>((const T*)Y, (T*)y, N, S, C, scale, w_interior, &m,
                                                       (hipStream_t)stream));
}

// floats of scratch the two-pass forms need (the half-spectrum between the row and the column kernel); 0: S has no
// two-pass form
long ud_fft2_two_pass_ws_floats(int N, int S, int C) {
    if (N < 1 || C < 1 || (S != 32 && S != 64)) return 0;
    return (long)N * S * (S / 2 + 1) * 2 * C;
}

int ud_rfft2_two_pass(const void* x, void* Y, float* ws, int N, int S, int C, float scale, float w_interior,
                      const ud_bn_ref* bn, void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc,
                      float* gate_grad, int f16, uint32_t* absmax, ud_stream_t stream) {
    if (N < 1 || C < 1 || !x || !Y || !ws || (S != 32 && S != 64)) return UD_EINVAL;
    if (gate_mode < 0 || gate_mode > 2 || (gate_mode != 0 && !gate_alpha)) return UD_EINVAL;
    if (bn && bn->G != 1) return UD_EINVAL;
    if (act_out && !bn) return UD_EINVAL;
    if (gate_grad && (!gate_acc || !gate_alpha)) return UD_EINVAL;
    RfftEx exv{bn, act_out, gate_alpha, gate_mode, gate_acc, gate_grad, absmax};
    const RfftEx* ex = (bn || gate_mode != 0 || gate_grad || absmax) ? &exv : nullptr;
    hipStream_t s = (hipStream_t)stream;
    UD_STORAGE_DISPATCH(f16, if (S == 64) return rfft2_two_pass<T, 64>((const T*)x, (T*)Y, ws, N, C, scale, w_interior, ex, s);
                        return rfft2_two_pass<T, 32>((const T*)x, (T*)Y, ws, N, C, scale, w_interior, ex, s));
}

int ud_irfft2_two_pass(const void* Y, void* y, float* ws, int N, int S, int C, float scale, float w_interior,
                       const void* spat, const float* alpha, void* freq_out, double* sum, double* sumsq, int f16,
                       ud_stream_t stream) {
    if (N < 1 || C < 1 || !Y || !y || !ws || (S != 32 && S != 64)) return UD_EINVAL;
    if (spat && (!alpha || !freq_out || !sum || !sumsq)) return UD_EINVAL;
    IrfftMix mv{spat, alpha, freq_out, sum, sumsq};
    const IrfftMix* m = spat ? &mv : nullptr;
    hipStream_t s = (hipStream_t)stream;
    UD_STORAGE_DISPATCH(f16, if (S == 64) return irfft2_two_pass<T, 64>((const T*)Y, (T*)y, ws, N, C, scale, w_interior, m, s);
                        return irfft2_two_pass<T, 32>((const T*)Y, (T*)y, ws, N, C, scale, w_interior, m, s));
}

}  // extern "C"
