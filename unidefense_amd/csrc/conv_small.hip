// Direct 3x3 convolution for SMALL channel counts (<= 48): the image-resolution end of the reconstruction decoder
// (model/unidefense.py:59-102: 40 -> 20 -> 3 channels at 64x64 / 128x128), its data gradients, and the stem conv
// (3 -> 48, model/efficientnet/model.py:185).
//
// On the implicit-GEMM path (ud_gemm, a_mode 2) these shapes have N = 3..40 output columns: a 32-wide MFMA tile is
// 10-60 % padding and the 9x gather amplification makes the launch L1/L2-bound (measured 4-37 TFLOP/s).  Here:
// one thread = one output pixel, all COUT accumulators in registers, the input taps read as float4 runs straight
// from global memory (neighbouring lanes overlap in L1), the weights wave-uniform -> scalar loads feeding the FMAs
// as SGPR operands (no LDS, no matrix core).  Same gather rule as ud_conv_geom (stride, asymmetric padding,
// transposed), so forward, data gradient and the transposed conv all use this one kernel.
// Measured (bs 32): 20->20 at 128x128 62 us vs 160 us on the GEMM path, 20->3 25 vs 146, 3->20 25 vs 82, stem 3->48
// 54 vs 99.  NOT used for the 40-channel stage at 64x64 (131072 pixels = 2 waves per SIMD: 162 us vs 107 us).
//   y[m][co] = sum_{tap,ci} x[src(m, tap)][ci] * wmat[co][tap*CIN + ci]        m = (n, oh, ow)
#include "ud_common.h"

namespace {

constexpr int NT = 256;

template <int CIN, int COUT>
__global__ __launch_bounds__(NT) void conv_small(ud_conv_geom g, const float* __restrict__ x,
                                                 const float* __restrict__ wmat, float* __restrict__ y, long M) {
    constexpr int K = 9 * CIN;
    const long m = (long)blockIdx.x * NT + threadIdx.x;
    if (m >= M) return;
    const int ow = (int)(m % g.Wout);
    const long t = m / g.Wout;
    const int oh = (int)(t % g.Hout);
    const int n = (int)(t / g.Hout);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        int ih, iw;
        bool ok;
        if (!g.transposed) {
            ih = oh * g.stride - g.pad_t + kh;
            iw = ow * g.stride - g.pad_l + kw;
            ok = ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win;
        } else {
            const int th = oh + g.pad_t - kh, tw = ow + g.pad_l - kw;
            ok = th >= 0 && tw >= 0 && (th % g.stride) == 0 && (tw % g.stride) == 0;
            ih = th / g.stride;
            iw = tw / g.stride;
            ok = ok && ih < g.Hin && iw < g.Win;
        }
        if (!ok) continue;
        const float* src = x + (((long)n * g.Hin + ih) * g.Win + iw) * CIN;
        float xv[CIN];
        if (CIN % 4 == 0) {
#pragma unroll
            for (int c = 0; c < CIN / 4; ++c) {
                const f32x4 v = reinterpret_cast<const f32x4*>(src)[c];
                xv[4 * c] = v[0]; xv[4 * c + 1] = v[1]; xv[4 * c + 2] = v[2]; xv[4 * c + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) xv[c] = src[c];
        }
        const float* wt = wmat + tap * CIN;            // wave-uniform: scalar loads
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[co] += xv[c] * wt[co * K + c];
        }
    }
    float* dst = y + m * COUT;
    if (COUT % 4 == 0) {
#pragma unroll
        for (int c = 0; c < COUT / 4; ++c)
            reinterpret_cast<f32x4*>(dst)[c] = f32x4{acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
    } else {
#pragma unroll
        for (int c = 0; c < COUT; ++c) dst[c] = acc[c];
    }
}

template <int CIN, int COUT>
int launch(const ud_conv_geom& g, const float* x, const float* wmat, float* y, hipStream_t s) {
    const long M = (long)g.N * g.Hout * g.Wout;
    hipLaunchKernelGGL((conv_small<CIN, COUT>), dim3((unsigned)ud_cdiv(M, NT)), dim3(NT), 0, s, g, x, wmat, y, M);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

// 1 when ud_conv_small has a kernel for (Cin, Cout) with a 3x3 window, else 0
int ud_conv_small_supported(int Cin, int Cout, int KH, int KW) {
    if (KH != 3 || KW != 3) return 0;
    const int key = Cin * 100 + Cout;
    switch (key) {
        case 2020: case 2003: case 320: case 348: return 1;
        default: return 0;
    }
}

// y[N][Hout][Wout][Cout] = gather-conv(x[N][Hin][Win][Cin], wmat[Cout][9*Cin]) under geometry g (see ud_conv_geom)
int ud_conv_small(const ud_conv_geom* g, const float* x, const float* wmat, float* y, int Cout, ud_stream_t stream) {
    if (!g || !ud_conv_small_supported(g->Cin, Cout, g->KH, g->KW) || g->stride < 1) return UD_EINVAL;
    if (g->N < 1 || g->Hout < 1 || g->Wout < 1) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (g->Cin * 100 + Cout) {
        case 2020: return launch<20, 20>(*g, x, wmat, y, s);
        case 2003: return launch<20, 3>(*g, x, wmat, y, s);
        case 320: return launch<3, 20>(*g, x, wmat, y, s);
        case 348: return launch<3, 48>(*g, x, wmat, y, s);
        default: return UD_EINVAL;
    }
}

}  // extern "C"
