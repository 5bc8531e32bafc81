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


// ---- weight gradient of the same convs ---------------------------------------------------------------------------
//   out[ma][k] = sum_m a[m][ma] * patch(x)[m][k],   m = (n, oh, ow) over g's output grid, k = tap*CIN + ci
// (ud_gemm's b_mode 2).  As a GEMM this is a 20 x 180 (or 3 x 180, 48 x 27) result reduced over 524 288 rows: one
// or two MFMA tiles and a split-K of 256+ with atomics — 146-263 us per call.  Here a workgroup streams a chunk of
// rows through LDS in tiles of TP pixels (the a-rows and the gathered patches), every thread keeps a 4 x 4 block of
// the result in registers, and the per-workgroup results go to a partial buffer that a second launch sums.
constexpr int TP = 28;            // pixels per tile: TP * 9 (pixel, tap) staging items = 252 <= one per thread

template <int CIN, int MA>
__global__ __launch_bounds__(NT) void conv_small_wgrad(ud_conv_geom g, const float* __restrict__ a,
                                                       const float* __restrict__ x, float* __restrict__ part, long M,
                                                       long rows_per_block) {
    constexpr int K = 9 * CIN, KP = (K + 3) / 4 * 4, MAP = (MA + 3) / 4 * 4;
    constexpr int KB = KP / 4, CB = MAP / 4;
    constexpr int AV = (TP * MAP + NT - 1) / NT;          // a-values staged per thread
    static_assert(KB * CB <= NT, "one 4x4 register block per thread");
    static_assert(TP * 9 <= NT, "one (pixel, tap) staging item per thread");
    __shared__ __attribute__((aligned(16))) float As[TP][MAP];
    __shared__ __attribute__((aligned(16))) float Xs[TP][KP];
    const int t = threadIdx.x;
    const bool worker = t < KB * CB;
    const int kb = t % KB, cb = t / KB;
    const bool stager = t < TP * 9;
    const int sp = t / 9, stap = t - 9 * sp;              // this thread's staging item: pixel sp of the tile, tap stap
    const int skh = stap / 3, skw = stap - 3 * skh;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const long m_begin = (long)blockIdx.x * rows_per_block;
    long m_end = m_begin + rows_per_block;
    if (m_end > M) m_end = M;
    if (KP > K) {       // zero the padding columns once
        for (int i = t; i < TP * (KP - K); i += NT) Xs[i / (KP - K)][K + i % (KP - K)] = 0.f;
    }
    float xr[CIN], ar[AV];
    // global loads of one tile into registers (issued a whole tile ahead of their use)
    auto fetch = [&](long m0) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int i = t + j * NT;
            const int p = i / MAP, c = i - p * MAP;
            const long m = m0 + p;
            ar[j] = (i < TP * MAP && m < m_end && c < MA) ? a[m * MA + c] : 0.f;
        }
        bool ok = stager && (m0 + sp) < m_end;
        int n = 0, ih = 0, iw = 0;
        if (ok) {
            const long m = m0 + sp;
            const int ow = (int)(m % g.Wout);
            const long r = m / g.Wout;
            const int oh = (int)(r % g.Hout);
            n = (int)(r / g.Hout);
            if (!g.transposed) {
                ih = oh * g.stride - g.pad_t + skh;
                iw = ow * g.stride - g.pad_l + skw;
                ok = ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win;
            } else {
                const int th = oh + g.pad_t - skh, tw = ow + g.pad_l - skw;
                ok = th >= 0 && tw >= 0 && (th % g.stride) == 0 && (tw % g.stride) == 0;
                ih = th / g.stride;
                iw = tw / g.stride;
                ok = ok && ih < g.Hin && iw < g.Win;
            }
        }
        if (ok) {
            const float* src = x + (((long)n * g.Hin + ih) * g.Win + iw) * CIN;
            if (CIN % 4 == 0) {
#pragma unroll
                for (int c = 0; c < CIN / 4; ++c) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(src)[c];
                    xr[4 * c] = v[0]; xr[4 * c + 1] = v[1]; xr[4 * c + 2] = v[2]; xr[4 * c + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CIN; ++c) xr[c] = src[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) xr[c] = 0.f;
        }
    };
    fetch(m_begin);
    for (long m0 = m_begin; m0 < m_end; m0 += TP) {
        // ---- registers -> LDS
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int i = t + j * NT;
            if (i < TP * MAP) (&As[0][0])[i] = ar[j];
        }
        if (stager) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) Xs[sp][stap * CIN + c] = xr[c];
        }
        __syncthreads();
        if (m0 + TP < m_end) fetch(m0 + TP);              // next tile's loads fly during this tile's FMAs
        if (worker) {
#pragma unroll 7
            for (int p = 0; p < TP; ++p) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&As[p][cb * 4]);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&Xs[p][kb * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * xv[j];
            }
        }
        __syncthreads();
    }
    if (worker) {
        float* dst = part + (long)blockIdx.x * MA * K;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ma = cb * 4 + i, k = kb * 4 + j;
                if (ma < MA && k < K) dst[ma * K + k] = acc[i][j];
            }
    }
}

// out[i] = sum_b part[b][i]; 16 entries x 16 partial-lanes per workgroup, fp64 accumulation
__global__ __launch_bounds__(NT) void sum_partials(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
    __shared__ double sm[NT];
    const int el = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el;
    double acc = 0.0;
    if (i < n)
        for (int b = pl; b < nparts; b += 16) acc += (double)part[(long)b * n + i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0 && i < n) {
        for (int k = 1; k < 16; ++k) acc += sm[k * 16 + el];
        out[i] = (float)acc;
    }
}

constexpr int WGRAD_BLOCKS = 1024;

template <int CIN, int MA>
int launch_wgrad(const ud_conv_geom& g, const float* a, const float* x, float* part, float* out, hipStream_t s) {
    const long M = (long)g.N * g.Hout * g.Wout;
    long rows = (M + WGRAD_BLOCKS - 1) / WGRAD_BLOCKS;
    rows = (rows + TP - 1) / TP * TP;
    const int blocks = (int)((M + rows - 1) / rows);
    hipLaunchKernelGGL((conv_small_wgrad<CIN, MA>), dim3(blocks), dim3(NT), 0, s, g, a, x, part, M, rows);
    UD_LAUNCH_CHECK();
    const int n = MA * 9 * CIN;
    hipLaunchKernelGGL(sum_partials, dim3(ud_cdiv(n, 16)), dim3(NT), 0, s, part, blocks, n, out);
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
        case 2020: case 2003: case 320: case 348: case 3203: case 332: return 1;          // (32 <-> 3: the UDR50 decoder's image end)
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
        case 3203: return launch<32, 3>(*g, x, wmat, y, s);
        case 332: return launch<3, 32>(*g, x, wmat, y, s);
        default: return UD_EINVAL;
    }
}

// 1 when ud_conv_small_wgrad has a kernel for gathering Cin channels against Ma columns of `a`, else 0
int ud_conv_small_wgrad_supported(int Cin, int Ma, int KH, int KW) {
    if (KH != 3 || KW != 3) return 0;
    switch (Cin * 100 + Ma) {
        case 2020: case 2003: case 348: case 320: case 3203: return 1;
        default: return 0;
    }
}

// floats of scratch `part` must hold (per-workgroup partial results)
long ud_conv_small_wgrad_ws_floats(int Cin, int Ma) { return (long)WGRAD_BLOCKS * Ma * 9 * Cin; }

// out[Ma][9*Cin] = sum over rows m = (n,oh,ow) of g's output grid of a[m][Ma] (x) patch(x)[m][9*Cin]
int ud_conv_small_wgrad(const ud_conv_geom* g, const float* a, const float* x, float* part, float* out, int Ma,
                        ud_stream_t stream) {
    if (!g || !ud_conv_small_wgrad_supported(g->Cin, Ma, g->KH, g->KW) || g->stride < 1 || !part) return UD_EINVAL;
    if (g->N < 1 || g->Hout < 1 || g->Wout < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (g->Cin * 100 + Ma) {
        case 2020: return launch_wgrad<20, 20>(*g, a, x, part, out, s);
        case 2003: return launch_wgrad<20, 3>(*g, a, x, part, out, s);
        case 348: return launch_wgrad<3, 48>(*g, a, x, part, out, s);
        case 320: return launch_wgrad<3, 20>(*g, a, x, part, out, s);
        case 3203: return launch_wgrad<32, 3>(*g, a, x, part, out, s);
        default: return UD_EINVAL;
    }
}

}  // extern "C"
