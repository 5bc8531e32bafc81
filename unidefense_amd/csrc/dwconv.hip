// Depthwise k x k convolution (k = 3, 5; stride 1, 2; static asymmetric zero padding) on pixel-major
// [N][H][W][C] fp32, forward + data gradient + weight gradient.
//
// Serves the spatial branch of SFConv2dStaticSamePadding.forward (model/efficientnet/exp.py:49-51) and the
// plain depthwise Conv2dStaticSamePadding (model/efficientnet/utils.py:277-280) — ZeroPad2d(left, right,
// top, bottom) followed by an unpadded grouped conv; the pads are passed explicitly (pad_t, pad_l; the
// bottom/right pads are implied by the output size).
//
// HBM-bound: lanes run over channels (4 per lane, 16-B loads), so every global access of a wave is one
// contiguous segment; the k*k taps re-read neighbouring pixels through L1/L2.  Weights are passed
// tap-major wt[k*k][C] so that the per-lane weight loads are coalesced too.
#include "ud_common.h"

namespace {

constexpr int NT = 256;

struct DwGeom {
    int N, H, W, C4, Ho, Wo, stride, pad_t, pad_l;
};

template <typename T, int K>
__global__ __launch_bounds__(NT) void dw_bwd_data(DwGeom q, const T* __restrict__ dy, const float* __restrict__ wt,
                                                  const T* __restrict__ add, T* __restrict__ dx) {
    const In4<T> dy4{dy}, add4{add};
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wt);
    const Out4<T> dx4{dx};
    const long total = (long)q.N * q.H * q.W * q.C4;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % q.C4);
        long pix = e / q.C4;
        int w = (int)(pix % q.W);
        long t = pix / q.W;
        int h = (int)(t % q.H);
        int n = (int)(t / q.H);
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            int th = h + q.pad_t - kh;
            if (th < 0 || (th % q.stride) != 0) continue;
            int ho = th / q.stride;
            if (ho >= q.Ho) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                int tw = w + q.pad_l - kw;
                if (tw < 0 || (tw % q.stride) != 0) continue;
                int wo = tw / q.stride;
                if (wo >= q.Wo) continue;
                f32x4 g = dy4[(((long)n * q.Ho + ho) * q.Wo + wo) * q.C4 + c4];
                f32x4 ww = w4[(kh * K + kw) * q.C4 + c4];
                acc += g * ww;
            }
        }
        if (add) acc += add4[e];
        dx4.st(e, acc);
    }
}

// ---- strip forms: one thread = 4 channels x TW consecutive output columns ---------------------------------------
// A row of the window is loaded once ((TW-1)*S + K float4) and serves all TW outputs, the K taps of the row are
// loaded once per thread: (TW-1)*S+K + K loads per TW*K multiply-adds instead of 2 per multiply-add.  The plain
// kernels above issue K*K (9 / 25) 16-byte loads per output and sit at ~30 % of their HBM roofline (L1-bound).
constexpr int TW = 8;

template <typename T, int K, int S>
__global__ __launch_bounds__(NT) void dw_fwd_strip(DwGeom q, const T* __restrict__ x, const float* __restrict__ wt,
                                                   T* __restrict__ y) {
    constexpr int NCOL = (TW - 1) * S + K;
    const In4<T> x4{x};
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wt);
    const Out4<T> y4{y};
    const int WoB = (q.Wo + TW - 1) / TW;
    const long total = (long)q.N * q.Ho * WoB * q.C4;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % q.C4);
        long r = e / q.C4;
        int wb = (int)(r % WoB);
        long t2 = r / WoB;
        int ho = (int)(t2 % q.Ho);
        int n = (int)(t2 / q.Ho);
        const int wo0 = wb * TW;
        const int ih0 = ho * S - q.pad_t, iw0 = wo0 * S - q.pad_l;
        f32x4 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = ih0 + kh;
            if (ih < 0 || ih >= q.H) continue;
            const In4<T> row = x4 + ((((long)n * q.H + ih) * q.W) * q.C4 + c4);
            f32x4 in[NCOL], w[K];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int iw = iw0 + j;
                in[j] = (iw >= 0 && iw < q.W) ? row[(long)iw * q.C4] : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw) w[kw] = w4[(kh * K + kw) * q.C4 + c4];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[t] += in[t * S + kw] * w[kw];
        }
        const Out4<T> out = y4 + ((((long)n * q.Ho + ho) * q.Wo + wo0) * q.C4 + c4);
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (wo0 + t < q.Wo) out.st((long)t * q.C4, acc[t]);
    }
}

// data gradient, stride 1: dx[h][w] = sum dy[h + pad_t - kh][w + pad_l - kw] * w[kh][kw]
template <typename T, int K>
__global__ __launch_bounds__(NT) void dw_bwd_data_strip(DwGeom q, const T* __restrict__ dy,
                                                        const float* __restrict__ wt, const T* __restrict__ add,
                                                        T* __restrict__ dx) {
    constexpr int NCOL = TW + K - 1;
    const In4<T> dy4{dy};
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wt);
    const Out4<T> dx4{dx};
    const int WB = (q.W + TW - 1) / TW;
    const long total = (long)q.N * q.H * WB * q.C4;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % q.C4);
        long r = e / q.C4;
        int wb = (int)(r % WB);
        long t2 = r / WB;
        int h = (int)(t2 % q.H);
        int n = (int)(t2 / q.H);
        const int w0 = wb * TW;
        const int col0 = w0 + q.pad_l - (K - 1);          // dy column of window slot 0
        f32x4 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ho = h + q.pad_t - kh;
            if (ho < 0 || ho >= q.Ho) continue;
            const In4<T> row = dy4 + ((((long)n * q.Ho + ho) * q.Wo) * q.C4 + c4);
            f32x4 g[NCOL], w[K];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int wo = col0 + j;
                g[j] = (wo >= 0 && wo < q.Wo) ? row[(long)wo * q.C4] : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw) w[kw] = w4[(kh * K + kw) * q.C4 + c4];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[t] += g[t - kw + K - 1] * w[kw];
        }
        const long o0 = (((long)n * q.H + h) * q.W + w0) * q.C4 + c4;
        const Out4<T> out = dx4 + o0;
        const In4<T> add4 = In4<T>{add} + o0;
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (w0 + t < q.W) out.st((long)t * q.C4, add ? acc[t] + add4[(long)t * q.C4] : acc[t]);
    }
}

// partial weight gradient: part[(p * rpi + ri)][tap][C]
template <int K>
__global__ __launch_bounds__(NT) void dw_bwd_weight_partial(DwGeom q, int rpi, int pix_per_chunk,
                                                            const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    int t = threadIdx.x, ri, c4;
    bool active;
    if (q.C4 <= NT) {
        ri = t / q.C4; c4 = t % q.C4; active = ri < rpi;
    } else {
        ri = 0; c4 = blockIdx.y * NT + t; active = c4 < q.C4;
    }
    if (!active) return;
    const long npix = (long)q.N * q.Ho * q.Wo;
    const long p_begin = (long)blockIdx.x * pix_per_chunk;
    long p_end = p_begin + pix_per_chunk;
    if (p_end > npix) p_end = npix;
    f32x4 acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (long pix = p_begin + ri; pix < p_end; pix += rpi) {
        int wo = (int)(pix % q.Wo);
        long tt = pix / q.Wo;
        int ho = (int)(tt % q.Ho);
        int n = (int)(tt / q.Ho);
        f32x4 g = dy4[pix * q.C4 + c4];
        const int ih0 = ho * q.stride - q.pad_t, iw0 = wo * q.stride - q.pad_l;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            int ih = ih0 + kh;
            bool okh = (ih >= 0) && (ih < q.H);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                int iw = iw0 + kw;
                if (okh && iw >= 0 && iw < q.W) {
                    f32x4 a = x4[(((long)n * q.H + ih) * q.W + iw) * q.C4 + c4];
                    acc[kh * K + kw] += a * g;
                }
            }
        }
    }
    const long pidx = (long)blockIdx.x * rpi + ri;
    f32x4* out = reinterpret_cast<f32x4*>(part) + pidx * (K * K) * q.C4;
#pragma unroll
    for (int i = 0; i < K * K; ++i) out[(long)i * q.C4 + c4] = acc[i];
}

// Weight gradient, sliding-window form: one thread = one channel (lane = channel, 4-B coalesced accesses) and
// a group of output rows.  Walking along an output row it keeps the K x K input window in registers, so each
// output pixel costs 1 dy load + K*S new window loads instead of K*K.  part[(p * K*K + tap)][C].
template <int K, int S>
__global__ void dw_bwd_weight_rows(DwGeom q, int C, int rows_per_group, const float* __restrict__ x,
                                   const float* __restrict__ dy, float* __restrict__ part) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int p = blockIdx.x;
    const int rows_total = q.N * q.Ho;
    const int row0 = p * rows_per_group;
    int row1 = row0 + rows_per_group;
    if (row1 > rows_total) row1 = rows_total;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int row = row0; row < row1; ++row) {
        const int n = row / q.Ho, ho = row % q.Ho;
        const int ih0 = ho * S - q.pad_t;
        const float* xr[K];
        bool vh[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = ih0 + kh;
            vh[kh] = (ih >= 0) && (ih < q.H);
            xr[kh] = x + (((long)n * q.H + (vh[kh] ? ih : 0)) * q.W) * C + c;
        }
        float w[K][K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int iw = kw - q.pad_l;
                w[kh][kw] = (vh[kh] && iw >= 0 && iw < q.W) ? xr[kh][(long)iw * C] : 0.f;
            }
        const float* dyr = dy + ((long)row * q.Wo) * C + c;
#pragma unroll 4
        for (int wo = 0; wo < q.Wo; ++wo) {
            const float g = dyr[(long)wo * C];
#pragma unroll
            for (int kh = 0; kh < K; ++kh)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[kh * K + kw] += g * w[kh][kw];
            // slide the window by S columns
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
#pragma unroll
                for (int kw = 0; kw + S < K; ++kw) w[kh][kw] = w[kh][kw + S];
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const int iw = (wo + 1) * S - q.pad_l + (K - S) + j;
                    w[kh][K - S + j] = (vh[kh] && iw >= 0 && iw < q.W) ? xr[kh][(long)iw * C] : 0.f;
                }
            }
        }
    }
    float* out = part + (long)p * (K * K) * C + c;
#pragma unroll
    for (int i = 0; i < K * K; ++i) out[(long)i * C] = acc[i];
}

// 16 outputs x 16 part-lanes per block; fp64 accumulation of the per-chunk partial sums
// partials are tap-major [tap][C]; the result is written in the PARAMETER's layout dw[C][K*K] (weight [C,1,k,k])
__global__ __launch_bounds__(NT) void dw_bwd_weight_finalize(int nparts, int KK, int C, const float* __restrict__ part,
                                                             float* __restrict__ dw) {
    __shared__ double sm[NT];
    const int KKC = KK * C;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double a = 0.0;
    if (i < KKC)
        for (int p = pl; p < nparts; p += 16) a += (double)part[(long)p * KKC + i];
    sm[threadIdx.x] = a;
    __syncthreads();
    if (i < KKC && pl == 0) {
        for (int k = 1; k < 16; ++k) a += sm[k * 16 + cl];
        const int tap = i / C, c = i % C;
        dw[(long)c * KK + tap] = (float)a;
    }
}

int ew_blocks(long total) {
    long b = (total + NT - 1) / NT;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

bool geom_ok(const DwGeom& q, int K) {
    return q.N > 0 && q.H > 0 && q.W > 0 && q.C4 > 0 && q.Ho > 0 && q.Wo > 0 && (q.stride == 1 || q.stride == 2) &&
           (K == 3 || K == 5);
}

}  // namespace

extern "C" {

int ud_dwconv_fwd(const void* xv, const float* wt, void* yv, int N, int H, int W, int C, int Ho, int Wo, int K,
                  int stride, int pad_t, int pad_l, int f16, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    DwGeom q{N, H, W, C / 4, Ho, Wo, stride, pad_t, pad_l};
    if (!geom_ok(q, K)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    {
        long total = (long)N * Ho * ((Wo + TW - 1) / TW) * q.C4;
        dim3 g(ew_blocks(total));
        UD_STORAGE_DISPATCH(f16, const T* x = (const T*)xv; T* y = (T*)yv;
                            if (K == 3 && stride == 1) hipLaunchKernelGGL((dw_fwd_strip<T, 3, 1>), g, dim3(NT), 0, s, q, x, wt, y);
                            else if (K == 3) hipLaunchKernelGGL((dw_fwd_strip<T, 3, 2>), g, dim3(NT), 0, s, q, x, wt, y);
                            else if (stride == 1) hipLaunchKernelGGL((dw_fwd_strip<T, 5, 1>), g, dim3(NT), 0, s, q, x, wt, y);
                            else hipLaunchKernelGGL((dw_fwd_strip<T, 5, 2>), g, dim3(NT), 0, s, q, x, wt, y));
    }
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_dwconv_bwd_data(const void* dyv, const float* wt, const void* addv, void* dxv, int N, int H, int W, int C, int Ho,
                       int Wo, int K, int stride, int pad_t, int pad_l, int f16, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    DwGeom q{N, H, W, C / 4, Ho, Wo, stride, pad_t, pad_l};
    if (!geom_ok(q, K)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1) {
        long total = (long)N * H * ((W + TW - 1) / TW) * q.C4;
        dim3 g(ew_blocks(total));
        UD_STORAGE_DISPATCH(f16, const T* dy = (const T*)dyv; const T* add = (const T*)addv; T* dx = (T*)dxv;
                            if (K == 3) hipLaunchKernelGGL((dw_bwd_data_strip<T, 3>), g, dim3(NT), 0, s, q, dy, wt, add, dx);
                            else hipLaunchKernelGGL((dw_bwd_data_strip<T, 5>), g, dim3(NT), 0, s, q, dy, wt, add, dx));
    } else {       // stride 2 (4 of the 32 blocks): one output per thread
        long total = (long)N * H * W * q.C4;
        dim3 g(ew_blocks(total));
        UD_STORAGE_DISPATCH(f16, const T* dy = (const T*)dyv; const T* add = (const T*)addv; T* dx = (T*)dxv;
                            if (K == 3) hipLaunchKernelGGL((dw_bwd_data<T, 3>), g, dim3(NT), 0, s, q, dy, wt, add, dx);
                            else hipLaunchKernelGGL((dw_bwd_data<T, 5>), g, dim3(NT), 0, s, q, dy, wt, add, dx));
    }
    UD_LAUNCH_CHECK();
    return 0;
}

// number of float partial rows (each K*K*C floats) the weight-gradient pass needs for `chunks` chunks
int ud_dwconv_bwd_weight_parts(int C, int chunks) {
    (void)C;
    return chunks;      // one partial row of K*K*C floats per row-group
}

int ud_dwconv_bwd_weight(const float* x, const float* dy, float* dwt, float* part, int chunks, int N, int H, int W,
                         int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, ud_stream_t stream) {
    if (C % 4 || chunks < 1) return UD_EINVAL;
    DwGeom q{N, H, W, C / 4, Ho, Wo, stride, pad_t, pad_l};
    if (!geom_ok(q, K)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int rows_total = N * Ho;
    if (chunks > rows_total) return UD_EINVAL;
    const int rpg = (rows_total + chunks - 1) / chunks;     // output rows per group; trailing groups may be empty
    int bt = ((C < NT ? C : NT) + 63) / 64 * 64;             // threads per block: whole waves covering min(C, 256)
    dim3 grid((unsigned)chunks, (unsigned)ud_cdiv(C, bt), 1);
    // groups beyond the row range write zeros (their loops are empty), so the finalize can sum all parts
    if (K == 3 && stride == 1) hipLaunchKernelGGL((dw_bwd_weight_rows<3, 1>), grid, dim3(bt), 0, s, q, C, rpg, x, dy, part);
    else if (K == 3) hipLaunchKernelGGL((dw_bwd_weight_rows<3, 2>), grid, dim3(bt), 0, s, q, C, rpg, x, dy, part);
    else if (stride == 1) hipLaunchKernelGGL((dw_bwd_weight_rows<5, 1>), grid, dim3(bt), 0, s, q, C, rpg, x, dy, part);
    else hipLaunchKernelGGL((dw_bwd_weight_rows<5, 2>), grid, dim3(bt), 0, s, q, C, rpg, x, dy, part);
    UD_LAUNCH_CHECK();
    int KKC = K * K * C;
    hipLaunchKernelGGL(dw_bwd_weight_finalize, dim3(ud_cdiv(KKC, 16)), dim3(NT), 0, s, chunks, K * K, C, part, dwt);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
