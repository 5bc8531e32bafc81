// Pooling / residual-ReLU / channel-concat kernels of the ResNet variants (pixel-major [N][H][W][C] fp32,
// 4 channels per lane).  Reference call sites: F.adaptive_avg_pool2d to an integer fraction of the size
// (model/resnet/module_exp.py:30-31; also the SFConv 2x2 pool), nn.MaxPool2d(3, 2, 1) (module_exp.py:73-75),
// `x += shortcut; act(x)` (model/resnet/exp.py:146-147, module_exp.py:86-88,108-109) and torch.cat(dim=1)
// (module_exp.py:32).  All HBM-bound, one coalesced pass each, no atomics.
#include "ud_common.h"

namespace {

constexpr int NT = 256;

inline int ew_blocks(long total, int cap = 8192) {
    long b = (total + NT - 1) / NT;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// y[n][ho][wo][c] = mean over the k x k block
__global__ __launch_bounds__(NT) void avgpool_fwd(long total4, int Ho, int Wo, int C4, int k,
                                                  const float* __restrict__ x, float* __restrict__ y) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    const int W = Wo * k, H = Ho * k;
    const float inv = 1.f / (float)(k * k);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % C4);
        long pix = e / C4;
        int wo = (int)(pix % Wo);
        long t = pix / Wo;
        int ho = (int)(t % Ho);
        long n = t / Ho;
        f32x4 acc = {0, 0, 0, 0};
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) acc += x4[((n * H + ho * k + i) * W + wo * k + j) * C4 + c4];
        y4[e] = acc * inv;
    }
}

// dx[n][h][w][c] = dy[n][h/k][w/k][c] / k^2
__global__ __launch_bounds__(NT) void avgpool_bwd(long total4, int Ho, int Wo, int C4, int k,
                                                  const float* __restrict__ dy, float* __restrict__ dx) {
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* o4 = reinterpret_cast<f32x4*>(dx);
    const int W = Wo * k, H = Ho * k;
    const float inv = 1.f / (float)(k * k);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % C4);
        long pix = e / C4;
        int w = (int)(pix % W);
        long t = pix / W;
        int h = (int)(t % H);
        long n = t / H;
        o4[e] = d4[((n * Ho + h / k) * Wo + w / k) * C4 + c4] * inv;
    }
}

// F.adaptive_avg_pool2d to ANY output size (model/efficientnet/exp.py:61-62 on the 95 x 95 map of the 380 x 380 trunk's stride-2 SF
// block: 95 -> 48, windows of 2 and 3 rows that overlap): output index o averages input rows [floor(o * H / Ho), ceil((o + 1) *
// H / Ho)) — ATen's start_index / end_index (aten/src/ATen/native/AdaptivePooling.h).  Rounds 3-4 called ATen's kernel here.
__device__ __forceinline__ int ada_start(int o, int out, int in) { return (int)(((long)o * in) / out); }
__device__ __forceinline__ int ada_end(int o, int out, int in) { return (int)(((long)(o + 1) * in + out - 1) / out); }

__global__ __launch_bounds__(NT) void adaptive_avgpool_fwd(long total4, int H, int W, int Ho, int Wo, int C4,
                                                           const float* __restrict__ x, float* __restrict__ y) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        const int c4 = (int)(e % C4);
        long pix = e / C4;
        const int wo = (int)(pix % Wo);
        long t = pix / Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const int h0 = ada_start(ho, Ho, H), h1 = ada_end(ho, Ho, H), w0 = ada_start(wo, Wo, W), w1 = ada_end(wo, Wo, W);
        f32x4 acc = {0, 0, 0, 0};
        for (int i = h0; i < h1; ++i)
            for (int j = w0; j < w1; ++j) acc += x4[((n * H + i) * W + j) * C4 + c4];
        y4[e] = acc / (float)((h1 - h0) * (w1 - w0));
    }
}

// adjoint: dx[h][w] = sum over the output windows that contain (h, w) of dy[o] / |window(o)|  (a gather: no atomics); the
// windows containing row h are the outputs oh with start(oh) <= h < end(oh): a run around floor(h * Ho / H)
__global__ __launch_bounds__(NT) void adaptive_avgpool_bwd(long total4, int H, int W, int Ho, int Wo, int C4,
                                                           const float* __restrict__ dy, float* __restrict__ dx) {
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* o4 = reinterpret_cast<f32x4*>(dx);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        const int c4 = (int)(e % C4);
        long pix = e / C4;
        const int w = (int)(pix % W);
        long t = pix / W;
        const int h = (int)(t % H);
        const long n = t / H;
        int oh0 = (int)(((long)h * Ho) / H), ow0 = (int)(((long)w * Wo) / W);
        while (oh0 > 0 && ada_end(oh0 - 1, Ho, H) > h) --oh0;
        while (ow0 > 0 && ada_end(ow0 - 1, Wo, W) > w) --ow0;
        f32x4 acc = {0, 0, 0, 0};
        for (int oh = oh0; oh < Ho && ada_start(oh, Ho, H) <= h; ++oh) {
            const int nh = ada_end(oh, Ho, H) - ada_start(oh, Ho, H);
            for (int ow = ow0; ow < Wo && ada_start(ow, Wo, W) <= w; ++ow) {
                const int nw = ada_end(ow, Wo, W) - ada_start(ow, Wo, W);
                acc += d4[((n * Ho + oh) * Wo + ow) * C4 + c4] / (float)(nh * nw);
            }
        }
        o4[e] = acc;
    }
}

// max over the 3x3 window of stride 2, padding 1 (padding never wins: -inf); arg = winning tap (kh*3 + kw),
// first maximum in scan order like ATen's max_pool2d
__global__ __launch_bounds__(NT) void maxpool3s2_fwd(long total, int H, int W, int Ho, int Wo, int C,
                                                     const float* __restrict__ x, float* __restrict__ y,
                                                     unsigned char* __restrict__ arg) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int c = (int)(e % C);
        long pix = e / C;
        int wo = (int)(pix % Wo);
        long t = pix / Wo;
        int ho = (int)(t % Ho);
        long n = t / Ho;
        float best = -INFINITY;
        int bi = 0;
        for (int kh = 0; kh < 3; ++kh) {
            int ih = 2 * ho - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int iw = 2 * wo - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                float v = x[((n * H + ih) * W + iw) * C + c];
                if (v > best) { best = v; bi = kh * 3 + kw; }
            }
        }
        y[e] = best;
        arg[e] = (unsigned char)bi;
    }
}

// gather form of the max-pool gradient: each input pixel sums the dy of the (<= 4) windows it won
__global__ __launch_bounds__(NT) void maxpool3s2_bwd(long total, int H, int W, int Ho, int Wo, int C,
                                                     const float* __restrict__ dy,
                                                     const unsigned char* __restrict__ arg, float* __restrict__ dx) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int c = (int)(e % C);
        long pix = e / C;
        int w = (int)(pix % W);
        long t = pix / W;
        int h = (int)(t % H);
        long n = t / H;
        float acc = 0.f;
        for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {          // windows with 2*oh-1 <= h <= 2*oh+1
            if (oh >= Ho) continue;
            int kh = h - (2 * oh - 1);
            for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
                if (ow >= Wo) continue;
                int kw = w - (2 * ow - 1);
                long o = ((n * Ho + oh) * Wo + ow) * C + c;
                if (arg[o] == kh * 3 + kw) acc += dy[o];
            }
        }
        dx[e] = acc;
    }
}

// y = act(a + b), act: 0 identity, 2 ReLU
__global__ __launch_bounds__(NT) void add_act_fwd(long total4, const float* __restrict__ a, const float* __restrict__ b,
                                                  int act, float* __restrict__ y) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        f32x4 v = a4[e] + b4[e];
        if (act == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        y4[e] = v;
    }
}

// g = dy * [y > 0]   (gradient of both summands)
__global__ __launch_bounds__(NT) void relu_bwd(long total4, const float* __restrict__ dy, const float* __restrict__ y,
                                               float* __restrict__ g) {
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        f32x4 d = d4[e], v = y4[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = v[k] > 0.f ? d[k] : 0.f;
        g4[e] = d;
    }
}

// dir 0: wide[m][off .. off+Cn) = narrow[m][:]      (concat forward)
// dir 1: narrow[m][:] = wide[m][off .. off+Cn)       (concat backward = slice)
__global__ __launch_bounds__(NT) void copy_cols(long total4, int Cn4, int Cw4, int off4, int dir,
                                                float* __restrict__ narrow, float* __restrict__ wide) {
    f32x4* n4 = reinterpret_cast<f32x4*>(narrow);
    f32x4* w4 = reinterpret_cast<f32x4*>(wide);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long m = e / Cn4;
        int c = (int)(e - m * Cn4);
        if (dir == 0) w4[m * Cw4 + off4 + c] = n4[e];
        else n4[e] = w4[m * Cw4 + off4 + c];
    }
}

}  // namespace

extern "C" {

int ud_avgpool_fwd(const float* x, float* y, int N, int Ho, int Wo, int C, int k, ud_stream_t stream) {
    if (C % 4 || k < 1) return UD_EINVAL;
    long total4 = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avgpool_fwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, Ho, Wo, C / 4, k,
                       x, y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_avgpool_bwd(const float* dy, float* dx, int N, int Ho, int Wo, int C, int k, ud_stream_t stream) {
    if (C % 4 || k < 1) return UD_EINVAL;
    long total4 = (long)N * Ho * k * Wo * k * (C / 4);
    hipLaunchKernelGGL(avgpool_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, Ho, Wo, C / 4, k,
                       dy, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int Ho, int Wo, int C, ud_stream_t stream) {
    if (C % 4 || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || Ho > H || Wo > W || !x || !y) return UD_EINVAL;
    long total4 = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(adaptive_avgpool_fwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, H, W, Ho, Wo,
                       C / 4, x, y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int Ho, int Wo, int C, ud_stream_t stream) {
    if (C % 4 || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || Ho > H || Wo > W || !dy || !dx) return UD_EINVAL;
    long total4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(adaptive_avgpool_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, H, W, Ho, Wo,
                       C / 4, dy, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_maxpool3s2_fwd(const float* x, float* y, unsigned char* arg, int N, int H, int W, int C, ud_stream_t stream) {
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long total = (long)N * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool3s2_fwd, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, H, W, Ho, Wo, C,
                       x, y, arg);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_maxpool3s2_bwd(const float* dy, const unsigned char* arg, float* dx, int N, int H, int W, int C,
                      ud_stream_t stream) {
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long total = (long)N * H * W * C;
    hipLaunchKernelGGL(maxpool3s2_bwd, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, H, W, Ho, Wo, C,
                       dy, arg, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_add_act_fwd(const float* a, const float* b, int act, float* y, long total, ud_stream_t stream) {
    if (total % 4) return UD_EINVAL;
    hipLaunchKernelGGL(add_act_fwd, dim3(ew_blocks(total / 4)), dim3(NT), 0, (hipStream_t)stream, total / 4, a, b, act,
                       y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_relu_bwd(const float* dy, const float* y, float* g, long total, ud_stream_t stream) {
    if (total % 4) return UD_EINVAL;
    hipLaunchKernelGGL(relu_bwd, dim3(ew_blocks(total / 4)), dim3(NT), 0, (hipStream_t)stream, total / 4, dy, y, g);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_copy_cols(float* narrow, float* wide, long M, int Cn, int Cw, int off, int dir, ud_stream_t stream) {
    if (Cn % 4 || Cw % 4 || off % 4 || off + Cn > Cw) return UD_EINVAL;
    long total4 = M * (Cn / 4);
    hipLaunchKernelGGL(copy_cols, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, Cn / 4, Cw / 4,
                       off / 4, dir, narrow, wide);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
