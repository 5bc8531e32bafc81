// Small / elementwise kernels of the UniDefense step: squeeze-excite FCs and gating, SFConv branch mixing,
// residual + drop-connect, dropout-by-mask, layout changes, bilinear resize (align_corners=True), the
// dynamic-filter mask, and the L1 reductions of the reconstruction losses.  All fp32; HBM- or
// latency-bound; one coalesced pass each.
#include "ud_common.h"

namespace {

constexpr int NT = 256;

inline int ew_blocks(long total, int cap = 8192) {
    long b = (total + NT - 1) / NT;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float act_in_f(float x, int act) { return act == 1 ? ud_swish(x) : x; }

// ------------------------------------------------------------------------------------------------
// Small fully-connected layer:  y[n][o] = sum_i act_in(x[n][i]) * W[o][i] + b[o]
// (SE reduce / expand 1x1 convs on [N,C,1,1], model/efficientnet/model.py:119-121; classifier
// nn.Linear, model/modules.py:27).  One wave per output element.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void fc_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                             const float* __restrict__ b, float* __restrict__ y, int N, int I, int O,
                                             int act_in) {
    const int wave = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N * O) return;
    const int n = wave / O, o = wave % O;
    float acc = 0.f;
#pragma unroll 4
    for (int i = lane; i < I; i += 64) acc += act_in_f(x[(long)n * I + i], act_in) * W[(long)o * I + i];
    acc = ud_wave_sum(acc);
    if (lane == 0) y[(long)n * O + o] = acc + (b ? b[o] : 0.f);
}

// Same result for a short input and a wide output (SE expand: I = C/24 <= 128, O = C up to 2688): one wave per output
// would spend 64 lanes on <= 128 products.  grid (O / 256, N): act_in(x[n][:]) staged in LDS, one thread per output
// walking its own weight row.
__global__ __launch_bounds__(NT) void fc_fwd_wide(const float* __restrict__ x, const float* __restrict__ W,
                                                  const float* __restrict__ b, float* __restrict__ y, int N, int I, int O,
                                                  int act_in) {
    __shared__ float sx[128];
    const int n = blockIdx.y;
    if ((int)threadIdx.x < I) sx[threadIdx.x] = act_in_f(x[(long)n * I + threadIdx.x], act_in);
    __syncthreads();
    const int o = blockIdx.x * NT + threadIdx.x;
    if (o >= O) return;
    const float* w = W + (long)o * I;
    float acc = 0.f;
    if ((I & 3) == 0) {
#pragma unroll 4
        for (int i = 0; i < I; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + i);
            acc += sx[i] * v[0] + sx[i + 1] * v[1] + sx[i + 2] * v[2] + sx[i + 3] * v[3];
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < I; ++i) acc += sx[i] * w[i];
    }
    y[(long)n * O + o] = acc + (b ? b[o] : 0.f);
}

// dx[n][i] = act_in'(x[n][i]) * sum_o dy[n][o] * W[o][i]
__global__ __launch_bounds__(NT) void fc_bwd_x(const float* __restrict__ dy, const float* __restrict__ W,
                                               const float* __restrict__ x, float* __restrict__ dx, int N, int I, int O,
                                               int act_in) {
    const long e = (long)blockIdx.x * NT + threadIdx.x;
    if (e >= (long)N * I) return;
    const int n = (int)(e / I), i = (int)(e % I);
    float acc = 0.f;
#pragma unroll 8
    for (int o = 0; o < O; ++o) acc += dy[(long)n * O + o] * W[(long)o * I + i];
    if (act_in == 1) acc *= ud_swish_grad(x[e]);
    dx[e] = acc;
}

// same result, one wave per (n, i) with the lanes striding over o: for wide O (SE expand: O = C up to 2688)
__global__ __launch_bounds__(NT) void fc_bwd_x_wave(const float* __restrict__ dy, const float* __restrict__ W,
                                                    const float* __restrict__ x, float* __restrict__ dx, int N, int I,
                                                    int O, int act_in) {
    const int wave = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N * I) return;
    const int n = wave / I, i = wave % I;
    float acc = 0.f;
#pragma unroll 4
    for (int o = lane; o < O; o += 64) acc += dy[(long)n * O + o] * W[(long)o * I + i];
    acc = ud_wave_sum(acc);
    if (lane == 0) {
        if (act_in == 1) acc *= ud_swish_grad(x[(long)n * I + i]);
        dx[(long)n * I + i] = acc;
    }
}

// dW[o][i] = sum_n dy[n][o] * act_in(x[n][i]);  db[o] = sum_n dy[n][o]
__global__ __launch_bounds__(NT) void fc_bwd_w(const float* __restrict__ dy, const float* __restrict__ x,
                                               float* __restrict__ dW, float* __restrict__ db, int N, int I, int O,
                                               int act_in) {
    const long e = (long)blockIdx.x * NT + threadIdx.x;
    if (e >= (long)O * I) return;
    const int o = (int)(e / I), i = (int)(e % I);
    float acc = 0.f, accb = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
        float g = dy[(long)n * O + o];
        acc += g * act_in_f(x[(long)n * I + i], act_in);
        accb += g;
    }
    dW[e] = acc;
    if (db && i == 0) db[o] = accb;
}

// ------------------------------------------------------------------------------------------------
// SE gating  y = x * sigmoid(s[n][c])          (model/efficientnet/model.py:122)
// backward   dx = dy * sigmoid(s[n][c]) + dpool[n][c] * inv_hw      (dpool = grad of the avg-pool branch)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void se_scale_fwd(long total4, int HW, int C4, const float* __restrict__ x,
                                                   const float* __restrict__ s, float* __restrict__ y) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long n = row / HW;
        f32x4 a = x4[e], g = s4[n * C4 + c4], o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k] * ud_sigmoid(g[k]);
        y4[e] = o;
    }
}

__global__ __launch_bounds__(NT) void se_scale_bwd(long total4, int HW, int C4, const float* __restrict__ dy,
                                                   const float* __restrict__ s, const float* __restrict__ dpool,
                                                   float inv_hw, float* __restrict__ dx) {
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(dpool);
    f32x4* o4 = reinterpret_cast<f32x4*>(dx);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long n = row / HW;
        f32x4 d = d4[e], g = s4[n * C4 + c4], p = p4[n * C4 + c4], o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = d[k] * ud_sigmoid(g[k]) + p[k] * inv_hw;
        o4[e] = o;
    }
}

// v[i] *= sigmoid'(s[i])
__global__ void sigmoid_grad_mul(long n, const float* __restrict__ s, float* __restrict__ v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float g = ud_sigmoid(s[i]);
        v[i] *= g * (1.f - g);
    }
}

// ------------------------------------------------------------------------------------------------
// SFConv mixing  y = (1 - a) * spat + a * P(freq),  a = sigmoid(alpha),  P = identity or 2x2 average pool
// (model/efficientnet/exp.py:61-65; adaptive_avg_pool2d from an even size to half of it == 2x2 mean).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ f32x4 pooled_freq(const In4<T>& f4, int pool, long n, int ho, int wo, int Ho, int Wo, int C4,
                                             int c4) {
    if (!pool) return f4[((n * Ho + ho) * Wo + wo) * C4 + c4];
    const int H = Ho * 2, W = Wo * 2;
    const long b = ((n * H + 2 * ho) * W + 2 * wo) * C4 + c4;
    f32x4 v = f4[b] + f4[b + C4] + f4[b + (long)W * C4] + f4[b + (long)W * C4 + C4];
    return v * 0.25f;
}

template <typename T>
__global__ __launch_bounds__(NT) void sfmix_fwd(long total4, int Ho, int Wo, int C4, int pool,
                                                const T* __restrict__ spat, const T* __restrict__ freq,
                                                const float* __restrict__ alpha, T* __restrict__ y) {
    const float a = ud_sigmoid(alpha[0]);
    const In4<T> s4{spat}, f4{freq};
    const Out4<T> y4{y};
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % C4);
        long pix = e / C4;
        int wo = (int)(pix % Wo);
        long t = pix / Wo;
        int ho = (int)(t % Ho);
        long n = t / Ho;
        f32x4 fp = pooled_freq(f4, pool, n, ho, wo, Ho, Wo, C4, c4);
        y4.st(e, s4[e] * (1.f - a) + fp * a);
    }
}

// dspat = (1-a) dy;  dfreq = a * U(dy) (/4 when pooled);  part[block] = sum dy * (P(freq) - spat)
template <typename T>
__global__ __launch_bounds__(NT) void sfmix_bwd(long total4, int Ho, int Wo, int C4, int pool,
                                                const T* __restrict__ spat, const T* __restrict__ freq,
                                                const float* __restrict__ alpha, const T* __restrict__ dy,
                                                T* __restrict__ dspat, T* __restrict__ dfreq,
                                                double* __restrict__ part) {
    const float a = ud_sigmoid(alpha[0]);
    const In4<T> s4{spat}, f4{freq}, d4{dy};
    const Out4<T> ds4{dspat}, df4{dfreq};
    double acc = 0.0;   // fp64: the sum cancels heavily (sf_coef gradient)
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        int c4 = (int)(e % C4);
        long pix = e / C4;
        int wo = (int)(pix % Wo);
        long t = pix / Wo;
        int ho = (int)(t % Ho);
        long n = t / Ho;
        f32x4 fp = pooled_freq(f4, pool, n, ho, wo, Ho, Wo, C4, c4);
        f32x4 d = d4[e], s = s4[e];
        ds4.st(e, d * (1.f - a));
        if (!pool) {
            df4.st(e, d * a);
        } else {
            const int H = Ho * 2, W = Wo * 2;
            const long b = ((n * H + 2 * ho) * W + 2 * wo) * C4 + c4;
            f32x4 g = d * (0.25f * a);
            df4.st(b, g); df4.st(b + C4, g); df4.st(b + (long)W * C4, g); df4.st(b + (long)W * C4 + C4, g);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += (double)d[k] * ((double)fp[k] - (double)s[k]);
    }
    __shared__ double sm[NT / 64];
    acc = ud_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < NT / 64; ++i) tot += sm[i];
        part[blockIdx.x] = tot;
    }
}

// out[0] (+)= sigmoid'(alpha) * sum(part[0..n))        (gradient of a sigmoid-gated scalar coefficient)
__global__ __launch_bounds__(NT) void gate_grad_finalize(int n, const double* __restrict__ part,
                                                         const float* __restrict__ alpha, float* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) acc += part[i];
    __shared__ double sm[NT / 64];
    acc = ud_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < NT / 64; ++i) tot += sm[i];
        double a = 1.0 / (1.0 + exp(-(double)alpha[0]));
        out[0] = (float)(tot * a * (1.0 - a));
    }
}

// ------------------------------------------------------------------------------------------------
// generic gated mix of two same-shape tensors: y = (1-a) p + a q  (fuse_coef, model/unidefense.py:153-154)
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// elementwise helpers
// ------------------------------------------------------------------------------------------------
// out = x * (keep[n] * inv_keep) + skip      (drop_connect + residual, model/efficientnet/model.py:130-134)
__global__ __launch_bounds__(NT) void residual_fwd(long total4, long per_sample4, const float* __restrict__ x,
                                                   const float* __restrict__ skip, const float* __restrict__ keep,
                                                   float inv_keep, float* __restrict__ out) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* k4 = reinterpret_cast<const f32x4*>(skip);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        float sc = keep ? keep[e / per_sample4] * inv_keep : 1.f;
        f32x4 v = x4[e] * sc;
        if (skip) v += k4[e];
        o4[e] = v;
    }
}

// out = a * alpha + b * beta   (b may be null)
template <typename T>
__global__ __launch_bounds__(NT) void axpby(long total, const T* __restrict__ a, float alpha,
                                            const T* __restrict__ b, float beta, T* __restrict__ out) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        float v = (float)a[e] * alpha;
        if (b) v += (float)b[e] * beta;
        out[e] = (T)v;
    }
}

// out[i] (+)= sum_s ws[s][i], s ascending: the fixed-order end of a deterministic split-K GEMM (ud_gemm out_mode 3)
__global__ __launch_bounds__(NT) void sum_slices(long total4, int slices, long stride4, const float* __restrict__ ws,
                                                 float* __restrict__ out, int accumulate) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        f32x4 v = accumulate ? o4[e] : f32x4{0, 0, 0, 0};
        for (int s = 0; s < slices; ++s) v += w4[(long)s * stride4 + e];
        o4[e] = v;
    }
}

// out = x * mask * scale        (dropout with an explicit keep-mask)
__global__ __launch_bounds__(NT) void mask_scale(long total, const float* __restrict__ x, const float* __restrict__ m,
                                                 float scale, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT)
        out[e] = x[e] * m[e] * scale;
}

// out[n][p][c] = g[n][c] * scale      (gradient of a mean over the HW rows of each sample)
__global__ __launch_bounds__(NT) void bcast_rows(long total4, int HW, int C4, const float* __restrict__ g, float scale,
                                                 float* __restrict__ out) {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long n = row / HW;
        o4[e] = g4[n * C4 + c4] * scale;
    }
}

// out = |a - b|  (b may be null)
__global__ __launch_bounds__(NT) void absdiff(long total, const float* __restrict__ a, const float* __restrict__ b,
                                              float* __restrict__ out) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT)
        out[e] = fabsf(a[e] - (b ? b[e] : 0.f));
}

// y = (1-a) p + a q, a = sigmoid(alpha)
__global__ __launch_bounds__(NT) void gate_mix_fwd(long total, const float* __restrict__ p, const float* __restrict__ q,
                                                   const float* __restrict__ alpha, float* __restrict__ y) {
    const float a = ud_sigmoid(alpha[0]);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT)
        y[e] = (1.f - a) * p[e] + a * q[e];
}

// dp = (1-a) dy, dq = a dy, part[block] = sum dy (q - p)
__global__ __launch_bounds__(NT) void gate_mix_bwd(long total, const float* __restrict__ p, const float* __restrict__ q,
                                                   const float* __restrict__ alpha, const float* __restrict__ dy,
                                                   float* __restrict__ dp, float* __restrict__ dq,
                                                   double* __restrict__ part) {
    const float a = ud_sigmoid(alpha[0]);
    double acc = 0.0;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        float d = dy[e];
        dp[e] = (1.f - a) * d;
        dq[e] = a * d;
        acc += (double)d * ((double)q[e] - (double)p[e]);
    }
    __shared__ double sm[NT / 64];
    acc = ud_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < NT / 64; ++i) tot += sm[i];
        part[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// layout: planes [N][C][HW]  <->  pixel-major [N][HW][C]   (C small, e.g. 3)
// mode 0: copy; mode 1 (to planes): out = tanh(in);  mode 2 (to pixels): out = in * (1 - aux^2)  (tanh grad,
// aux = saved tanh output in the planes layout)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void pix_to_planes(long total, int C, int HW, const float* __restrict__ in,
                                                    float* __restrict__ out, int mode) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        // e indexes the OUTPUT [n][c][p]
        int p = (int)(e % HW);
        long t = e / HW;
        int c = (int)(t % C);
        long n = t / C;
        float v = in[(n * HW + p) * C + c];
        out[e] = (mode == 1) ? tanhf(v) : v;
    }
}

__global__ __launch_bounds__(NT) void planes_to_pix(long total, int C, int HW, const float* __restrict__ in,
                                                    const float* __restrict__ aux, float* __restrict__ out, int mode) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        // e indexes the OUTPUT [n][p][c]
        int c = (int)(e % C);
        long t = e / C;
        int p = (int)(t % HW);
        long n = t / HW;
        long src = (n * C + c) * HW + p;
        float v = in[src];
        if (mode == 2) {
            float y = aux[src];
            v *= (1.f - y * y);
        }
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// bilinear resize of planes, align_corners=True  (model/unidefense.py:16,126-127,244)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_coord(int o, int in, int out, int& i0, int& i1, float& l1) {
    float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(NT) void bilinear_fwd(long total, int Hi, int Wi, int Ho, int Wo,
                                                   const float* __restrict__ x, float* __restrict__ y) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int wo = (int)(e % Wo);
        long t = e / Wo;
        int ho = (int)(t % Ho);
        long p = t / Ho;
        int h0, h1, w0, w1; float lh, lw;
        bil_coord(ho, Hi, Ho, h0, h1, lh);
        bil_coord(wo, Wi, Wo, w0, w1, lw);
        const float* b = x + p * Hi * Wi;
        float v = (1.f - lh) * ((1.f - lw) * b[h0 * Wi + w0] + lw * b[h0 * Wi + w1]) +
                  lh * ((1.f - lw) * b[h1 * Wi + w0] + lw * b[h1 * Wi + w1]);
        y[e] = v;
    }
}

// output positions o in [lo, hi] whose interpolation may touch input position i (a superset; the weight test below
// is exact because it re-derives (i0, i1, l) with bil_coord)
__device__ __forceinline__ void bil_sources(int i, int in, int out, int& lo, int& hi) {
    if (in <= 1 || out <= 1) { lo = 0; hi = out - 1; return; }
    const float inv = (float)(out - 1) / (float)(in - 1);
    lo = (int)floorf((float)(i - 1) * inv) - 1;
    hi = (int)ceilf((float)(i + 1) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

__device__ __forceinline__ float bil_weight(int o, int i, int in, int out) {
    int i0, i1; float l;
    bil_coord(o, in, out, i0, i1, l);
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

// gather form (deterministic, no atomics, no zero fill): one thread per INPUT pixel sums the <= ~5x5 output pixels
// that interpolate from it.  The atomic scatter form took 609 us for the 128 -> 256 upsample of the loss tail
// (25 M atomics); this one moves 6.3 M + 1.6 M floats.
__global__ __launch_bounds__(NT) void bilinear_bwd(long total, int Hi, int Wi, int Ho, int Wo,
                                                   const float* __restrict__ dy, float* __restrict__ dx) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int w = (int)(e % Wi);
        long t = e / Wi;
        int h = (int)(t % Hi);
        long p = t / Hi;
        int ho_lo, ho_hi, wo_lo, wo_hi;
        bil_sources(h, Hi, Ho, ho_lo, ho_hi);
        bil_sources(w, Wi, Wo, wo_lo, wo_hi);
        const float* g = dy + p * Ho * Wo;
        float acc = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            const float wh = bil_weight(ho, h, Hi, Ho);
            if (wh == 0.f) continue;
            float row = 0.f;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) row += bil_weight(wo, w, Wi, Wo) * g[(long)ho * Wo + wo];
            acc += wh * row;
        }
        dx[e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// per-sample L1:  part[n][blockIdx.x] = sum |a - b| over the block's share of sample n   (b may be null)
// and its gradient  da = g[n] * scale * sign(a - b)
// (model/unidefense.py:245,251-253: torch.abs(...).mean(dim=[-3,-2,-1]))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void l1_partial(long per_sample, const float* __restrict__ a,
                                                 const float* __restrict__ b, float* __restrict__ part) {
    const long n = blockIdx.y;
    const float* pa = a + n * per_sample;
    const float* pb = b ? b + n * per_sample : nullptr;
    float acc = 0.f;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < per_sample; e += (long)gridDim.x * NT)
        acc += fabsf(pa[e] - (pb ? pb[e] : 0.f));
    __shared__ float sm[NT / 64];
    acc = ud_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < NT / 64; ++i) tot += sm[i];
        part[n * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ void l1_finalize(int N, int P, const float* __restrict__ part, float scale, float* __restrict__ out) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int p = 0; p < P; ++p) a += part[(long)n * P + p];
    out[n] = a * scale;
}

// da (+)= g[n] * scale * sign(a - b)
__global__ __launch_bounds__(NT) void l1_bwd(long total, long per_sample, const float* __restrict__ a,
                                             const float* __restrict__ b, const float* __restrict__ g, float scale,
                                             int accumulate, float* __restrict__ da) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        float d = a[e] - (b ? b[e] : 0.f);
        float s = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        float v = g[e / per_sample] * scale * s;
        da[e] = accumulate ? da[e] + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// Dynamic-filter mask (model/modules.py:94-102, 123-131), one wave per pixel row m:
//   pre = [mean_c proj, max_c proj, diff[0..D)];  mask = sigmoid(w2 . pre);  out = mask * x
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void dynfilter_fwd(int M, int C, int D, int Cx, const float* __restrict__ proj,
                                                    const float* __restrict__ diff, const float* __restrict__ w2,
                                                    const float* __restrict__ x, float* __restrict__ pre,
                                                    int* __restrict__ argmax, float* __restrict__ mask,
                                                    float* __restrict__ out) {
    const int m = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (m >= M) return;
    const float* pr = proj + (long)m * C;
    float s = 0.f, mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        float v = pr[c];
        s += v;
        if (v > mx) { mx = v; am = c; }
    }
    s = ud_wave_sum(s);
    // wave arg-max, first occurrence on ties (torch.max semantics)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float omx = __shfl_xor(mx, o, 64);
        int oam = __shfl_xor(am, o, 64);
        if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
    }
    const float mean = s / (float)C;
    float logit = w2[0] * mean + w2[1] * mx;
    for (int j = 0; j < D; ++j) logit += w2[2 + j] * diff[(long)m * D + j];
    const float mk = ud_sigmoid(logit);
    if (lane == 0) {
        pre[(long)m * (2 + D)] = mean;
        pre[(long)m * (2 + D) + 1] = mx;
        for (int j = 0; j < D; ++j) pre[(long)m * (2 + D) + 2 + j] = diff[(long)m * D + j];
        argmax[m] = am;
        mask[m] = mk;
    }
    for (int c = lane; c < Cx; c += 64) out[(long)m * Cx + c] = mk * x[(long)m * Cx + c];
}

// dx = mask * dout;  dmask = dmask_ext + sum_c dout * x;  dlogit = dmask * mask (1 - mask)
// dproj[c] = dlogit * (w2[0] / C + w2[1] * [c == argmax])
__global__ __launch_bounds__(NT) void dynfilter_bwd(int M, int C, int Cx, const float* __restrict__ dout,
                                                    const float* __restrict__ dmask_ext, const float* __restrict__ x,
                                                    const float* __restrict__ mask, const int* __restrict__ argmax,
                                                    const float* __restrict__ w2, float* __restrict__ dx,
                                                    float* __restrict__ dlogit, float* __restrict__ dproj) {
    const int m = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (m >= M) return;
    const float mk = mask[m];
    float acc = 0.f;
    for (int c = lane; c < Cx; c += 64) {
        float d = dout[(long)m * Cx + c];
        acc += d * x[(long)m * Cx + c];
        dx[(long)m * Cx + c] = mk * d;
    }
    acc = ud_wave_sum(acc);
    const float dm = acc + (dmask_ext ? dmask_ext[m] : 0.f);
    const float dl = dm * mk * (1.f - mk);
    if (lane == 0) dlogit[m] = dl;
    const float gm = dl * w2[0] / (float)C, gx = dl * w2[1];
    const int am = argmax[m];
    for (int c = lane; c < C; c += 64) dproj[(long)m * C + c] = gm + ((c == am) ? gx : 0.f);
}

// All depthwise weights of the network to tap-major in ONE launch: table[l] = (src pointer, C, KK, dst offset);
// dst[off + tap*C + c] = src[c*KK + tap]   (the [C,1,k,k] parameter -> wt[k*k][C] of the dw kernels).  Replaces 32
// per-layer transpose copies per step.
struct DwWtEntry { const float* src; long C, KK, dst_off; };
__global__ __launch_bounds__(NT) void dw_wt_tapmajor(const DwWtEntry* __restrict__ tab, float* __restrict__ dst) {
    const DwWtEntry e = tab[blockIdx.y];
    const long n = e.C * e.KK;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const long tap = i / e.C, c = i - tap * e.C;
    dst[e.dst_off + i] = e.src[c * e.KK + tap];
}

// ---- all k x k conv weights of a step into their GEMM layouts in ONE launch ------------------------------------------------
// The implicit-GEMM convs (decoder, dynamic filters, stem; the ResNet trunks) read their weights as [rows][tap][reduced channel]
// matrices: W[A][B][KH][KW] -> mode 0: [a][kh][kw][b] (forward of a conv; data gradient of a transposed conv), mode 1:
// [b][KH-1-kh][KW-1-kw][a] (data gradient of a stride-1 conv: flipped and transposed), mode 2: [b][kh][kw][a] (forward of a
// transposed conv; data gradient of a strided conv).  Rounds 1-4 made each with torch's permute / flip + contiguous: 2-3
// launches per conv and pass, ~45 per UDEB4 step.  Device table of items; block -> item by bisection of the block prefixes.
__global__ __launch_bounds__(NT) void weight_layouts_kernel(const ud_layout_item* __restrict__ items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].block0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const ud_layout_item it = items[lo];
    const long e = (long)(blockIdx.x - it.block0) * NT + threadIdx.x;
    const long total = (long)it.A * it.B * it.KH * it.KW;
    if (e >= total) return;
    const int taps = it.KH * it.KW;
    // dst index e -> (row, tap, col); mode 0: row = a, col = b; modes 1 / 2: row = b, col = a
    const int inner = it.mode == 0 ? it.B : it.A;
    const int col = (int)(e % inner);
    const int tap = (int)((e / inner) % taps);
    const int row = (int)(e / ((long)inner * taps));
    const int a = it.mode == 0 ? row : col, b = it.mode == 0 ? col : row;
    const int st = it.mode == 1 ? taps - 1 - tap : tap;          // (KH-1-kh) * KW + (KW-1-kw) = taps - 1 - (kh * KW + kw)
    it.dst[e] = it.src[((long)a * it.B + b) * taps + st];
}

}  // namespace

extern "C" {

int ud_weight_layouts_multi(const ud_layout_item* items_dev, int n, int blocks_total, ud_stream_t stream) {
    if (!items_dev || n < 1 || blocks_total < n) return UD_EINVAL;
    hipLaunchKernelGGL(weight_layouts_kernel, dim3((unsigned)blocks_total), dim3(NT), 0, (hipStream_t)stream, items_dev, n);
    UD_LAUNCH_CHECK();
    return 0;
}


int ud_fc_fwd(const float* x, const float* W, const float* b, float* y, int N, int I, int O, int act_in,
              ud_stream_t stream) {
    if (I <= 128 && O >= 256) {
        hipLaunchKernelGGL(fc_fwd_wide, dim3(ud_cdiv(O, NT), N), dim3(NT), 0, (hipStream_t)stream, x, W, b, y, N, I, O,
                           act_in);
        UD_LAUNCH_CHECK();
        return 0;
    }
    long waves = (long)N * O;
    hipLaunchKernelGGL(fc_fwd, dim3(ud_cdiv(waves * 64, NT)), dim3(NT), 0, (hipStream_t)stream, x, W, b, y, N, I, O,
                       act_in);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_fc_bwd(const float* dy, const float* W, const float* x, float* dx, float* dW, float* db, int N, int I, int O,
              int act_in, ud_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        if (O >= 256)
            hipLaunchKernelGGL(fc_bwd_x_wave, dim3(ud_cdiv((long)N * I * 64, NT)), dim3(NT), 0, s, dy, W, x, dx, N, I,
                               O, act_in);
        else
            hipLaunchKernelGGL(fc_bwd_x, dim3(ud_cdiv((long)N * I, NT)), dim3(NT), 0, s, dy, W, x, dx, N, I, O, act_in);
        UD_LAUNCH_CHECK();
    }
    if (dW) {
        hipLaunchKernelGGL(fc_bwd_w, dim3(ud_cdiv((long)O * I, NT)), dim3(NT), 0, s, dy, x, dW, db, N, I, O, act_in);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

int ud_se_scale_fwd(const float* x, const float* s, float* y, int N, int HW, int C, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    long total4 = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(se_scale_fwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, HW, C / 4, x, s,
                       y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_scale_bwd(const float* dy, const float* s, const float* dpool, float* dx, int N, int HW, int C,
                    ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    long total4 = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(se_scale_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, HW, C / 4, dy,
                       s, dpool, 1.f / (float)HW, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_sigmoid_grad_mul(const float* s, float* v, long n, ud_stream_t stream) {
    hipLaunchKernelGGL(sigmoid_grad_mul, dim3(ud_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, s, v);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_sfmix_blocks(int N, int Ho, int Wo, int C) { return ew_blocks((long)N * Ho * Wo * (C / 4), 1024); }

int ud_sfmix_fwd(const void* spat, const void* freq, const float* alpha, void* y, int N, int Ho, int Wo, int C,
                 int pool, int f16, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    long total4 = (long)N * Ho * Wo * (C / 4);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(sfmix_fwd<T>, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream,
                                                total4, Ho, Wo, C / 4, pool, (const T*)spat, (const T*)freq, alpha,
                                                (T*)y));
    UD_LAUNCH_CHECK();
    return 0;
}

// part must hold ud_sfmix_blocks(...) floats; dalpha receives sigmoid'(alpha) * sum dy (P(freq) - spat)
int ud_sfmix_bwd(const void* spat, const void* freq, const float* alpha, const void* dy, void* dspat, void* dfreq,
                 double* part, float* dalpha, int N, int Ho, int Wo, int C, int pool, int f16, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    long total4 = (long)N * Ho * Wo * (C / 4);
    int nb = ew_blocks(total4, 1024);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(sfmix_bwd<T>, dim3(nb), dim3(NT), 0, s, total4, Ho, Wo, C / 4, pool,
                                                (const T*)spat, (const T*)freq, alpha, (const T*)dy, (T*)dspat,
                                                (T*)dfreq, part));
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gate_grad_finalize, dim3(1), dim3(NT), 0, s, nb, part, alpha, dalpha);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_gate_mix_blocks(long total) { return ew_blocks(total, 1024); }

int ud_gate_mix_fwd(const float* p, const float* q, const float* alpha, float* y, long total, ud_stream_t stream) {
    hipLaunchKernelGGL(gate_mix_fwd, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, p, q, alpha, y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_gate_mix_bwd(const float* p, const float* q, const float* alpha, const float* dy, float* dp, float* dq,
                    double* part, float* dalpha, long total, ud_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    int nb = ew_blocks(total, 1024);
    hipLaunchKernelGGL(gate_mix_bwd, dim3(nb), dim3(NT), 0, s, total, p, q, alpha, dy, dp, dq, part);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gate_grad_finalize, dim3(1), dim3(NT), 0, s, nb, part, alpha, dalpha);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_residual_fwd(const float* x, const float* skip, const float* keep, float inv_keep, float* out, long total,
                    long per_sample, ud_stream_t stream) {
    if (total % 4 || per_sample % 4) return UD_EINVAL;
    hipLaunchKernelGGL(residual_fwd, dim3(ew_blocks(total / 4)), dim3(NT), 0, (hipStream_t)stream, total / 4,
                       per_sample / 4, x, skip, keep, inv_keep, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_axpby(const void* a, float alpha, const void* b, float beta, void* out, long total, int f16, ud_stream_t stream) {
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(axpby<T>, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total,
                                                (const T*)a, alpha, (const T*)b, beta, (T*)out));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_sum_slices(const float* ws, float* out, int slices, long total, long slice_stride, int accumulate,
                  ud_stream_t stream) {
    if (!ws || !out || slices < 1 || total < 4 || total % 4 || slice_stride % 4 || slice_stride < total) return UD_EINVAL;
    hipLaunchKernelGGL(sum_slices, dim3(ew_blocks(total / 4)), dim3(NT), 0, (hipStream_t)stream, total / 4, slices,
                       slice_stride / 4, ws, out, accumulate);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_mask_scale(const float* x, const float* mask, float scale, float* out, long total, ud_stream_t stream) {
    hipLaunchKernelGGL(mask_scale, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, x, mask, scale,
                       out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_bcast_rows(const float* g, float scale, float* out, int N, int HW, int C, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    long total4 = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(bcast_rows, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, HW, C / 4, g,
                       scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_absdiff(const float* a, const float* b, float* out, long total, ud_stream_t stream) {
    hipLaunchKernelGGL(absdiff, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, a, b, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_pix_to_planes(const float* in, float* out, int N, int C, int HW, int mode, ud_stream_t stream) {
    long total = (long)N * C * HW;
    hipLaunchKernelGGL(pix_to_planes, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, C, HW, in, out,
                       mode);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_planes_to_pix(const float* in, const float* aux, float* out, int N, int C, int HW, int mode,
                     ud_stream_t stream) {
    long total = (long)N * C * HW;
    hipLaunchKernelGGL(planes_to_pix, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, C, HW, in, aux,
                       out, mode);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_bilinear_fwd(const float* x, float* y, int P, int Hi, int Wi, int Ho, int Wo, ud_stream_t stream) {
    long total = (long)P * Ho * Wo;
    hipLaunchKernelGGL(bilinear_fwd, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, Hi, Wi, Ho, Wo, x,
                       y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_bilinear_bwd(const float* dy, float* dx, int P, int Hi, int Wi, int Ho, int Wo, ud_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    long total = (long)P * Hi * Wi;
    hipLaunchKernelGGL(bilinear_bwd, dim3(ew_blocks(total)), dim3(NT), 0, s, total, Hi, Wi, Ho, Wo, dy, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_l1_chunks(long per_sample) { return ew_blocks(per_sample, 64); }

// out[n] = scale * sum |a - b| over sample n;  part: N * ud_l1_chunks(per_sample) floats
int ud_l1_fwd(const float* a, const float* b, float* part, float* out, int N, long per_sample, float scale,
              ud_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    int P = ew_blocks(per_sample, 64);
    hipLaunchKernelGGL(l1_partial, dim3(P, N), dim3(NT), 0, s, per_sample, a, b, part);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(l1_finalize, dim3(ud_cdiv(N, 64)), dim3(64), 0, s, N, P, part, scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_l1_bwd(const float* a, const float* b, const float* g, float scale, int accumulate, float* da, int N,
              long per_sample, ud_stream_t stream) {
    long total = (long)N * per_sample;
    hipLaunchKernelGGL(l1_bwd, dim3(ew_blocks(total)), dim3(NT), 0, (hipStream_t)stream, total, per_sample, a, b, g,
                       scale, accumulate, da);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_dynfilter_fwd(const float* proj, const float* diff, const float* w2, const float* x, float* pre, int* argmax,
                     float* mask, float* out, int M, int C, int D, int Cx, ud_stream_t stream) {
    hipLaunchKernelGGL(dynfilter_fwd, dim3(ud_cdiv((long)M * 64, NT)), dim3(NT), 0, (hipStream_t)stream, M, C, D, Cx,
                       proj, diff, w2, x, pre, argmax, mask, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_dynfilter_bwd(const float* dout, const float* dmask_ext, const float* x, const float* mask, const int* argmax,
                     const float* w2, float* dx, float* dlogit, float* dproj, int M, int C, int Cx,
                     ud_stream_t stream) {
    hipLaunchKernelGGL(dynfilter_bwd, dim3(ud_cdiv((long)M * 64, NT)), dim3(NT), 0, (hipStream_t)stream, M, C, Cx, dout,
                       dmask_ext, x, mask, argmax, w2, dx, dlogit, dproj);
    UD_LAUNCH_CHECK();
    return 0;
}

// table: layers x 4 int64 on the DEVICE (source pointer, C, K*K, destination offset in floats)
int ud_dw_weights_tapmajor(const void* table, int layers, long max_elems, float* dst, ud_stream_t stream) {
    if (layers < 1 || max_elems < 1 || !table || !dst) return UD_EINVAL;
    hipLaunchKernelGGL(dw_wt_tapmajor, dim3((unsigned)ud_cdiv(max_elems, NT), (unsigned)layers), dim3(NT), 0,
                       (hipStream_t)stream, reinterpret_cast<const DwWtEntry*>(table), dst);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
