// Deferred-BatchNorm helpers shared by csrc/fused.hip and csrc/dwtile.hip: coefficients of a channel quad from the fp64
// sums (ud_bn_ref), act(bn(x)) on a loaded quad, the SF gate factor, the fold of fp64 partials into an accumulator.
#pragma once
#include "colgeom.h"

namespace {

constexpr int NT = UD_COL_NT;

// rows in flight per thread: a half quad is an 8-byte access, so twice the rows keep the same bytes in flight
template <typename T> constexpr int kRowUnroll = sizeof(T) == 2 ? 8 : 4;

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

struct Bn4 { f32x4 mu, is, ga, be; };

// Coefficients of channel quad c4 from the fp64 sums; the designated thread of a kernel (update == true for exactly
// one thread per channel quad per launch) also moves the running statistics (nn.BatchNorm2d training forward).
__device__ __forceinline__ Bn4 bn_load(const ud_bn_ref& b, int g, int C4, int c4, bool update) {
    Bn4 o;
    const long i0 = ((long)(b.G == 1 ? 0 : g) * C4 + c4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double m = b.sum[i0 + e] * b.inv_count;
        double v = b.sumsq[i0 + e] * b.inv_count - m * m;
        if (v < 0.0) v = 0.0;
        o.mu[e] = (float)m;
        o.is[e] = rsqrtf((float)(v + (double)b.eps));
        if (update && b.running_mean) {
            const int c = c4 * 4 + e;
            b.running_mean[c] = (1.f - b.momentum) * b.running_mean[c] + b.momentum * (float)m;
            b.running_var[c] = (1.f - b.momentum) * b.running_var[c] + b.momentum * (float)(v * b.unbias);
        }
    }
    o.ga = reinterpret_cast<const f32x4*>(b.gamma)[c4];
    o.be = reinterpret_cast<const f32x4*>(b.beta)[c4];
    return o;
}

__device__ __forceinline__ f32x4 bn_apply(const f32x4& a, const Bn4& b, int act) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ud_act_fast(b.ga[e] * ((a[e] - b.mu[e]) * b.is[e]) + b.be[e], act);
    return o;
}

__device__ __forceinline__ float gate_factor(const float* alpha, int mode) {
    if (mode == 0 || !alpha) return 1.f;
    const float a = ud_sigmoid(alpha[0]);          // accurate: a is 4.5e-5 at the initial sf_coef = -10
    return mode == 1 ? a : 1.f - a;
}

// acc[q][o] += sum_p part[q][(o / C) * P + p][o % C]   (8 outputs x 32 chunk-lanes per block, four partials in flight
// per lane: 512 partials per output are four rounds of loads)
__global__ __launch_bounds__(NT) void partials_to_acc(int nq, int G, int C, int P, const double* __restrict__ part,
                                                      double* __restrict__ a1, double* __restrict__ a2) {
    __shared__ double sm[2][NT];
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int idx = blockIdx.x * 8 + cl;
    const bool ok = idx < G * C;
    const long plane = (long)G * P * C;
    double a = 0.0, b = 0.0;
    if (ok) {
        const int g = idx / C, c = idx % C;
        const long base = (long)g * P * C + c, step = 32L * C;
        int p = pl;
        for (; p + 96 < P; p += 128) {
            const long o = base + (long)p * C;
            a += (part[o] + part[o + step]) + (part[o + 2 * step] + part[o + 3 * step]);
            if (nq == 2) b += (part[plane + o] + part[plane + o + step]) + (part[plane + o + 2 * step] + part[plane + o + 3 * step]);
        }
        for (; p < P; p += 32) {
            const long o = base + (long)p * C;
            a += part[o];
            if (nq == 2) b += part[plane + o];
        }
    }
    sm[0][threadIdx.x] = a;
    sm[1][threadIdx.x] = b;
    __syncthreads();
    if (ok && pl == 0) {
        for (int k = 1; k < 32; ++k) {
            a += sm[0][k * 8 + cl];
            b += sm[1][k * 8 + cl];
        }
        a1[idx] += a;
        if (nq == 2) a2[idx] += b;
    }
}

}  // namespace
