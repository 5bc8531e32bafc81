// Column (per-channel) reductions and normalisation kernels on pixel-major [G groups][R rows][C] fp32.
//
// Serves nn.BatchNorm2d/1d in training mode (G = 1; model/efficientnet/model.py:67,77,91,186,222,
// model/unidefense.py:104, model/modules.py:83,112), nn.InstanceNorm2d (G = N; model/unidefense.py:54),
// their backward, the fused swish (model/efficientnet/utils.py:66-82), F.adaptive_avg_pool2d(x, 1) and
// x.mean([-2,-1]) (G = N; model/efficientnet/model.py:118, model/unidefense.py:226,232-236).
//
// All of these are HBM-bound: one coalesced 16-B-per-lane pass over the tensor per kernel.  A reduction
// is two launches: a partial pass (grid = chunks x channel-slabs x groups; every thread owns 4 adjacent
// channels and strides over rows, then the block folds its row-lanes through LDS) and a tiny finalize.
// Deterministic (no atomics).
#include "ud_common.h"

namespace {

constexpr int NT = 256;

struct RedGeom {
    int G, R, C4, P;      // groups, rows per group, float4 channels, chunks per group
    int rpi;              // rows per block iteration
    int rows_per_chunk;
};

__device__ __forceinline__ bool thread_coords(const RedGeom& q, int& ri, int& c4) {
    int t = threadIdx.x;
    if (q.C4 <= NT) {
        ri = t / q.C4;
        c4 = t % q.C4;
        return ri < q.rpi;
    }
    ri = 0;
    c4 = blockIdx.y * NT + t;
    return c4 < q.C4;
}

// fold the row-lanes of a block: vals[8] per thread -> thread (ri == 0) holds the block total
template <int NQ>
__device__ __forceinline__ void block_fold(const RedGeom& q, int ri, int c4, bool active, float (&v)[8]) {
    if (q.C4 > NT || q.rpi == 1) return;
    __shared__ float sm[NT * 8];
    if (active) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) sm[threadIdx.x * 8 + i] = v[i];
    }
    __syncthreads();
    if (active && ri == 0) {
        for (int r = 1; r < q.rpi; ++r) {
            int t = r * q.C4 + c4;
#pragma unroll
            for (int i = 0; i < NQ; ++i) v[i] += sm[t * 8 + i];
        }
    }
}

enum { RED_STATS = 0, RED_NORMBWD = 1, RED_SUM = 2, RED_DOT = 3 };

// part1/part2: [(g*P + p)][C]
template <int MODE>
__global__ __launch_bounds__(NT) void colreduce_partial(RedGeom q, const float* __restrict__ x,
                                                        const float* __restrict__ y2,      // dy (NORMBWD) / b (DOT)
                                                        const float* __restrict__ mean,    // [G][C]
                                                        const float* __restrict__ invstd,  // [G][C]
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, float* __restrict__ part1, float* __restrict__ part2) {
    int ri, c4;
    bool active = thread_coords(q, ri, c4);
    const int g = blockIdx.z, p = blockIdx.x;
    const int r_begin = p * q.rows_per_chunk;
    int r_end = r_begin + q.rows_per_chunk;
    if (r_end > q.R) r_end = q.R;
    const long gbase = (long)g * q.R * q.C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y2);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        f32x4 sh = {0, 0, 0, 0}, mu = sh, is = sh, ga = sh, be = sh;
        if (MODE == RED_STATS) sh = x4[gbase + c4];   // shift = first row of the group (conditioning)
        if (MODE == RED_NORMBWD) {
            mu = reinterpret_cast<const f32x4*>(mean)[(long)g * q.C4 + c4];
            is = reinterpret_cast<const f32x4*>(invstd)[(long)g * q.C4 + c4];
            ga = reinterpret_cast<const f32x4*>(gamma)[c4];
            be = reinterpret_cast<const f32x4*>(beta)[c4];
        }
        for (int r = r_begin + ri; r < r_end; r += q.rpi) {
            const long idx = gbase + (long)r * q.C4 + c4;
            f32x4 a = x4[idx];
            if (MODE == RED_STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = a[e] - sh[e];
                    v[e] += d;
                    v[4 + e] += d * d;
                }
            } else if (MODE == RED_NORMBWD) {
                f32x4 dy = y4[idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xh = (a[e] - mu[e]) * is[e];
                    float dz = dy[e];
                    if (act == 1) dz *= ud_swish_grad(ga[e] * xh + be[e]);
                    v[e] += dz;
                    v[4 + e] += dz * xh;
                }
            } else if (MODE == RED_SUM) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += a[e];
            } else {
                f32x4 b = y4[idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += a[e] * b[e];
            }
        }
    }
    constexpr int NQ = (MODE == RED_STATS || MODE == RED_NORMBWD) ? 8 : 4;
    block_fold<NQ>(q, ri, c4, active, v);
    if (active && ri == 0) {
        const long o = ((long)g * q.P + p) * q.C4 + c4;
        f32x4 o1 = {v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(part1)[o] = o1;
        if (NQ == 8) {
            f32x4 o2 = {v[4], v[5], v[6], v[7]};
            reinterpret_cast<f32x4*>(part2)[o] = o2;
        }
    }
}

// ---- finalize kernels: one thread per (g, c) -------------------------------------------------
__global__ void stats_finalize(int G, int R, int C, int P, const float* __restrict__ x, const float* __restrict__ part1,
                               const float* __restrict__ part2, float eps, float* __restrict__ mean,
                               float* __restrict__ invstd, float* __restrict__ var_out, float momentum,
                               float* __restrict__ running_mean, float* __restrict__ running_var) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    int g = i / C, c = i % C;
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) {
        s1 += part1[((long)g * P + p) * C + c];
        s2 += part2[((long)g * P + p) * C + c];
    }
    float sh = x[(long)g * R * C + c];
    float n = (float)R;
    float m1 = s1 / n;
    float var = s2 / n - m1 * m1;
    if (var < 0.f) var = 0.f;
    float mu = sh + m1;
    mean[i] = mu;
    invstd[i] = rsqrtf(var + eps);
    if (var_out) var_out[i] = var;
    if (running_mean && G == 1) {   // nn.BatchNorm: unbiased variance goes into the running estimate
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        float unb = (R > 1) ? var * n / (n - 1.f) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

// s1,s2: [G][C] sums over chunks; dgamma/dbeta: [C] sums over groups too (may be null)
__global__ void normbwd_finalize(int G, int C, int P, const float* __restrict__ part1, const float* __restrict__ part2,
                                 float* __restrict__ s1, float* __restrict__ s2, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float tg = 0.f, tb = 0.f;
    for (int g = 0; g < G; ++g) {
        float a = 0.f, b = 0.f;
        for (int p = 0; p < P; ++p) {
            a += part1[((long)g * P + p) * C + c];
            b += part2[((long)g * P + p) * C + c];
        }
        s1[(long)g * C + c] = a;
        s2[(long)g * C + c] = b;
        tb += a;
        tg += b;
    }
    if (dgamma) dgamma[c] = tg;
    if (dbeta) dbeta[c] = tb;
}

__global__ void partial_sum_finalize(int G, int C, int P, const float* __restrict__ part1, float scale,
                                     float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    int g = i / C, c = i % C;
    float a = 0.f;
    for (int p = 0; p < P; ++p) a += part1[((long)g * P + p) * C + c];
    out[i] = a * scale;
}

// ---- elementwise passes -----------------------------------------------------------------------
// y = act(gamma * (x - mean[g]) * invstd[g] + beta)
__global__ __launch_bounds__(NT) void norm_apply_fwd(long total4, int R, int C4, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, float* __restrict__ y) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long g = row / R;
        f32x4 a = x4[e];
        f32x4 mu = reinterpret_cast<const f32x4*>(mean)[g * C4 + c4];
        f32x4 is = reinterpret_cast<const f32x4*>(invstd)[g * C4 + c4];
        f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        f32x4 be = reinterpret_cast<const f32x4*>(beta)[c4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float z = ga[k] * ((a[k] - mu[k]) * is[k]) + be[k];
            o[k] = (act == 1) ? ud_swish(z) : z;
        }
        y4[e] = o;
    }
}

// dx = gamma*invstd*(dz - s1/R - xhat*s2/R),  dz = dy*act'(z)          (batch statistics)
// with s1 == nullptr: dx = gamma*invstd*dz                               (fixed statistics, eval mode)
__global__ __launch_bounds__(NT) void norm_apply_bwd(long total4, int R, int C4, const float* __restrict__ x,
                                                     const float* __restrict__ dy, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ s1,
                                                     const float* __restrict__ s2, int act, float* __restrict__ dx) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
    const float invR = 1.f / (float)R;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long g = row / R;
        f32x4 a = x4[e], d = dy4[e];
        f32x4 mu = reinterpret_cast<const f32x4*>(mean)[g * C4 + c4];
        f32x4 is = reinterpret_cast<const f32x4*>(invstd)[g * C4 + c4];
        f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        f32x4 be = reinterpret_cast<const f32x4*>(beta)[c4];
        f32x4 t1 = {0, 0, 0, 0}, t2 = t1;
        if (s1) {
            t1 = reinterpret_cast<const f32x4*>(s1)[g * C4 + c4];
            t2 = reinterpret_cast<const f32x4*>(s2)[g * C4 + c4];
        }
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xh = (a[k] - mu[k]) * is[k];
            float dz = d[k];
            if (act == 1) dz *= ud_swish_grad(ga[k] * xh + be[k]);
            o[k] = ga[k] * is[k] * (dz - t1[k] * invR - xh * t2[k] * invR);
        }
        dx4[e] = o;
    }
}

RedGeom make_geom(int G, int R, int C, int P) {
    RedGeom q;
    q.G = G; q.R = R; q.C4 = C / 4; q.P = P;
    q.rpi = (q.C4 <= NT) ? (NT / q.C4) : 1;
    q.rows_per_chunk = (R + P - 1) / P;
    return q;
}

dim3 red_grid(const RedGeom& q) { return dim3((unsigned)q.P, (unsigned)((q.C4 + NT - 1) / NT), (unsigned)q.G); }

int ew_blocks(long total4) {
    long b = (total4 + NT - 1) / NT;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int ud_reduce_chunks(int G, int R, int C) {
    if (C <= 0 || C % 4) return UD_EINVAL;
    int C4 = C / 4;
    int rpi = (C4 <= NT) ? NT / C4 : 1;
    int slabs = (C4 + NT - 1) / NT;
    long want = 2048 / ((long)G * slabs);        // aim at ~2048 blocks in flight
    if (want < 1) want = 1;
    long maxp = (R + (long)rpi * 4 - 1) / ((long)rpi * 4);   // at least 4 iterations per block
    if (maxp < 1) maxp = 1;
    long P = want < maxp ? want : maxp;
    if (P > 1024) P = 1024;
    return (int)P;
}

int ud_norm_stats(const float* x, int G, int R, int C, int P, float eps, float* part1, float* part2, float* mean,
                  float* invstd, float* var_out, float momentum, float* running_mean, float* running_var,
                  ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_STATS>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_finalize, dim3(ud_cdiv((long)G * C, 256)), dim3(256), 0, s, G, R, C, P, x, part1, part2,
                       eps, mean, invstd, var_out, momentum, running_mean, running_var);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_norm_apply_fwd(const float* x, int G, int R, int C, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, int act, float* y, ud_stream_t stream) {
    if (C % 4) return UD_EINVAL;
    long total4 = (long)G * R * (C / 4);
    hipLaunchKernelGGL(norm_apply_fwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, R, C / 4, x,
                       mean, invstd, gamma, beta, act, y);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_norm_bwd(const float* x, const float* dy, int G, int R, int C, int P, const float* mean, const float* invstd,
                const float* gamma, const float* beta, int act, float* part1, float* part2, float* s1, float* s2,
                float* dgamma, float* dbeta, float* dx, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_NORMBWD>, red_grid(q), dim3(NT), 0, s, q, x, dy, mean, invstd, gamma, beta,
                       act, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(normbwd_finalize, dim3(ud_cdiv(C, 256)), dim3(256), 0, s, G, C, P, part1, part2, s1, s2, dgamma,
                       dbeta);
    UD_LAUNCH_CHECK();
    if (dx) {
        long total4 = (long)G * R * (C / 4);
        hipLaunchKernelGGL(norm_apply_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, s, total4, R, C / 4, x, dy, mean,
                           invstd, gamma, beta, s1, s2, act, dx);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

// out[g][c] = scale * sum_r x[g][r][c]
int ud_group_colsum(const float* x, int G, int R, int C, int P, float scale, float* part1, float* out,
                    ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_SUM>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, part1, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, dim3(ud_cdiv((long)G * C, 256)), dim3(256), 0, s, G, C, P, part1, scale,
                       out);
    UD_LAUNCH_CHECK();
    return 0;
}

// out[g][c] = scale * sum_r a[g][r][c] * b[g][r][c]
int ud_group_coldot(const float* a, const float* b, int G, int R, int C, int P, float scale, float* part1, float* out,
                    ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_DOT>, red_grid(q), dim3(NT), 0, s, q, a, b, nullptr, nullptr, nullptr,
                       nullptr, 0, part1, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, dim3(ud_cdiv((long)G * C, 256)), dim3(256), 0, s, G, C, P, part1, scale,
                       out);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
