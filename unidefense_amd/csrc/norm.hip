// Column (per-channel) reductions and normalisation kernels on pixel-major [G groups][R rows][C] fp32.
//
// Serves nn.BatchNorm2d/1d in training mode (G = 1; model/efficientnet/model.py:67,77,91,186,222,
// model/unidefense.py:104, model/modules.py:83,112), nn.InstanceNorm2d (G = N; model/unidefense.py:54),
// their backward, the fused swish (model/efficientnet/utils.py:66-82), F.adaptive_avg_pool2d(x, 1) and
// x.mean([-2,-1]) (G = N; model/efficientnet/model.py:118, model/unidefense.py:226,232-236).
//
// All of these are HBM-bound: one coalesced 16-B-per-lane pass over the tensor per kernel.  A reduction
// is two launches: a partial pass (grid = chunks x channel-slabs x groups; every thread owns 4 adjacent
// channels and strides over rows, then the block folds its row-lanes through LDS) and a finalize that
// sums the chunk partials with 16 lanes per channel.  Accumulation is in fp64 (the kernels are bandwidth
// bound, so it is free) — batch statistics over millions of rows and the cancelling sums of the backward
// then carry no summation error of their own.  Deterministic (no atomics).
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

// fold the row-lanes of a block: v[NQ] per thread -> thread (ri == 0) holds the block total
template <int NQ>
__device__ __forceinline__ void block_fold(const RedGeom& q, int ri, int c4, bool active, double (&v)[8]) {
    if (q.C4 > NT || q.rpi == 1) return;
    __shared__ double sm[NT * NQ];
    if (active) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) sm[threadIdx.x * NQ + i] = v[i];
    }
    __syncthreads();
    if (active && ri == 0) {
        for (int r = 1; r < q.rpi; ++r) {
            int t = r * q.C4 + c4;
#pragma unroll
            for (int i = 0; i < NQ; ++i) v[i] += sm[t * NQ + i];
        }
    }
}

enum { RED_STATS = 0, RED_NORMBWD = 1, RED_SUM = 2, RED_DOT = 3 };

// part1/part2: double [(g*P + p)][C]
template <int MODE>
__global__ __launch_bounds__(NT) void colreduce_partial(RedGeom q, const float* __restrict__ x,
                                                        const float* __restrict__ y2,      // dy (NORMBWD) / b (DOT)
                                                        const float* __restrict__ mean,    // [G][C]
                                                        const float* __restrict__ invstd,  // [G][C]
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, double* __restrict__ part1, double* __restrict__ part2) {
    int ri, c4;
    bool active = thread_coords(q, ri, c4);
    const int g = blockIdx.z, p = blockIdx.x;
    const int r_begin = p * q.rows_per_chunk;
    int r_end = r_begin + q.rows_per_chunk;
    if (r_end > q.R) r_end = q.R;
    const long gbase = (long)g * q.R * q.C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y2);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        f32x4 mu = {0, 0, 0, 0}, is = mu, ga = mu, be = mu;
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
                    double d = (double)a[e];
                    v[e] += d;
                    v[4 + e] += d * d;
                }
            } else if (MODE == RED_NORMBWD) {
                f32x4 dy = y4[idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xh = (a[e] - mu[e]) * is[e];
                    float dz = dy[e];
                    if (act) dz *= ud_act_grad(ga[e] * xh + be[e], act);
                    v[e] += (double)dz;
                    v[4 + e] += (double)dz * (double)xh;
                }
            } else if (MODE == RED_SUM) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (double)a[e];
            } else {
                f32x4 b = y4[idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (double)a[e] * (double)b[e];
            }
        }
    }
    constexpr int NQ = (MODE == RED_STATS || MODE == RED_NORMBWD) ? 8 : 4;
    block_fold<NQ>(q, ri, c4, active, v);
    if (active && ri == 0) {
        const long o = (((long)g * q.P + p) * q.C4 + c4) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) part1[o + e] = v[e];
        if (NQ == 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) part2[o + e] = v[4 + e];
        }
    }
}

// ---- finalize: 16 channels x 16 chunk-lanes per block -----------------------------------------------
// returns (in lane pl == 0) the sums over the P chunks of part1/part2 for entry idx = (g, c)
__device__ __forceinline__ void chunk_sums(int G, int C, int P, const double* __restrict__ part1,
                                           const double* __restrict__ part2, int& idx, bool& lead, double& s1,
                                           double& s2) {
    __shared__ double sm1[NT], sm2[NT];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    idx = blockIdx.x * 16 + cl;
    const bool ok = idx < G * C;
    double a = 0.0, b = 0.0;
    if (ok) {
        const int g = idx / C, c = idx % C;
        for (int p = pl; p < P; p += 16) {
            const long o = ((long)g * P + p) * C + c;
            a += part1[o];
            if (part2) b += part2[o];
        }
    }
    sm1[threadIdx.x] = a;
    sm2[threadIdx.x] = b;
    __syncthreads();
    lead = ok && pl == 0;
    if (lead) {
        for (int k = 1; k < 16; ++k) {
            a += sm1[k * 16 + cl];
            b += sm2[k * 16 + cl];
        }
    }
    s1 = a;
    s2 = b;
}

__global__ __launch_bounds__(NT) void stats_finalize(int G, int R, int C, int P, const double* __restrict__ part1,
                                                     const double* __restrict__ part2, float eps,
                                                     float* __restrict__ mean, float* __restrict__ invstd,
                                                     float* __restrict__ var_out, float momentum,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var) {
    int idx; bool lead; double s1, s2;
    chunk_sums(G, C, P, part1, part2, idx, lead, s1, s2);
    if (!lead) return;
    const double n = (double)R;
    const double mu = s1 / n;
    double var = s2 / n - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[idx] = (float)mu;
    invstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
    if (var_out) var_out[idx] = (float)var;
    if (running_mean && G == 1) {   // nn.BatchNorm: the unbiased variance goes into the running estimate
        const int c = idx;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        const double unb = (R > 1) ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// s1,s2: float [G][C] sums over chunks
__global__ __launch_bounds__(NT) void normbwd_finalize(int G, int C, int P, const double* __restrict__ part1,
                                                       const double* __restrict__ part2, float* __restrict__ s1o,
                                                       float* __restrict__ s2o) {
    int idx; bool lead; double s1, s2;
    chunk_sums(G, C, P, part1, part2, idx, lead, s1, s2);
    if (!lead) return;
    s1o[idx] = (float)s1;
    s2o[idx] = (float)s2;
}

// dgamma[c] = sum_g s2[g][c], dbeta[c] = sum_g s1[g][c]
__global__ void group_sum(int G, int C, const float* __restrict__ s1, const float* __restrict__ s2,
                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int g = 0; g < G; ++g) {
        a += (double)s1[(long)g * C + c];
        b += (double)s2[(long)g * C + c];
    }
    if (dbeta) dbeta[c] = (float)a;
    if (dgamma) dgamma[c] = (float)b;
}

__global__ __launch_bounds__(NT) void partial_sum_finalize(int G, int C, int P, const double* __restrict__ part1,
                                                           float scale, float* __restrict__ out) {
    int idx; bool lead; double s1, s2;
    chunk_sums(G, C, P, part1, nullptr, idx, lead, s1, s2);
    if (lead) out[idx] = (float)(s1 * (double)scale);
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
            o[k] = ud_act(z, act);
        }
        y4[e] = o;
    }
}

// dx = gamma*invstd*(dz - s1*invR - xhat*s2*invR),  dz = dy*act'(z)
__global__ __launch_bounds__(NT) void norm_apply_bwd(long total4, int R, int C4, const float* __restrict__ x,
                                                     const float* __restrict__ dy, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ s1,
                                                     const float* __restrict__ s2, float invR, int act,
                                                     float* __restrict__ dx) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total4; e += (long)gridDim.x * NT) {
        long row = e / C4;
        int c4 = (int)(e - row * C4);
        long g = row / R;
        f32x4 a = x4[e], d = dy4[e];
        f32x4 mu = reinterpret_cast<const f32x4*>(mean)[g * C4 + c4];
        f32x4 is = reinterpret_cast<const f32x4*>(invstd)[g * C4 + c4];
        f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        f32x4 be = reinterpret_cast<const f32x4*>(beta)[c4];
        f32x4 t1 = reinterpret_cast<const f32x4*>(s1)[g * C4 + c4];
        f32x4 t2 = reinterpret_cast<const f32x4*>(s2)[g * C4 + c4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xh = (a[k] - mu[k]) * is[k];
            float dz = d[k];
            if (act) dz *= ud_act_grad(ga[k] * xh + be[k], act);
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

inline dim3 fin_grid(int G, int C) { return dim3((unsigned)ud_cdiv((long)G * C, 16)); }

}  // namespace

extern "C" {

int ud_reduce_chunks(int G, int R, int C) {
    if (C <= 0 || C % 4) return UD_EINVAL;
    int C4 = C / 4;
    int rpi = (C4 <= NT) ? NT / C4 : 1;
    int slabs = (C4 + NT - 1) / NT;
    long want = 1024 / ((long)G * slabs);        // aim at ~1024 blocks in flight
    if (want < 1) want = 1;
    long maxp = (R + (long)rpi * 4 - 1) / ((long)rpi * 4);   // at least 4 iterations per block
    if (maxp < 1) maxp = 1;
    long P = want < maxp ? want : maxp;
    if (P > 256) P = 256;
    return (int)P;
}

int ud_norm_stats(const float* x, int G, int R, int C, int P, float eps, double* part1, double* part2, float* mean,
                  float* invstd, float* var_out, float momentum, float* running_mean, float* running_var,
                  ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_STATS>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_finalize, fin_grid(G, C), dim3(NT), 0, s, G, R, C, P, part1, part2, eps, mean, invstd,
                       var_out, momentum, running_mean, running_var);
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
                const float* gamma, const float* beta, int act, double* part1, double* part2, float* s1, float* s2,
                float* dgamma, float* dbeta, float* dx, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_NORMBWD>, red_grid(q), dim3(NT), 0, s, q, x, dy, mean, invstd, gamma, beta,
                       act, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(normbwd_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, P, part1, part2, s1, s2);
    UD_LAUNCH_CHECK();
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(group_sum, dim3(ud_cdiv(C, 256)), dim3(256), 0, s, G, C, s1, s2, dgamma, dbeta);
        UD_LAUNCH_CHECK();
    }
    if (dx) {
        long total4 = (long)G * R * (C / 4);
        hipLaunchKernelGGL(norm_apply_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, s, total4, R, C / 4, x, dy, mean,
                           invstd, gamma, beta, s1, s2, 1.f / (float)R, act, dx);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

// The elementwise half of ud_norm_bwd alone, with caller-provided sums and 1/count: used when s1/s2 were
// summed over all ranks first (SyncBatchNorm backward; count = rows of ALL ranks).
int ud_norm_bwd_apply(const float* x, const float* dy, int G, int R, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, const float* s1, const float* s2, float inv_count,
                      int act, float* dx, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1) return UD_EINVAL;
    long total4 = (long)G * R * (C / 4);
    hipLaunchKernelGGL(norm_apply_bwd, dim3(ew_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, total4, R, C / 4, x,
                       dy, mean, invstd, gamma, beta, s1, s2, inv_count, act, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

// out[g][c] = scale * sum_r x[g][r][c]
int ud_group_colsum(const float* x, int G, int R, int C, int P, float scale, double* part1, float* out,
                    ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_SUM>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, part1, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, P, part1, scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

// out[g][c] = scale * sum_r a[g][r][c] * b[g][r][c]
int ud_group_coldot(const float* a, const float* b, int G, int R, int C, int P, float scale, double* part1, float* out,
                    ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || P < 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C, P);
    hipLaunchKernelGGL(colreduce_partial<RED_DOT>, red_grid(q), dim3(NT), 0, s, q, a, b, nullptr, nullptr, nullptr,
                       nullptr, 0, part1, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, P, part1, scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
