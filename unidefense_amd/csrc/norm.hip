// Column (per-channel) reductions and normalisation kernels on pixel-major [G groups][R rows][C] fp32.
//
// Serves nn.BatchNorm2d/1d in training mode (G = 1; model/efficientnet/model.py:67,77,91,186,222,
// model/unidefense.py:104, model/modules.py:83,112), nn.InstanceNorm2d (G = N; model/unidefense.py:54),
// their backward, the fused swish (model/efficientnet/utils.py:66-82), F.adaptive_avg_pool2d(x, 1) and
// x.mean([-2,-1]) (G = N; model/efficientnet/model.py:118, model/unidefense.py:226,232-236).
//
// All of these are HBM-bound: one coalesced 16-B-per-lane pass over the tensor per kernel.  A reduction is two
// launches.  Partial pass: grid = row-chunks x channel-groups x groups; a workgroup owns <= 32 float4 columns
// (<= 512 contiguous bytes of a row) and one row-chunk, every thread strides over rows with four independent
// 16-byte loads in flight, the block folds its row-lanes through LDS and stores its totals as fp64 partials.
// Finalize: 16 lanes per channel sum the chunk partials and produce the op's outputs.  fp64 accumulation (free:
// the kernels are bandwidth bound), no atomics: deterministic.
// A single-launch variant (fp64 atomics + "last workgroup finalizes" ticket) was measured and dropped: on this
// multi-XCD part the device-scope fence every workgroup needs costs more than the second launch (3-6x slower).
#include "colgeom.h"

namespace {

constexpr int NT = UD_COL_NT;

enum { RED_STATS = 0, RED_NORMBWD = 1, RED_SUM = 2, RED_DOT = 3 };

template <int MODE>
__device__ __forceinline__ void accumulate(const f32x4& a, const f32x4& b, const f32x4& mu, const f32x4& is,
                                           const f32x4& ga, const f32x4& be, int act, double (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (MODE == RED_STATS) {
            double d = (double)a[e];
            v[e] += d;
            v[4 + e] += d * d;
        } else if (MODE == RED_NORMBWD) {
            float xh = (a[e] - mu[e]) * is[e];
            float dz = b[e];
            if (act) dz *= ud_act_grad(ga[e] * xh + be[e], act);
            v[e] += (double)dz;
            v[4 + e] += (double)dz * (double)xh;
        } else if (MODE == RED_SUM) {
            v[e] += (double)a[e];
        } else {
            v[e] += (double)a[e] * (double)b[e];
        }
    }
}

// part1/part2: double [(g*P + p)][C]
template <int MODE>
__global__ __launch_bounds__(NT) void colreduce_partial(RedGeom q, const float* __restrict__ x,
                                                        const float* __restrict__ y2,      // dy (NORMBWD) / b (DOT)
                                                        const float* __restrict__ mean,    // [G][C]
                                                        const float* __restrict__ invstd,  // [G][C]
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, double* __restrict__ part1, double* __restrict__ part2) {
    constexpr bool HAS_Y = (MODE == RED_NORMBWD || MODE == RED_DOT);
    constexpr int NQ = (MODE == RED_STATS || MODE == RED_NORMBWD) ? 8 : 4;
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
        const long step = (long)q.rpi * q.C4;
        int r = r_begin + ri;
        long idx = gbase + (long)r * q.C4 + c4;
        for (; r + 3 * q.rpi < r_end; r += 4 * q.rpi, idx += 4 * step) {     // four loads in flight per thread
            f32x4 a0 = x4[idx], a1 = x4[idx + step], a2 = x4[idx + 2 * step], a3 = x4[idx + 3 * step];
            f32x4 b0 = a0, b1 = a0, b2 = a0, b3 = a0;
            if (HAS_Y) { b0 = y4[idx]; b1 = y4[idx + step]; b2 = y4[idx + 2 * step]; b3 = y4[idx + 3 * step]; }
            accumulate<MODE>(a0, b0, mu, is, ga, be, act, v);
            accumulate<MODE>(a1, b1, mu, is, ga, be, act, v);
            accumulate<MODE>(a2, b2, mu, is, ga, be, act, v);
            accumulate<MODE>(a3, b3, mu, is, ga, be, act, v);
        }
        for (; r < r_end; r += q.rpi, idx += step) {
            f32x4 a = x4[idx];
            f32x4 b = a;
            if (HAS_Y) b = y4[idx];
            accumulate<MODE>(a, b, mu, is, ga, be, act, v);
        }
    }
    block_fold<NQ>(q, ri, active, v);
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
        const long base = (long)g * P * C + c, step = 16L * C;
        int p = pl;
        // four independent chunk partials in flight per lane (the kernel is pure load latency)
        for (; p + 48 < P; p += 64) {
            const long o = base + (long)p * C;
            double a0 = part1[o], a1 = part1[o + step], a2 = part1[o + 2 * step], a3 = part1[o + 3 * step];
            double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
            if (part2) { b0 = part2[o]; b1 = part2[o + step]; b2 = part2[o + 2 * step]; b3 = part2[o + 3 * step]; }
            a += (a0 + a1) + (a2 + a3);
            b += (b0 + b1) + (b2 + b3);
        }
        for (; p < P; p += 16) {
            const long o = base + (long)p * C;
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

// s1,s2: float [G][C] sums over chunks; for G == 1 also dgamma = s2, dbeta = s1 (saves the group_sum launch)
__global__ __launch_bounds__(NT) void normbwd_finalize(int G, int C, int P, const double* __restrict__ part1,
                                                       const double* __restrict__ part2, float* __restrict__ s1o,
                                                       float* __restrict__ s2o, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta) {
    int idx; bool lead; double s1, s2;
    chunk_sums(G, C, P, part1, part2, idx, lead, s1, s2);
    if (!lead) return;
    s1o[idx] = (float)s1;
    s2o[idx] = (float)s2;
    if (G == 1) {
        if (dbeta) dbeta[idx] = (float)s1;
        if (dgamma) dgamma[idx] = (float)s2;
    }
}

// dgamma[c] = sum_g s2[g][c], dbeta[c] = sum_g s1[g][c]   (G > 1: InstanceNorm)
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

// ---- SyncBatchNorm: combine the gathered per-rank (mean, biased var) of equally sized shards ------------------
// st: [ws][2][C];  mean = avg_r mean_r;  var = avg_r (var_r + (mean_r - mean)^2)
__global__ void syncbn_combine(const float* __restrict__ st, int world, int C, double n_total, float eps,
                               float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                               float* __restrict__ mean, float* __restrict__ invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = 0.0;
    for (int r = 0; r < world; ++r) m += (double)st[((long)r * 2) * C + c];
    m /= (double)world;
    double var = 0.0;
    for (int r = 0; r < world; ++r) {
        double d = (double)st[((long)r * 2) * C + c] - m;
        var += (double)st[((long)r * 2 + 1) * C + c] + d * d;
    }
    var /= (double)world;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        const double unb = (n_total > 1.0) ? var * n_total / (n_total - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
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

// ---------------------------------------------------------------------------------------------------------
// One-launch normalisation (round 6): statistics + apply (forward), sums + apply (backward) in ONE kernel.
//
// The three-launch forms above cost their launches, not their bytes, on the tensors of the reconstruction decoder
// (InstanceNorm on 0.3 ... 42 MB: 22-36 us forward, 45-58 us backward) and of the ResNet variants' BatchNorms.  Here the
// P workgroups that share a (group g, column group) form a CLUSTER: each reduces its row chunk, PUBLISHES its fp64 partials
// (returning atomic exchanges at agent scope: the returned value is the acknowledgement that the word is visible
// device-wide — no fence, no L2 write-back; round 1's "last workgroup finalizes" paid 3-6x a launch for its device-scope
// fences), counts itself in on the cluster's counter and waits for the other P - 1; then every workgroup folds the P partials
// of its columns IN THE SAME ORDER (bit-identical statistics in all of them, deterministic), and walks its row chunk a second
// time — out of L2 — to apply.  Progress: workgroups are dispatched in order of their linear index and a cluster's indices
// are consecutive (p = blockIdx.x), so the lowest-indexed incomplete cluster is always resident as a whole; the wait is
// bounded by wall-clock time all the same (2 s: results become NaN, never a hung queue).  counters: P-cluster counters,
// zero before the launch (kernels.zeros: part of a captured step's fills).
// ---------------------------------------------------------------------------------------------------------
// all NQ exchanges are issued back to back, THEN their returned words are waited for (one round trip, not NQ: the first version
// waited per word and cost 20-25 us per kernel)
template <int NQ>
__device__ __forceinline__ void publish64(double* p, const double (&v)[8]) {
    unsigned long long old[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        old[i] = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p + i), (unsigned long long)__double_as_longlong(v[i]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < NQ; ++i) asm volatile("" ::"v"(old[i]));          // the returned word: the exchange has been performed
}
__device__ __forceinline__ double peek64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// v[0..NQ): this thread's sums over its rows -> the cluster's totals for its channel quad, in EVERY thread of the column.
// slots: [(g * CG + cg) * P + p][CW][8] doubles.  Returns false after a timed-out wait (a peer never arrived).
template <int NQ>
__device__ __forceinline__ bool cluster_allsum(const RedGeom& q, int ri, bool active, double (&v)[8],
                                               double* __restrict__ slots, unsigned* __restrict__ counters) {
    __shared__ double tot[UD_COL_NT / 4 * 8];          // [column lane][8] (CW <= 64)
    __shared__ int ok_flag;
    const int cl = threadIdx.x % q.CW;
    const int CG = gridDim.y;
    const long cluster = (long)blockIdx.z * CG + blockIdx.y;
    block_fold<NQ>(q, ri, active, v);
    bool ok = true;
    if (q.P > 1) {
        double* mine = slots + ((cluster * q.P + blockIdx.x) * q.CW + cl) * 8;
        if (active && ri == 0) publish64<NQ>(mine, v);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* ctr = counters + cluster;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            int good = 1;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)q.P) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 200000000LL) { good = 0; break; }          // 2 s of the 100 MHz clock
            }
            ok_flag = good;
        }
        __syncthreads();
        ok = ok_flag != 0;
        // fold the P partials of this column: lane ri takes p = ri, ri + rpi, ... (ascending), the lanes fold through LDS in
        // ascending ri — the same order in every workgroup of the cluster
        double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active) {
            const double* base = slots + (cluster * q.P * q.CW + cl) * 8;
            for (int p = ri; p < q.P; p += q.rpi) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) w[i] += peek64(base + (long)p * q.CW * 8 + i);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[i];
        __syncthreads();                                   // (block_fold's LDS is reused)
        block_fold<NQ>(q, ri, active, v);
    }
    __syncthreads();
    if (active && ri == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) tot[cl * 8 + i] = v[i];
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) v[i] = tot[cl * 8 + i];
    }
    return ok;
}

__global__ __launch_bounds__(NT) void norm_fwd_fused(RedGeom q, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, float eps, double* __restrict__ slots,
                                                     unsigned* __restrict__ counters, float* __restrict__ mean_o,
                                                     float* __restrict__ invstd_o, float momentum,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var,
                                                     float* __restrict__ y) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const int g = blockIdx.z, p = blockIdx.x;
    const int r_begin = p * q.rows_per_chunk;
    int r_end = r_begin + q.rows_per_chunk;
    if (r_end > q.R) r_end = q.R;
    const long gbase = (long)g * q.R * q.C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    const long step = (long)q.rpi * q.C4;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, is = mu, ga = mu, be = mu;
    if (active) {
        int r = r_begin + ri;
        long idx = gbase + (long)r * q.C4 + c4;
        for (; r + 3 * q.rpi < r_end; r += 4 * q.rpi, idx += 4 * step) {
            f32x4 a0 = x4[idx], a1 = x4[idx + step], a2 = x4[idx + 2 * step], a3 = x4[idx + 3 * step];
            accumulate<RED_STATS>(a0, a0, mu, is, ga, be, 0, v);
            accumulate<RED_STATS>(a1, a1, mu, is, ga, be, 0, v);
            accumulate<RED_STATS>(a2, a2, mu, is, ga, be, 0, v);
            accumulate<RED_STATS>(a3, a3, mu, is, ga, be, 0, v);
        }
        for (; r < r_end; r += q.rpi, idx += step) {
            f32x4 a = x4[idx];
            accumulate<RED_STATS>(a, a, mu, is, ga, be, 0, v);
        }
    }
    const bool ok = cluster_allsum<8>(q, ri, active, v, slots, counters);
    if (!active) return;
    const double n = (double)q.R;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double m = v[e] / n;
        double var = v[4 + e] / n - m * m;
        if (var < 0.0) var = 0.0;
        mu[e] = (float)m;
        is[e] = ok ? (float)(1.0 / sqrt(var + (double)eps)) : __builtin_nanf("");
        if (p == 0 && ri == 0) {
            const int c = c4 * 4 + e;
            mean_o[(long)g * q.C4 * 4 + c] = mu[e];
            invstd_o[(long)g * q.C4 * 4 + c] = is[e];
            if (running_mean && q.G == 1) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                const double unb = (q.R > 1) ? var * n / (n - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
    ga = reinterpret_cast<const f32x4*>(gamma)[c4];
    be = reinterpret_cast<const f32x4*>(beta)[c4];
    int r = r_begin + ri;
    long idx = gbase + (long)r * q.C4 + c4;
#pragma unroll 4
    for (; r < r_end; r += q.rpi, idx += step) {
        const f32x4 a = x4[idx];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ud_act(ga[k] * ((a[k] - mu[k]) * is[k]) + be[k], act);
        y4[idx] = o;
    }
}

// s1o / s2o: [G][C] sums (G > 1: ud_norm_bwd's group_sum turns them into dgamma / dbeta); G == 1: dgamma / dbeta written here
__global__ __launch_bounds__(NT) void norm_bwd_fused(RedGeom q, const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, double* __restrict__ slots, unsigned* __restrict__ counters,
                                                     float* __restrict__ s1o, float* __restrict__ s2o,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dx) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const int g = blockIdx.z, p = blockIdx.x;
    const int r_begin = p * q.rows_per_chunk;
    int r_end = r_begin + q.rows_per_chunk;
    if (r_end > q.R) r_end = q.R;
    const long gbase = (long)g * q.R * q.C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* o4 = reinterpret_cast<f32x4*>(dx);
    const long step = (long)q.rpi * q.C4;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, is = mu, ga = mu, be = mu;
    if (active) {
        mu = reinterpret_cast<const f32x4*>(mean)[(long)g * q.C4 + c4];
        is = reinterpret_cast<const f32x4*>(invstd)[(long)g * q.C4 + c4];
        ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        be = reinterpret_cast<const f32x4*>(beta)[c4];
        int r = r_begin + ri;
        long idx = gbase + (long)r * q.C4 + c4;
        for (; r + 3 * q.rpi < r_end; r += 4 * q.rpi, idx += 4 * step) {
            f32x4 a0 = x4[idx], a1 = x4[idx + step], a2 = x4[idx + 2 * step], a3 = x4[idx + 3 * step];
            f32x4 b0 = d4[idx], b1 = d4[idx + step], b2 = d4[idx + 2 * step], b3 = d4[idx + 3 * step];
            accumulate<RED_NORMBWD>(a0, b0, mu, is, ga, be, act, v);
            accumulate<RED_NORMBWD>(a1, b1, mu, is, ga, be, act, v);
            accumulate<RED_NORMBWD>(a2, b2, mu, is, ga, be, act, v);
            accumulate<RED_NORMBWD>(a3, b3, mu, is, ga, be, act, v);
        }
        for (; r < r_end; r += q.rpi, idx += step) accumulate<RED_NORMBWD>(x4[idx], d4[idx], mu, is, ga, be, act, v);
    }
    const bool ok = cluster_allsum<8>(q, ri, active, v, slots, counters);
    if (!active) return;
    f32x4 t1, t2;
    const float invR = 1.f / (float)q.R;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // (the three-launch form rounds the sums to fp32 in its finalize: the same values here)
        const float s1 = ok ? (float)v[e] : __builtin_nanf(""), s2 = (float)v[4 + e];
        t1[e] = s1 * invR;
        t2[e] = s2 * invR;
        if (p == 0 && ri == 0) {
            const long o = (long)g * q.C4 * 4 + c4 * 4 + e;
            if (s1o) s1o[o] = s1;
            if (s2o) s2o[o] = s2;
            if (q.G == 1) {
                if (dbeta) dbeta[o] = s1;
                if (dgamma) dgamma[o] = s2;
            }
        }
    }
    if (!dx) return;
    int r = r_begin + ri;
    long idx = gbase + (long)r * q.C4 + c4;
#pragma unroll 4
    for (; r < r_end; r += q.rpi, idx += step) {
        const f32x4 a = x4[idx], d = d4[idx];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (a[k] - mu[k]) * is[k];
            float dz = d[k];
            if (act) dz *= ud_act_grad(ga[k] * xh + be[k], act);
            o[k] = ga[k] * is[k] * (dz - t1[k] - xh * t2[k]);
        }
        o4[idx] = o;
    }
}

// geometry of the one-launch forms: at most 64 workgroups per cluster (the fold reads P partials per column), ~768 workgroups
// in all (they must be resident together: 3 per CU)
inline RedGeom fused_norm_geom(int G, int R, int C) { return make_geom_ex(G, R, C, 768, 64, 4); }

int ew_blocks(long total4) {
    long b = (total4 + NT - 1) / NT;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

inline dim3 fin_grid(int G, int C) { return dim3((unsigned)ud_cdiv((long)G * C, 16)); }

}  // namespace

extern "C" {

// scratch doubles a reduction over [G][R][C] needs (two partial arrays of G*P*C); contents need no initialisation
int ud_reduce_ws_doubles(int G, int R, int C) {
    if (G < 1 || R < 1 || C < 4 || C % 4) return UD_EINVAL;
    RedGeom q = make_geom(G, R, C);
    const long n = 2L * G * q.P * C;
    return n > 0x7fffffffL ? UD_EINVAL : (int)n;
}

int ud_norm_stats(const float* x, int G, int R, int C, float eps, double* ws, float* mean, float* invstd,
                  float* var_out, float momentum, float* running_mean, float* running_var, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C);
    double* part1 = ws;
    double* part2 = ws + (long)G * q.P * C;
    hipLaunchKernelGGL(colreduce_partial<RED_STATS>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_finalize, fin_grid(G, C), dim3(NT), 0, s, G, R, C, q.P, part1, part2, eps, mean, invstd,
                       var_out, momentum, running_mean, running_var);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_syncbn_combine(const float* gathered, int world, int C, long rows_per_rank, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, ud_stream_t stream) {
    if (world < 1 || C < 1 || rows_per_rank < 1) return UD_EINVAL;
    hipLaunchKernelGGL(syncbn_combine, dim3(ud_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gathered, world, C,
                       (double)rows_per_rank * world, eps, momentum, running_mean, running_var, mean, invstd);
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

int ud_norm_bwd(const float* x, const float* dy, int G, int R, int C, const float* mean, const float* invstd,
                const float* gamma, const float* beta, int act, double* ws, float* s1, float* s2, float* dgamma,
                float* dbeta, float* dx, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || !ws || !s1 || !s2) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C);
    double* part1 = ws;
    double* part2 = ws + (long)G * q.P * C;
    hipLaunchKernelGGL(colreduce_partial<RED_NORMBWD>, red_grid(q), dim3(NT), 0, s, q, x, dy, mean, invstd, gamma, beta,
                       act, part1, part2);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(normbwd_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, q.P, part1, part2, s1, s2, dgamma, dbeta);
    UD_LAUNCH_CHECK();
    if (G > 1 && (dgamma || dbeta)) {
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

// One-launch forms (see norm_fwd_fused).  slots: ud_norm_fused_ws_doubles(G, R, C) doubles of scratch (contents arbitrary);
// counters: ud_norm_fused_counters(G, R, C) ZERO 32-bit words.  Results equal ud_norm_stats + ud_norm_apply_fwd (ud_norm_bwd)
// up to the summation order of the fp64 partials.
long ud_norm_fused_ws_doubles(int G, int R, int C) {
    if (C % 4 || G < 1 || R < 1 || C < 4) return UD_EINVAL;
    RedGeom q = fused_norm_geom(G, R, C);
    const dim3 gr = red_grid(q);
    return (long)gr.z * gr.y * q.P * q.CW * 8;
}
int ud_norm_fused_counters(int G, int R, int C) {
    if (C % 4 || G < 1 || R < 1 || C < 4) return UD_EINVAL;
    RedGeom q = fused_norm_geom(G, R, C);
    const dim3 gr = red_grid(q);
    return (int)(gr.z * gr.y);
}
int ud_norm_fwd_fused(const float* x, int G, int R, int C, const float* gamma, const float* beta, int act, float eps,
                      double* slots, uint32_t* counters, float* mean, float* invstd, float momentum, float* running_mean,
                      float* running_var, float* y, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || C < 4 || !x || !gamma || !beta || !slots || !counters || !mean || !invstd || !y)
        return UD_EINVAL;
    RedGeom q = fused_norm_geom(G, R, C);
    hipLaunchKernelGGL(norm_fwd_fused, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q, x, gamma, beta, act, eps, slots,
                       counters, mean, invstd, momentum, running_mean, running_var, y);
    UD_LAUNCH_CHECK();
    return 0;
}
int ud_norm_bwd_fused(const float* x, const float* dy, int G, int R, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, double* slots, uint32_t* counters, float* s1,
                      float* s2, float* dgamma, float* dbeta, float* dx, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || C < 4 || !x || !dy || !mean || !invstd || !gamma || !beta || !slots || !counters ||
        (G > 1 && (!s1 || !s2)))
        return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = fused_norm_geom(G, R, C);
    hipLaunchKernelGGL(norm_bwd_fused, red_grid(q), dim3(NT), 0, s, q, x, dy, mean, invstd, gamma, beta, act, slots, counters,
                       s1, s2, dgamma, dbeta, dx);
    UD_LAUNCH_CHECK();
    if (G > 1 && (dgamma || dbeta)) {
        hipLaunchKernelGGL(group_sum, dim3(ud_cdiv(C, 256)), dim3(256), 0, s, G, C, s1, s2, dgamma, dbeta);
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
int ud_group_colsum(const float* x, int G, int R, int C, float scale, double* ws, float* out, ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C);
    hipLaunchKernelGGL(colreduce_partial<RED_SUM>, red_grid(q), dim3(NT), 0, s, q, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, 0, ws, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, q.P, ws, scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

// out[g][c] = scale * sum_r a[g][r][c] * b[g][r][c]
int ud_group_coldot(const float* a, const float* b, int G, int R, int C, float scale, double* ws, float* out,
                    ud_stream_t stream) {
    if (C % 4 || G < 1 || R < 1 || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedGeom q = make_geom(G, R, C);
    hipLaunchKernelGGL(colreduce_partial<RED_DOT>, red_grid(q), dim3(NT), 0, s, q, a, b, nullptr, nullptr, nullptr,
                       nullptr, 0, ws, nullptr);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_finalize, fin_grid(G, C), dim3(NT), 0, s, G, C, q.P, ws, scale, out);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
