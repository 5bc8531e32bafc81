// Deferred-normalisation kernels of the fused MBConv path (model/efficientnet/model.py:94-135 of the reference).
//
// A training-mode BatchNorm costs the unfused path three passes over the expanded activation (statistics, apply +
// swish, and the consumer's own read) plus two tiny finalize launches, forward and backward each.  Here the
// statistics are fp64 sums accumulated with atomic adds into a zeroed accumulator by whoever produces / first reads
// the tensor, and every consumer applies act(gamma (x - mean) invstd + beta) while loading x (ud_bn_ref): no finalize
// launches, no normalised copy.  All kernels use the column-owning decomposition of colgeom.h: a thread keeps one
// channel quad for its whole life, so the per-channel coefficients are derived once per thread from the sums.
//
// HBM-bound; 16-byte accesses, one contiguous run of channels per wave access.  fp64 atomics execute at the memory
// side (global_atomic_add_f64): one wave-instruction of 64 lanes per quantity and workgroup.
#include "bnref.h"

namespace {

// rows [r_begin, r_end) of this workgroup's chunk, float4 index of (g, r, c4) = gbase + r * C4 + c4
struct Rows {
    int r, r_end;
    long idx, step;
};
__device__ __forceinline__ Rows rows_of(const RedGeom& q, int ri, int c4) {
    Rows w;
    const int g = blockIdx.z, p = blockIdx.x;
    const int r_begin = p * q.rows_per_chunk;
    w.r_end = r_begin + q.rows_per_chunk;
    if (w.r_end > q.R) w.r_end = q.R;
    w.step = (long)q.rpi * q.C4;
    w.r = r_begin + ri;
    w.idx = (long)g * q.R * q.C4 + (long)w.r * q.C4 + c4;
    return w;
}

__device__ __forceinline__ bool is_updater(int ri) { return blockIdx.x == 0 && blockIdx.z == 0 && ri == 0; }

// Epilogue of every reducing kernel: fold the block's row-lanes, then either add the totals atomically into the
// accumulators (few contributors per address: ~12 ns per fp64 add and address, serialised) or store them as fp64
// partials part[q][(g * P + p)][C] for a finalize launch (partials_to_acc) when hundreds of workgroups would queue on
// the same addresses.  per_group: accumulators are [G][C] (else one [C] set for all groups).
template <int NQ>
__device__ __forceinline__ void red_out(const RedGeom& q, int ri, bool active, int c4, double (&v)[8], double* a1,
                                        double* a2, double* part, bool per_group) {
    block_fold<NQ>(q, ri, active, v);
    if (!(active && ri == 0)) return;
    if (part) {
        const long o = (((long)blockIdx.z * q.P + blockIdx.x) * q.C4 + c4) * 4;
        const long plane = (long)q.G * q.P * q.C4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[o + e] = v[e];
            if (NQ == 8) part[plane + o + e] = v[4 + e];
        }
    } else {
        const long o = ((long)(per_group ? blockIdx.z : 0) * q.C4 + c4) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomic_add_f64(a1 + o + e, v[e]);
            if (NQ == 8) atomic_add_f64(a2 + o + e, v[4 + e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// statistics and lazy reductions
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void colstats_kernel(RedGeom q, const T* __restrict__ x, double* __restrict__ sum,
                                                      double* __restrict__ sumsq, double* __restrict__ part) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> x4{x};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        Rows w = rows_of(q, ri, c4);
        for (; w.r + 3 * q.rpi < w.r_end; w.r += 4 * q.rpi, w.idx += 4 * w.step) {
            f32x4 a0 = x4[w.idx], a1 = x4[w.idx + w.step], a2 = x4[w.idx + 2 * w.step], a3 = x4[w.idx + 3 * w.step];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d0 = a0[e], d1 = a1[e], d2 = a2[e], d3 = a3[e];
                v[e] += (d0 + d1) + (d2 + d3);
                v[4 + e] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) {
            f32x4 a = x4[w.idx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = a[e];
                v[e] += d;
                v[4 + e] += d * d;
            }
        }
    }
    red_out<8>(q, ri, active, c4, v, sum, sumsq, part, true);
}

// DOT = false: out[g][c] += sum_r act(bn(x));  DOT = true: out[g][c] += sum_r dy * act(bn(x))
template <typename T, bool DOT>
__global__ __launch_bounds__(NT) void colsum_bn_kernel(RedGeom q, const T* __restrict__ x,
                                                       const T* __restrict__ dy, ud_bn_ref bn,
                                                       double* __restrict__ out, double* __restrict__ part,
                                                       uint32_t* __restrict__ amax) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> x4{x}, d4{dy};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mo = 0.f;          // !DOT, amax: max |act(bn(x))| as a side output (the scale of the planes ud_se_scale_bn_planes writes)
    if (active) {
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, !DOT && is_updater(ri));
        Rows w = rows_of(q, ri, c4);
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) {
            f32x4 a = bn_apply(x4[w.idx], cb, bn.act);
            if (DOT) {
                f32x4 d = d4[w.idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (double)d[e] * (double)a[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (double)a[e];
                mo = fmaxf(mo, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
            }
        }
    }
    red_out<4>(q, ri, active, c4, v, out, nullptr, part, true);
    if (!DOT) ud_absmax_commit(mo, amax);
}

// y[n][o] = sum_i (xsum[n][i] * xscale) W[o][i] + b[o]; one wave per output
__global__ __launch_bounds__(NT) void fc_fwd_d_kernel(const double* __restrict__ xsum, float xscale,
                                                      const float* __restrict__ W, const float* __restrict__ b,
                                                      float* __restrict__ y, int N, int I, int O) {
    const int wave = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N * O) return;
    const int n = wave / O, o = wave % O;
    float acc = 0.f;
#pragma unroll 4
    for (int i = lane; i < I; i += 64) acc += ((float)xsum[(long)n * I + i] * xscale) * W[(long)o * I + i];
    acc = ud_wave_sum(acc);
    if (lane == 0) y[(long)n * O + o] = acc + (b ? b[o] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// elementwise consumers
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void se_scale_bn_kernel(RedGeom q, const T* __restrict__ x, ud_bn_ref bn,
                                                         const float* __restrict__ s, T* __restrict__ y,
                                                         uint32_t* __restrict__ amax) {
    int ri, c4;
    float m = 0.f;
    if (thread_coords(q, ri, c4)) {
        const In4<T> x4{x};
        const Out4<T> y4{y};
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
        f32x4 gate = reinterpret_cast<const f32x4*>(s)[(long)blockIdx.z * q.C4 + c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gate[e] = ud_sigmoid_fast(gate[e]);
        Rows w = rows_of(q, ri, c4);
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) {
            const f32x4 v = bn_apply(x4[w.idx], cb, bn.act) * gate;
            y4.st(w.idx, v);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    ud_absmax_commit(m, amax);
}

// The same pass writing its result DIRECTLY as the fp16 x 2 planes of the project conv's GEMM operand (ud_gemm_p3 prec 2, P32
// layout over [G R] x C): |y| <= max |act(bn(x))| (the gate is a sigmoid), and that maximum is the side output of ud_colsum_bn_amax,
// the SE squeeze pass that read the whole tensor a moment ago — the scale is the one the exact maximum of y would give, or one
// binade or two above.  amax_in: its 256 slots; *inv_scale receives 1 / scale.
__global__ __launch_bounds__(NT) void se_scale_bn_planes_kernel(RedGeom q, const float* __restrict__ x, ud_bn_ref bn,
                                                                const float* __restrict__ s, uint16_t* __restrict__ planes,
                                                                long panel, long plane, const uint32_t* __restrict__ amax_in,
                                                                float* __restrict__ inv_scale) {
    uint32_t mb = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) mb = max(mb, amax_in[(threadIdx.x & 63) + 64 * i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, o, 64));
    float ps, pinv;
    ud_h2_scale(mb, ps, pinv);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *inv_scale = pinv;
    int ri, c4;
    if (!thread_coords(q, ri, c4)) return;
    const In4<float> x4{x};
    const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
    f32x4 gate = reinterpret_cast<const f32x4*>(s)[(long)blockIdx.z * q.C4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gate[e] = ud_sigmoid_fast(gate[e]) * ps;
    Rows w = rows_of(q, ri, c4);
    uint16_t* o = planes + (long)(c4 >> 3) * panel + (c4 & 7) * 4;
    long row = (long)blockIdx.z * q.R + w.r;
#pragma unroll 4
    for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step, row += q.rpi) {
        const f32x4 v = bn_apply(x4[w.idx], cb, bn.act) * gate;
        uint16_t h0[4], h1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ud_split_h2(v[e], h0[e], h1[e]);
        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u16x4*>(o + row * 32) = u16x4{h0[0], h0[1], h0[2], h0[3]};
        *reinterpret_cast<u16x4*>(o + row * 32 + plane) = u16x4{h1[0], h1[1], h1[2], h1[3]};
    }
}

// half storage (the mixed-precision mode): the gated tensor laid straight into the ONE fp16 plane ud_gemm_p3 prec 1 reads
// (scale 1, pad columns of the last panel zero) — bit for bit se_scale_bn_kernel<_Float16>'s values
__global__ __launch_bounds__(NT) void se_scale_bn_plane_half_kernel(RedGeom q, const _Float16* __restrict__ x, ud_bn_ref bn,
                                                                    const float* __restrict__ s, uint16_t* __restrict__ plane,
                                                                    long panel, float* __restrict__ inv_scale) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *inv_scale = 1.f;
    int ri, c4;
    if (!thread_coords(q, ri, c4)) return;
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const In4<_Float16> x4{x};
    const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
    f32x4 gate = reinterpret_cast<const f32x4*>(s)[(long)blockIdx.z * q.C4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gate[e] = ud_sigmoid_fast(gate[e]);
    Rows w = rows_of(q, ri, c4);
    uint16_t* o = plane + (long)(c4 >> 3) * panel + (c4 & 7) * 4;
    const int npad = (c4 == q.C4 - 1) ? 7 - (c4 & 7) : 0;
    long row = (long)blockIdx.z * q.R + w.r;
#pragma unroll 4
    for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step, row += q.rpi) {
        const f32x4 v = bn_apply(x4[w.idx], cb, bn.act) * gate;
        Quad<_Float16>::st(reinterpret_cast<_Float16*>(o + row * 32), 0, v);
        for (int z = 1; z <= npad; ++z) *reinterpret_cast<u16x4*>(o + row * 32 + 4 * z) = u16x4{0, 0, 0, 0};
    }
}

// PL (round 6): ALSO written as the fp16 x 2 planes of the NEXT block's expand conv (ud_gemm_p3 prec 2, P32 layout over [G R] x C)
// — the block output is that conv's operand, and splitting it took a pass of its own (17 launches per step).  The scale comes from
// an a-priori bound: |bn(p)_c| <= |gamma_c| sqrt(count) + |beta_c| (a z-score never exceeds sqrt(count - 1)), times the
// drop-connect factor, plus the skip's exact maximum (its producer's absmax slots); every workgroup derives the same scale.
struct ResPlanes {
    uint16_t* buf;
    long panel, plane;
    float* inv_scale;
    const uint32_t* skip_amax;          // 256 slots: max |skip| (NULL: no skip)
};

template <typename T, int PL = 0>
__global__ __launch_bounds__(NT) void residual_bn_kernel(RedGeom q, const T* __restrict__ x, ud_bn_ref bn,
                                                         const float* __restrict__ keep, float inv_keep,
                                                         const T* __restrict__ skip, T* __restrict__ out,
                                                         uint32_t* __restrict__ amax, ResPlanes rp) {
    float ps = 1.f;
    if constexpr (PL == 1) {
        __shared__ float pl_red[NT / 64];
        const float rc = sqrtf((float)(1.0 / bn.inv_count));
        float gm = 0.f;
        for (int i = threadIdx.x; i < q.C4 * 4; i += NT) gm = fmaxf(gm, fabsf(bn.gamma[i]) * rc + fabsf(bn.beta[i]));
        gm *= keep ? inv_keep : 1.f;
        gm = ud_wave_max(gm);
        if ((threadIdx.x & 63) == 0) pl_red[threadIdx.x >> 6] = gm;
        __syncthreads();
        gm = pl_red[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) gm = fmaxf(gm, pl_red[i]);
        float sk = 0.f;
        if (rp.skip_amax) {
            uint32_t mb = rp.skip_amax[threadIdx.x & 255];          // NT = 256 slots: one each
            sk = __uint_as_float(mb);
            sk = ud_wave_max(sk);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) pl_red[threadIdx.x >> 6] = sk;
            __syncthreads();
            sk = pl_red[0];
#pragma unroll
            for (int i = 1; i < NT / 64; ++i) sk = fmaxf(sk, pl_red[i]);
        }
        float pinv;
        ud_h2_scale(__float_as_uint((gm + sk) * 1.001f), ps, pinv);
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *rp.inv_scale = pinv;
    }
    int ri, c4;
    float m = 0.f;
    if (thread_coords(q, ri, c4)) {
        const In4<T> x4{x}, k4{skip};
        const Out4<T> o4{out};
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, is_updater(ri));
        const float sc = keep ? keep[blockIdx.z] * inv_keep : 1.f;
        Rows w = rows_of(q, ri, c4);
        uint16_t* po = PL ? rp.buf + (long)(c4 >> 3) * rp.panel + (c4 & 7) * 4 : nullptr;
        const int npad = (PL && c4 == q.C4 - 1) ? 7 - (c4 & 7) : 0;
        long prow = (long)blockIdx.z * q.R + w.r;
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step, prow += q.rpi) {
            f32x4 v = bn_apply(x4[w.idx], cb, bn.act) * sc;
            if (skip) v += k4[w.idx];
            o4.st(w.idx, v);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            if constexpr (PL == 1) {
                typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                uint16_t h0[4], h1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ud_split_h2(v[e] * ps, h0[e], h1[e]);
                *reinterpret_cast<u16x4*>(po + prow * 32) = u16x4{h0[0], h0[1], h0[2], h0[3]};
                *reinterpret_cast<u16x4*>(po + prow * 32 + rp.plane) = u16x4{h1[0], h1[1], h1[2], h1[3]};
                for (int z = 1; z <= npad; ++z) {
                    *reinterpret_cast<u16x4*>(po + prow * 32 + 4 * z) = u16x4{0, 0, 0, 0};
                    *reinterpret_cast<u16x4*>(po + prow * 32 + 4 * z + rp.plane) = u16x4{0, 0, 0, 0};
                }
            }
        }
    }
    ud_absmax_commit(m, amax);
}

// y = act(bn(x)): the materialised form, for consumers that re-read their input many times (plain depthwise convs
// and their weight gradient: re-evaluating the swish per tap costs more than this pass)
template <typename T>
__global__ __launch_bounds__(NT) void bn_apply_kernel(RedGeom q, const T* __restrict__ x, ud_bn_ref bn,
                                                      T* __restrict__ y) {
    int ri, c4;
    if (!thread_coords(q, ri, c4)) return;
    const In4<T> x4{x};
    const Out4<T> y4{y};
    const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, is_updater(ri));
    Rows w = rows_of(q, ri, c4);
#pragma unroll kRowUnroll<T>
    for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) y4.st(w.idx, bn_apply(x4[w.idx], cb, bn.act));
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm backward
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dz_terms(const f32x4& a, const f32x4& d, const Bn4& cb, int act, bool is_dz, float sc,
                                         f32x4& dz, f32x4& xh) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = (a[e] - cb.mu[e]) * cb.is[e];
        float g = d[e];
        if (!is_dz) {
            g *= sc;
            if (act) g *= ud_act_grad_fast(cb.ga[e] * xh[e] + cb.be[e], act);
        }
        dz[e] = g;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void normbwd_sums_kernel(RedGeom q, const T* __restrict__ x,
                                                          const T* __restrict__ dy, const float* __restrict__ keep,
                                                          float inv_keep, ud_bn_ref bn, int dy_is_dz,
                                                          double* __restrict__ s1, double* __restrict__ s2,
                                                          double* __restrict__ s3, double* __restrict__ part) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> x4{x}, d4{dy};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 en = {0.f, 0.f, 0.f, 0.f};          // s3 != NULL: sum dz^2 (the energy bound of ud_normbwd_apply_planes' scale)
    if (active) {
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
        const float sc = keep ? keep[blockIdx.z] * inv_keep : 1.f;
        Rows w = rows_of(q, ri, c4);
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) {
            f32x4 dz, xh;
            dz_terms(x4[w.idx], d4[w.idx], cb, bn.act, dy_is_dz != 0, sc, dz, xh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += (double)dz[e];
                v[4 + e] += (double)dz[e] * (double)xh[e];
            }
            en += dz * dz;
        }
    }
    if (s3) {          // (uniform) rounded UP a little: the fp32 partial sums must never under-estimate
        double ev[8] = {(double)en[0] * 1.0001, (double)en[1] * 1.0001, (double)en[2] * 1.0001, (double)en[3] * 1.0001, 0, 0, 0, 0};
        block_fold<4>(q, ri, active, ev);
        if (active && ri == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomic_add_f64(s3 + (long)c4 * 4 + e, ev[e]);
        }
    }
    red_out<8>(q, ri, active, c4, v, s1, s2, part, false);      // batch-norm sums: one set for all groups
}

// PL (round 5): the result written DIRECTLY as the fp16 x 2 planes of the GEMMs that consume it (the conv's weight and data
// gradient: ud_gemm_p3 prec 2, P32 layout over [G R] x C, pad columns of the last panel zero) instead of fp32 + a split pass.
// The scale comes from an a-priori bound: dx_c = gamma_c invstd_c P(dz_c), P a contraction (the projection off 1 and xhat), so
// |dx| <= max_c |gamma_c invstd_c| sqrt(E_c), E_c >= sum_rows dz_c^2 over the WHOLE batch (all ranks) — `energy`, summed by the
// kernel that produced dz (ud_normbwd_sums' s3, ud_irfft2_dwbwd's sum_dz2).  Every workgroup derives the same bound (a scan of C
// values); block (0,0,0) stores 1 / scale.
struct PlanesDst {
    uint16_t* buf;
    long panel, plane;
    float* inv_scale;
    const double* energy;
};

// MIX: also the SF-mix gradient: acc += dd * diff, diff = freq - spat as written by ud_irfft2_mix
// PL = 2 (the mixed-precision mode, half storage): the half result itself laid into the P32 plane ud_gemm_p3 prec 1 reads (scale 1) —
// the row-major tensor had no other reader than the layout pass.
template <typename T, bool MIX, int PL = 0>
__global__ __launch_bounds__(NT) void normbwd_apply_kernel(RedGeom q, const T* __restrict__ x,
                                                           const T* __restrict__ dy, const float* __restrict__ keep,
                                                           float inv_keep, ud_bn_ref bn, int dy_is_dz,
                                                           const double* __restrict__ s1, const double* __restrict__ s2,
                                                           const double* __restrict__ s1l, const double* __restrict__ s2l,
                                                           const T* __restrict__ freq,
                                                           T* __restrict__ dx, double* __restrict__ dalpha_acc,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           uint32_t* __restrict__ amax, double* __restrict__ energy,
                                                           PlanesDst pd) {
    float ps = 1.f;
    if constexpr (PL == 2) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *pd.inv_scale = 1.f;
    }
    if constexpr (PL == 1) {
        __shared__ float pl_red[NT / 64];
        float gm = 0.f;
        for (int i = threadIdx.x; i < q.C4 * 4; i += NT) {
            const double m = bn.sum[i] * bn.inv_count;
            double vv = bn.sumsq[i] * bn.inv_count - m * m;
            if (vv < 0.0) vv = 0.0;
            const float gi = bn.gamma[i] * rsqrtf((float)(vv + (double)bn.eps));          // (bn_load's own form)
            gm = fmaxf(gm, gi * gi * (float)pd.energy[i]);
        }
        gm = ud_wave_max(gm);
        if ((threadIdx.x & 63) == 0) pl_red[threadIdx.x >> 6] = gm;
        __syncthreads();
        gm = pl_red[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) gm = fmaxf(gm, pl_red[i]);
        float pinv;
        ud_h2_scale(__float_as_uint(sqrtf(gm) * 1.002f), ps, pinv);          // (the 0.2 %: rsqrtf / fp32 rounding of the apply)
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *pd.inv_scale = pinv;
    }
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> x4{x}, d4{dy}, fr4{freq};
    const Out4<T> o4{dx};
    double acc = 0.0;
    float mo = 0.f;
    f32x4 en = {0.f, 0.f, 0.f, 0.f};          // MIX, energy != NULL: sum of dx^2 per channel over this thread's rows
    if (active) {
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
        const float sc = keep ? keep[blockIdx.z] * inv_keep : 1.f;
        f32x4 t1, t2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t1[e] = (float)s1[c4 * 4 + e] * (float)bn.inv_count;
            t2[e] = (float)s2[c4 * 4 + e] * (float)bn.inv_count;
        }
        if (is_updater(ri)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dbeta) dbeta[c4 * 4 + e] = (float)s1l[c4 * 4 + e];
                if (dgamma) dgamma[c4 * 4 + e] = (float)s2l[c4 * 4 + e];
            }
        }
        Rows w = rows_of(q, ri, c4);
        // PL: this thread's quad of the panel; the thread of the LAST quad also zeroes the panel's pad quads
        uint16_t* po = PL ? pd.buf + (long)(c4 >> 3) * pd.panel + (c4 & 7) * 4 : nullptr;
        const int npad = (PL && c4 == q.C4 - 1) ? 7 - (c4 & 7) : 0;
        long prow = (long)blockIdx.z * q.R + w.r;
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step, prow += q.rpi) {
            f32x4 dz, xh, o;
            dz_terms(x4[w.idx], d4[w.idx], cb, bn.act, dy_is_dz != 0, sc, dz, xh);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = cb.ga[e] * cb.is[e] * (dz[e] - t1[e] - xh[e] * t2[e]);
            if constexpr (PL == 2) {
                typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                Quad<T>::st(reinterpret_cast<T*>(po + prow * 32), 0, o);
                for (int z = 1; z <= npad; ++z) *reinterpret_cast<u16x4*>(po + prow * 32 + 4 * z) = u16x4{0, 0, 0, 0};
                continue;
            }
            if constexpr (PL == 1) {
                typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                uint16_t h0[4], h1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ud_split_h2(o[e] * ps, h0[e], h1[e]);
                *reinterpret_cast<u16x4*>(po + prow * 32) = u16x4{h0[0], h0[1], h0[2], h0[3]};
                *reinterpret_cast<u16x4*>(po + prow * 32 + pd.plane) = u16x4{h1[0], h1[1], h1[2], h1[3]};
                for (int z = 1; z <= npad; ++z) {
                    *reinterpret_cast<u16x4*>(po + prow * 32 + 4 * z) = u16x4{0, 0, 0, 0};
                    *reinterpret_cast<u16x4*>(po + prow * 32 + 4 * z + pd.plane) = u16x4{0, 0, 0, 0};
                }
                continue;
            }
            o4.st(w.idx, o);
            mo = fmaxf(mo, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            if (MIX) {
                f32x4 f = fr4[w.idx];                    // freq - spat (ud_irfft2_mix)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += (double)ud_rounded<T>(o[e]) * (double)f[e];
                en += o * o;
            }
        }
    }
    if (MIX && energy) {
        // per-channel energy of the result (the a-priori bound of the transform that follows: ud_rfft2_ex_planes); rounded UP a
        // little so that the fp32 partial sums never under-estimate
        double v[8] = {(double)en[0] * 1.0001, (double)en[1] * 1.0001, (double)en[2] * 1.0001, (double)en[3] * 1.0001, 0, 0, 0, 0};
        block_fold<4>(q, ri, active, v);
        if (active && ri == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomic_add_f64(energy + (long)c4 * 4 + e, v[e]);
        }
    }
    if (MIX) {
        __shared__ double sm[NT / 64];
        acc = ud_wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int i = 0; i < NT / 64; ++i) tot += sm[i];
            // 64 slots: ~13 workgroups per address instead of ~800 queued on one
            atomic_add_f64(dalpha_acc + ((blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & 63), tot);
        }
    }
    ud_absmax_commit(mo, amax);
}

__global__ void gate_grad_from_acc_kernel(const double* __restrict__ acc, const float* __restrict__ alpha,
                                          float* __restrict__ out) {
    double t = ud_wave_sum_d(acc[threadIdx.x]);          // 64 slots, one lane each
    if (threadIdx.x == 0) {
        const double a = 1.0 / (1.0 + exp(-(double)alpha[0]));
        out[0] = (float)(t * a * (1.0 - a));
    }
}

// db = dc * sigmoid(s) + dpool * inv_hw;  dz = db * act'(bn(x));  sums
template <typename T>
__global__ __launch_bounds__(NT) void se_scale_bwd_bn_kernel(RedGeom q, const T* __restrict__ dc,
                                                             const T* __restrict__ x, ud_bn_ref bn,
                                                             const float* __restrict__ s, const float* __restrict__ dpool,
                                                             float inv_hw, T* __restrict__ dzo,
                                                             double* __restrict__ s1, double* __restrict__ s2,
                                                             double* __restrict__ part) {
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> x4{x}, d4{dc};
    const Out4<T> o4{dzo};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const Bn4 cb = bn_load(bn, blockIdx.z, q.C4, c4, false);
        f32x4 gate = reinterpret_cast<const f32x4*>(s)[(long)blockIdx.z * q.C4 + c4];
        f32x4 dp = reinterpret_cast<const f32x4*>(dpool)[(long)blockIdx.z * q.C4 + c4] * inv_hw;
#pragma unroll
        for (int e = 0; e < 4; ++e) gate[e] = ud_sigmoid_fast(gate[e]);
        Rows w = rows_of(q, ri, c4);
#pragma unroll kRowUnroll<T>
        for (; w.r < w.r_end; w.r += q.rpi, w.idx += w.step) {
            f32x4 a = x4[w.idx], d = d4[w.idx], dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (a[e] - cb.mu[e]) * cb.is[e];
                float g = d[e] * gate[e] + dp[e];
                if (bn.act) g *= ud_act_grad_fast(cb.ga[e] * xh + cb.be[e], bn.act);
                g = ud_rounded<T>(g);                  // the sums are those of the stored gradient
                dz[e] = g;
                v[e] += (double)g;
                v[4 + e] += (double)g * (double)xh;
            }
            o4.st(w.idx, dz);
        }
    }
    red_out<8>(q, ri, active, c4, v, s1, s2, part, false);
}

// ---------------------------------------------------------------------------------------------------------
// SE backward FCs
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpre_of(const double* dgate, const float* s2, long i) {
    const float g = ud_sigmoid_fast(s2[i]);
    return (float)dgate[i] * g * (1.f - g);
}

// blocks [0, N * chunks): sample n, chunk of CH = 32 * NSL channels: ds1_acc[n][i] += sum_{c in chunk} dpre[n][c] We[c][i]
//            (dpre of the chunk staged in LDS; threads = (i, channel slice); fp64 atomics, C / CH adds per address).  A thread
//            walks 32 channels: round 4's chunks of 256 left each of the NSL = 2 slices of the wide (IP = 128) form a chain of 128
//            dependent-latency loads — 19 us for the 8 x 8 stage's 1632 x 68 FC against 8 us for everything else in the launch
// blocks beyond: dW_e[c][i] = sum_n dpre[n][c] swish(s1[n][i]), db_e[c] = sum_n dpre[n][c] for NT / IP channels each
template <int IP>
__global__ __launch_bounds__(NT) void se_bwd_a_kernel(const double* __restrict__ dgate, const float* __restrict__ s2,
                                                      const float* __restrict__ s1, const float* __restrict__ We,
                                                      double* __restrict__ ds1_acc, float* __restrict__ dWe,
                                                      float* __restrict__ dbe, int N, int C, int Cs) {
    constexpr int NSL = NT / IP;
    constexpr int CH = 32 * NSL;
    __shared__ float sh[NT];
    __shared__ float red[NT];
    const int i = threadIdx.x % IP, sl = threadIdx.x / IP;
    const int chunks = (C + CH - 1) / CH;
    if ((int)blockIdx.x < N * chunks) {
        const int n = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * CH;
        const int cn = min(CH, C - c0);
        if ((int)threadIdx.x < cn) sh[threadIdx.x] = dpre_of(dgate, s2, (long)n * C + c0 + threadIdx.x);
        __syncthreads();
        float acc = 0.f;
        if (i < Cs) {
#pragma unroll 4
            for (int c = sl; c < cn; c += NSL) acc += sh[c] * We[(long)(c0 + c) * Cs + i];
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (sl == 0 && i < Cs) {
            for (int k = 1; k < NSL; ++k) acc += red[k * IP + i];
            unsafeAtomicAdd(ds1_acc + (long)n * Cs + i, (double)acc);
        }
        return;
    }
    // weight gradient: the block's NSL channels x all i
    const int wb = blockIdx.x - N * chunks;
    const int c = wb * NSL + sl;
    const bool own = c < C && i < Cs;
    float acc = 0.f, accb = 0.f;
    for (int n0 = 0; n0 < N; n0 += IP) {                          // IP samples per round: sh[channel slot][sample]
        const int nn = min(IP, N - n0);
        __syncthreads();
        {
            const int cc = c;                                     // thread (i, sl) stages sample n0 + i of channel slot sl
            sh[sl * IP + i] = (cc < C && i < nn) ? dpre_of(dgate, s2, (long)(n0 + i) * C + cc) : 0.f;
        }
        __syncthreads();
        if (own) {
#pragma unroll 8
            for (int n = 0; n < nn; ++n) {
                const float g = sh[sl * IP + n];
                acc += g * ud_act_fast(s1[(long)(n0 + n) * Cs + i], 1);
                accb += g;
            }
        }
    }
    if (!own) return;
    dWe[(long)c * Cs + i] = acc;
    if (i == 0) dbe[c] = accb;
}

// ds1[n][i] = ds1_acc[n][i] * swish'(s1[n][i]).  grid (x, y):
//   y <  N : dpool[n][c] = sum_i ds1[n][i] Wr[i][c] for sample n = y, channels x * 256 ..
//   y >= N : i = y - N: dW_r[i][c] = sum_n ds1[n][i] pool[n][c] * pool_scale;  db_r[i] = sum_n ds1[n][i]
__global__ __launch_bounds__(NT) void se_bwd_b_kernel(const double* __restrict__ ds1_acc, const float* __restrict__ s1,
                                                      const float* __restrict__ Wr, const double* __restrict__ pool,
                                                      float pool_scale, float* __restrict__ dpool,
                                                      float* __restrict__ dWr, float* __restrict__ dbr, int N, int C,
                                                      int Cs) {
    __shared__ float sh[NT];
    const int c = blockIdx.x * NT + threadIdx.x;
    if ((int)blockIdx.y < N) {
        const int n = blockIdx.y;
        if ((int)threadIdx.x < Cs)
            sh[threadIdx.x] = (float)ds1_acc[(long)n * Cs + threadIdx.x] * ud_act_grad_fast(s1[(long)n * Cs + threadIdx.x], 1);
        __syncthreads();
        if (c >= C) return;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < Cs; ++i) acc += sh[i] * Wr[(long)i * C + c];
        dpool[(long)n * C + c] = acc;
        return;
    }
    const int i = blockIdx.y - N;
    if ((int)threadIdx.x < N)
        sh[threadIdx.x] = (float)ds1_acc[(long)threadIdx.x * Cs + i] * ud_act_grad_fast(s1[(long)threadIdx.x * Cs + i], 1);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float b = 0.f;
        for (int n = 0; n < N; ++n) b += sh[n];
        dbr[i] = b;
    }
    if (c >= C) return;
    float acc = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) acc += sh[n] * ((float)pool[(long)n * C + c] * pool_scale);
    dWr[(long)i * C + c] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// depthwise conv with a deferred BatchNorm on its input (forward) / behind its data gradient (backward)
// Column-owning decomposition over "items": forward items = (n, ho, strip of TW output columns).
// ---------------------------------------------------------------------------------------------------------
struct DwGeom {
    int N, H, W, C4, Ho, Wo, stride, pad_t, pad_l;
};
constexpr int TW = 8;

// Data gradient.  STRIP (stride 1): items = (n, h, strip of TW input columns); otherwise one input pixel per item.
// BN: push the result through act'(bn(x)) of the conv's input and accumulate the BatchNorm backward sums.
template <typename T, int K, bool STRIP, bool BN>
__global__ __launch_bounds__(NT) void dw_bwd_data_ex_kernel(RedGeom q, DwGeom d, const T* __restrict__ dy,
                                                            const float* __restrict__ gate_alpha, int gate_mode,
                                                            const float* __restrict__ wt, const T* __restrict__ add,
                                                            const T* __restrict__ x, ud_bn_ref bn,
                                                            T* __restrict__ out, double* __restrict__ s1,
                                                            double* __restrict__ s2, double* __restrict__ part) {
    constexpr int TWB = STRIP ? TW : 1;
    constexpr int NCOL = TWB + K - 1;
    int ri, c4;
    const bool active = thread_coords(q, ri, c4);
    const In4<T> dy4{dy}, add4{add}, x4{x};
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wt);
    const Out4<T> o4{out};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        Bn4 cb;
        if (BN) cb = bn_load(bn, 0, q.C4, c4, false);
        const float gs = gate_factor(gate_alpha, gate_mode);
        const int WB = (d.W + TWB - 1) / TWB;
        Rows it = rows_of(q, ri, c4);
        for (; it.r < it.r_end; it.r += q.rpi) {
            const int wb = it.r % WB;
            const int t2 = it.r / WB;
            const int h = t2 % d.H, n = t2 / d.H;
            const int w0 = wb * TWB;
            f32x4 acc[TWB];
#pragma unroll
            for (int t = 0; t < TWB; ++t) acc[t] = f32x4{0, 0, 0, 0};
            if (STRIP) {
                const int col0 = w0 + d.pad_l - (K - 1);          // dy column of window slot 0
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int ho = h + d.pad_t - kh;
                    if (ho < 0 || ho >= d.Ho) continue;
                    const In4<T> row = dy4 + ((((long)n * d.Ho + ho) * d.Wo) * q.C4 + c4);
                    f32x4 g[NCOL], w[K];
#pragma unroll
                    for (int j = 0; j < NCOL; ++j) {
                        const int wo = col0 + j;
                        g[j] = (wo >= 0 && wo < d.Wo) ? row[(long)wo * q.C4] : f32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) w[kw] = w4[(kh * K + kw) * q.C4 + c4];
#pragma unroll
                    for (int t = 0; t < TWB; ++t)
#pragma unroll
                        for (int kw = 0; kw < K; ++kw) acc[t] += g[t - kw + K - 1] * w[kw];
                }
            } else {
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int th = h + d.pad_t - kh;
                    if (th < 0 || (th % d.stride) != 0) continue;
                    const int ho = th / d.stride;
                    if (ho >= d.Ho) continue;
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const int tw = w0 + d.pad_l - kw;
                        if (tw < 0 || (tw % d.stride) != 0) continue;
                        const int wo = tw / d.stride;
                        if (wo >= d.Wo) continue;
                        acc[0] += dy4[(((long)n * d.Ho + ho) * d.Wo + wo) * q.C4 + c4] * w4[(kh * K + kw) * q.C4 + c4];
                    }
                }
            }
            const long o0 = (((long)n * d.H + h) * d.W + w0) * q.C4 + c4;
#pragma unroll
            for (int t = 0; t < TWB; ++t) {
                if (w0 + t >= d.W) continue;
                const long o = o0 + (long)t * q.C4;
                f32x4 da = acc[t] * gs;
                if (add) da += add4[o];
                if (BN) {
                    const f32x4 a = x4[o];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (a[e] - cb.mu[e]) * cb.is[e];
                        float g = da[e];
                        if (bn.act) g *= ud_act_grad_fast(cb.ga[e] * xh + cb.be[e], bn.act);
                        g = ud_rounded<T>(g);
                        da[e] = g;
                        v[e] += (double)g;
                        v[4 + e] += (double)g * (double)xh;
                    }
                }
                o4.st(o, da);
            }
        }
    }
    if (BN) red_out<8>(q, ri, active, c4, v, s1, s2, part, false);
}

// Weight gradient, sliding-window form (as dwconv.hip:dw_bwd_weight_rows; its finalize below applies the gate factor).
// x is the MATERIALISED activated input: re-evaluating a deferred swish for every window load made this kernel 5x slower.
template <typename T, int K, int S>
__global__ __launch_bounds__(NT) void dw_bwd_weight_rows_ex(DwGeom q, int C, int rows_per_group, const T* __restrict__ x,
                                      const T* __restrict__ dy, float* __restrict__ part) {
    // a lane owns CPL adjacent channels: 1 float or 2 halves — a 4-byte access either way (2-byte accesses ran this
    // kernel at a third of its fp32 speed)
    constexpr int CPL = sizeof(T) == 2 ? 2 : 1;
    typedef T VT __attribute__((ext_vector_type(CPL == 2 ? 2 : 1)));
    typedef float VF __attribute__((ext_vector_type(CPL == 2 ? 2 : 1)));
    const int c = (blockIdx.y * blockDim.x + threadIdx.x) * CPL;
    if (c >= C) return;
    const int p = blockIdx.x;
    const int rows_total = q.N * q.Ho;
    const int row0 = p * rows_per_group;
    int row1 = row0 + rows_per_group;
    if (row1 > rows_total) row1 = rows_total;
    const VF zero = {};
    auto ld = [](const T* ptr) { return __builtin_convertvector(*reinterpret_cast<const VT*>(ptr), VF); };
    VF acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = zero;
    for (int row = row0; row < row1; ++row) {
        const int n = row / q.Ho, ho = row % q.Ho;
        const int ih0 = ho * S - q.pad_t;
        const T* xr[K];
        bool vh[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = ih0 + kh;
            vh[kh] = (ih >= 0) && (ih < q.H);
            xr[kh] = x + (((long)n * q.H + (vh[kh] ? ih : 0)) * q.W) * C + c;
        }
        VF w[K][K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int iw = kw - q.pad_l;
                w[kh][kw] = (vh[kh] && iw >= 0 && iw < q.W) ? ld(xr[kh] + (long)iw * C) : zero;
            }
        const T* dyr = dy + ((long)row * q.Wo) * C + c;
#pragma unroll 4
        for (int wo = 0; wo < q.Wo; ++wo) {
            const VF g = ld(dyr + (long)wo * C);
#pragma unroll
            for (int kh = 0; kh < K; ++kh)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[kh * K + kw] += g * w[kh][kw];
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
#pragma unroll
                for (int kw = 0; kw + S < K; ++kw) w[kh][kw] = w[kh][kw + S];
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const int iw = (wo + 1) * S - q.pad_l + (K - S) + j;
                    w[kh][K - S + j] = (vh[kh] && iw >= 0 && iw < q.W) ? ld(xr[kh] + (long)iw * C) : zero;
                }
            }
        }
    }
    float* out = part + (long)p * (K * K) * C + c;
#pragma unroll
    for (int i = 0; i < K * K; ++i) *reinterpret_cast<VF*>(out + (long)i * C) = acc[i];
}

// 16 lanes x 4 consecutive outputs (one 16-byte load per partial) x 16 part-slices per block, 4 partials in flight per
// thread, fp64 accumulation; result in the parameter's layout dw[C][K*K], scaled by the gate.  KKC % 4 == 0 (C % 4 == 0).
__global__ __launch_bounds__(NT) void dw_bwd_weight_finalize_ex(int nparts, int KK, int C, const float* __restrict__ part,
                                                                const float* __restrict__ gate_alpha, int gate_mode,
                                                                float* __restrict__ dw) {
    const int KKC = KK * C;
    __shared__ double sm[16][16][4];
    const int lane = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (blockIdx.x * 16 + lane) * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < KKC) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(part + i);
        const long step = (long)KKC / 4;          // f32x4 stride between partials
        int p = sl;
        for (; p + 48 < nparts; p += 64) {
            const f32x4 v0 = p4[(long)p * step], v1 = p4[(long)(p + 16) * step];
            const f32x4 v2 = p4[(long)(p + 32) * step], v3 = p4[(long)(p + 48) * step];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += ((double)v0[e] + (double)v1[e]) + ((double)v2[e] + (double)v3[e]);
        }
        for (; p < nparts; p += 16) {
            const f32x4 v = p4[(long)p * step];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += (double)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[sl][lane][e] = a[e];
    __syncthreads();
    if (sl < 4 && i < KKC) {                     // thread (lane, e = sl) folds the 16 slices of output i + e
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][lane][sl];
        const int o = i + sl, tap = o / C, c = o % C;
        dw[(long)c * KK + tap] = (float)(t * (double)gate_factor(gate_alpha, gate_mode));
    }
}

inline bool dw_geom_ok(const DwGeom& q, int K) {
    return q.N > 0 && q.H > 0 && q.W > 0 && q.C4 > 0 && q.Ho > 0 && q.Wo > 0 && (q.stride == 1 || q.stride == 2) &&
           (K == 3 || K == 5);
}

inline bool shape_ok(int G, int R, int C) { return G >= 1 && R >= 1 && C >= 4 && C % 4 == 0; }

// Reducing kernels: one launch with fp64 atomics while at most 64 workgroups add to the same addresses (about
// 12 ns per add and address, serialised: a sub-microsecond tail); otherwise the two-launch form with fp64 partials in
// the caller's scratch `ws` (ud_fused_reduce_ws_doubles) — on the large early-stage tensors the finalize launch is
// noise, on the small late-stage ones the single launch halves the launch count.
struct RedPlan {
    RedGeom q;
    bool use_part;
};
inline RedPlan plan_reduce(int G, int R, int C, bool per_group, const double* ws, int min_rows = 8,
                           int row_elems = 1) {
    constexpr int lim = 64;                 // most contributions per column that still go through fp64 atomics
    constexpr long big_elems = 1L << 20;
    RedGeom qa = make_geom_ex(G, R, C, 1024, lim, min_rows);
    const long contrib = per_group ? qa.P : (long)qa.G * qa.P;
    // >= 16 MB (row_elems pixels per row item): the pass is bandwidth-bound and wants ~2000 workgroups; its finalize
    // launch is noise
    // (per-group accumulators — the squeeze-excite sums: [G][C] — see only P adds per address whatever the size: one launch)
    const bool big = !per_group && (long)G * R * row_elems * (C / 4) >= big_elems;
    if ((contrib <= lim && !big) || !ws) return RedPlan{qa, false};
    return RedPlan{make_geom_ex(G, R, C, 2048, 512, min_rows), true};
}
inline int finish_reduce(const RedPlan& pl, int nq, bool per_group, int C, const double* ws, double* a1, double* a2,
                         hipStream_t s) {
    if (!pl.use_part) return 0;
    const int G = per_group ? pl.q.G : 1;
    const int P = per_group ? pl.q.P : pl.q.G * pl.q.P;
    hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv((long)G * C, 8)), dim3(NT), 0, s, nq, G, C, P, ws, a1, a2);
    UD_LAUNCH_CHECK();
    return 0;
}
// The element-wise kernels (no reduction) take whole rows up to 512 channels per workgroup (128 float4 columns): a workgroup then
// streams contiguous memory like a plain copy, where the reductions' 256-byte column strips touch 16 rows x 256 B spread over
// 16 row pitches per iteration — normbwd_apply on 32 x 1024 x 336: 29.8 -> 23.7 us (4.4 -> 5.6 TB/s; torch's add on the same
// bytes: 6.4), whole step 26.67 -> 26.52 ms (tools/probe_stream_bw.py; the reductions are faster on the narrow strips).
inline RedGeom geom_ew(int G, int R, int C, int min_rows = 4) { return make_geom_ex(G, R, C, 2048, 4096, min_rows, 128); }

}  // namespace

extern "C" {

// doubles of scratch the reducing entry points below may need for their two-launch form (0: single launch)
long ud_fused_reduce_ws_doubles(int G, int R, int C, int per_group, int min_rows) {
    if (!shape_ok(G, R, C) || min_rows < 1) return UD_EINVAL;
    static double dummy;
    RedPlan pl = plan_reduce(G, R, C, per_group != 0, &dummy, min_rows);
    return pl.use_part ? 2L * pl.q.G * pl.q.P * C : 0;
}

// acc[n] += sum over the slots of the GEMM epilogue statistics (ud_gemm_desc.stat_sum with > 64 row tiles).  The two slot
// arrays need not be adjacent: two launches of the one-quantity form.
int ud_stat_slots_fold(const double* slot_sum, const double* slot_sumsq, int slots, int N, double* acc_sum,
                       double* acc_sumsq, ud_stream_t stream) {
    if (!slot_sum || !slot_sumsq || slots < 1 || N < 1 || !acc_sum || !acc_sumsq) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (slot_sumsq == slot_sum + (long)slots * N) {
        hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv(N, 8)), dim3(NT), 0, s, 2, 1, N, slots, slot_sum, acc_sum, acc_sumsq);
    } else {
        hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv(N, 8)), dim3(NT), 0, s, 1, 1, N, slots, slot_sum, acc_sum, nullptr);
        hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv(N, 8)), dim3(NT), 0, s, 1, 1, N, slots, slot_sumsq, acc_sumsq, nullptr);
    }
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_colstats(const void* x, int G, int R, int C, double* sum, double* sumsq, double* ws, int f16,
                ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !sum || !sumsq) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // G == 1 is ONE set of accumulators: planned like the other whole-batch reductions, so that a large tensor (the stem conv's
    // 100 MB output: 64 workgroups of atomics took 48 us) runs as ~500 workgroups of partials + a fold
    RedPlan pl = plan_reduce(G, R, C, G != 1, ws);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(colstats_kernel<T>, red_grid(pl.q), dim3(NT), 0, s, pl.q, (const T*)x, sum,
                                                sumsq, pl.use_part ? ws : nullptr));
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 2, G != 1, C, ws, sum, sumsq, s);
}

int ud_colsum_bn(const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double* ws, int f16,
                 ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !out) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedPlan pl = plan_reduce(G, R, C, true, ws);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL((colsum_bn_kernel<T, false>), red_grid(pl.q), dim3(NT), 0, s, pl.q,
                                                (const T*)x, (const T*)nullptr, *bn, out, pl.use_part ? ws : nullptr,
                                                (uint32_t*)nullptr));
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 1, true, C, ws, out, nullptr, s);
}

int ud_colsum_bn_amax(const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double* ws, uint32_t* absmax,
                      ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !out || !absmax) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedPlan pl = plan_reduce(G, R, C, true, ws);
    hipLaunchKernelGGL((colsum_bn_kernel<float, false>), red_grid(pl.q), dim3(NT), 0, s, pl.q, (const float*)x,
                       (const float*)nullptr, *bn, out, pl.use_part ? ws : nullptr, absmax);
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 1, true, C, ws, out, nullptr, s);
}

int ud_coldot_bn(const void* dy, const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double* ws,
                 int f16, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dy || !bn || !out) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedPlan pl = plan_reduce(G, R, C, true, ws);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL((colsum_bn_kernel<T, true>), red_grid(pl.q), dim3(NT), 0, s, pl.q,
                                                (const T*)x, (const T*)dy, *bn, out, pl.use_part ? ws : nullptr,
                                                (uint32_t*)nullptr));
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 1, true, C, ws, out, nullptr, s);
}

int ud_fc_fwd_d(const double* xsum, float xscale, const float* W, const float* b, float* y, int N, int I, int O,
                ud_stream_t stream) {
    if (N < 1 || I < 1 || O < 1 || !xsum || !W || !y) return UD_EINVAL;
    long waves = (long)N * O;
    hipLaunchKernelGGL(fc_fwd_d_kernel, dim3(ud_cdiv(waves * 64, NT)), dim3(NT), 0, (hipStream_t)stream, xsum, xscale, W,
                       b, y, N, I, O);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_bn_apply(const void* x, const ud_bn_ref* bn, void* y, int G, int R, int C, int f16, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !y) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(bn_apply_kernel<T>, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q,
                                                (const T*)x, *bn, (T*)y));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_scale_bn(const void* x, const ud_bn_ref* bn, const float* s, void* y, int G, int R, int C, int f16,
                   uint32_t* absmax, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !s || !y) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(se_scale_bn_kernel<T>, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q,
                                                (const T*)x, *bn, s, (T*)y, absmax));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_scale_bn_planes(const void* x, const ud_bn_ref* bn, const float* s, uint16_t* planes, long panel_stride,
                          long plane_stride, float* inv_scale, const uint32_t* amax_in, int G, int R, int C,
                          ud_stream_t stream) {
    if (!shape_ok(G, R, C) || C % 32 || !x || !bn || !s || !planes || !inv_scale || !amax_in) return UD_EINVAL;
    if (panel_stride < (long)G * R * 32 || panel_stride % 8 || plane_stride % 8 || plane_stride < (long)(C / 32) * panel_stride)
        return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    hipLaunchKernelGGL(se_scale_bn_planes_kernel, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q, (const float*)x, *bn, s,
                       planes, panel_stride, plane_stride, amax_in, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_residual_bn(const void* x, const ud_bn_ref* bn, const float* keep, float inv_keep, const void* skip,
                   void* out, int G, int R, int C, int f16, uint32_t* absmax, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !out) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(residual_bn_kernel<T>, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q,
                                                (const T*)x, *bn, keep, inv_keep, (const T*)skip, (T*)out, absmax, ResPlanes{}));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_residual_bn_planes(const float* x, const ud_bn_ref* bn, const float* keep, float inv_keep, const float* skip,
                          const uint32_t* skip_absmax, float* out, uint16_t* planes, long panel_stride, long plane_stride,
                          float* inv_scale, int G, int R, int C, uint32_t* absmax, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !out || !planes || !inv_scale || panel_stride < 32L * G * R ||
        plane_stride < panel_stride * ((C + 31) / 32) || (skip && !skip_absmax))
        return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    hipLaunchKernelGGL((residual_bn_kernel<float, 1>), red_grid(q), dim3(NT), 0, (hipStream_t)stream, q, x, *bn, keep, inv_keep,
                       skip, out, absmax, ResPlanes{planes, panel_stride, plane_stride, inv_scale, skip ? skip_absmax : nullptr});
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_scale_bn_plane_half(const void* x, const ud_bn_ref* bn, const float* s, uint16_t* plane, long panel_stride,
                              float* inv_scale, int G, int R, int C, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !bn || !s || !plane || !inv_scale || (bn->G != 1 && bn->G != G)) return UD_EINVAL;
    if (panel_stride < 32L * G * R) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    hipLaunchKernelGGL(se_scale_bn_plane_half_kernel, red_grid(q), dim3(NT), 0, (hipStream_t)stream, q, (const _Float16*)x, *bn, s,
                       plane, panel_stride, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_normbwd_sums(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                    int dy_is_dz, int G, int R, int C, double* s1, double* s2, double* s3, double* ws, int f16,
                    ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dy || !bn || !s1 || !s2 || bn->G != 1) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RedPlan pl = plan_reduce(G, R, C, false, ws);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(normbwd_sums_kernel<T>, red_grid(pl.q), dim3(NT), 0, s, pl.q, (const T*)x,
                                                (const T*)dy, keep, inv_keep, *bn, dy_is_dz, s1, s2, s3,
                                                pl.use_part ? ws : nullptr));
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 2, false, C, ws, s1, s2, s);
}

int ud_normbwd_apply(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                     int dy_is_dz, const double* s1, const double* s2, const double* s1_local,
                     const double* s2_local, int G, int R, int C, void* dx, float* dgamma, float* dbeta, int f16,
                     uint32_t* absmax, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dy || !bn || !s1 || !s2 || !dx || bn->G != 1) return UD_EINVAL;
    if ((dgamma || dbeta) && (!s1_local || !s2_local)) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL((normbwd_apply_kernel<T, false>), red_grid(q), dim3(NT), 0,
                                                (hipStream_t)stream, q, (const T*)x, (const T*)dy, keep, inv_keep, *bn,
                                                dy_is_dz, s1, s2, s1_local, s2_local, (const T*)nullptr, (T*)dx,
                                                (double*)nullptr, dgamma, dbeta, absmax, (double*)nullptr,
                                                PlanesDst{nullptr, 0, 0, nullptr, nullptr}));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_normbwd_apply_planes(const float* x, const float* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                            int dy_is_dz, const double* s1, const double* s2, const double* s1_local,
                            const double* s2_local, const double* energy, int G, int R, int C, uint16_t* planes,
                            long panel_stride, long plane_stride, float* inv_scale, float* dgamma, float* dbeta,
                            ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dy || !bn || !s1 || !s2 || !energy || !planes || !inv_scale || bn->G != 1 || !bn->gamma)
        return UD_EINVAL;
    if ((dgamma || dbeta) && (!s1_local || !s2_local)) return UD_EINVAL;
    const long rows = (long)G * R;
    if (panel_stride < 32 * rows || plane_stride < (long)ud_cdiv(C, 32) * panel_stride) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    hipLaunchKernelGGL((normbwd_apply_kernel<float, false, 1>), red_grid(q), dim3(NT), 0, (hipStream_t)stream, q, x, dy, keep,
                       inv_keep, *bn, dy_is_dz, s1, s2, s1_local, s2_local, (const float*)nullptr, (float*)nullptr,
                       (double*)nullptr, dgamma, dbeta, (uint32_t*)nullptr, (double*)nullptr,
                       PlanesDst{planes, panel_stride, plane_stride, inv_scale, energy});
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_normbwd_apply_plane_half(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                                int dy_is_dz, const double* s1, const double* s2, const double* s1_local,
                                const double* s2_local, int G, int R, int C, uint16_t* plane, long panel_stride,
                                float* inv_scale, float* dgamma, float* dbeta, ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dy || !bn || !s1 || !s2 || !plane || !inv_scale || bn->G != 1) return UD_EINVAL;
    if ((dgamma || dbeta) && (!s1_local || !s2_local)) return UD_EINVAL;
    if (panel_stride < 32L * G * R) return UD_EINVAL;
    RedGeom q = geom_ew(G, R, C);
    hipLaunchKernelGGL((normbwd_apply_kernel<_Float16, false, 2>), red_grid(q), dim3(NT), 0, (hipStream_t)stream, q,
                       (const _Float16*)x, (const _Float16*)dy, keep, inv_keep, *bn, dy_is_dz, s1, s2, s1_local, s2_local,
                       (const _Float16*)nullptr, (_Float16*)nullptr, (double*)nullptr, dgamma, dbeta, (uint32_t*)nullptr,
                       (double*)nullptr, PlanesDst{plane, panel_stride, 0, inv_scale, nullptr});
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_normbwd_apply_mix(const void* x, const void* dz, const ud_bn_ref* bn, const double* s1, const double* s2,
                         const double* s1_local, const double* s2_local, const void* diff, int G, int R, int C,
                         void* dd, double* dalpha_acc, float* dgamma, float* dbeta, double* energy, int f16,
                         ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !x || !dz || !bn || !s1 || !s2 || !dd || !diff || !dalpha_acc || bn->G != 1)
        return UD_EINVAL;
    if ((dgamma || dbeta) && (!s1_local || !s2_local)) return UD_EINVAL;
    RedGeom q = make_geom_ex(G, R, C, 512, 4096, 4);      // one fp64 atomic per workgroup onto dalpha_acc
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL((normbwd_apply_kernel<T, true>), red_grid(q), dim3(NT), 0,
                                                (hipStream_t)stream, q, (const T*)x, (const T*)dz, (const float*)nullptr,
                                                1.f, *bn, 1, s1, s2, s1_local, s2_local, (const T*)diff, (T*)dd,
                                                dalpha_acc, dgamma, dbeta, (uint32_t*)nullptr, energy,
                                                PlanesDst{nullptr, 0, 0, nullptr, nullptr}));
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_gate_grad_from_acc(const double* acc, const float* alpha, float* out, ud_stream_t stream) {
    if (!acc || !alpha || !out) return UD_EINVAL;
    hipLaunchKernelGGL(gate_grad_from_acc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, alpha, out);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_bwd_a(const double* dgate, const float* s2, const float* s1, const float* We, double* ds1_acc, float* dWe,
                float* dbe, int N, int C, int Cs, ud_stream_t stream) {
    if (N < 1 || C < 1 || Cs < 1 || Cs > 128 || !dgate || !s2 || !s1 || !We || !ds1_acc || !dWe || !dbe) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int role1 = N * ud_cdiv(C, Cs <= 64 ? 32 * (NT / 64) : 32 * (NT / 128));
    if (Cs <= 64) {
        dim3 grid((unsigned)(role1 + ud_cdiv(C, NT / 64)));
        hipLaunchKernelGGL(se_bwd_a_kernel<64>, grid, dim3(NT), 0, s, dgate, s2, s1, We, ds1_acc, dWe, dbe, N, C, Cs);
    } else {
        dim3 grid((unsigned)(role1 + ud_cdiv(C, NT / 128)));
        hipLaunchKernelGGL(se_bwd_a_kernel<128>, grid, dim3(NT), 0, s, dgate, s2, s1, We, ds1_acc, dWe, dbe, N, C, Cs);
    }
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_bwd_b(const double* ds1_acc, const float* s1, const float* Wr, const double* pool, float pool_scale,
                float* dpool, float* dWr, float* dbr, int N, int C, int Cs, ud_stream_t stream) {
    if (N < 1 || N > NT || C < 1 || Cs < 1 || Cs > NT || !ds1_acc || !s1 || !Wr || !pool || !dpool || !dWr || !dbr)
        return UD_EINVAL;
    dim3 grid((unsigned)ud_cdiv(C, NT), (unsigned)(N + Cs));
    hipLaunchKernelGGL(se_bwd_b_kernel, grid, dim3(NT), 0, (hipStream_t)stream, ds1_acc, s1, Wr, pool, pool_scale, dpool,
                       dWr, dbr, N, C, Cs);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_se_scale_bwd_bn(const void* dc, const void* x, const ud_bn_ref* bn, const float* s, const float* dpool,
                       float inv_hw, void* dz, double* s1, double* s2, double* ws, int G, int R, int C, int f16,
                       ud_stream_t stream) {
    if (!shape_ok(G, R, C) || !dc || !x || !bn || !s || !dpool || !dz || !s1 || !s2 || bn->G != 1) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    RedPlan pl = plan_reduce(G, R, C, false, ws);
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL(se_scale_bwd_bn_kernel<T>, red_grid(pl.q), dim3(NT), 0, st, pl.q,
                                                (const T*)dc, (const T*)x, *bn, s, dpool, inv_hw, (T*)dz, s1, s2,
                                                pl.use_part ? ws : nullptr));
    UD_LAUNCH_CHECK();
    return finish_reduce(pl, 2, false, C, ws, s1, s2, st);
}

// items of the depthwise data-gradient kernels (rows of their column-owning decomposition) and its scratch need
static long dw_bwd_items(int N, int H, int W, int stride) {
    return stride == 1 ? (long)N * H * ((W + TW - 1) / TW) : (long)N * H * W;
}
long ud_dwconv_bwd_data_bn_ws_doubles(int N, int H, int W, int C, int stride) {
    const long items = dw_bwd_items(N, H, W, stride);
    if (items < 1 || items > 0x7fffffffL || C % 4 || C < 4) return UD_EINVAL;
    static double dummy;
    RedPlan pl = plan_reduce(1, (int)items, C, false, &dummy, stride == 1 ? 1 : 4, stride == 1 ? TW : 1);
    return pl.use_part ? 2L * pl.q.G * pl.q.P * C : 0;
}

}  // extern "C"

template <typename T>
static int dw_bwd_data_launch(const T* dy, const float* gate_alpha, int gate_mode, const float* wt, const T* add,
                              const T* x, const ud_bn_ref* bn, T* out, double* s1, double* s2, double* ws, int N,
                              int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, hipStream_t s) {
    if (C % 4 || !dy || !wt || !out) return UD_EINVAL;
    DwGeom d{N, H, W, C / 4, Ho, Wo, stride, pad_t, pad_l};
    if (!dw_geom_ok(d, K)) return UD_EINVAL;
    const bool strip = stride == 1;
    const long items = dw_bwd_items(N, H, W, stride);
    if (items > 0x7fffffffL) return UD_EINVAL;
    ud_bn_ref none{};
    const bool has_bn = bn != nullptr;
    if (has_bn && (!x || !s1 || !s2 || bn->G != 1)) return UD_EINVAL;
    RedPlan pl{make_geom_ex(1, (int)items, C, 2048, 8192, strip ? 1 : 4), false};
    if (has_bn) pl = plan_reduce(1, (int)items, C, false, ws, strip ? 1 : 4, strip ? TW : 1);
    const RedGeom& q = pl.q;
    double* part = pl.use_part ? ws : nullptr;
    const ud_bn_ref& b = has_bn ? *bn : none;
#define UD_DW_BWD(KK, ST, BB)                                                                                        \
    hipLaunchKernelGGL((dw_bwd_data_ex_kernel<T, KK, ST, BB>), red_grid(q), dim3(NT), 0, s, q, d, dy, gate_alpha,    \
                       gate_mode, wt, add, x, b, out, s1, s2, part)
    if (K == 3) {
        if (strip) { if (has_bn) UD_DW_BWD(3, true, true); else UD_DW_BWD(3, true, false); }
        else { if (has_bn) UD_DW_BWD(3, false, true); else UD_DW_BWD(3, false, false); }
    } else {
        if (strip) { if (has_bn) UD_DW_BWD(5, true, true); else UD_DW_BWD(5, true, false); }
        else { if (has_bn) UD_DW_BWD(5, false, true); else UD_DW_BWD(5, false, false); }
    }
#undef UD_DW_BWD
    UD_LAUNCH_CHECK();
    return has_bn ? finish_reduce(pl, 2, false, C, ws, s1, s2, s) : 0;
}

extern "C" {

int ud_dwconv_bwd_data_bn(const void* dy, const float* gate_alpha, int gate_mode, const float* wt, const void* add,
                          const void* x, const ud_bn_ref* bn, void* dz, double* s1, double* s2, double* ws, int N,
                          int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int f16,
                          ud_stream_t stream) {
    if (!bn) return UD_EINVAL;
    UD_STORAGE_DISPATCH(f16, return dw_bwd_data_launch<T>((const T*)dy, gate_alpha, gate_mode, wt, (const T*)add,
                                                          (const T*)x, bn, (T*)dz, s1, s2, ws, N, H, W, C, Ho, Wo, K,
                                                          stride, pad_t, pad_l, (hipStream_t)stream));
}

int ud_dwconv_bwd_data_ex(const void* dy, const float* gate_alpha, int gate_mode, const float* wt, const void* add,
                          void* dx, int N, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t,
                          int pad_l, int f16, ud_stream_t stream) {
    UD_STORAGE_DISPATCH(f16, return dw_bwd_data_launch<T>((const T*)dy, gate_alpha, gate_mode, wt, (const T*)add,
                                                          (const T*)nullptr, nullptr, (T*)dx, nullptr, nullptr, nullptr,
                                                          N, H, W, C, Ho, Wo, K, stride, pad_t, pad_l,
                                                          (hipStream_t)stream));
}

int ud_dwconv_bwd_weight_ex(const void* x, const void* dy, const float* gate_alpha, int gate_mode, float* dwt,
                            float* part, int chunks, int N, int H, int W, int C, int Ho, int Wo, int K, int stride,
                            int pad_t, int pad_l, int f16, ud_stream_t stream) {
    if (C % 4 || chunks < 1 || !x || !dy || !dwt || !part) return UD_EINVAL;
    DwGeom q{N, H, W, C / 4, Ho, Wo, stride, pad_t, pad_l};
    if (!dw_geom_ok(q, K)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int rows_total = N * Ho;
    if (chunks > rows_total) return UD_EINVAL;
    const int rpg = (rows_total + chunks - 1) / chunks;
    const int lanes = f16 ? C / 2 : C;          // half: two adjacent channels per lane (C % 4 == 0)
    int bt = ((lanes < NT ? lanes : NT) + 63) / 64 * 64;
    dim3 grid((unsigned)chunks, (unsigned)ud_cdiv(lanes, bt), 1);
#define UD_DW_WG(KK, SS)                                                                                              \
    UD_STORAGE_DISPATCH(f16, hipLaunchKernelGGL((dw_bwd_weight_rows_ex<T, KK, SS>), grid, dim3(bt), 0, s, q, C, rpg,         \
                                                (const T*)x, (const T*)dy, part))
    if (K == 3 && stride == 1) UD_DW_WG(3, 1);
    else if (K == 3) UD_DW_WG(3, 2);
    else if (stride == 1) UD_DW_WG(5, 1);
    else UD_DW_WG(5, 2);
#undef UD_DW_WG
    UD_LAUNCH_CHECK();
    const int KKC = K * K * C;
    hipLaunchKernelGGL(dw_bwd_weight_finalize_ex, dim3(ud_cdiv(KKC, 64)), dim3(NT), 0, s, chunks, K * K, C, part,
                       gate_alpha, gate_mode, dwt);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
