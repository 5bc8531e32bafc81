// Backward of a THIN expand 1x1 conv with its BatchNorm's backward applied ON LOAD (round 6).
//
// The first blocks of the trunk (model/efficientnet/model.py:94-112: _expand_conv 24 -> 144 on 128 x 128, 32 -> 192 on 64 x 64) hold
// the step's largest tensors: e = expand(x) [M][CE] and the gradient dz of swish(bn0(e)) [M][CE] with M = N H W = 131072 ... 524288.
// Their backward used to be three passes over them:
//     ud_normbwd_apply      de = gamma invstd (dz - mean(dz) - xhat mean(dz xhat))        reads dz, e     writes de
//     gemm_x3 (tn)          dW[CE][CIN] = de^T x                                          reads de, x
//     gemm_x3 (nn)          dx[M][CIN]  = de W (+ the skip path's gradient)               reads de        writes dx
// — 5 tensor passes of M x CE for 0.1 % of the step's FLOPs.  This kernel makes ONE pass: a workgroup takes 32-row tiles, forms de
// from (dz, e) in registers (the statistics sums s1 / s2 come from the kernel that produced dz, as before), splits it into gemm_x3's
// three exact bf16 pieces, lays them into ONE row-major LDS image and runs both products from that image: the data gradient reads
// rows (ds_read_b128: k = channels), the weight gradient reads the SAME image column-wise (ds_read_b64_tr_b16: k = rows).  W lives
// in registers as bf16 fragments for the whole kernel; the weight gradient accumulates in registers over all the workgroup's tiles
// and leaves one partial per workgroup, folded in a fixed order by pw_bwd_fold_kernel (deterministic by construction).
// Arithmetic = gemm_x3's: six piece products of order <= 2^-16 per k-step on v_mfma_f32_16x16x32_bf16, fp32 accumulation.
#include "pw_common.h"

#include <type_traits>

namespace {

constexpr int NTH = 256;          // 4 waves
constexpr int R = 32;             // rows per tile = the k of one weight-gradient MFMA

using namespace pw;

template <int CE, int CIN> struct PwCfg {
    static_assert(CE % 16 == 0 && CIN % 8 == 0 && CIN > 16 && CIN <= 32, "thin expand convs: 16 < CIN <= 32");
    static constexpr int KD = (CE + 31) / 32 * 32;      // k of the data gradient, padded to whole MFMA k-steps (pad channels stay zero)
    static constexpr int KS = KD / 32;
    static constexpr int DYS = KD * 2 + 16;             // bytes of an image row (the + 16 spreads rows over the banks)
    static constexpr int Q = CE / 4;                    // channel quads per row
    static constexpr int NIT = (R * Q + NTH - 1) / NTH; // float4 items of (dz, e) per thread and tile
    static constexpr int RI = (NIT * NTH + Q - 1) / Q;  // image rows: the items past the tile's 32 rows land in rows that are never read
    static constexpr int DYP = RI * DYS;                // one bf16 plane of the de tile
    static constexpr int NB = 2;                        // 16-column blocks of the padded CIN
    static constexpr int XS = NB * 32 + 16;
    static constexpr int XP = (NTH / (CIN / 4) + 1) * XS;          // (the x tile's items past its 32 rows: rows never read, too)
    static constexpr int XQ = CIN / 4;
    static constexpr int MB = CE / 16;                  // 16-channel row blocks of the weight gradient
    static constexpr int NBW = (MB + 1) / 2;            // ... per wave: wave w owns (row block (w >> 1) + 2 i, column block w & 1)
    static constexpr int IMG = 3 * DYP + 3 * XP;
    static constexpr int LDS = IMG + 5 * CE * 4;
};

struct PwArgs {
    const float* e;            // [M][CE] expand conv output (pre-BatchNorm)
    const float* dz;           // [M][CE] gradient of the BatchNorm's OUTPUT (activation already differentiated: dy_is_dz)
    const float* x;            // [M][CIN] the conv's input
    const float* w;            // [CE][CIN]
    const float* add;          // [M][CIN] a term of dx to add (may alias dx), or NULL
    float* dx;                 // [M][CIN]
    float* part;               // [grid][CE][CIN] weight-gradient partials
    const double* bsum;        // BatchNorm statistics of e (ud_bn_ref.sum / sumsq)
    const double* bsumsq;
    const float* gamma;
    const double* s1;          // sum dz, sum dz xhat over all rows (and ranks)
    const double* s2;
    double inv_count;
    float eps;
    long M;
    long tiles;
};

// D: tiles of global loads in flight per thread (a register ring: slot t % D holds tile t's (dz, e, x) quads until they are converted);
// WPC: workgroups per CU the register budget is cut for.  The kernel is latency-bound without the ring: one tile per workgroup in
// flight only while its MFMA phase runs left the 128 x 128 block at 3.5 TB/s (profiles/r06/pw_bwd_fused.txt).
template <int CE, int CIN, int D, int WPC>
__global__ __launch_bounds__(NTH, WPC) void pw_bwd_kernel(PwArgs a) {
    using CF = PwCfg<CE, CIN>;
    extern __shared__ __attribute__((aligned(16))) char L[];
    const lds_char* Lp = (const lds_char*)L;
    float* coef = reinterpret_cast<float*>(L + CF::IMG);          // mu, invstd, gamma invstd, mean(dz), mean(dz xhat): [5][CE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4;
    const int mb_d = wave >> 1, nbk = wave & 1;          // data gradient: 16-row block / 16-column block of this wave; weight gradient: column block

    for (int i = tid; i < CF::IMG / 16; i += NTH) reinterpret_cast<u32x4*>(L)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int c = tid; c < CE; c += NTH) {
        const double m = a.bsum[c] * a.inv_count;
        double v = a.bsumsq[c] * a.inv_count - m * m;
        if (v < 0.0) v = 0.0;
        const float is = rsqrtf((float)(v + (double)a.eps));          // (bnref.h: bn_load's own form)
        coef[c] = (float)m;
        coef[CE + c] = is;
        coef[2 * CE + c] = a.gamma[c] * is;
        coef[3 * CE + c] = (float)a.s1[c] * (float)a.inv_count;
        coef[4 * CE + c] = (float)a.s2[c] * (float)a.inv_count;
    }

    // W as the data gradient's B operand: lane holds W[k = 32 ks + 8 g + j][n = 16 nbk + u], j = 0..7, three pieces
    u32x4 wf[CF::KS][3];
    {
        const int n = 16 * nbk + u;
#pragma unroll
        for (int ks = 0; ks < CF::KS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * ks + 8 * g + j;
                const float wv = a.w[min(k, CE - 1) * CIN + min(n, CIN - 1)];          // (branch-free: clamped, then selected)
                v[j] = (n < CIN && k < CE) ? wv : 0.f;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                uint32_t p0, p1, p2;
                split2(v[2 * h], v[2 * h + 1], p0, p1, p2);
                wf[ks][0][h] = p0; wf[ks][1][h] = p1; wf[ks][2][h] = p2;
            }
        }
    }

    f32x4 accw[CF::NBW];
#pragma unroll
    for (int i = 0; i < CF::NBW; ++i) accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f32x4* dz4 = reinterpret_cast<const f32x4*>(a.dz);
    const f32x4* e4 = reinterpret_cast<const f32x4*>(a.e);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
    const long lim4 = a.M * CF::Q, limx = a.M * CF::XQ;
    f32x4 rdz[D][CF::NIT], re[D][CF::NIT], rx[D];
    float rav[D][4];
    const float* const add_src = a.add ? a.add : a.dx;          // (no add: the loads read dx's own memory and are discarded)
    const bool has_add = a.add != nullptr;
    const int dcol = 16 * nbk + u;                              // this lane's dx column
    // Loads WITHOUT any branch: a tile past the end is clamped to the last one (it is never converted), addresses are clamped into the
    // tensors and validity is re-derived at store time — so every s_waitcnt of the loop is a static count and a wait for one slot
    // leaves the younger slots' loads in flight (vmcnt counts in issue order; a conditional load forces vmcnt(0) at the join).
    auto prefetch = [&](auto slot_c, long tile_) {
        constexpr int S = decltype(slot_c)::value;
        const long tile = min(tile_, a.tiles - 1);
        const long b4 = tile * (R * CF::Q);
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) {
            const int idx = tid + it * NTH;
            long i = b4 + idx;
            i = (idx < R * CF::Q && i < lim4) ? i : 0;
            rdz[S][it] = dz4[i];
            re[S][it] = e4[i];
        }
        long ix = tile * (R * CF::XQ) + tid;
        ix = (tid < R * CF::XQ && ix < limx) ? ix : 0;
        rx[S] = x4[ix];
        const long row0 = tile * R + 16 * mb_d + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) rav[S][r] = add_src[min(row0 + r, a.M - 1) * CIN + min(dcol, CIN - 1)];
    };

    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    // one tile: slot S -> image, barrier, slot S refilled with tile + D grid, both products, barrier
    auto step = [&](auto slot_c, long tile) {
        constexpr int S = decltype(slot_c)::value;
        // ---- de = gamma invstd (dz - t1 - xhat t2), split, into the image
        {
            const long b4 = tile * (R * CF::Q);
#pragma unroll
            for (int it = 0; it < CF::NIT; ++it) {
                const int idx = tid + it * NTH;
                {          // (no predicate: an item past row 31 converts whatever its clamped load returned into a row nobody reads)
                    const int row = idx / CF::Q, q = idx - row * CF::Q;
                    const bool ok = b4 + idx < lim4;
                    const f32x4 mu = c4[q], is = c4[CF::Q + q], gi = c4[2 * CF::Q + q], t1 = c4[3 * CF::Q + q], t2 = c4[4 * CF::Q + q];
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (re[S][it][k] - mu[k]) * is[k];
                        o[k] = ok ? gi[k] * (rdz[S][it][k] - t1[k] - xh * t2[k]) : 0.f;
                    }
                    uint32_t p0[2], p1[2], p2[2];
                    split2(o[0], o[1], p0[0], p1[0], p2[0]);
                    split2(o[2], o[3], p0[1], p1[1], p2[1]);
                    char* dst = L + row * CF::DYS + q * 8;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{p0[0], p0[1]};
                    *reinterpret_cast<u32x2*>(dst + CF::DYP) = u32x2{p1[0], p1[1]};
                    *reinterpret_cast<u32x2*>(dst + 2 * CF::DYP) = u32x2{p2[0], p2[1]};
                }
            }
            {
                const int row = tid / CF::XQ, q = tid - row * CF::XQ;
                const bool ok = tile * (R * CF::XQ) + tid < limx;
                uint32_t p0[2], p1[2], p2[2];
                split2(ok ? rx[S][0] : 0.f, ok ? rx[S][1] : 0.f, p0[0], p1[0], p2[0]);
                split2(ok ? rx[S][2] : 0.f, ok ? rx[S][3] : 0.f, p0[1], p1[1], p2[1]);
                char* dst = L + 3 * CF::DYP + row * CF::XS + q * 8;
                *reinterpret_cast<u32x2*>(dst) = u32x2{p0[0], p0[1]};
                *reinterpret_cast<u32x2*>(dst + CF::XP) = u32x2{p1[0], p1[1]};
                *reinterpret_cast<u32x2*>(dst + 2 * CF::XP) = u32x2{p2[0], p2[1]};
            }
        }
        __syncthreads();
        float av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = has_add ? rav[S][r] : 0.f;
        prefetch(slot_c, tile + (long)D * gridDim.x);          // in flight under this and the next D - 1 tiles' phases

        // ---- data gradient: dx[16 rows][16 columns] of this wave, k = channels
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int col = dcol;
            const long row0 = tile * R + 16 * mb_d + 4 * g;
            const lds_char* arow = Lp + (16 * mb_d + u) * CF::DYS + g * 16;
#pragma unroll
            for (int ks = 0; ks < CF::KS; ++ks) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(arow + p * CF::DYP + ks * 64));
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, wf[ks][0]), b1 = __builtin_bit_cast(bf16x8, wf[ks][1]),
                             b2 = __builtin_bit_cast(bf16x8, wf[ks][2]);
                // smallest terms first (gemm_x3.hip's order)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col < CIN && row0 + r < a.M) a.dx[(row0 + r) * CIN + col] = acc[r] + av[r];
        }
        // ---- weight gradient: dW[16 channels][16 columns] blocks of this wave, k = the tile's 32 rows (transposed reads)
        {
            const int q = u >> 2, pp = u & 3;
            bf16x8 bf[3];
            const lds_char* xb = Lp + 3 * CF::DYP + (8 * g + q) * CF::XS + (16 * nbk + 4 * pp) * 2;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xb + p * CF::XP));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xb + p * CF::XP + 4 * CF::XS));
                bf[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            const lds_char* ab = Lp + (8 * g + q) * CF::DYS + 4 * pp * 2;
#pragma unroll
            for (int i = 0; i < CF::NBW; ++i) {
                // (a wave's block past the last one — odd MB — recomputes the last block into an accumulator that is never stored:
                //  no branch, so the reads of block i + 1 are scheduled under the MFMAs of block i)
                const int mbk = min((wave >> 1) + 2 * i, CF::MB - 1);
                {
                    bf16x8 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const lds_char* s = ab + p * CF::DYP + mbk * 32;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + 4 * CF::DYS));
                        af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    f32x4 c = accw[i];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0], c, 0, 0, 0);
                    accw[i] = c;
                }
            }
        }
        __syncthreads();          // every wave is done with the image before the next tile overwrites it
    };

    using std::integral_constant;
    {
        long t = blockIdx.x;
        prefetch(integral_constant<int, 0>{}, t);
        if constexpr (D > 1) { t += gridDim.x; prefetch(integral_constant<int, 1>{}, t); }
        if constexpr (D > 2) { t += gridDim.x; prefetch(integral_constant<int, 2>{}, t); }
        if constexpr (D > 3) { t += gridDim.x; prefetch(integral_constant<int, 3>{}, t); }
    }
    __syncthreads();          // images zeroed, coefficients in place
    for (long tile = blockIdx.x;;) {
        if (tile >= a.tiles) break;
        step(integral_constant<int, 0>{}, tile);
        tile += gridDim.x;
        if constexpr (D > 1) {
            if (tile >= a.tiles) break;
            step(integral_constant<int, 1>{}, tile);
            tile += gridDim.x;
        }
        if constexpr (D > 2) {
            if (tile >= a.tiles) break;
            step(integral_constant<int, 2>{}, tile);
            tile += gridDim.x;
        }
        if constexpr (D > 3) {
            if (tile >= a.tiles) break;
            step(integral_constant<int, 3>{}, tile);
            tile += gridDim.x;
        }
    }

    // ---- this workgroup's weight-gradient partial
    {
        float* part = a.part + (long)blockIdx.x * (CE * CIN);
        const int col = 16 * nbk + u;
        if (col < CIN) {
#pragma unroll
            for (int i = 0; i < CF::NBW; ++i) {
                const int mbk = (wave >> 1) + 2 * i;
                if (mbk < CF::MB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[(16 * mbk + 4 * g + r) * CIN + col] = accw[i][r];
                }
            }
        }
    }
}

// dW = the partials summed in a FIXED order; dgamma / dbeta from this rank's sums (ud_normbwd_apply's side outputs).
// Workgroup = 16 consecutive elements x 16 lanes of partials: lane group j adds partials j, j + 16, ... (independent loads, 64-byte
// segments), the 16 group sums are added in order through LDS.  (A first form — one thread per element walking all 512 partials —
// took 40-50 us: 14 workgroups of dependent loads; profiles/r06/pw_bwd_fused.txt.)
__global__ __launch_bounds__(256) void pw_bwd_fold_kernel(const float* __restrict__ part, int nparts, int numel, float* __restrict__ dw,
                                                          const double* __restrict__ s1l, const double* __restrict__ s2l, int CE,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f;
    if (i < numel) {
        int p = grp;
        for (; p + 16 < nparts; p += 32) {
            s0 += part[(long)p * numel + i];
            s1 += part[(long)(p + 16) * numel + i];
        }
        if (p < nparts) s0 += part[(long)p * numel + i];
    }
    red[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0 && i < numel) {
        float t = red[0][el];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][el];
        dw[i] = t;
    }
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < CE) {
        if (dbeta) dbeta[c] = (float)s1l[c];
        if (dgamma) dgamma[c] = (float)s2l[c];
    }
}

template <int CE, int CIN, int D, int WPC>
int launch(const PwArgs& a, int grid, hipStream_t s) {
    using CF = PwCfg<CE, CIN>;
    static_assert(CF::LDS <= 64 * 1024, "dynamic LDS within the default limit");
    hipLaunchKernelGGL((pw_bwd_kernel<CE, CIN, D, WPC>), dim3((unsigned)grid), dim3(NTH), CF::LDS, s, a);
    return 0;
}

int g_pw_form = 0;          // tools/bench_pwbwd.py: 0 the shipped form per shape, 1..4 = (D, WPC) in {(1,2), (2,2), (3,1), (4,1)}

}  // namespace

int pw::fold_launch(const float* part, int nparts, int numel, float* dw, const double* s1l, const double* s2l, int CE, float* dgamma,
                    float* dbeta, hipStream_t s) {
    hipLaunchKernelGGL(pw_bwd_fold_kernel, dim3((unsigned)ud_cdiv(numel, 16)), dim3(256), 0, s, part, nparts, numel, dw, s1l, s2l, CE,
                       dgamma, dbeta);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int ud_pw_bwd_set_form(int form) {
    if (form < 0 || form > 4) return UD_EINVAL;
    g_pw_form = form;
    return 0;
}

int ud_pw_bwd_fused_ok(int CE, int CIN) { return (CE == 144 && CIN == 24) || (CE == 192 && CIN == 32); }

long ud_pw_bwd_fused_grid(long M) {
    if (M < 1) return UD_EINVAL;
    const long tiles = (M + R - 1) / R;
    return tiles < 512 ? tiles : 512;
}

int ud_pw_bwd_fused(const float* e, const float* dz, const ud_bn_ref* bn, const double* s1, const double* s2, const double* s1_local,
                    const double* s2_local, const float* x, const float* w, const float* add, long M, int CE, int CIN, float* dx,
                    float* dw, float* part, float* dgamma, float* dbeta, ud_stream_t stream) {
    if (!e || !dz || !bn || !s1 || !s2 || !x || !w || !dx || !dw || !part || M < 1 || !ud_pw_bwd_fused_ok(CE, CIN) || bn->G != 1 ||
        !bn->gamma)
        return UD_EINVAL;
    if ((dgamma || dbeta) && (!s1_local || !s2_local)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PwArgs a;
    a.e = e; a.dz = dz; a.x = x; a.w = w; a.add = add; a.dx = dx; a.part = part;
    a.bsum = bn->sum; a.bsumsq = bn->sumsq; a.gamma = bn->gamma; a.s1 = s1; a.s2 = s2;
    a.inv_count = bn->inv_count; a.eps = bn->eps; a.M = M; a.tiles = (M + R - 1) / R;
    const int grid = (int)ud_pw_bwd_fused_grid(M);
    const int form = g_pw_form ? g_pw_form : 1;
    int rc;
    if (CE == 144)
        rc = form == 1 ? launch<144, 24, 1, 2>(a, grid, s) : form == 2 ? launch<144, 24, 2, 2>(a, grid, s)
           : form == 3 ? launch<144, 24, 3, 1>(a, grid, s) : launch<144, 24, 4, 1>(a, grid, s);
    else
        rc = form == 1 ? launch<192, 32, 1, 2>(a, grid, s) : form == 2 ? launch<192, 32, 2, 2>(a, grid, s)
           : form == 3 ? launch<192, 32, 3, 1>(a, grid, s) : launch<192, 32, 4, 1>(a, grid, s);
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return pw::fold_launch(part, grid, CE * CIN, dw, s1_local, s2_local, CE, dgamma, dbeta, s);
}

}  // extern "C"
