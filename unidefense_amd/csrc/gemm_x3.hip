// fp32 GEMM on the gfx950 BF16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 pieces
//   x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (8 + 8 + 8 significand bits)
// and a*b is accumulated in fp32 from the six piece products of order <= 2^-16
//   a0b0 + (a0b1 + a1b0) + (a1b1 + a0b2 + a2b0);
// the dropped terms (a1b2, a2b1, a2b2) are below 2^-26 |ab|, i.e. under the rounding of an fp32 FMA chain.  Each
// piece product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result has fp32 GEMM
// accuracy (tests/test_a_kernels_gpu.py compares both kernels with float64) while six bf16 MFMAs (6 x 32 cycles per
// 32x32x16) replace eight fp32 MFMAs (8 x 64 cycles): 2.67x the fp32 matrix rate.
//
// Used by ud_gemm for the plain GEMMs (a_mode, b_mode in {0,1}) and for the implicit-GEMM conv gather on the A side
// (a_mode 2: forward / data gradient of the 3x3 and transposed convs) and on the B side (b_mode 2: their weight gradient);
// tiny / skinny / unaligned shapes stay on gemm.hip's v_mfma_f32_32x32x2_f32 kernel.
//
// Structure: 256 threads = 4 wave64, block tile BM x BN x 16, wave tile (TM x TN) x 32x32.  Operand tiles go
// global -> registers (3 K-tiles in flight) -> split -> LDS as three bf16 planes per operand, each plane
// [k-group h = 0,1][row][8 bf16 = 16 B], so that lane (r = l&31, h = l>>5) fetches its whole MFMA operand
// (row r, k = 8h..8h+7) with ONE conflict-free ds_read_b128.  Two LDS stages, one barrier per K-tile.
#include "gemm_internal.h"
#include "ud_common.h"

#include <type_traits>

namespace {

constexpr int BK = 16;
constexpr int NTHREADS = 256;
constexpr int PD = 3;                    // K-tiles of global loads in flight per thread (register ring)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (lo = bf16(x), hi = bf16(y)), round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    bf16x2 v = {(__bf16)x, (__bf16)y};
    return __builtin_bit_cast(uint32_t, v);
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// (lo = fp16(x), hi = fp16(y)), round-to-nearest-even
__device__ __forceinline__ uint32_t pack_f16(float x, float y) {
    f16x2 v = {(_Float16)x, (_Float16)y};
    return __builtin_bit_cast(uint32_t, v);
}

// exact three-way split of two floats; p[i] packs piece i of (x, y)
__device__ __forceinline__ void split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(sx, sy);
}

// issue-order pipeline for one K-tile: (1 MFMA, 6 VALU, NW/NM LDS writes) x NM   (sched_group_barrier wants
// literal constants, hence the recursion)
template <int N, int NM, int NW>
struct SchedPipe {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        constexpr int W = (N + 1) * NW / NM - N * NW / NM;
        if constexpr (W > 0) __builtin_amdgcn_sched_group_barrier(0x200, W, 0);
        SchedPipe<N + 1, NM, NW>::run();
    }
};
template <int NM, int NW>
struct SchedPipe<NM, NM, NW> {
    static __device__ __forceinline__ void run() {}
};

// rows 8..15 of every 16 swap neighbours: keeps the 32-lane operand reads contiguous and spreads the 4-byte
// transposing stores of the row-contiguous operands over all banks
__device__ __forceinline__ int phys_row(int row) { return row ^ ((row >> 3) & 1); }

// ---------------------------------------------------------------------------------------------------------
// Operand tile: ROWS x 16.  MODE 0: source [row][k], K-contiguous.  MODE 1: source [k][row], row-contiguous.
// MODE 2: conv gather with rows = output pixels, k = (tap, ci) (K-contiguous inside a tap).  MODE 3: conv gather with
// k = output pixels and rows = (tap, ci) — the B operand of a conv's weight gradient: row-contiguous like MODE 1, the
// thread's (tap, ci) fixed for the whole kernel, the pixel of its two k's walked incrementally.
// Loads are branch-free (addresses clamped into the operand, validity re-derived at store time): a load under a
// branch made hipcc wait for it on the spot, which serialised the whole prefetch ring.
// ---------------------------------------------------------------------------------------------------------
// PREC 3: the exact three-way bf16 split (fp32 GEMM accuracy).
// PREC 1: ONE fp16 piece per operand (round to nearest even): the mixed-precision mode of BASELINE configs[4] — fp16 MFMA
//         operands, fp32 accumulation, one v_mfma_f32_32x32x16_f16 instead of six bf16 MFMAs.
// (Measured in round 2 with the split's arithmetic compiled out: it costs 4-8 % for B alone, 10-12 % for both operands —
//  not the limiter.)
// H: the operand is stored as _Float16 (half storage of the activations, PREC 1 only): the same element-to-thread map
// with 8-byte / 4-byte loads, and the store is a masked copy (K-contiguous) or a pair interleave (row-contiguous).
template <int ROWS, int MODE, int PREC = 3, bool H = false>
struct XLoader {
    static_assert(!H || PREC == 1, "half operands feed the fp16 MFMA directly");
    static constexpr int NPL = PREC == 1 ? 1 : 3;      // bf16 / fp16 planes per operand
    static constexpr int GS = ROWS * 16 + 32;          // bytes of one (plane, k-group) image
    static constexpr int STAGE = 2 * NPL * GS;         // planes x 2 k-groups
    static constexpr int NV0 = ROWS / 64;              // MODE 0: float4 (4 k of one row) per thread
    static constexpr int VEC = ROWS / 32;              // MODE 1: rows per thread (x 2 consecutive k)
    static constexpr bool KC = MODE == 0 || MODE == 2;                  // K-contiguous source (MODE 2: inside a tap)
    static constexpr int NV = KC ? NV0 : 2;
    static constexpr int NWRITE = KC ? NPL * NV0 : NPL * VEC;           // ds_write instructions per stage
    using E = typename std::conditional<H, _Float16, float>::type;
    using VF = typename std::conditional<(KC || VEC == 4), f32x4, f32x2>::type;
    using VH = typename std::conditional<(KC || VEC == 4), f16x4, f16x2>::type;
    using V = typename std::conditional<H, VH, VF>::type;

    const E* base;
    long ld;
    int k_last;           // last valid k of this workgroup's K range
    long off[NV0 > 2 ? NV0 : 2];   // MODE 0: element offset of (clamped row, kq*4); MODE 1: [0] = clamped row offset
    bool rowok[NV0 > 2 ? NV0 : 2];
    int kloc;             // this thread's k offset inside a tile: MODE 0 / 2: 4*kq; MODE 1: 2*kb
    V regs[PD][NV];
    // MODE 2 (implicit-GEMM conv gather, rows = output pixels, k = (tap, ci); Cin % 4 == 0 so a float4 never straddles
    // taps): the (tap, ci) of the thread's NEXT load, advanced incrementally (no division in the loop), the rows'
    // pixel origins, and one validity bit per (ring slot, row) for the store
    // MODE 3: g_kh / g_kw / g_ci = the thread's fixed tap and first channel, (g_img, g_oh, g_ow) = output pixel of g_k
    ud_conv_geom g;
    int g_k, g_ci, g_kh, g_kw;
    int g_img, g_oh, g_ow;
    int g_nbase[NV0 > 2 ? NV0 : 2], g_ih0[NV0 > 2 ? NV0 : 2], g_iw0[NV0 > 2 ? NV0 : 2];
    unsigned vmask[PD];
    // FAST path (MODE 0 / 1, a workgroup whose rows are all inside the operand and whose K range is whole tiles): the
    // address of a load is a UNIFORM pointer (operand base + this tile's k, in scalar registers) plus a per-thread 32-bit
    // element offset fixed for the whole kernel — no per-load vector arithmetic (the general path pays a clamp, a 64-bit
    // add and, for the row-contiguous operand, a quarter-rate 32-bit multiply per load) and no validity selects at store
    // time (one v_cndmask per element there).
    unsigned voff[NV];

    // can this workgroup take the FAST path for this operand?  (uniform; voff is a BYTE offset below 2^31)
    __device__ __forceinline__ static bool fast_ok(long ld_, int dim, int row0, int k_begin, int k_end) {
        if constexpr (MODE > 1) return false;
        return row0 + ROWS <= dim && (k_end - k_begin) % BK == 0 &&
               (MODE == 0 ? (long)dim * ld_ : 16 * ld_ + dim) * (long)sizeof(E) < (1L << 31);
    }

    __device__ __forceinline__ void init_fast(int row0, int tid) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NV0; ++i)
                voff[i] = (unsigned)(((row0 + ((tid + i * NTHREADS) >> 2)) * ld + (tid & 3) * 4) * (long)sizeof(E));
        } else if constexpr (MODE == 1) {
            const int q = (tid >> 5) * 4 + (tid & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                voff[i] = (unsigned)(((2 * ((tid >> 2) & 7) + i) * ld + row0 + q * VEC) * (long)sizeof(E));
        }
    }

    template <int S>
    __device__ __forceinline__ void load_fast(int k0) {
        const char* ub = reinterpret_cast<const char*>(MODE == 0 ? base + k0 : base + (long)k0 * ld);     // uniform
#pragma unroll
        for (int i = 0; i < NV; ++i) regs[S][i] = *reinterpret_cast<const V*>(ub + voff[i]);
    }

    __device__ __forceinline__ void init(const float* p, long ld_, int dim, int row0, int k_end, int tid,
                                         const ud_conv_geom& geom, int k_begin) {
        base = reinterpret_cast<const E*>(p); ld = ld_; k_last = k_end - 1;
        if constexpr (MODE == 2) {
            g = geom;
            kloc = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < NV0; ++i) {
                const int f = tid + i * NTHREADS;
                const int m = row0 + (f >> 2);
                rowok[i] = m < dim;
                const int mm = rowok[i] ? m : 0;
                const int ow = mm % g.Wout, t = mm / g.Wout;
                const int oh = t % g.Hout, n = t / g.Hout;
                g_nbase[i] = n * g.Hin * g.Win;
                g_ih0[i] = g.transposed ? oh + g.pad_t : oh * g.stride - g.pad_t;
                g_iw0[i] = g.transposed ? ow + g.pad_l : ow * g.stride - g.pad_l;
            }
            g_k = k_begin + kloc;
            const int tap = g_k / g.Cin;
            g_ci = g_k - tap * g.Cin;
            g_kh = tap / g.KW;
            g_kw = tap - g_kh * g.KW;
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NV0; ++i) {
                int f = tid + i * NTHREADS;
                int r = row0 + (f >> 2);
                rowok[i] = r < dim;
                off[i] = (long)(rowok[i] ? r : 0) * ld;
            }
            kloc = (tid & 3) * 4;          // (tid + i*256) & 3 == tid & 3
        } else if constexpr (MODE == 3) {
            g = geom;
            const int q = (tid >> 5) * 4 + (tid & 3);
            const int r = row0 + q * VEC;              // first of the thread's VEC columns (tap, ci .. ci + VEC - 1)
            rowok[0] = r < dim;
            const int rr = rowok[0] ? r : 0;
            const int tap = rr / g.Cin;
            g_ci = rr - tap * g.Cin;
            g_kh = tap / g.KW;
            g_kw = tap - g_kh * g.KW;
            kloc = 2 * ((tid >> 2) & 7);
            g_k = k_begin + kloc;
            g_ow = g_k % g.Wout;
            const int t = g_k / g.Wout;
            g_oh = t % g.Hout;
            g_img = t / g.Hout;
        } else {
            const int q = (tid >> 5) * 4 + (tid & 3);
            int r = row0 + q * VEC;
            rowok[0] = r < dim;
            off[0] = rowok[0] ? r : 0;
            kloc = 2 * ((tid >> 2) & 7);
        }
    }

    // tile whose first k is k0 (k0 < k_end; k0 % 16 == 0)
    template <int S>
    __device__ __forceinline__ void load(int k0) {
        if constexpr (MODE == 2) {
            // advance (tap, ci) to this tile's k (tiles are requested in non-decreasing order; the surplus prefetches
            // past the end repeat the last one)
            int delta = k0 + kloc - g_k;
            g_k += delta;
            g_ci += delta;
            while (g_ci >= g.Cin) {
                g_ci -= g.Cin;
                if (++g_kw == g.KW) { g_kw = 0; ++g_kh; }
            }
            const bool kok = g_k <= k_last;
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < NV0; ++i) {
                int ih, iw;
                bool ok;
                if (!g.transposed) {
                    ih = g_ih0[i] + g_kh;
                    iw = g_iw0[i] + g_kw;
                    ok = (ih >= 0) && (ih < g.Hin) && (iw >= 0) && (iw < g.Win);
                } else {
                    const int th = g_ih0[i] - g_kh, tw = g_iw0[i] - g_kw;
                    ok = (th >= 0) && (tw >= 0) && (th % g.stride == 0) && (tw % g.stride == 0);
                    ih = th / g.stride;
                    iw = tw / g.stride;
                    ok = ok && (ih < g.Hin) && (iw < g.Win);
                }
                ok = ok && kok && rowok[i];
                const long o = ok ? ((long)g_nbase[i] + (long)ih * g.Win + iw) * (long)g.Cin + g_ci : 0;   // branch-free
                regs[S][i] = *reinterpret_cast<const V*>(base + o);
                mask |= (ok ? 1u : 0u) << i;
            }
            vmask[S] = mask;
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NV0; ++i) {
                // K % 4 == 0: the float4 at k0 + kq4 is entirely inside or entirely outside [.., k_end); an outside
                // one is redirected to the last inside one (k_last - 3) and zeroed at store time
                regs[S][i] = *reinterpret_cast<const V*>(base + off[i] + min(k0 + kloc, k_last - 3));
            }
        } else if constexpr (MODE == 3) {
            // walk the pixel to this tile's k (non-decreasing; surplus prefetches past the end repeat the last tile)
            const int delta = k0 + kloc - g_k;
            g_k += delta;
            g_ow += delta;
            while (g_ow >= g.Wout) {
                g_ow -= g.Wout;
                if (++g_oh == g.Hout) { g_oh = 0; ++g_img; }
            }
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ow = g_ow + i, oh = g_oh, img = g_img;
                if (ow >= g.Wout) { ow -= g.Wout; if (++oh == g.Hout) { oh = 0; ++img; } }
                int ih, iw;
                bool ok;
                if (!g.transposed) {
                    ih = oh * g.stride - g.pad_t + g_kh;
                    iw = ow * g.stride - g.pad_l + g_kw;
                    ok = (ih >= 0) && (ih < g.Hin) && (iw >= 0) && (iw < g.Win);
                } else {
                    const int th = oh + g.pad_t - g_kh, tw = ow + g.pad_l - g_kw;
                    ok = (th >= 0) && (tw >= 0) && (th % g.stride == 0) && (tw % g.stride == 0);
                    ih = th / g.stride;
                    iw = tw / g.stride;
                    ok = ok && (ih < g.Hin) && (iw < g.Win);
                }
                ok = ok && rowok[0] && (g_k + i <= k_last);
                const long o = ok ? (((long)img * g.Hin + ih) * g.Win + iw) * (long)g.Cin + g_ci : 0;     // branch-free
                regs[S][i] = *reinterpret_cast<const V*>(base + o);
                mask |= (ok ? 1u : 0u) << i;
            }
            vmask[S] = mask;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int k = min(k0 + kloc + i, k_last);
                regs[S][i] = *reinterpret_cast<const V*>(base + (long)k * ld + off[0]);
            }
        }
    }

    template <int S, bool FAST = false>
    __device__ __forceinline__ void store(char* L, int tid, int k0) const {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NV0; ++i) {
                int f = tid + i * NTHREADS;
                int row = f >> 2, kq = f & 3;
                const bool ok = FAST ? true : MODE == 2 ? ((vmask[S] >> i) & 1u) != 0 : rowok[i] && (k0 + kq * 4 <= k_last);
                V v = regs[S][i];
                char* p = L + (kq >> 1) * GS + phys_row(row) * 16 + (kq & 1) * 8;
                if constexpr (H) {
                    const u32x2 raw = __builtin_bit_cast(u32x2, v);
                    *reinterpret_cast<u32x2*>(p) = ok ? raw : u32x2{0u, 0u};
                } else if constexpr (PREC == 1) {
                    *reinterpret_cast<u32x2*>(p) = u32x2{pack_f16(ok ? v[0] : 0.f, ok ? v[1] : 0.f),
                                                         pack_f16(ok ? v[2] : 0.f, ok ? v[3] : 0.f)};
                } else {
                    uint32_t a0, a1, a2, b0, b1, b2;
                    split2(ok ? v[0] : 0.f, ok ? v[1] : 0.f, a0, a1, a2);
                    split2(ok ? v[2] : 0.f, ok ? v[3] : 0.f, b0, b1, b2);
                    *reinterpret_cast<u32x2*>(p) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(p + 2 * GS) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(p + 4 * GS) = u32x2{a2, b2};
                }
            }
        } else {
            const int q = (tid >> 5) * 4 + (tid & 3), kb = (tid >> 2) & 7;
            const bool ok0 = FAST ? true : MODE == 3 ? (vmask[S] & 1u) != 0 : rowok[0] && (k0 + kloc <= k_last);
            const bool ok1 = FAST ? true : MODE == 3 ? (vmask[S] & 2u) != 0 : rowok[0] && (k0 + kloc + 1 <= k_last);
            char* p0 = L + (kb >> 2) * GS + (kb & 3) * 4;
            V v0 = regs[S][0], v1 = regs[S][1];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                char* p = p0 + phys_row(q * VEC + e) * 16;
                if constexpr (H) {
                    const f16x2 pr = {ok0 ? v0[e] : (_Float16)0, ok1 ? v1[e] : (_Float16)0};
                    *reinterpret_cast<uint32_t*>(p) = __builtin_bit_cast(uint32_t, pr);
                } else if constexpr (PREC == 1) {
                    *reinterpret_cast<uint32_t*>(p) = pack_f16(ok0 ? v0[e] : 0.f, ok1 ? v1[e] : 0.f);
                } else {
                    uint32_t a0, a1, a2;
                    split2(ok0 ? v0[e] : 0.f, ok1 ? v1[e] : 0.f, a0, a1, a2);
                    *reinterpret_cast<uint32_t*>(p) = a0;
                    *reinterpret_cast<uint32_t*>(p + 2 * GS) = a1;
                    *reinterpret_cast<uint32_t*>(p + 4 * GS) = a2;
                }
            }
        }
    }
};

// resident waves per SIMD the register budget must allow: the small tiles are chosen where a launch is latency-bound, and
// live on occupancy (with the general and the fast k-loop in one kernel hipcc otherwise spends up to 205 VGPRs on the
// 128x64 tile: two waves per SIMD instead of three, -6 % on the UDR18 step)
template <int BM, int BN>
constexpr int kWavesPerSimd = BM * BN >= 128 * 128 ? 2 : BM * BN >= 128 * 64 ? 3 : 4;

template <int BM, int BN, int WGM, int WGN, int AMODE, int BMODE, int PREC = 3, bool AH = false, bool BH = false>
__global__ __launch_bounds__(NTHREADS, (kWavesPerSimd<BM, BN>)) void gemm_x3_kernel(const ud_gemm_desc d, int tiles_m,
                                                                                   int tiles_n) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int NPL = PREC == 1 ? 1 : 3;
    using LA = XLoader<BM, AMODE, PREC, AH>;
    using LB = XLoader<BN, BMODE, PREC, BH>;
    __shared__ __attribute__((aligned(16))) char As[2][LA::STAGE];
    __shared__ __attribute__((aligned(16))) char Bs[2][LB::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), each with its own 4 MiB L2.  With
    // tile_cfg bit 8 every XCD gets one CONTIGUOUS range of the (column-major) tile order instead of every eighth tile, so
    // that the A / B panels its workgroups share are fetched into ONE L2 (a bijection for any tile count: no padding).
    int bt = blockIdx.x;
    if (d.tile_cfg & 0x100) {
        const int T = tiles_m * tiles_n, q = T >> 3, r = T & 7, x = bt & 7;
        bt = x * q + (x < r ? x : r) + (bt >> 3);
    }
    const int split = blockIdx.y, bz = blockIdx.z;
    const int kt_total = (d.K + BK - 1) / BK;
    const int kt_per = (kt_total + d.split_k - 1) / d.split_k;
    const int k_begin = split * kt_per * BK;
    int k_end = k_begin + kt_per * BK;
    if (k_end > d.K) k_end = d.K;
    const int tile_m = bt % tiles_m, tile_n = bt / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
    float* Cp = d.C + (long)bz * d.strideC + (d.out_mode == 3 ? (long)split * d.slice_stride : 0L);   // mode 3: own slice

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto kloop = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        LA la; LB lb;
        la.init(d.A + (long)bz * d.strideA, d.lda, d.M, m0, k_end, tid, d.g, k_begin);
        lb.init(d.B + (long)bz * d.strideB, d.ldb, d.N, n0, k_end, tid, d.g, k_begin);
        if constexpr (FAST) { la.init_fast(m0, tid); lb.init_fast(n0, tid); }
        const int kt_max = nkt - 1;
        auto k0_of = [&](int kt) { return k_begin + min(kt, kt_max) * BK; };   // clamped: surplus prefetches re-read the last tile
        auto ld_a = [&](auto slot_c, int k0) {
            constexpr int S = decltype(slot_c)::value;
            if constexpr (FAST) la.template load_fast<S>(k0); else la.template load<S>(k0);
        };
        auto ld_b = [&](auto slot_c, int k0) {
            constexpr int S = decltype(slot_c)::value;
            if constexpr (FAST) lb.template load_fast<S>(k0); else lb.template load<S>(k0);
        };
        using std::integral_constant;
        using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>; using I2 = integral_constant<int, 2>;

        // prologue: K-tiles 0..PD-1 in flight, tile 0 into LDS stage 0, then tile PD into the freed slot
        ld_a(I0{}, k0_of(0)); ld_b(I0{}, k0_of(0));
        ld_a(I1{}, k0_of(1)); ld_b(I1{}, k0_of(1));
        ld_a(I2{}, k0_of(2)); ld_b(I2{}, k0_of(2));
        la.template store<0, FAST>(As[0], tid, k0_of(0));
        lb.template store<0, FAST>(Bs[0], tid, k0_of(0));
        ld_a(I0{}, k0_of(PD)); ld_b(I0{}, k0_of(PD));
        __syncthreads();

        const int pr = phys_row(l31);
        const int a_off = half * LA::GS + (wm * TM * 32 + pr) * 16;
        const int b_off = half * LB::GS + (wn * TN * 32 + pr) * 16;
        constexpr int NM = TM * TN * (PREC == 1 ? 1 : 6);  // MFMAs per K-tile and wave
        constexpr int NW = LA::NWRITE + LB::NWRITE;        // LDS writes per K-tile and thread

        // one K-tile: operands of tile kt from LDS stage kt&1 -> NM MFMAs; meanwhile (STAGE_NEXT) the register
        // slot S holding tile kt+1 is split and written to the other LDS stage, then refilled with tile kt+1+PD
        auto stage = [&](auto slot_c, auto stage_next_c, int kt) {
            constexpr int S = decltype(slot_c)::value;
            constexpr bool STAGE_NEXT = decltype(stage_next_c)::value;
            const char* Ab = As[kt & 1] + a_off;
            const char* Bb = Bs[kt & 1] + b_off;
            bf16x8 fa[TM][NPL], fb[TN][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i][p] = *reinterpret_cast<const bf16x8*>(Ab + 2 * p * LA::GS + i * 512);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j][p] = *reinterpret_cast<const bf16x8*>(Bb + 2 * p * LB::GS + j * 512);
            }
            if constexpr (STAGE_NEXT) {
                la.template store<S, FAST>(As[(kt + 1) & 1], tid, k0_of(kt + 1));
                lb.template store<S, FAST>(Bs[(kt + 1) & 1], tid, k0_of(kt + 1));
            }
            // smallest terms first into the running sum
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (PREC == 1) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][0]),
                                                                   __builtin_bit_cast(f16x8, fb[j][0]), c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    }
                    acc[i][j] = c;
                }
            if constexpr (STAGE_NEXT) {
                // issue order: the split / LDS-write work rides in the issue slots between the MFMAs (an MFMA
                // holds the vector issue port for 8 of its 32 cycles): 1 MFMA, 6 VALU, NW/NM LDS writes, repeat
                SchedPipe<0, NM, NW>::run();
                ld_a(slot_c, k0_of(kt + 1 + PD));
                ld_b(slot_c, k0_of(kt + 1 + PD));
            }
            __syncthreads();
        };
        int kt = 0;
        for (; kt + PD <= kt_max; kt += PD) {          // steady state: branch-free body, register slots static
            stage(integral_constant<int, 1>{}, integral_constant<bool, true>{}, kt);
            stage(integral_constant<int, 2>{}, integral_constant<bool, true>{}, kt + 1);
            stage(integral_constant<int, 0>{}, integral_constant<bool, true>{}, kt + 2);
        }
        // tail: kt is a multiple of PD, at most PD-1 staging iterations and the final compute-only one remain
        if (kt < kt_max) { stage(integral_constant<int, 1>{}, integral_constant<bool, true>{}, kt); ++kt; }
        if (kt < kt_max) { stage(integral_constant<int, 2>{}, integral_constant<bool, true>{}, kt); ++kt; }
        stage(integral_constant<int, 0>{}, integral_constant<bool, false>{}, kt);
    };
    if (nkt > 0) {
        // uniform: every row of both tiles inside the operands, whole K-tiles, 32-bit element offsets
        const bool fast = AMODE <= 1 && BMODE <= 1 &&
                          LA::fast_ok(d.lda, d.M, m0, k_begin, k_end) && LB::fast_ok(d.ldb, d.N, n0, k_begin, k_end);
        if (fast) kloop(std::integral_constant<bool, true>{});
        else kloop(std::integral_constant<bool, false>{});
    }

    // epilogue: D[i][j], j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (nkt == 0 && (d.out_mode == 1 || d.out_mode == 2)) return;          // (modes 0 / 3 store the zeros)
    const bool c_half = (d.half_mask & 4) != 0;
    if (c_half) {          // statistics and consumers see the rounded values
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = ud_rounded<_Float16>(acc[i][j][r]);
    }
    if (d.stat_sum) {
        // per-column sums of this wave's TM*32 rows (rows beyond M hold exact zeros: their A rows were zero-filled), the
        // two lane halves (different rows, same column) folded by one shuffle, one fp64 atomic per column and quantity;
        // more than 64 row tiles: slot (tile_m % 64) of [64][N], so that at most ~2 * tiles/64 adds queue per address
        const long slot = (tiles_m > 64) ? (long)(tile_m & 63) * d.N : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = (double)acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int col = n0 + wn * (TN * 32) + j * 32 + l31;
            if (half == 0 && col < d.N) {
                unsafeAtomicAdd(d.stat_sum + slot + col, s1);
                unsafeAtomicAdd(d.stat_sumsq + slot + col, s2);
            }
        }
    }
    // The output form and "whole tile inside C" are decided ONCE per workgroup (descriptor fields as values, a row pointer
    // advanced by constants): the form decided per element cost ~40 scalar instructions and ten branches per stored value —
    // a third of a thin-K workgroup's life.
    const int dM = d.M, dN = d.N;
    const long ldc = d.ldc;
    const int mode = d.out_mode == 3 ? 0 : d.out_mode;
    auto store_all = [&](auto mode_c, auto inside_c) {
        constexpr int MODE = decltype(mode_c)::value;          // 0 store, 1 +=, 2 atomic, 4 half store, 5 half +=
        constexpr bool INSIDE = decltype(inside_c)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (TN * 32) + j * 32 + l31;
                const int row0 = m0 + wm * (TM * 32) + i * 32 + 4 * half;
                const long off0 = (long)row0 * ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (INSIDE || (row0 + dr < dM && col < dN)) {
                        const float v = acc[i][j][r];
                        if constexpr (MODE >= 4) {
                            _Float16* p = reinterpret_cast<_Float16*>(Cp) + off0 + (long)dr * ldc;
                            *p = (_Float16)(MODE == 4 ? v : v + (float)*p);
                        } else {
                            float* p = Cp + off0 + (long)dr * ldc;
                            if constexpr (MODE == 0) *p = v;
                            else if constexpr (MODE == 1) *p += v;
                            else atomicAdd(p, v);
                        }
                    }
                }
            }
        }
    };
    using std::integral_constant;
    auto store_mode = [&](auto inside_c) {
        if (c_half) {          // out_mode 0 / 1 only (ud_gemm rejects atomics onto half)
            if (mode == 0) store_all(integral_constant<int, 4>{}, inside_c);
            else store_all(integral_constant<int, 5>{}, inside_c);
        } else if (mode == 0) store_all(integral_constant<int, 0>{}, inside_c);
        else if (mode == 1) store_all(integral_constant<int, 1>{}, inside_c);
        else store_all(integral_constant<int, 2>{}, inside_c);
    };
    if (m0 + BM <= dM && n0 + BN <= dN) store_mode(integral_constant<bool, true>{});
    else store_mode(integral_constant<bool, false>{});
}

template <int BM, int BN, int AMODE, int BMODE, int PREC, bool AH = false, bool BH = false>
int launch_tile(const ud_gemm_desc& d, hipStream_t s) {
    int tiles_m = ud_cdiv(d.M, BM), tiles_n = ud_cdiv(d.N, BN);
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)d.split_k, (unsigned)d.batch);
    hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, 2, 2, AMODE, BMODE, PREC, AH, BH>), grid, dim3(NTHREADS), 0, s, d, tiles_m,
                       tiles_n);
    UD_LAUNCH_CHECK();
    return 0;
}

struct XCfg { int bm, bn; double penalty; };
// Half-size tiles split every A (or B) row twice as often: 19-23 % more time per flop (tools/check_gemm_paths.py
// with UD_GEMM_X3_CFG=0/1/2: 4096^3 179 vs 150 TFLOP/s), so they only win where they fill the chip better.
constexpr int NXCFG = 4;
constexpr XCfg kX[NXCFG] = {{128, 128, 1.00}, {128, 64, 1.20}, {64, 128, 1.20}, {64, 64, 1.60}};

// tile configuration of a descriptor: d.tile_cfg (1 + index, set by the caller's per-shape tuner) > UD_GEMM_X3_CFG > the
// cost model over the first three (the 64x64 tile is only ever chosen by measurement)
int pick_cfg(const ud_gemm_desc& d) {
    const int want = d.tile_cfg & 0xff;          // bit 8: XCD-contiguous tile order (gemm_x3_kernel)
    if (want >= 1 && want <= NXCFG) return want - 1;
    static const int forced = [] {
        const char* e = getenv("UD_GEMM_X3_CFG");
        return (e && e[0] >= '0' && e[0] < '0' + NXCFG) ? e[0] - '0' : -1;
    }();
    if (forced >= 0) return forced;
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 3; ++i) {
        long tiles = (long)ud_cdiv(d.M, kX[i].bm) * ud_cdiv(d.N, kX[i].bn) * d.split_k * d.batch;
        // a CU's matrix pipe is shared by its resident workgroups, so a launch lasts about ceil(tiles / 256 CUs)
        // single-workgroup tile times whether the workgroups of a CU run side by side or one after the other
        long rounds = (tiles + 255) / 256;
        double cost = (double)rounds * kX[i].bm * kX[i].bn * kX[i].penalty;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

template <int AMODE, int BMODE, int PREC, bool AH = false, bool BH = false>
int launch_modes(const ud_gemm_desc& d, hipStream_t s) {
    switch (pick_cfg(d)) {
        case 1: return launch_tile<128, 64, AMODE, BMODE, PREC, AH, BH>(d, s);
        case 2: return launch_tile<64, 128, AMODE, BMODE, PREC, AH, BH>(d, s);
        case 3: return launch_tile<64, 64, AMODE, BMODE, PREC, AH, BH>(d, s);
        default: return launch_tile<128, 128, AMODE, BMODE, PREC, AH, BH>(d, s);
    }
}

}  // namespace

bool ud_gemm_x3_eligible(const ud_gemm_desc& d, bool a_vec, bool b_vec) {
    if (d.a_mode == 2 && d.b_mode == 0) {
        // implicit-GEMM conv (forward and data gradient of the 3x3 / transposed convs): Cin % 4 == 0 (a_vec), batch 1
        return a_vec && b_vec && d.K % 4 == 0 && d.batch == 1;
    }
    if (d.a_mode == 1 && d.b_mode == 2) {
        // weight gradient of a conv: dY^T (row-contiguous A) x gathered input (B rows = (tap, ci)); Cin % 4 == 0 (b_vec)
        return a_vec && b_vec && d.M % 4 == 0 && d.batch == 1;
    }
    if (d.a_mode > 1 || d.b_mode > 1) return false;
    if (!(d.a_mode == 0 && d.b_mode == 0) && !(d.a_mode == 0 && d.b_mode == 1) && !(d.a_mode == 1 && d.b_mode == 1))
        return false;
    if (!a_vec || !b_vec) return false;
    if ((d.a_mode == 0 || d.b_mode == 0) && d.K % 4 != 0) return false;
    // row-contiguous A ([K][lda], rows of the product along the contiguous index): the loads are 2 / 4 consecutive m wide, so M is
    // a multiple of 4 — or the rows carry slack up to the next multiple (lda >= ceil4(M): the DFT matrices of kernels.py, padded
    // with zeros by their maker); what the slack holds only reaches product rows >= M, which are not stored
    if (d.a_mode == 1 && d.M % 4 != 0 && d.lda < ((d.M + 3) & ~3)) return false;
    if (d.b_mode == 1 && d.N % 4 != 0) return false;
    return true;
}

// rows per tile of the configuration launch_modes picks (for the epilogue-statistics slot rule)
int ud_gemm_x3_tile_rows(const ud_gemm_desc& d) { return kX[pick_cfg(d)].bm; }

// Half-stored operands (ud_gemm_desc.half_mask): activations / activation gradients are _Float16 in memory, weights and
// weight gradients fp32 — the three products of a 1x1 conv: forward (0,0) and data gradient (0,1) with A half, weight
// gradient (1,1) with both half.  Always the fp16 MFMA.
int ud_gemm_x3_launch_half(const ud_gemm_desc& d, hipStream_t s) {
    const int ab = d.half_mask & 3;
    if (d.a_mode == 0 && d.b_mode == 0 && ab == 1) return launch_modes<0, 0, 1, true, false>(d, s);
    if (d.a_mode == 0 && d.b_mode == 1 && ab == 1) return launch_modes<0, 1, 1, true, false>(d, s);
    if (d.a_mode == 1 && d.b_mode == 1 && ab == 3) return launch_modes<1, 1, 1, true, true>(d, s);
    if (ab == 0) return ud_gemm_x3_launch(d, s, true);          // fp32 operands, half result
    return UD_EINVAL;
}

// f16: one fp16 piece per operand (mixed precision) instead of the exact three-way bf16 split
int ud_gemm_x3_launch(const ud_gemm_desc& d, hipStream_t s, bool f16) {
    if (d.a_mode == 2 && d.b_mode == 0) return f16 ? launch_modes<2, 0, 1>(d, s) : launch_modes<2, 0, 3>(d, s);
    if (d.a_mode == 1 && d.b_mode == 2) return f16 ? launch_modes<1, 3, 1>(d, s) : launch_modes<1, 3, 3>(d, s);
    if (f16) {
        if (d.a_mode == 0 && d.b_mode == 0) return launch_modes<0, 0, 1>(d, s);
        if (d.a_mode == 0 && d.b_mode == 1) return launch_modes<0, 1, 1>(d, s);
        if (d.a_mode == 1 && d.b_mode == 1) return launch_modes<1, 1, 1>(d, s);
        return UD_EINVAL;
    }
    if (d.a_mode == 0 && d.b_mode == 0) return launch_modes<0, 0, 3>(d, s);
    if (d.a_mode == 0 && d.b_mode == 1) return launch_modes<0, 1, 3>(d, s);
    if (d.a_mode == 1 && d.b_mode == 1) return launch_modes<1, 1, 3>(d, s);
    return UD_EINVAL;
}
