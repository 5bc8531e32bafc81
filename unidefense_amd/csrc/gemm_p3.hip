// fp32-accurate GEMM on the gfx950 matrix pipe from PRE-SPLIT operands (round 4).
//
// gemm_x3.hip splits every fp32 operand tile into three bf16 pieces inside the k-loop: global -> VGPR -> ~160 vector
// instructions -> ds_write, per K-tile and wave, against 24 MFMAs.  Here the operands ARRIVE split: a matrix X[R][C] is
// stored by its producer (ud_split_planes*) as NPL 16-bit planes in the "P32" panel layout
//     plane p, panel c/32, row r, column c%32        at   p*plane + (c/32)*panel + r*32 + (c%32)        (16-bit elements)
// so that BOTH uses of the matrix see contiguous 1-KiB pieces of whole cache lines:
//     mode 0 (GEMM rows = rows of X, k = columns of X):  a 128 x 32 tile is ONE contiguous 8-KiB run of a panel;
//     mode 1 (GEMM rows = columns of X, k = rows of X):   a 32 x 128 tile is four 2-KiB runs (one per panel).
// Two precisions:
//   PREC 3: x = x0 + x1 + x2 in bf16 (the exact split of gemm_x3.hip), six piece products per 32x32x16 in gemm_x3's order —
//           bitwise the results of gemm_x3_kernel.
//   PREC 2: s*x = h0 + 2^-11 h1 in fp16 with a power-of-two scale s PER GEMM ROW (row maximum -> [2^14, 2^15)); the second piece
//           is stored scaled by 2^11, so that it is a NORMAL fp16 wherever the first is (the matrix pipe flushes fp16 subnormals:
//           unscaled, every element more than 2^-18 below its row's maximum lost its second piece — measured 1.4e-5 on
//           heavy-tailed rows).  Three products on v_mfma_f32_32x32x16_f16 into TWO fp32 accumulators per tile — a0b0, and
//           a1b0 + a0b1 — combined as (hi + 2^-11 lo) / (sa sb) in the epilogue.  h0 + 2^-11 h1 carries 22 significand bits of every
//           element within 2^-18 of the scale's maximum (below that the error is ABSOLUTE: 2^-40 of the maximum — a residual
//           under 2^-25 of the scaled value flushes), the dropped a1b1 is below 2^-24 |ab|: the error of a product is that of an
//           fp32 multiply, the sum is accumulated in fp32 as before — HALF the matrix-pipe work and two thirds of the operand
//           bytes of PREC 3, which is what counts on a part that holds its clock down under MFMA load (measured 1.5-1.6 GHz).
//
// The k-loop is LDS-DMA + ds_read + MFMA only.  Roles: waves 0-3 compute (one per SIMD), waves 4-7 only move data — the ISSUE of
// an LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave-instruction) costs the issuing wave 60-180 cycles while the CU's
// texture-address path is busy, which in a computing wave is time the matrix pipe idles (measured: +15-18 % with the DMA
// compiled out of a combined loop, +7 % from giving it to loader waves).  A ring of NSTAGE stages, counted vmcnt, ONE raw
// s_barrier per K-tile.  The mode-0 image is XOR-swizzled on the SOURCE address (the DMA destination is lane-linear) so that the
// ds_read_b128 operand fetch is conflict-free (SQ_LDS_BANK_CONFLICT = 0), the mode-1 image is read with ds_read_b64_tr_b16.
//
// Serves the spectral 1x1 convs of the SF blocks (model/efficientnet/exp.py:57 freq_conv: forward, data gradient, weight
// gradient) — the large launches of the step.
#include "gemm_internal.h"
#include "ud_common.h"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int NT = 512;                       // 4 MFMA waves (one per SIMD) + 4 loader waves
constexpr int BK = 32;
constexpr int BM = 128, BN = 128;
constexpr int PLANE_IMG = 128 * 64;           // bytes of one plane of one operand tile (128 rows x 32 16-bit elements)

template <int PREC> struct Cfg {
    static constexpr int NPL = PREC;                    // planes per operand
    static constexpr int OP_IMG = NPL * PLANE_IMG;      // one operand, all planes
    static constexpr int STAGE = 2 * OP_IMG;            // A then B
    // 144 KiB / 128 KiB of LDS; PREC 1: 4 stages of 16 KiB = 64 KiB, TWO workgroups per CU — 3 / 6 / 8 stages measured slower
    // (f16 bs-64 step 32.2 ms against 32.6 / 33.2 / 33.3: the deeper rings leave one workgroup per CU)
    static constexpr int NSTAGE = PREC == 3 ? 3 : 4;
    static constexpr int PT = 4 * NPL;                  // DMA pieces per tile and loader wave
    static constexpr int NTERM = PREC == 3 ? 6 : PREC == 2 ? 3 : 1;     // piece products per 32x32x16
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;

// One LDS-DMA piece: 64 lanes x 16 B from gbase + voff (per lane) to LDS byte address lds_dst + 16 * lane.
// Absent from hipcc's waitcnt bookkeeping: completion is counted by hand (vmcnt in issue order) — see the loader.
// No "memory" clobber: the statement is ordered against the wait / barrier statements (all volatile).
// s_nop 1 + the three scalar instructions = the 5 wait states between a scalar write of %2 / %3 and the load reading them.
__device__ __forceinline__ void dma_piece(unsigned voff, unsigned lds_dst, const char* gbase) {
    unsigned keep;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_dst), "s"(gbase));
}

// all but the N youngest DMA pieces of this wave have landed (and its LDS reads returned), then the workgroup barrier
template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// ... with N = tiles * PT chosen at run time (uniform)
template <int PT>
__device__ __forceinline__ void wait_tiles_and_barrier(int tiles) {
    if (tiles <= 0) wait_dma_and_barrier<0>();
    else if (tiles == 1) wait_dma_and_barrier<PT>();
    else if (tiles == 2) wait_dma_and_barrier<2 * PT>();
    else if (tiles == 3 || PT > 4) wait_dma_and_barrier<3 * PT>();          // (NSTAGE <= 4: never more than 2)
    else if (tiles == 4) wait_dma_and_barrier<4 * PT>();
    else if (tiles == 5) wait_dma_and_barrier<5 * PT>();
    else wait_dma_and_barrier<6 * PT>();
}
__device__ __forceinline__ void wait_lds_and_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Operand fragments of one 16-deep k-step: [row block 0/1][plane]
template <int NPL> struct Frags {
    s16x8 v[2][NPL];
};

// MODE 0 image of a plane: [row 128][64 B], 16-byte chunk c of row r stored at slot c ^ ((r >> 2) & 3).
// MODE 1 image of a plane: [panel 4][k 32][64 B = 32 GEMM rows], linear.
template <int MODE>
struct Reader {
    unsigned b0, b1;          // MODE 0: byte offsets of the lane's chunk for k-step 0 / 1 (row block 0, plane 0); MODE 1: b0 only

    __device__ __forceinline__ void init(int lane, int wrow) {          // wrow: first of the wave's 64 rows inside the tile
        if constexpr (MODE == 0) {
            const int r = lane & 31, h = lane >> 5, x = (r >> 2) & 3;
            b0 = (unsigned)((wrow + r) * 64 + ((h ^ x) << 4));
            b1 = (unsigned)((wrow + r) * 64 + (((2 + h) ^ x) << 4));
        } else {
            const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
            b0 = (unsigned)((wrow >> 5) * 2048 + (8 * (g >> 1) + q) * 64 + (g & 1) * 32 + pp * 8);
            b1 = 0;
        }
    }

    // fragment (row block I, plane PL) of k-step S
    template <int S, int I, int PL, int NPL>
    __device__ __forceinline__ void read_one(const lds_char* img, Frags<NPL>& f) const {
        if constexpr (MODE == 0) {
            const lds_char* p = img + (S == 0 ? b0 : b1) + I * 2048 + PL * PLANE_IMG;
            f.v[I][PL] = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(p);
        } else {
            const lds_char* p = img + b0 + S * 1024 + I * 2048 + PL * PLANE_IMG;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 256));
            f.v[I][PL] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }

    template <int S, int NPL>
    __device__ __forceinline__ void read(const lds_char* img, Frags<NPL>& f) const {
        read_one<S, 0, 0>(img, f); read_one<S, 1, 0>(img, f);
        if constexpr (NPL >= 2) { read_one<S, 0, 1>(img, f); read_one<S, 1, 1>(img, f); }
        if constexpr (NPL == 3) { read_one<S, 0, 2>(img, f); read_one<S, 1, 2>(img, f); }
    }
};

// per-lane source offset of a DMA piece (bytes from the piece's uniform base)
template <int MODE>
__device__ __forceinline__ unsigned dma_lane_offset(int lane) {
    if constexpr (MODE == 0) return (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    else return (unsigned)(lane * 16);
}

// planes of the TERM-th piece product, smallest first.  PREC 3 (gemm_x3.hip's order): a2b0, a0b2, a1b1, a1b0, a0b1, a0b0;
// PREC 2: a1b0, a0b1, a0b0
template <int PREC, int TERM> struct Term {
    static constexpr int pa = PREC == 3 ? (TERM == 0 ? 2 : (TERM == 1 || TERM >= 4) ? 0 : 1) : PREC == 2 ? (TERM == 0 ? 1 : 0) : 0;
    static constexpr int pb = PREC == 3 ? (TERM == 0 ? 0 : TERM == 1 ? 2 : (TERM == 2 || TERM == 4) ? 1 : 0) : PREC == 2 ? (TERM == 1 ? 1 : 0) : 0;
};

// MFMA number Mi of a k-step: product term Mi / 4 of accumulator Mi % 4 (the accumulators are interleaved, every one sees
// its terms in order).  PREC 2: the cross terms (a1b0, a0b1: pieces scaled by 2^11) go to `lo`, a0b0 to `acc`.
template <int PREC, int Mi>
__device__ __forceinline__ void mma_one(f32x16 (&acc)[2][2], f32x16 (&lo)[2][2], const Frags<PREC>& a, const Frags<PREC>& b) {
    constexpr int term = Mi / 4, i = (Mi % 4) / 2, j = Mi % 2;
    using Tm = Term<PREC, term>;
    if constexpr (PREC == 3) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.v[i][Tm::pa]),
                                                            __builtin_bit_cast(bf16x8, b.v[j][Tm::pb]), acc[i][j], 0, 0, 0);
    } else if constexpr (PREC == 1) {          // the first fp16 piece only: one product (mixed precision, BASELINE configs[4])
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v[i][0]),
                                                           __builtin_bit_cast(f16x8, b.v[j][0]), acc[i][j], 0, 0, 0);
    } else if constexpr (term < 2) {
        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v[i][Tm::pa]),
                                                          __builtin_bit_cast(f16x8, b.v[j][Tm::pb]), lo[i][j], 0, 0, 0);
    } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v[i][Tm::pa]),
                                                           __builtin_bit_cast(f16x8, b.v[j][Tm::pb]), acc[i][j], 0, 0, 0);
    }
}

// SK = false: grid (tiles, split_k); workgroup (tile, s) reduces K-tiles [s * kt_per, (s+1) * kt_per) — ud_gemm's launch forms
//             (out_mode 0 / 1 / 2 / 3, epilogue statistics).
// SK = true ("stream-K"): ONE workgroup per CU (the kernel owns most of the LDS, so a CU never holds two) and tile counts like 260
//             or 540 on 256 CUs waste up to half a round.  The (tile, K-tile) units are dealt evenly instead: grid G (a multiple
//             of 8, <= CUs), worker g takes units [g U / G, (g+1) U / G) of the tile-major order — the tail of one tile, whole tiles,
//             the head of another.  A segment that covers its tile's whole K range stores (out_mode 0) or adds (1); a partial one
//             adds atomically — C must be zero (out_mode 0) or hold the term to add to (1) before the launch.  Workers of one XCD
//             (blockIdx % 8) take CONSECUTIVE ranges and the tile order walks 8-wide column bands row by row.  Measured: it pays
//             only where the tile count sits just above a multiple of the CUs (260 tiles: 1.2x); elsewhere workers that share a
//             panel no longer stream the same k at the same time and the L2 hit rate falls (540 tiles: 0.95x).
// The kernel's body: workgroup bx of gx (and split-K slice by) of the problem d; L: the workgroup's NSTAGE * STAGE bytes of LDS.
// A device function so that gemm_p3_pair_kernel can run two problems' workgroups in one launch.
template <int PREC, int AMODE, int BMODE, bool SK>
__device__ __forceinline__ void p3_body(const ud_gemm_p3_desc& d, int tiles_m, int tiles_n, char* L, int bx, int by, int gx) {
    using CF = Cfg<PREC>;
    constexpr int NPL = CF::NPL, OP_IMG = CF::OP_IMG, STAGE = CF::STAGE, NSTAGE = CF::NSTAGE, PT = CF::PT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int kt_total = d.K / BK;
    // the descriptor fields the epilogue needs, as values: read through the lambda's reference to the kernel argument hipcc
    // re-loaded d.ldc from the argument segment (and waited for it) before EVERY one of a lane's 64 stores — 9 us per tile
    const int dM = d.M, dN = d.N;
    const long ldc = d.ldc;
    const bool c_half = d.c_half != 0;
    double* const stat_sum = d.stat_sum;
    double* const stat_sumsq = d.stat_sumsq;
    const float* const a_inv = d.a_inv_scale;
    const float* const b_inv = d.b_inv_scale;
    const long ast = d.a_scale_stride, bst = d.b_scale_stride;
    const lds_char* Lp = (const lds_char*)L;          // generic -> LDS address space: the low 32 bits are the LDS byte address
    const unsigned lds0 = (unsigned)(uintptr_t)Lp;
    const int lw = wave & 3;                                        // loader wave lw moves tile rows 32 lw .. 32 lw + 31
    const unsigned dst_w = lds0 + (unsigned)lw * 2048u;            // its pieces inside a plane image
    const long a_plane = d.a_plane * 2, b_plane = d.b_plane * 2;
    const unsigned a_voff = dma_lane_offset<AMODE>(lane), b_voff = dma_lane_offset<BMODE>(lane);
    Reader<AMODE> ra;
    Reader<BMODE> rb;
    ra.init(lane, wm * 64);
    rb.init(lane, wn * 64);
    using std::integral_constant;
    using H0 = integral_constant<int, 0>;
    using H1 = integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    // K-tiles [kt0, kt0 + nkt) of tile (tile_m, tile_n); ep: 0 store, 1 add, 2 atomic add; stats only with ep 0
    auto segment = [&](int tile_m, int tile_n, int kt0, int nkt, int ep, float* Cp, bool first) {
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        f32x16 acc[2][2], lo[PREC == 2 ? 2 : 1][2];          // lo: PREC 2's cross terms
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.f;
                    if constexpr (PREC == 2) lo[i][j][r] = 0.f;
                }

        if (nkt > 0) {
            if (wave >= 4) {
                // ---- loader.  Source of this wave: 32 tile rows (mode 0: rows 32 lw .. 32 lw + 31 = pieces 2 lw, 2 lw + 1; mode 1:
                // panel lw, both 16-deep halves of the K-tile); g(t, plane, jj) = g0 + t * step + plane * plane_bytes + jj * 1024
                const char* a_g0;
                const char* b_g0;
                long a_step, b_step;
                if constexpr (AMODE == 0) {
                    a_g0 = reinterpret_cast<const char*>(d.A) + (long)kt0 * d.a_panel * 2 + (long)(m0 + 32 * lw) * 64;
                    a_step = d.a_panel * 2;
                } else {
                    a_g0 = reinterpret_cast<const char*>(d.A) + (long)min(m0 / 32 + lw, d.a_npanel - 1) * d.a_panel * 2 +
                           (long)kt0 * BK * 64;
                    a_step = BK * 64;
                }
                if constexpr (BMODE == 0) {
                    b_g0 = reinterpret_cast<const char*>(d.B) + (long)kt0 * d.b_panel * 2 + (long)(n0 + 32 * lw) * 64;
                    b_step = d.b_panel * 2;
                } else {
                    b_g0 = reinterpret_cast<const char*>(d.B) + (long)min(n0 / 32 + lw, d.b_npanel - 1) * d.b_panel * 2 +
                           (long)kt0 * BK * 64;
                    b_step = BK * 64;
                }
                // tile t: the wave's 2 NPL pieces of A (plane, 16-row half), then those of B
                auto issue = [&](int t) {
                    const unsigned dst = dst_w + (unsigned)(t % NSTAGE) * STAGE;
                    const char* ga = a_g0 + (long)t * a_step;
                    const char* gb = b_g0 + (long)t * b_step;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            dma_piece(a_voff, dst + pl * PLANE_IMG + jj * 1024, ga + pl * a_plane + jj * 1024);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            dma_piece(b_voff, dst + OP_IMG + pl * PLANE_IMG + jj * 1024, gb + pl * b_plane + jj * 1024);
                };
                // One workgroup barrier per K-tile is the whole protocol:
                //   barrier(t):  loaders arrive after THEIR pieces of tile t+1 have landed (counted vmcnt; later tiles stay in
                //                flight), MFMA waves after their last read of stage t % NSTAGE (tile t, k-step 1)  ->  after it tile
                //                t+1 is visible and stage t % NSTAGE is refilled with tile t + NSTAGE.
                if (!first) wait_dma_and_barrier<0>();          // later stream-K segment: everyone is done with the stages
                const int pro = min(nkt, NSTAGE);
                for (int t = 0; t < pro; ++t) issue(t);
                wait_tiles_and_barrier<PT>(pro - 1);          // tile 0 landed
                for (int t = 0; t + 1 < nkt; ++t) {
                    wait_tiles_and_barrier<PT>(min(nkt, t + NSTAGE) - (t + 2));
                    if (t + NSTAGE < nkt) issue(t + NSTAGE);
                }
                return;          // loaders hold no accumulators
            }
            // ---- MFMA wave
            if (!first) wait_lds_and_barrier();
            wait_lds_and_barrier();          // tile 0 landed

            f32x16 (&lo2)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&lo[0][0]);          // (PREC 3: never touched)
            Frags<NPL> fa0, fb0, fa1, fb1;
            ra.template read<0>(Lp, fa0);
            rb.template read<0>(Lp + OP_IMG, fb0);

            // One PHASE = the 4 NTERM MFMAs of a k-step, with the reads of the next k-step's fragments spread between them
            // (pinned by sched_barrier: two reads and 4 / 3 MFMAs per slot), in the order the next phase first needs them.
            auto phase = [&](auto rs_c, auto do_read_c, const Frags<NPL>& ca, const Frags<NPL>& cb, Frags<NPL>& na, Frags<NPL>& nb,
                             const lds_char* img) {
                constexpr int RS = decltype(rs_c)::value;
                constexpr bool RD = decltype(do_read_c)::value;
                constexpr int NSLOT = 2 * NPL, PER = 4 * CF::NTERM / NSLOT;          // 6 slots of 4 MFMAs / 4 slots of 3 / 2 slots of 2
                auto slot = [&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    // fragment pair k: row block k & 1; planes in the order of first use (PREC 3: A 2,0,1 / B 0,2,1; PREC 2: A 1,0 / B 0,1)
                    constexpr int ai = k & 1, bi = k & 1;
                    constexpr int apl = NPL == 3 ? (k < 2 ? 2 : k < 4 ? 0 : 1) : NPL == 2 ? (k < 2 ? 1 : 0) : 0;
                    constexpr int bpl = NPL == 3 ? (k < 2 ? 0 : k < 4 ? 2 : 1) : NPL == 2 ? (k < 2 ? 0 : 1) : 0;
                    if constexpr (RD) {
                        ra.template read_one<RS, ai, apl>(img, na);
                        rb.template read_one<RS, bi, bpl>(img + OP_IMG, nb);
                    }
                    mma_one<PREC, PER * k>(acc, lo2, ca, cb);
                    mma_one<PREC, PER * k + 1>(acc, lo2, ca, cb);
                    if constexpr (PER >= 3) mma_one<PREC, PER * k + 2>(acc, lo2, ca, cb);
                    if constexpr (PER == 4) mma_one<PREC, PER * k + 3>(acc, lo2, ca, cb);
                    __builtin_amdgcn_sched_barrier(0);
                };
                slot(integral_constant<int, 0>{}); slot(integral_constant<int, 1>{});
                if constexpr (NSLOT >= 4) { slot(integral_constant<int, 2>{}); slot(integral_constant<int, 3>{}); }
                if constexpr (NSLOT == 6) { slot(integral_constant<int, 4>{}); slot(integral_constant<int, 5>{}); }
            };
            // iteration t (fragments of (t, k-step 0) in fa0 / fb0):
            //   phase A: MFMAs of k-step 0 | reads of (t, k-step 1);  barrier(t);  phase B: MFMAs of k-step 1 | reads of (t+1, k-step 0)
            for (int t = 0; t + 1 < nkt; ++t) {
                phase(H1{}, T{}, fa0, fb0, fa1, fb1, Lp + (t % NSTAGE) * STAGE);
                wait_lds_and_barrier();
                phase(H0{}, T{}, fa1, fb1, fa0, fb0, Lp + ((t + 1) % NSTAGE) * STAGE);
            }
            phase(H1{}, T{}, fa0, fb0, fa1, fb1, Lp + ((nkt - 1) % NSTAGE) * STAGE);
            phase(H0{}, F{}, fa1, fb1, fa0, fb0, Lp);
        } else if (wave >= 4) {
            return;
        }

        // ---- epilogue: D[i][j], j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (nkt <= 0 && ep != 0) return;
        if constexpr (PREC <= 2) {
            // undo the operands' power-of-two scales (exact); rows / columns beyond the matrix were fed from slack: zero.
            // The scales are loaded UNCONDITIONALLY from clamped addresses, all of them before the first use: a load under a
            // per-element select (`ok ? scale[row] : 0`) made hipcc branch around each one and wait for it on the spot —
            // 64 dependent round trips, 13 us per tile with nothing else resident on the CU (a third of a 1920-deep tile).
            const float* __restrict__ as = a_inv;
            const float* __restrict__ bs = b_inv;
            float ib[2], ia[2][16];
#pragma unroll
            for (int j = 0; j < 2; ++j) ib[j] = bs[(long)min(n0 + wn * 64 + j * 32 + l31, dN - 1) * bst];
            if (ast == 0) {
                const float a0 = as[0];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ia[i][r] = a0;
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ia[i][r] = as[min(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, dM - 1)];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool cok = n0 + wn * 64 + j * 32 + l31 < dN;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[i][j][r];
                        if constexpr (PREC == 2) v += 0x1p-11f * lo[i][j][r];
                        v *= ia[i][r] * ib[j];
                        asm volatile("" : "+v"(v));          // computed, then selected: no branch around the arithmetic
                        acc[i][j][r] = (cok && row < dM) ? v : 0.f;
                    }
            }
        }
        if (!SK && stat_sum) {
            if (PREC == 3 && m0 + BM > dM) {          // rows beyond M were fed from the allocation's slack: not part of the statistics
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row >= dM) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
                    }
            }
            const long slot = (tiles_m > 64) ? (long)(tile_m & 63) * dN : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const double v = (double)acc[i][j][r];
                        s1 += v;
                        s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                const int col = n0 + wn * 64 + j * 32 + l31;
                if (half == 0 && col < dN) {
                    unsafeAtomicAdd(stat_sum + slot + col, s1);
                    unsafeAtomicAdd(stat_sumsq + slot + col, s2);
                }
            }
        }
        auto store_all = [&](auto mode_c, auto inside_c) {
            constexpr int MODE = decltype(mode_c)::value;          // the output form decided ONCE, not per element
            constexpr bool INSIDE = decltype(inside_c)::value;     // the whole tile inside C: no per-element predicate
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + l31;
                    const long off0 = (long)(m0 + wm * 64 + i * 32 + 4 * half) * ldc + col;
                    float* pc = Cp + off0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (INSIDE || (m0 + wm * 64 + i * 32 + 4 * half + dr < dM && col < dN)) {
                            const float v = acc[i][j][r];
                            if constexpr (MODE >= 4) {          // half-stored result (the mixed-precision mode's activations)
                                _Float16* p = reinterpret_cast<_Float16*>(Cp) + off0 + (long)dr * ldc;
                                *p = (_Float16)(MODE == 4 ? v : v + (float)*p);
                            } else {
                                float* p = pc + (long)dr * ldc;
                                if constexpr (MODE == 0) *p = v;
                                else if constexpr (MODE == 1) *p += v;
                                else atomicAdd(p, v);
                            }
                        }
                    }
                }
            }
        };
        const bool inside = m0 + BM <= dM && n0 + BN <= dN;
        if (PREC == 1 && c_half) {          // out_mode 0 / 1 only (ud_gemm_p3 rejects atomics onto a half result)
            if (inside) {
                if (ep == 0) store_all(integral_constant<int, 4>{}, T{});
                else store_all(integral_constant<int, 5>{}, T{});
            } else {
                if (ep == 0) store_all(integral_constant<int, 4>{}, F{});
                else store_all(integral_constant<int, 5>{}, F{});
            }
        } else if (inside) {
            if (ep == 0) store_all(integral_constant<int, 0>{}, T{});
            else if (ep == 1) store_all(integral_constant<int, 1>{}, T{});
            else store_all(integral_constant<int, 2>{}, T{});
        } else {
            if (ep == 0) store_all(integral_constant<int, 0>{}, F{});
            else if (ep == 1) store_all(integral_constant<int, 1>{}, F{});
            else store_all(integral_constant<int, 2>{}, F{});
        }
    };

    if constexpr (!SK) {
        int bt = bx;
        if (d.tile_cfg & 0x100) {          // each XCD takes a contiguous range of the tile order (see gemm_x3.hip)
            const int Tn = tiles_m * tiles_n, q = Tn >> 3, r = Tn & 7, x = bt & 7;
            bt = x * q + (x < r ? x : r) + (bt >> 3);
        }
        int tile_m = bt % tiles_m, tile_n = bt / tiles_m;
        if (d.tile_cfg & 0x200) {
            // XCD-aware raster: XCD x (= blockIdx % 8, its own L2) takes a CONTIGUOUS range of an order that walks groups of GM
            // tile rows column by column, so the ~32 workgroups it runs at a time cover a GM x (32 / GM) block of tiles — they
            // step through K together and share GM A panels + 32 / GM B panels in that L2 instead of ~9 + 8 of the round-robin deal
            const int GM = (d.tile_cfg >> 12) & 15 ? (d.tile_cfg >> 12) & 15 : 4;
            const int Tn = tiles_m * tiles_n, q = Tn >> 3, r = Tn & 7, x = bx & 7;
            const int o = x * q + (x < r ? x : r) + (bx >> 3);
            const int per_group = GM * tiles_n, g = o / per_group, first_m = g * GM;
            const int gm = min(GM, tiles_m - first_m), in = o - g * per_group;
            tile_n = in / gm;
            tile_m = first_m + in - tile_n * gm;
        }
        const int split = by;
        const int kt_per = (kt_total + d.split_k - 1) / d.split_k;
        const int kt0 = split * kt_per;
        const int nkt = min(kt_per, kt_total - kt0);
        segment(tile_m, tile_n, kt0, nkt, d.out_mode == 3 ? 0 : d.out_mode,
                d.C + (d.out_mode == 3 ? (long)split * d.slice_stride : 0L), true);
    } else {
        const int G = gx, b = bx;
        const int g = (b & 7) * (G >> 3) + (b >> 3);
        const long U = (long)tiles_m * tiles_n * kt_total;
        long u = g * U / G;
        const long u1 = (g + 1) * U / G;
        // tile order: column bands of 8 tiles, row by row inside a band (the last band may be narrower)
        const int band_tiles = 8 * tiles_m;
        bool first = true;
        while (u < u1) {
            const int tile = (int)(u / kt_total), kt_b = (int)(u - (long)tile * kt_total);
            const int kt_e = (int)min((long)kt_total, kt_b + (u1 - u));
            const int band = tile / band_tiles, in_band = tile - band * band_tiles;
            const int bw = min(8, tiles_n - band * 8);
            const int tile_m = in_band / bw, tile_n = band * 8 + in_band - tile_m * bw;
            const bool whole = kt_b == 0 && kt_e == kt_total;
            segment(tile_m, tile_n, kt_b, kt_e - kt_b, whole ? d.out_mode : 2, d.C, first);
            first = false;
            u += kt_e - kt_b;
        }
    }
}

template <int PREC, int AMODE, int BMODE, bool SK = false>
__global__ __launch_bounds__(NT, 1) void gemm_p3_kernel(const ud_gemm_p3_desc d, int tiles_m, int tiles_n) {
    using CF = Cfg<PREC>;
    __shared__ __attribute__((aligned(1024))) char L[CF::NSTAGE * CF::STAGE];
    p3_body<PREC, AMODE, BMODE, SK>(d, tiles_m, tiles_n, L, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// TWO problems in one launch (ud_gemm_p3_pair): the data gradient (modes 0 / 1) and the weight gradient (modes 1 / 1) of one 1x1
// conv share dy and nothing else, and each alone leaves CUs idle in its last round of tiles (540 tiles on 256 CUs are 2.1 rounds: the
// third runs 28 workgroups) — two queues would fill the holes but a fork / join costs ~24 us on this runtime (profiles/r05/
// spectral_pair_streams_ab.txt).  Here the second problem's workgroups simply follow the first's in ONE grid: workgroups
// [0, n0) = problem 0 (tile, split-K slice) pairs, the rest problem 1; the dispatcher hands them out in order, so problem 1 starts on
// the CUs problem 0's last round leaves free.  Each problem keeps its own XCD-aware raster over its LOCAL index (the hardware XCD
// is (local + n0) % 8: the ranges stay contiguous per XCD, rotated).
// Round 6: the operand modes of the two problems are template parameters.  <0, 1, 1, 1>: data + weight gradient (round 5).
// <0, 0, 0, 0>: the TAIL PAIR of one forward product — problem 0 the leading row tiles that fill whole rounds of the CUs as plain
// tiles, problem 1 the last row tiles split over K (atomics onto zeroed rows) whose short workgroups fill what problem 0's last
// round leaves free: 540 tiles on 256 CUs are 2.1 rounds instead of 3, in ONE launch (the "tail" plan needed two + a fill).
template <int PREC, int A0 = 0, int B0 = 1, int A1 = 1, int B1 = 1>
__global__ __launch_bounds__(NT, 1) void gemm_p3_pair_kernel(const ud_gemm_p3_desc d0, int tm0, int tn0, int n0,
                                                              const ud_gemm_p3_desc d1, int tm1, int tn1) {
    using CF = Cfg<PREC>;
    __shared__ __attribute__((aligned(1024))) char L[CF::NSTAGE * CF::STAGE];
    // n0 > 0: the data gradient's workgroups first; n0 < 0: the weight gradient's first (its |n0| workgroups)
    const int b = (int)blockIdx.x;
    const bool first = b < (n0 < 0 ? -n0 : n0);
    const int l = first ? b : b - (n0 < 0 ? -n0 : n0);
    if (first == (n0 > 0)) {
        const int T = tm0 * tn0;
        p3_body<PREC, A0, B0, false>(d0, tm0, tn0, L, l % T, l / T, T);
    } else {
        const int T = tm1 * tn1;
        p3_body<PREC, A1, B1, false>(d1, tm1, tn1, L, l % T, l / T, T);
    }
}

int num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        return cus;
    }();
    return n;
}

template <int PREC, int AMODE, int BMODE>
int launch(const ud_gemm_p3_desc& d, hipStream_t s) {
    const int tiles_m = ud_cdiv(d.M, BM), tiles_n = ud_cdiv(d.N, BN);
    if (d.tile_cfg & 0x800) {
        // stream-K: one worker per CU, at least 4 K-tiles of work each, a multiple of 8 (the XCD count)
        const long U = (long)tiles_m * tiles_n * (d.K / BK);
        long G = std::min<long>(num_cus(), U / 4);
        G = std::max<long>(8, G / 8 * 8);
        hipLaunchKernelGGL((gemm_p3_kernel<PREC, AMODE, BMODE, true>), dim3((unsigned)G), dim3(NT), 0, s, d, tiles_m, tiles_n);
    } else {
        dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)d.split_k, 1);
        hipLaunchKernelGGL((gemm_p3_kernel<PREC, AMODE, BMODE>), grid, dim3(NT), 0, s, d, tiles_m, tiles_n);
    }
    UD_LAUNCH_CHECK();
    return 0;
}

// ---- fp32 [R][C] (row stride ld) -> three bf16 planes in the P32 layout; one thread = 8 consecutive columns of one row
__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)x, (__bf16)y};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(sx, sy);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long R, int C, long ld,
                                                           uint16_t* __restrict__ out, long panel, long plane) {
    // workgroup: 64 rows x one panel; thread: row = tid >> 2, chunk (8 columns) = tid & 3
    const int tid = threadIdx.x;
    const long row = (long)blockIdx.x * 64 + (tid >> 2);
    const int pan = blockIdx.y, c0 = pan * 32 + (tid & 3) * 8;
    if (row >= R) return;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 8 <= C) {
        v0 = *reinterpret_cast<const f32x4*>(x + row * ld + c0);
        v1 = *reinterpret_cast<const f32x4*>(x + row * ld + c0 + 4);
    } else if (c0 + 4 <= C) {
        v0 = *reinterpret_cast<const f32x4*>(x + row * ld + c0);
    }
    uint32_t a[4], b[4], c[4];
    split2(v0[0], v0[1], a[0], b[0], c[0]);
    split2(v0[2], v0[3], a[1], b[1], c[1]);
    split2(v1[0], v1[1], a[2], b[2], c[2]);
    split2(v1[2], v1[3], a[3], b[3], c[3]);
    const u32x4 p0 = {a[0], a[1], a[2], a[3]}, p1 = {b[0], b[1], b[2], b[3]}, p2 = {c[0], c[1], c[2], c[3]};
    uint16_t* o = out + (long)pan * panel + row * 32 + (tid & 3) * 8;
    *reinterpret_cast<u32x4*>(o) = p0;
    *reinterpret_cast<u32x4*>(o + plane) = p1;
    *reinterpret_cast<u32x4*>(o + 2 * plane) = p2;
}

// ---- fp32 [R][C] -> two fp16 planes of s[r] * x[r][:], s[r] = the power of two that takes the row maximum into [2^14, 2^15),
// and inv[r] = 1 / s[r].  One wave per row (the row lives in registers between the maximum and the split: C <= 256 * MAXQ),
// four consecutive rows per workgroup, so that each panel receives 256 contiguous bytes.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float x, float y, uint32_t& h0, uint32_t& h1) {
    const f16x2 a = {(_Float16)x, (_Float16)y};
    h0 = __builtin_bit_cast(uint32_t, a);
    const f16x2 b = {(_Float16)((x - (float)a[0]) * 2048.f), (_Float16)((y - (float)a[1]) * 2048.f)};          // exact residual x 2^11
    h1 = __builtin_bit_cast(uint32_t, b);
}

constexpr int H2_MAXQ = 16;          // float4 per lane: rows of up to 4096 columns

__global__ __launch_bounds__(256) void split_h2_rows_kernel(const float* __restrict__ x, long R, int C, long ld,
                                                            uint16_t* __restrict__ out, long panel, long plane,
                                                            float* __restrict__ inv_scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int nq = C >> 2;                      // float4 of the row
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ld);
    f32x4 v[H2_MAXQ];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < H2_MAXQ; ++i) {
        const int q = lane + 64 * i;
        v[i] = q < nq ? xr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
    }
    m = ud_wave_max(m);
    // exponent field e of the maximum -> scale 2^(14 - (e - 127)) (field 268 - e), kept inside the normal range; an all-zero row: 1
    const int e = (int)(__float_as_uint(m) >> 23) & 0xff;
    int fs = m > 0.f ? 268 - e : 127;
    fs = fs < 1 ? 1 : fs > 254 ? 254 : fs;
    const float s = __uint_as_float((uint32_t)fs << 23), inv = __uint_as_float((uint32_t)(254 - fs) << 23);
    if (lane == 0) inv_scale[row] = inv;
    const int nq_pad = ((C + 31) >> 5) << 3;          // float4 up to the end of the last panel: zero columns
#pragma unroll
    for (int i = 0; i < H2_MAXQ; ++i) {
        const int q = lane + 64 * i;
        if (q < nq_pad) {
            uint32_t a0, a1, b0, b1;
            split2h(v[i][0] * s, v[i][1] * s, a0, a1);
            split2h(v[i][2] * s, v[i][3] * s, b0, b1);
            uint16_t* o = out + (long)(q >> 3) * panel + row * 32 + (q & 7) * 4;
            *reinterpret_cast<u32x2*>(o) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(o + plane) = u32x2{a1, b1};
        }
    }
}

// ---- per-TENSOR scale: |x|max over the whole matrix, as ABSMAX_SLOTS partial maxima (bit patterns of non-negative floats:
// unsigned order = float order) that the split kernel folds — no atomics (8192 atomic maxima onto one word took 100 us), no
// zero fill.  A producer kernel may fill the slots itself.
constexpr int ABSMAX_SLOTS = 256;
__global__ __launch_bounds__(1024) void absmax_kernel(const float* __restrict__ x, long R, int C, long ld, uint32_t* __restrict__ out) {
    // 256 workgroups x 16 waves, two 16-byte loads in flight per lane: a reduction this short lives on loads in flight
    __shared__ float red[16];
    const long nq = R * (long)(C >> 2);
    const int cq = C >> 2;
    const long stride = (long)ABSMAX_SLOTS * 1024;
    float m = 0.f;
    auto at = [&](long q) {
        const long r = q / cq;
        return *reinterpret_cast<const f32x4*>(x + r * ld + (q - r * cq) * 4);
    };
    long q = (long)blockIdx.x * 1024 + threadIdx.x;
    for (; q + stride < nq; q += 2 * stride) {
        const f32x4 v = at(q), w = at(q + stride);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3]))));
    }
    if (q < nq) {
        const f32x4 v = at(q);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = ud_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) t = fmaxf(t, red[i]);
        out[blockIdx.x] = __float_as_uint(t);
    }
}

// scale of a tensor whose |x|max has the bit pattern `bits`: the power of two taking it into [2^14, 2^15) (exponent field
// 268 - e, kept inside the normal range; all zeros: 1); inv = 1 / scale.  NaN / inf maxima end up as NaN / inf pieces.
__device__ __forceinline__ void h2_scale(uint32_t bits, float& s, float& inv) { ud_h2_scale(bits, s, inv); }

// fp32 [R][C] -> two fp16 planes of s * x with ONE scale for the tensor (any GEMM mode may read them); thread = 8 columns of a row
__global__ __launch_bounds__(256) void split_h2_tensor_kernel(const float* __restrict__ x, long R, int C, long ld,
                                                              uint16_t* __restrict__ out, long panel, long plane,
                                                              const uint32_t* __restrict__ absmax, float* __restrict__ inv_scale) {
    const int tid = threadIdx.x;
    const long row = (long)blockIdx.x * 64 + (tid >> 2);
    const int pan = blockIdx.y, c0 = pan * 32 + (tid & 3) * 8;
    // fold the partial maxima (every wave by itself: 4 per lane; no barrier)
    uint32_t mb = 0;
#pragma unroll
    for (int i = 0; i < ABSMAX_SLOTS / 64; ++i) mb = max(mb, absmax[(tid & 63) + 64 * i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, o, 64));
    float s, inv;
    h2_scale(mb, s, inv);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *inv_scale = inv;
    if (row >= R) return;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 8 <= C) {
        v0 = *reinterpret_cast<const f32x4*>(x + row * ld + c0);
        v1 = *reinterpret_cast<const f32x4*>(x + row * ld + c0 + 4);
    } else if (c0 + 4 <= C) {
        v0 = *reinterpret_cast<const f32x4*>(x + row * ld + c0);
    }
    uint32_t a[4], b[4];
    split2h(v0[0] * s, v0[1] * s, a[0], b[0]);
    split2h(v0[2] * s, v0[3] * s, a[1], b[1]);
    split2h(v1[0] * s, v1[1] * s, a[2], b[2]);
    split2h(v1[2] * s, v1[3] * s, a[3], b[3]);
    uint16_t* o = out + (long)pan * panel + row * 32 + (tid & 3) * 8;
    *reinterpret_cast<u32x4*>(o) = u32x4{a[0], a[1], a[2], a[3]};
    *reinterpret_cast<u32x4*>(o + plane) = u32x4{b[0], b[1], b[2], b[3]};
}

// half-stored [R][C] (row stride ld) -> ONE fp16 plane in the P32 layout, values as they are (scale 1): the operand form of
// ud_gemm_p3 prec 1 for the mixed-precision mode's activations; thread = 8 columns of a row (16 bytes in, 16 bytes out)
__global__ __launch_bounds__(256) void planes_from_half_kernel(const uint16_t* __restrict__ x, long R, int C, long ld,
                                                               uint16_t* __restrict__ out, long panel,
                                                               float* __restrict__ inv_scale) {
    const int tid = threadIdx.x;
    const long row = (long)blockIdx.x * 64 + (tid >> 2);
    const int pan = blockIdx.y, c0 = pan * 32 + (tid & 3) * 8;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *inv_scale = 1.f;
    if (row >= R) return;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (c0 + 8 <= C) v = *reinterpret_cast<const u32x4*>(x + row * ld + c0);
    *reinterpret_cast<u32x4*>(out + (long)pan * panel + row * 32 + (tid & 3) * 8) = v;
}

// ---- a k x k conv as a 1x1 conv on the planes GEMM (round 5): its im2col matrix [N Hout Wout] x [KH KW Cin] written DIRECTLY as
// fp16 x 2 planes (one scale for the tensor: |x|max of the conv's input, `absmax` slots as for split_h2_tensor_kernel).  The deep
// 3 x 3 convs of the ResNet variants (model/resnet/exp.py:95-111; the 2048 -> 2048 filter conv of model/modules.py:111 at
// 10 x 10) ran at 93-135 TFLOP/s-equivalent on the in-kernel-split gather GEMM; from this matrix all three products of the conv
// (forward, data gradient as a GEMM + ud_col2im, weight gradient) run on ud_gemm_p3.  Cin % 32 == 0: a 32-column panel lies
// inside one tap.  thread = 8 columns of a row, as in the split kernel; out-of-image taps are zeros.
__global__ __launch_bounds__(256) void im2col_planes_kernel(const float* __restrict__ x, ud_conv_geom g, uint16_t* __restrict__ out,
                                                            long panel, long plane, const uint32_t* __restrict__ absmax,
                                                            float* __restrict__ inv_scale) {
    const int tid = threadIdx.x;
    const long M = (long)g.N * g.Hout * g.Wout;
    const long row = (long)blockIdx.x * 64 + (tid >> 2);
    const int pan = blockIdx.y;
    const int col0 = pan * 32;                          // first column of the panel: tap col0 / Cin, channel col0 % Cin
    const int tap = col0 / g.Cin, ci = col0 % g.Cin + (tid & 3) * 8;
    const int kh = tap / g.KW, kw = tap % g.KW;
    uint32_t mb = 0;
#pragma unroll
    for (int i = 0; i < ABSMAX_SLOTS / 64; ++i) mb = max(mb, absmax[(tid & 63) + 64 * i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, o, 64));
    float s, inv;
    h2_scale(mb, s, inv);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *inv_scale = inv;
    if (row >= M) return;
    const int ow = (int)(row % g.Wout);
    const long t = row / g.Wout;
    const int oh = (int)(t % g.Hout);
    const long n = t / g.Hout;
    const int ih = oh * g.stride - g.pad_t + kh, iw = ow * g.stride - g.pad_l + kw;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win) {
        const float* src = x + ((n * g.Hin + ih) * g.Win + iw) * g.Cin + ci;
        v0 = *reinterpret_cast<const f32x4*>(src);
        v1 = *reinterpret_cast<const f32x4*>(src + 4);
    }
    uint32_t a[4], b[4];
    split2h(v0[0] * s, v0[1] * s, a[0], b[0]);
    split2h(v0[2] * s, v0[3] * s, a[1], b[1]);
    split2h(v1[0] * s, v1[1] * s, a[2], b[2]);
    split2h(v1[2] * s, v1[3] * s, a[3], b[3]);
    uint16_t* o = out + (long)pan * panel + row * 32 + (tid & 3) * 8;
    *reinterpret_cast<u32x4*>(o) = u32x4{a[0], a[1], a[2], a[3]};
    *reinterpret_cast<u32x4*>(o + plane) = u32x4{b[0], b[1], b[2], b[3]};
}

// dx[n][ih][iw][ci] = sum over the taps of dcol[(n, oh, ow)][(kh KW + kw) Cin + ci] with (oh, ow) the output pixel whose window
// holds (ih, iw) at that tap (stride 1 or 2): the adjoint of the im2col gather, a gather itself (no atomics)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, ud_conv_geom g, float* __restrict__ dx) {
    const int C4 = g.Cin / 4;
    const long total = (long)g.N * g.Hin * g.Win * C4;
    const long Kc = (long)g.KH * g.KW * g.Cin;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long pix = e / C4;
        const int iw = (int)(pix % g.Win);
        long t = pix / g.Win;
        const int ih = (int)(t % g.Hin);
        const long n = t / g.Hin;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < g.KH; ++kh) {
            const int th = ih + g.pad_t - kh;
            if (th < 0 || th % g.stride) continue;
            const int oh = th / g.stride;
            if (oh >= g.Hout) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int tw = iw + g.pad_l - kw;
                if (tw < 0 || tw % g.stride) continue;
                const int ow = tw / g.stride;
                if (ow >= g.Wout) continue;
                acc += *reinterpret_cast<const f32x4*>(dcol + ((n * g.Hout + oh) * g.Wout + ow) * Kc +
                                                        (long)(kh * g.KW + kw) * g.Cin + c4 * 4);
            }
        }
        reinterpret_cast<f32x4*>(dx)[e] = acc;
    }
}

// ---- all weight matrices of a step in two launches -------------------------------------------------------------------
// A train step splits ~50 weight matrices, each with an absmax launch and a split launch of a few microseconds.  The items
// (pointers, shapes, first-block prefixes) live in a device table built once; block -> item by bisection of the prefixes.
__device__ __forceinline__ int item_of_block(const ud_split_item* __restrict__ it, int n, int b, bool split) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((split ? it[mid].split_block0 : it[mid].amax_block0) <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void absmax_multi_kernel(const ud_split_item* __restrict__ items, int n,
                                                            uint32_t* __restrict__ slots) {
    __shared__ float red[16];
    const int i = item_of_block(items, n, blockIdx.x, false);
    const ud_split_item it = items[i];
    const int bl = blockIdx.x - it.amax_block0, nb = it.amax_blocks;
    const int cq = it.C >> 2;
    const long nq = it.R * (long)cq, stride = (long)nb * 1024;
    float m = 0.f;
    for (long q = (long)bl * 1024 + threadIdx.x; q < nq; q += stride) {
        const long r = q / cq;
        const f32x4 v = *reinterpret_cast<const f32x4*>(it.x + r * it.ld + (q - r * cq) * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = ud_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = fmaxf(t, red[k]);
        slots[(long)i * ABSMAX_SLOTS + bl] = __float_as_uint(t);          // slots >= amax_blocks stay zero (zeroed once by the caller)
    }
}

__global__ __launch_bounds__(256) void split_h2_multi_kernel(const ud_split_item* __restrict__ items, int n,
                                                             const uint32_t* __restrict__ slots) {
    const int i = item_of_block(items, n, blockIdx.x, true);
    const ud_split_item it = items[i];
    const int bl = blockIdx.x - it.split_block0;
    const int pan = bl / it.split_bx, bx = bl - pan * it.split_bx;
    const int tid = threadIdx.x;
    const long row = (long)bx * 64 + (tid >> 2);
    const int c0 = pan * 32 + (tid & 3) * 8;
    const uint32_t* absmax = slots + (long)i * ABSMAX_SLOTS;
    uint32_t mb = 0;
#pragma unroll
    for (int k = 0; k < ABSMAX_SLOTS / 64; ++k) mb = max(mb, absmax[(tid & 63) + 64 * k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, o, 64));
    float s, inv;
    h2_scale(mb, s, inv);
    if (bl == 0 && tid == 0) *it.inv_scale = inv;
    if (row >= it.R) return;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 8 <= it.C) {
        v0 = *reinterpret_cast<const f32x4*>(it.x + row * it.ld + c0);
        v1 = *reinterpret_cast<const f32x4*>(it.x + row * it.ld + c0 + 4);
    } else if (c0 + 4 <= it.C) {
        v0 = *reinterpret_cast<const f32x4*>(it.x + row * it.ld + c0);
    }
    uint32_t a[4], b[4];
    split2h(v0[0] * s, v0[1] * s, a[0], b[0]);
    split2h(v0[2] * s, v0[3] * s, a[1], b[1]);
    split2h(v1[0] * s, v1[1] * s, a[2], b[2]);
    split2h(v1[2] * s, v1[3] * s, a[3], b[3]);
    uint16_t* o = it.out + (long)pan * it.panel + row * 32 + (tid & 3) * 8;
    *reinterpret_cast<u32x4*>(o) = u32x4{a[0], a[1], a[2], a[3]};
    *reinterpret_cast<u32x4*>(o + it.plane) = u32x4{b[0], b[1], b[2], b[3]};
}

}  // namespace

extern "C" int ud_split_planes_h2t_multi(const ud_split_item* items_dev, int n, uint32_t* slots, int amax_blocks_total,
                                         int split_blocks_total, ud_stream_t stream) {
    if (!items_dev || !slots || n < 1 || amax_blocks_total < n || split_blocks_total < n) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)amax_blocks_total), dim3(1024), 0, s, items_dev, n, slots);
    hipLaunchKernelGGL(split_h2_multi_kernel, dim3((unsigned)split_blocks_total), dim3(256), 0, s, items_dev, n, slots);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_im2col_planes(const float* x, const ud_conv_geom* g, uint16_t* planes, long panel_stride, long plane_stride,
                                const uint32_t* absmax, float* inv_scale, ud_stream_t stream) {
    if (!x || !g || !planes || !absmax || !inv_scale || g->transposed || g->Cin < 32 || g->Cin % 32 || g->N < 1 || g->Hin < 1 ||
        g->Win < 1 || g->Hout < 1 || g->Wout < 1 || g->KH < 1 || g->KW < 1 || g->stride < 1)
        return UD_EINVAL;
    const long M = (long)g->N * g->Hout * g->Wout, Kc = (long)g->KH * g->KW * g->Cin;
    if (panel_stride < M * 32 || panel_stride % 8 || plane_stride % 8 || plane_stride < (Kc / 32) * panel_stride) return UD_EINVAL;
    dim3 grid((unsigned)ud_cdiv(M, 64), (unsigned)(Kc / 32));
    hipLaunchKernelGGL(im2col_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, *g, planes, panel_stride, plane_stride,
                       absmax, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_col2im(const float* dcol, const ud_conv_geom* g, float* dx, ud_stream_t stream) {
    if (!dcol || !g || !dx || g->transposed || g->Cin % 4 || g->stride < 1) return UD_EINVAL;
    const long total = (long)g->N * g->Hin * g->Win * (g->Cin / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dcol, *g, dx);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_absmax(const float* x, long R, int C, long ld, uint32_t* out, ud_stream_t stream) {
    if (!x || !out || R <= 0 || C <= 0 || C % 4 != 0 || ld % 4 != 0 || ld < C) return UD_EINVAL;
    hipLaunchKernelGGL(absmax_kernel, dim3(ABSMAX_SLOTS), dim3(1024), 0, (hipStream_t)stream, x, R, C, ld, out);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_split_planes_h2t(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride,
                                   long plane_stride, const uint32_t* absmax, float* inv_scale, ud_stream_t stream) {
    if (!x || !planes || !absmax || !inv_scale || R <= 0 || C <= 0 || C % 4 != 0 || ld % 4 != 0 || ld < C ||
        panel_stride < R * 32 || panel_stride % 8 != 0 || plane_stride % 8 != 0 ||
        plane_stride < (long)ud_cdiv(C, 32) * panel_stride)
        return UD_EINVAL;
    dim3 grid((unsigned)ud_cdiv(R, 64), (unsigned)ud_cdiv(C, 32));
    hipLaunchKernelGGL(split_h2_tensor_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, R, C, ld, planes, panel_stride,
                       plane_stride, absmax, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_planes_from_half(const void* x, long R, int C, long ld, uint16_t* planes, long panel_stride, float* inv_scale,
                                   ud_stream_t stream) {
    if (!x || !planes || !inv_scale || R < 1 || C < 8 || C % 8 || ld < C || ld % 8 || panel_stride < 32 * R || panel_stride % 8 ||
        ((uintptr_t)x & 15))
        return UD_EINVAL;
    dim3 grid((unsigned)ud_cdiv(R, 64), (unsigned)ud_cdiv(C, 32));
    hipLaunchKernelGGL(planes_from_half_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, R, C, ld, planes,
                       panel_stride, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_split_planes(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride,
                               long plane_stride, ud_stream_t stream) {
    if (!x || !planes || R <= 0 || C <= 0 || C % 4 != 0 || ld % 4 != 0 || ld < C || panel_stride < R * 32 ||
        panel_stride % 8 != 0 || plane_stride % 8 != 0 || plane_stride < (long)ud_cdiv(C, 32) * panel_stride)
        return UD_EINVAL;
    dim3 grid((unsigned)ud_cdiv(R, 64), (unsigned)ud_cdiv(C, 32));
    hipLaunchKernelGGL(split_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, R, C, ld, planes, panel_stride,
                       plane_stride);
    UD_LAUNCH_CHECK();
    return 0;
}

extern "C" int ud_split_planes_h2(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride,
                                  long plane_stride, float* inv_scale, ud_stream_t stream) {
    if (!x || !planes || !inv_scale || R <= 0 || C <= 0 || C % 4 != 0 || C > 256 * H2_MAXQ || ld % 4 != 0 || ld < C ||
        panel_stride < R * 32 || panel_stride % 8 != 0 || plane_stride % 8 != 0 ||
        plane_stride < (long)ud_cdiv(C, 32) * panel_stride)
        return UD_EINVAL;
    hipLaunchKernelGGL(split_h2_rows_kernel, dim3((unsigned)ud_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, R, C, ld,
                       planes, panel_stride, plane_stride, inv_scale);
    UD_LAUNCH_CHECK();
    return 0;
}

static bool p3_desc_ok(const ud_gemm_p3_desc& d);

extern "C" int ud_gemm_p3_pair(const ud_gemm_p3_desc* nn, const ud_gemm_p3_desc* tn, ud_stream_t stream) {
    if (!nn || !tn || !p3_desc_ok(*nn) || !p3_desc_ok(*tn)) return UD_EINVAL;
    const ud_gemm_p3_desc &d0 = *nn, &d1 = *tn;
    const bool grads = d0.a_mode == 0 && d0.b_mode == 1 && d1.a_mode == 1 && d1.b_mode == 1;          // data + weight gradient
    const bool tailp = d0.a_mode == 0 && d0.b_mode == 0 && d1.a_mode == 0 && d1.b_mode == 0 && d0.prec == 2;      // tail pair of a forward product
    if ((d0.prec != 2 && d0.prec != 1) || d1.prec != d0.prec || !(grads || tailp)) return UD_EINVAL;
    if ((d0.tile_cfg | d1.tile_cfg) & 0x800) return UD_EINVAL;          // no stream-K form
    if (d0.stat_sum || d1.stat_sum || d0.out_mode == 3 || d1.out_mode == 3) return UD_EINVAL;
    const int tm0 = ud_cdiv(d0.M, BM), tn0 = ud_cdiv(d0.N, BN), tm1 = ud_cdiv(d1.M, BM), tn1 = ud_cdiv(d1.N, BN);
    const long n0 = (long)tm0 * tn0 * d0.split_k, n1 = (long)tm1 * tn1 * d1.split_k;
    if (n0 + n1 > 0x7fffffffL) return UD_EINVAL;
    // tile_cfg bit 16 of the weight gradient's descriptor: ITS workgroups go first (a long-K weight gradient with few tiles is the
    // pair's critical path: started first, the data gradient's many short tiles fill in around it)
    const int nfirst = (d1.tile_cfg & 0x10000) ? -(int)n1 : (int)n0;
    if (tailp)
        hipLaunchKernelGGL((gemm_p3_pair_kernel<2, 0, 0, 0, 0>), dim3((unsigned)(n0 + n1)), dim3(NT), 0, (hipStream_t)stream, d0, tm0,
                           tn0, nfirst, d1, tm1, tn1);
    else if (d0.prec == 1)
        hipLaunchKernelGGL((gemm_p3_pair_kernel<1>), dim3((unsigned)(n0 + n1)), dim3(NT), 0, (hipStream_t)stream, d0, tm0, tn0,
                           nfirst, d1, tm1, tn1);
    else
        hipLaunchKernelGGL((gemm_p3_pair_kernel<2>), dim3((unsigned)(n0 + n1)), dim3(NT), 0, (hipStream_t)stream, d0, tm0, tn0,
                           nfirst, d1, tm1, tn1);
    UD_LAUNCH_CHECK();
    return 0;
}

static bool p3_desc_ok(const ud_gemm_p3_desc& d) {
    if (!d.A || !d.B || !d.C || d.M <= 0 || d.N <= 0 || d.K <= 0 || d.K % BK != 0 || d.split_k < 1 ||
        d.split_k > d.K / BK || d.out_mode < 0 || d.out_mode > 3 || d.a_mode < 0 || d.a_mode > 1 || d.b_mode < 0 ||
        d.b_mode > 1 || d.a_panel % 8 != 0 || d.b_panel % 8 != 0 || d.a_plane % 8 != 0 || d.b_plane % 8 != 0 ||
        d.a_npanel < 1 || d.b_npanel < 1 || d.prec < 1 || d.prec > 3)
        return false;
    if (d.prec <= 2 && (!d.a_inv_scale || !d.b_inv_scale || d.a_scale_stride < 0 || d.a_scale_stride > 1 ||
                        d.b_scale_stride < 0 || d.b_scale_stride > 1))
        return false;
    if (d.stat_sum && (d.out_mode != 0 || d.split_k != 1 || !d.stat_sumsq || (d.tile_cfg & 0x800))) return false;
    if ((d.tile_cfg & 0x800) && (d.out_mode > 1 || d.split_k != 1)) return false;          // stream-K: store-onto-zeros or add
    if (d.out_mode == 3 && d.slice_stride < (long)d.M * d.ldc) return false;
    if (d.c_half && (d.prec != 1 || d.out_mode > 1 || d.split_k != 1 || (d.tile_cfg & 0x800))) return false;   // no atomics onto half
    return true;
}

extern "C" int ud_gemm_p3(const ud_gemm_p3_desc* dp, ud_stream_t stream) {
    if (!dp) return UD_EINVAL;
    const ud_gemm_p3_desc& d = *dp;
    if (!p3_desc_ok(d)) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (d.prec == 1) {          // the first plane of prec-2 operands: one fp16 product (mixed precision)
        if (d.a_mode == 0 && d.b_mode == 0) return launch<1, 0, 0>(d, s);
        if (d.a_mode == 0 && d.b_mode == 1) return launch<1, 0, 1>(d, s);
        if (d.a_mode == 1 && d.b_mode == 1) return launch<1, 1, 1>(d, s);
        return UD_EINVAL;
    }
    if (d.prec == 2) {
        if (d.a_mode == 0 && d.b_mode == 0) return launch<2, 0, 0>(d, s);
        if (d.a_mode == 0 && d.b_mode == 1) return launch<2, 0, 1>(d, s);
        if (d.a_mode == 1 && d.b_mode == 1) return launch<2, 1, 1>(d, s);
        return UD_EINVAL;
    }
    if (d.a_mode == 0 && d.b_mode == 0) return launch<3, 0, 0>(d, s);
    if (d.a_mode == 0 && d.b_mode == 1) return launch<3, 0, 1>(d, s);
    if (d.a_mode == 1 && d.b_mode == 1) return launch<3, 1, 1>(d, s);
    if (d.a_mode == 1 && d.b_mode == 0) return launch<3, 1, 0>(d, s);
    return UD_EINVAL;
}
