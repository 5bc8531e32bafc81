// fp32 GEMM / implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every GEMM-shaped op of the UniDefense step (include/unidefense_hip.h,
// ud_gemm): 1x1 convs incl. the spectral freq_conv (model/efficientnet/exp.py:57), dense 3x3 convs and
// the transposed conv of the decoder (model/unidefense.py:59-102) as implicit GEMMs with a gather
// loader, and all their data / weight gradients.
//
// Structure: 256 threads = 4 wave64; block tile BM x BN x 16; each wave owns TM x TN tiles of 32x32
// (16 accumulator VGPRs each).  Operand tiles are staged global -> registers -> LDS in a K-MAJOR LDS
// image  As[k][m], Bs[k][n]  whatever the global layout is, so the MFMA feed is the same conflict-free
// ds_read_b32 pattern for all modes (lane l reads [k0 + (l>>5)][col0 + (l&31)]: each 32-lane half reads
// 32 consecutive floats).  Register prefetch of tile t+1 overlaps the MFMAs of tile t; one barrier
// per K-tile.  fp32 MFMA is exact f32 FMA chaining (k ascending), the same arithmetic as the reference's
// fp32 path up to summation order.
#include "gemm_internal.h"
#include "ud_common.h"

#include <atomic>

namespace {

constexpr int BK = 32;                 // K-tile depth (two K-tiles of MFMA work cover one prefetch)
constexpr int KQ = BK / 4;             // float4 per K-contiguous tile row
constexpr int NTHREADS = 256;
constexpr int PATCH_M = 4, PATCH_N = 8;   // output-tile patch owned by one XCD's 32 consecutive workgroups

__device__ __forceinline__ void decode_row(const ud_conv_geom& g, int m, int M, int& nbase, int& ih0,
                                           int& iw0, bool& valid) {
    valid = m < M;
    int mm = valid ? m : 0;
    int ow = mm % g.Wout;
    int t = mm / g.Wout;
    int oh = t % g.Hout;
    int n = t / g.Hout;
    nbase = n * g.Hin * g.Win;
    if (!g.transposed) {
        ih0 = oh * g.stride - g.pad_t;
        iw0 = ow * g.stride - g.pad_l;
    } else {
        ih0 = oh + g.pad_t;
        iw0 = ow + g.pad_l;
    }
}

// element offset of source pixel's channel 0, ok=false when the tap falls outside the input
__device__ __forceinline__ long gather_pixel(const ud_conv_geom& g, int nbase, int ih0, int iw0, int kh,
                                             int kw, bool& ok) {
    int ih, iw;
    if (!g.transposed) {
        ih = ih0 + kh;
        iw = iw0 + kw;
        ok = (ih >= 0) && (ih < g.Hin) && (iw >= 0) && (iw < g.Win);
    } else {
        int th = ih0 - kh, tw = iw0 - kw;
        ok = (th >= 0) && (tw >= 0) && (th % g.stride == 0) && (tw % g.stride == 0);
        ih = th / g.stride;
        iw = tw / g.stride;
        ok = ok && (ih < g.Hin) && (iw < g.Win);
    }
    return ((long)nbase + (long)ih * g.Win + iw) * (long)g.Cin;
}

__device__ __forceinline__ f32x4 load4(const float* p, bool vec) {
    f32x4 v;
    if (vec) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Operand tile loader.  MODE 0: [row][k] K-contiguous.  MODE 1: [k][row] row-contiguous.
// MODE 2 (as A): conv gather, row = (n,oh,ow), k = (tap,ci)  -> K-contiguous inside a tap.
// MODE 2 (as B, "BGATHER"): conv gather, k = (n,oh,ow), row(col) = (tap,ci) -> row-contiguous.
// ROWS = BM or BN.  The loader keeps NV float4 per thread in registers.
// ---------------------------------------------------------------------------------------------
template <int ROWS, int MODE, bool IS_B>
struct Loader {
    static constexpr bool KCONTIG = (MODE == 0) || (MODE == 2 && !IS_B);
    static constexpr int TOTAL_V = ROWS * BK / 4;                     // float4 in one tile
    static constexpr int NV = (TOTAL_V + NTHREADS - 1) / NTHREADS;    // per thread
    // LDS row stride of the K-major image S[k][row].  K-contiguous operands are transposed on the way in with
    // 4-byte stores  S[4*kq + e][row]  (kq = 0..7 across lanes): a stride == 1 (mod 8) puts the 32 lanes of a
    // store on 32 different banks (stride ROWS + 4 was a 4-way conflict: SQ_LDS_BANK_CONFLICT = 50 % of the
    // LDS cycles).  Row-contiguous operands are stored 16 bytes at a time and need a multiple of 4.
    static constexpr int LDS_LD = KCONTIG ? ROWS + 1 : ROWS + 4;

    const float* base;
    long ld;
    int dim;        // number of valid rows (M or N)
    int K;
    bool vec;       // 16-byte loads allowed
    ud_conv_geom g;
    // per-thread cached gather state
    int c_nbase[NV], c_ih0[NV], c_iw0[NV];
    bool c_valid[NV];
    int c_kh[NV], c_kw[NV], c_ci[NV];
    f32x4 regs[NV];

    __device__ __forceinline__ void init(const float* p, long ld_, int dim_, int K_, bool vec_,
                                         const ud_conv_geom& g_, int row0, int tid) {
        base = p; ld = ld_; dim = dim_; K = K_; vec = vec_; g = g_;
        if constexpr (MODE == 2 && !IS_B) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int f = tid + i * NTHREADS;
                int row = f / KQ;
                decode_row(g, row0 + row, dim, c_nbase[i], c_ih0[i], c_iw0[i], c_valid[i]);
                c_valid[i] = c_valid[i] && (f < TOTAL_V);
            }
        }
        if constexpr (MODE == 2 && IS_B) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int f = tid + i * NTHREADS;
                int nq = f % (ROWS / 4);
                int col = row0 + nq * 4;
                c_valid[i] = (col < dim) && (f < TOTAL_V);
                int cc = c_valid[i] ? col : 0;
                int tap = cc / g.Cin;
                c_ci[i] = cc % g.Cin;
                c_kh[i] = tap / g.KW;
                c_kw[i] = tap % g.KW;
            }
        }
    }

    // issue the global loads of the tile whose first k is k0 (rows start at row0)
    __device__ __forceinline__ void load(int row0, int k0, int k_end, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int f = tid + i * NTHREADS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == 0) {
                int row = f / KQ, kq = f % KQ;
                int r = row0 + row, k = k0 + kq * 4;
                if (f < TOTAL_V && r < dim && k < k_end) {
                    const float* p = base + (long)r * ld + k;
                    if (k + 3 < k_end) {
                        v = load4(p, vec);
                    } else {
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) if (k + e < k_end) v[e] = p[e];
                    }
                }
            } else if constexpr (MODE == 1) {
                int kr = f / (ROWS / 4), q = f % (ROWS / 4);
                int k = k0 + kr, r = row0 + q * 4;
                if (f < TOTAL_V && k < k_end && r < dim) {
                    const float* p = base + (long)k * ld + r;
                    if (r + 3 < dim) {
                        v = load4(p, vec);
                    } else {
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) if (r + e < dim) v[e] = p[e];
                    }
                }
            } else if constexpr (MODE == 2 && !IS_B) {
                int kq = f % KQ;
                int k = k0 + kq * 4;
                if (c_valid[i] && k < k_end) {
                    if (vec) {   // Cin % 4 == 0: the 4 k's share one tap
                        int tap = k / g.Cin, ci = k % g.Cin;
                        bool ok;
                        long off = gather_pixel(g, c_nbase[i], c_ih0[i], c_iw0[i], tap / g.KW, tap % g.KW, ok);
                        if (ok) v = *reinterpret_cast<const f32x4*>(base + off + ci);
                    } else {
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                            int kk = k + e;
                            if (kk < k_end) {
                                int tap = kk / g.Cin, ci = kk % g.Cin;
                                bool ok;
                                long off = gather_pixel(g, c_nbase[i], c_ih0[i], c_iw0[i], tap / g.KW, tap % g.KW, ok);
                                if (ok) v[e] = base[off + ci];
                            }
                        }
                    }
                }
            } else {   // MODE 2 as B: k indexes pixels, columns index (tap, ci)
                int kr = f / (ROWS / 4), q = f % (ROWS / 4);
                int k = k0 + kr;
                if (c_valid[i] && k < k_end) {
                    int nbase, ih0, iw0; bool rv;
                    decode_row(g, k, K, nbase, ih0, iw0, rv);
                    if (vec) {
                        bool ok;
                        long off = gather_pixel(g, nbase, ih0, iw0, c_kh[i], c_kw[i], ok);
                        if (ok) v = *reinterpret_cast<const f32x4*>(base + off + c_ci[i]);
                    } else {
                        int col = row0 + q * 4;
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                            int cc = col + e;
                            if (cc < dim) {
                                int tap = cc / g.Cin, ci = cc % g.Cin;
                                bool ok;
                                long off = gather_pixel(g, nbase, ih0, iw0, tap / g.KW, tap % g.KW, ok);
                                if (ok) v[e] = base[off + ci];
                            }
                        }
                    }
                }
            }
            regs[i] = v;
        }
    }

    // write the registers into the K-major LDS image  S[k][row]
    __device__ __forceinline__ void store(float* S, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int f = tid + i * NTHREADS;
            if (f < TOTAL_V) {
                if constexpr (KCONTIG) {
                    int row = f / KQ, kq = f % KQ;
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[(kq * 4 + e) * LDS_LD + row] = regs[i][e];
                } else {
                    int kr = f / (ROWS / 4), q = f % (ROWS / 4);
                    *reinterpret_cast<f32x4*>(&S[kr * LDS_LD + q * 4]) = regs[i];
                }
            }
        }
    }
};

template <int BM, int BN, int WGM, int WGN, int AMODE, int BMODE>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const ud_gemm_desc d, int tiles_m, int tiles_n, int a_vec,
                                                        int b_vec) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile");
    using LA = Loader<BM, AMODE, false>;
    using LB = Loader<BN, BMODE, true>;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LA::LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LB::LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware rasterisation (speed only, never correctness): workgroups are dealt round-robin over the 8
    // XCDs, each with a private 4 MiB L2.  The 32 consecutive workgroups that land on one XCD get a 4 x 8
    // patch of output tiles, so they share 4 A-panels and 8 B-panels through that L2 instead of 32 + 32.
    int tile_m, tile_n;
    if (gridDim.x == (unsigned)(tiles_m * tiles_n)) {      // small grid: plain order (launcher decides)
        tile_m = blockIdx.x % tiles_m;
        tile_n = blockIdx.x / tiles_m;
    } else {
        const int patches_m = (tiles_m + PATCH_M - 1) / PATCH_M;
        const int patches_n = (tiles_n + PATCH_N - 1) / PATCH_N;
        const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
        const int patch = (i / (PATCH_M * PATCH_N)) * 8 + xcd, within = i % (PATCH_M * PATCH_N);
        if (patch >= patches_m * patches_n) return;
        tile_m = (patch % patches_m) * PATCH_M + (within % PATCH_M);
        tile_n = (patch / patches_m) * PATCH_N + (within / PATCH_M);
        if (tile_m >= tiles_m || tile_n >= tiles_n) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y, bz = blockIdx.z;

    // K range of this split
    const int kt_total = (d.K + BK - 1) / BK;
    const int kt_per = (kt_total + d.split_k - 1) / d.split_k;
    const int k_begin = split * kt_per * BK;
    int k_end = k_begin + kt_per * BK;
    if (k_end > d.K) k_end = d.K;
    const int nkt = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;

    const float* Ap = d.A + (long)bz * d.strideA;
    const float* Bp = d.B + (long)bz * d.strideB;
    float* Cp = d.C + (long)bz * d.strideC + (d.out_mode == 3 ? (long)split * d.slice_stride : 0L);   // mode 3: own slice

    LA la; LB lb;
    la.init(Ap, d.lda, d.M, d.K, (a_vec & 1) != 0, d.g, m0, tid);
    lb.init(Bp, d.ldb, d.N, d.K, (b_vec & 1) != 0, d.g, n0, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nkt > 0) {
        la.load(m0, k_begin, k_end, tid);
        lb.load(n0, k_begin, k_end, tid);
        la.store(As[0], tid);
        lb.store(Bs[0], tid);
    }
    __syncthreads();

    const int a_col = wm * (TM * 32) + l31;
    const int b_col = wn * (TN * 32) + l31;
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) {
            la.load(m0, k_begin + (kt + 1) * BK, k_end, tid);
            lb.load(n0, k_begin + (kt + 1) * BK, k_end, tid);
        }
        const float* Ab = As[cur];
        const float* Bb = Bs[cur];
        // The MFMA operands of k-pair p + FD are read from LDS while k-pair p is multiplied (a ring of FD
        // fragment sets, statically indexed): without it every pair of MFMAs waited out a full LDS round trip
        // (rocprofv3: matrix pipe 43 % busy, SQ_WAIT_INST_ANY 58 %).
        constexpr int NP = BK / 2, FD = 4;
        float fa[FD][TM], fb[FD][TN];
#pragma unroll
        for (int p = 0; p < FD; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = Ab[(2 * p + half) * LA::LDS_LD + a_col + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[p][j] = Bb[(2 * p + half) * LB::LDS_LD + b_col + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int s = p % FD;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
            if (p + FD < NP) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = Ab[(2 * (p + FD) + half) * LA::LDS_LD + a_col + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[s][j] = Bb[(2 * (p + FD) + half) * LB::LDS_LD + b_col + j * 32];
                // scheduling barrier: keeps these ds_reads HERE (hipcc otherwise sinks each read down to its
                // first use, which re-serialises read -> wait -> MFMA)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            la.store(As[cur ^ 1], tid);
            lb.store(Bs[cur ^ 1], tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: D[i][j], j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (nkt == 0 && (d.out_mode == 1 || d.out_mode == 2)) return;          // (modes 0 / 3 store the zeros)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < d.M && col < d.N) {
                    float* p = Cp + (long)row * d.ldc + col;
                    float v = acc[i][j][r];
                    if (d.out_mode == 0 || d.out_mode == 3) *p = v;
                    else if (d.out_mode == 1) *p += v;
                    else atomicAdd(p, v);
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int AMODE, int BMODE>
int launch_cfg(const ud_gemm_desc& d, int a_vec, int b_vec, hipStream_t s) {
    int tiles_m = ud_cdiv(d.M, BM), tiles_n = ud_cdiv(d.N, BN);
    // XCD patch order only pays on grids of >= 4 waves of tiles whose padding to whole patches stays small:
    // measured (tools/bench_gemm.py) 1280x3264x3264 = 510 tiles pads to 768 workgroups and drops from 94 to
    // 72 TFLOP/s when patched, while 4096^3 (1024 tiles, no padding) gains ~3 %
    unsigned gx = (unsigned)(tiles_m * tiles_n);
    if (tiles_m * tiles_n >= 1024) {
        int patches = ud_cdiv(tiles_m, PATCH_M) * ud_cdiv(tiles_n, PATCH_N);
        unsigned padded = (unsigned)(ud_cdiv(patches, 8) * 8 * PATCH_M * PATCH_N);
        if (padded != gx && padded <= gx + gx / 16) gx = padded;   // (equal sizes read as 'plain order': fine too)
    }
    dim3 grid(gx, (unsigned)d.split_k, (unsigned)d.batch);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WGM, WGN, AMODE, BMODE>), grid, dim3(NTHREADS), 0, s, d, tiles_m, tiles_n,
                       a_vec, b_vec);
    UD_LAUNCH_CHECK();
    return 0;
}

// Tile choice.  The fp32 matrix pipe is the bound, so a launch lasts about
//   ceil(tiles / 256 CUs) x (padded work of one tile) x (relative inefficiency of that tile shape).
// E.g. M=1280, N=3264 is 260 tiles of 128x128 (two rounds, 51 % of the chip) but 510 tiles of 128x64 (one
// round of two co-resident blocks per CU); a weight gradient with Cout = 3..32 rows wants a 32-row tile.
struct TileCfg { int bm, bn; double penalty; };
// penalties fitted to tools/bench_gemm.py on MI355X: the half-size tiles run as fast per flop as 128x128 at
// 4096^3 and up to 25 % faster on the mid-size shapes of this model (more workgroups per CU hide the loads)
constexpr TileCfg kCfgs[5] = {{128, 128, 1.00}, {128, 64, 0.93}, {256, 32, 1.10}, {32, 256, 1.10}, {64, 128, 0.93}};

inline int choose_cfg(const ud_gemm_desc& d) {
    static const int forced = [] {            // tuning aid: UD_GEMM_CFG=0..4 pins the tile configuration
        const char* e = getenv("UD_GEMM_CFG");
        return (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : -1;
    }();
    if (forced >= 0) return forced;
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 5; ++i) {
        const TileCfg& c = kCfgs[i];
        long tiles = (long)ud_cdiv(d.M, c.bm) * ud_cdiv(d.N, c.bn) * d.split_k * d.batch;
        long rounds = (tiles + 255) / 256;
        double cost = (double)rounds * c.bm * c.bn * c.penalty;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

template <int AMODE, int BMODE>
int launch_modes(const ud_gemm_desc& d, int a_vec, int b_vec, hipStream_t s) {
    switch (choose_cfg(d)) {
        case 1: return launch_cfg<128, 64, 2, 2, AMODE, BMODE>(d, a_vec, b_vec, s);
        case 2: return launch_cfg<256, 32, 4, 1, AMODE, BMODE>(d, a_vec, b_vec, s);
        case 3: return launch_cfg<32, 256, 1, 4, AMODE, BMODE>(d, a_vec, b_vec, s);
        case 4: return launch_cfg<64, 128, 2, 2, AMODE, BMODE>(d, a_vec, b_vec, s);
        default: return launch_cfg<128, 128, 2, 2, AMODE, BMODE>(d, a_vec, b_vec, s);
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 0 auto, 1 fp32 MFMA only (v_mfma_f32_32x32x2_f32), 2 split-bf16 wherever eligible (gemm_x3.hip),
// 3 mixed precision: fp16 MFMA operands + fp32 accumulation wherever the split-bf16 kernel would run (BASELINE configs[4])
std::atomic<int> g_path{[] {
    const char* e = getenv("UD_GEMM_PATH");
    return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 0;
}()};

}  // namespace

extern "C" int ud_gemm_set_path(int path) {
    if (path < 0 || path > 3) return UD_EINVAL;
    g_path.store(path);
    return 0;
}

extern "C" int ud_gemm_get_path(void) { return g_path.load(); }

// vector-load eligibility of the two operands
static void vec_flags(const ud_gemm_desc& d, int& a_vec, int& b_vec) {
    if (d.a_mode == 0) a_vec = aligned16(d.A) && d.lda % 4 == 0 && d.strideA % 4 == 0;
    else if (d.a_mode == 1) a_vec = aligned16(d.A) && d.lda % 4 == 0 && d.strideA % 4 == 0;
    else a_vec = aligned16(d.A) && d.g.Cin % 4 == 0;
    if (d.b_mode == 0) b_vec = aligned16(d.B) && d.ldb % 4 == 0 && d.strideB % 4 == 0;
    else if (d.b_mode == 1) b_vec = aligned16(d.B) && d.ldb % 4 == 0 && d.strideB % 4 == 0;
    else b_vec = aligned16(d.B) && d.g.Cin % 4 == 0;
}

// the split-bf16 kernel takes every eligible plain GEMM but the tiny shapes.  The thin early-stage pointwise convs
// (K or N = 24..56) are HBM-bound either way and stream better through its 3-deep register prefetch (A/B on the bench:
// minimum dimension 64 -> 16: 38.33 -> 37.85 ms/step; 1: no further change)
static bool takes_x3(const ud_gemm_desc& d, int a_vec, int b_vec) {
    if (d.half_mask) return ud_gemm_x3_eligible(d, a_vec != 0, b_vec != 0);     // ud_gemm rejects the others
    const int path = g_path.load();
    if (path == 1 || !ud_gemm_x3_eligible(d, a_vec != 0, b_vec != 0)) return false;
    constexpr int min_dim = 16;
    return path == 2 || path == 3 || (d.M >= min_dim && d.N >= min_dim && d.K >= min_dim);
}

// 2: this descriptor runs on the BF16 matrix pipe (gemm_x3.hip: 6 v_mfma_f32_32x32x16_bf16 per fp32 product tile),
// 3: the same kernel with ONE v_mfma_f32_32x32x16_f16 per product tile (path 3), 1: on the fp32 pipe (v_mfma_f32_32x32x2_f32) — what bench.py prices the executed MFMA work with
extern "C" int ud_gemm_query_path(const ud_gemm_desc* dp) {
    if (!dp) return UD_EINVAL;
    int a_vec = 0, b_vec = 0;
    vec_flags(*dp, a_vec, b_vec);
    return takes_x3(*dp, a_vec, b_vec) ? ((g_path.load() == 3 || dp->half_mask) ? 3 : 2) : 1;
}

// Epilogue statistics (ud_gemm_desc.stat_sum): only the split-bf16 kernel's plain-store epilogue sees whole column sums.
static bool stats_ok(const ud_gemm_desc& d, int a_vec, int b_vec) {
    return d.out_mode == 0 && d.split_k == 1 && d.batch == 1 && takes_x3(d, a_vec, b_vec);
}

extern "C" int ud_gemm_stats_slots(const ud_gemm_desc* dp) {
    if (!dp) return UD_EINVAL;
    int a_vec = 0, b_vec = 0;
    vec_flags(*dp, a_vec, b_vec);
    if (!stats_ok(*dp, a_vec, b_vec)) return 0;
    return ud_cdiv(dp->M, ud_gemm_x3_tile_rows(*dp)) > 64 ? 64 : 1;
}

extern "C" int ud_gemm(const ud_gemm_desc* dp, ud_stream_t stream) {
    if (!dp) return UD_EINVAL;
    ud_gemm_desc d = *dp;
    if (d.M <= 0 || d.N <= 0 || d.K < 0 || d.split_k < 1 || d.batch < 1) return UD_EINVAL;
    if (d.out_mode < 0 || d.out_mode > 3) return UD_EINVAL;
    if (d.split_k > 1 && d.out_mode != 2 && d.out_mode != 3) return UD_EINVAL;
    if (d.out_mode == 3 && (d.batch != 1 || d.slice_stride < (long)d.M * d.ldc || (d.half_mask & 4))) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int a_vec = 0, b_vec = 0;
    vec_flags(d, a_vec, b_vec);
    if (d.a_mode == 2 || d.b_mode == 2) {
        const ud_conv_geom& g = d.g;
        if (g.N <= 0 || g.Hin <= 0 || g.Win <= 0 || g.Cin <= 0 || g.Hout <= 0 || g.Wout <= 0 || g.KH <= 0 ||
            g.KW <= 0 || g.stride <= 0)
            return UD_EINVAL;
        long rows = (long)g.N * g.Hout * g.Wout, cols = (long)g.KH * g.KW * g.Cin;
        if (d.a_mode == 2 && (rows != d.M || cols != d.K)) return UD_EINVAL;
        if (d.b_mode == 2 && (rows != d.K || cols != d.N)) return UD_EINVAL;
    }
    if (d.stat_sum && (!d.stat_sumsq || !stats_ok(d, a_vec, b_vec))) return UD_EINVAL;
    if (d.half_mask) {
        // half-stored operands exist only on the fp16-MFMA kernel: plain modes, one batch, no atomics onto a half result
        if ((d.half_mask & ~7) || d.batch != 1 || d.a_mode > 1 || d.b_mode > 1 || !takes_x3(d, a_vec, b_vec) ||
            ((d.half_mask & 4) && d.out_mode == 2))
            return UD_EINVAL;
        return ud_gemm_x3_launch_half(d, s);
    }
    if (takes_x3(d, a_vec, b_vec)) return ud_gemm_x3_launch(d, s, g_path.load() == 3);
    if (d.a_mode == 0 && d.b_mode == 0) return launch_modes<0, 0>(d, a_vec, b_vec, s);
    if (d.a_mode == 0 && d.b_mode == 1) return launch_modes<0, 1>(d, a_vec, b_vec, s);
    if (d.a_mode == 1 && d.b_mode == 1) return launch_modes<1, 1>(d, a_vec, b_vec, s);
    if (d.a_mode == 2 && d.b_mode == 0) return launch_modes<2, 0>(d, a_vec, b_vec, s);
    if (d.a_mode == 1 && d.b_mode == 2) return launch_modes<1, 2>(d, a_vec, b_vec, s);
    return UD_EINVAL;
}
