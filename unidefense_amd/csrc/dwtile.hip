// LDS-tiled depthwise k x k convolution, stride 1 (k = 3, 5), on pixel-major [N][H][W][C] activations: forward, data
// gradient and weight gradient of the spatial branch of SFConv2dStaticSamePadding.forward (model/efficientnet/exp.py:49-51)
// and of the plain depthwise Conv2dStaticSamePadding (model/efficientnet/utils.py:277-280) inside MBConvBlock.forward
// (model/efficientnet/model.py:112-115).
//
// Why tiles: the strip kernels of dwconv.hip / fused.hip keep one (pixel strip, channel quad) per thread and re-read the
// k x k window through L1 / L2 — every input element is fetched k * (strip + k - 1) / strip times, the 25 taps of a 5 x 5
// window are re-loaded per strip row, and a deferred BatchNorm + swish on the input would be re-evaluated per tap (which
// is why the unfused path MATERIALISES swish(bn0(e)) for the plain blocks).  Here a workgroup stages a (TH + k - 1) x
// (TW + k - 1) halo tile of 32 channels in LDS ONCE — applying act(bn(x)) while staging, so the activated tensor never
// exists in HBM — and every thread (channel quad, strip of SW output columns) slides its window over LDS:
//   * forward      out = conv(act(bn(src)))                      [+ per-channel sum / sum of squares of out: BN1 statistics]
//   * data grad    da  = gate * conv_flipped(dy) [+ add];  dz = da * act'(bn(x))   [+ BatchNorm backward sums of dz]
//   * weight grad  dw[tap][c] = sum_pixels act(bn(src))[.. + tap] * dy           (fp32 partials per workgroup -> finalize)
// Layout: 256 threads = CQ channel quads x 256 / CQ pixel-threads — (8, 32): 16 x 16 output tiles of 32 channels (128
// contiguous bytes per pixel) for the maps of 16 x 16 and more, (16, 16): the whole 8 x 8 map of 64 channels; LDS holds fp32
// whatever the storage type; the padded row pitch (odd number of pixels) keeps the 16 lanes of a ds_read_b128 phase on
// distinct banks.  All global loads of the staging phase are issued before the first one is used (address-clamped, no
// branch around a load), and the epilogue's operands (added gradient, the BatchNorm's input) are fetched before the
// window loop, so that their latency hides behind the LDS traffic.  HBM-bound for k = 3; k = 5 (25 FMAs per element) is VALU / LDS-bound at
// ~60 % of the HBM roofline.
// Stride 2 (the four down-sampling blocks): forward and weight gradient step their window by 2 over an 8 x 8 output tile
// (halo 19 x 19 at k = 5); the data gradient runs the stride-1 kernel over a ZERO-STUFFED tile of dy (source pixel v of the
// up-sampled grid holds dy(v / 2) when v is even, else 0), which also folds the BatchNorm backward sums the strided path used
// to take in a pass of its own.
#include "bnref.h"

namespace {

struct TileGeom {
    int N, Hs, Ws, C4, Ho, Wo;             // source / output extents (stride 1)
    int P_t, P_l;                          // out(oh, ow) reads src(oh + i - P_t, ow + j - P_l), i, j in [0, K)
    int flip;                              // 1: tap of (i, j) is (K-1-i, K-1-j) (data gradient)
    int tiles_h, tiles_w;
    int up;                                // 1: the source is dy of a stride-2 conv read through a zero-stuffed grid
};

// A workgroup = CQ channel quads x PTH = 256 / CQ pixel-threads; every pixel-thread owns a strip of SW output columns of
// one tile row: (CQ, SW) = (8, 8): 16 x 16 outputs, (16, 4): 8 x 8 outputs.
template <int CQ, int SW> struct TileShape {
    static constexpr int PTH = NT / CQ;
    static constexpr int TW_ = SW == 8 ? 16 : 8;
    static constexpr int TH_ = PTH * SW / TW_;
};

template <int K, int CQ, int SW, int ST = 1> struct Lds {
    using S = TileShape<CQ, SW>;
    static constexpr int ROWS = (S::TH_ - 1) * ST + K;
    static constexpr int COLS = (S::TW_ - 1) * ST + K;
    static constexpr int PITCH = (COLS | 1);           // pixels per LDS row, odd: rows r and r + 1 differ by half a bank sweep
    static constexpr int TILE_Q = ROWS * PITCH * CQ;   // f32x4 slots of the halo tile
    static constexpr int W_Q = K * K * CQ;             // f32x4 slots of the tap table
    static constexpr int NPIX = ROWS * COLS;
    static constexpr int NL = (NPIX + S::PTH - 1) / S::PTH;      // staged pixels per thread
};

// Stage act(bn(src)) of the halo tile (zero outside the image).  Two phases: every load first (branch-free: an invalid
// element re-reads a valid address and is zeroed afterwards), then transform + LDS store.
template <typename T, int K, int CQ, int SW, int ST>
__device__ __forceinline__ void stage_tile(const TileGeom& g, const T* __restrict__ src, const ud_bn_ref& bn, bool has_bn,
                                           const Bn4& cb, int n, int oh0, int ow0, int cq0, f32x4* tile) {
    using L = Lds<K, CQ, SW, ST>;
    constexpr int PTH = L::S::PTH;
    const In4<T> s4{src};
    const int cq = threadIdx.x % CQ, p0 = threadIdx.x / CQ;
    const bool cok = cq0 + cq < g.C4;
    const int c4 = cok ? cq0 + cq : 0;
    const long img = (long)n * g.Hs * g.Ws;
    f32x4 v[L::NL];
    unsigned okmask = 0;
#pragma unroll
    for (int l = 0; l < L::NL; ++l) {
        const int p = p0 + l * PTH;
        const int r = p / L::COLS, c = p % L::COLS;
        int ih = oh0 * ST + r - g.P_t, iw = ow0 * ST + c - g.P_l;
        bool par = true;
        if (g.up) {                                             // zero-stuffed grid: only even positions hold a sample
            par = ((ih | iw) & 1) == 0;
            ih >>= 1;
            iw >>= 1;
        }
        const bool ok = cok && par && p < L::NPIX && ih >= 0 && ih < g.Hs && iw >= 0 && iw < g.Ws;
        const long pix = ok ? img + (long)ih * g.Ws + iw : img;
        v[l] = s4[pix * g.C4 + c4];
        okmask |= (ok ? 1u : 0u) << l;
    }
#pragma unroll
    for (int l = 0; l < L::NL; ++l) {
        const int p = p0 + l * PTH;
        if (p < L::NPIX) {
            f32x4 a = v[l];
            if (has_bn) a = bn_apply(a, cb, bn.act);
            if (!((okmask >> l) & 1u)) a = f32x4{0, 0, 0, 0};
            tile[((p / L::COLS) * L::PITCH + p % L::COLS) * CQ + cq] = a;
        }
    }
}

template <int K, int CQ>
__device__ __forceinline__ void stage_taps(const float* __restrict__ wt, int C4, int cq0, int flip, f32x4* taps) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wt);
    for (int i = threadIdx.x; i < K * K * CQ; i += NT) {
        const int tap = i / CQ, cq = i % CQ;
        const int src_tap = flip ? (K * K - 1 - tap) : tap;          // (K-1-i) * K + (K-1-j) = K*K - 1 - (i * K + j)
        taps[i] = (cq0 + cq < C4) ? w4[(long)src_tap * C4 + cq0 + cq] : f32x4{0, 0, 0, 0};
    }
}

// EPI 0: plain store.  EPI 1: + sum / sum of squares of the stored result (forward statistics).
// EPI 2: data gradient: da = gate * acc [+ add]; with a BatchNorm behind it (has_bn_out): dz = da * act'(bn(x)), sums of dz
//        and dz * xhat.
// waves per SIMD the register allocation has to leave room for: what the LDS tile allows (57 KB at K = 5 / 16 x 16: two
// workgroups per CU; 33 - 44 KB otherwise: three or four) — without the bound hipcc hoists every LDS read of the unrolled
// window loop and takes all 256 VGPRs
template <int K, int SW> constexpr int kMinWaves = (K == 5 && SW == 8) ? 2 : (SW == 4 ? 4 : 3);

template <typename T, int K, int CQ, int SW, int EPI, int ST = 1>
__global__ __launch_bounds__(NT, (kMinWaves<K, SW>)) void dw_tile_kernel(TileGeom g, const T* __restrict__ src, ud_bn_ref bn_in, int has_bn_in,
                                                     const float* __restrict__ wt, T* __restrict__ out,
                                                     const float* __restrict__ gate_alpha, int gate_mode,
                                                     const T* __restrict__ add, const T* __restrict__ xbn,
                                                     ud_bn_ref bn_out, int has_bn_out, double* __restrict__ part,
                                                     double* __restrict__ s1, double* __restrict__ s2) {
    using L = Lds<K, CQ, SW, ST>;
    using S = TileShape<CQ, SW>;
    constexpr int PTH = S::PTH;
    static_assert(ST == 1 || EPI != 2, "the strided data gradient runs the stride-1 kernel over a zero-stuffed tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* tile = reinterpret_cast<f32x4*>(smem);
    f32x4* taps = tile + L::TILE_Q;
    const int cq = threadIdx.x % CQ, p = threadIdx.x / CQ;
    const int cq0 = blockIdx.y * CQ;
    const int c4 = cq0 + cq;
    const bool cok = c4 < g.C4;
    int t = blockIdx.x;
    const int tw_i = t % g.tiles_w;
    t /= g.tiles_w;
    const int th_i = t % g.tiles_h, n = t / g.tiles_h;
    const int oh0 = th_i * S::TH_, ow0 = tw_i * S::TW_;
    // strip of this thread: row p % TH, columns (p / TH) * SW ..
    const int row = p % S::TH_, col0 = (p / S::TH_) * SW;
    const int oh = oh0 + row;
    const long obase = (((long)n * g.Ho + (oh < g.Ho ? oh : 0)) * g.Wo) * g.C4 + (cok ? c4 : 0);
    // epilogue operands first: their latency hides behind the staging and the window loop
    const In4<T> add4{add}, x4{xbn};
    f32x4 pre_add[EPI == 2 ? SW : 1], pre_x[EPI == 2 ? SW : 1];
    if (EPI == 2) {
#pragma unroll
        for (int o = 0; o < SW; ++o) {
            int ow = ow0 + col0 + o;
            if (ow >= g.Wo) ow = g.Wo - 1;                      // clamped: stored only when in range
            if (add) pre_add[o] = add4[obase + (long)ow * g.C4];
            if (has_bn_out) pre_x[o] = x4[obase + (long)ow * g.C4];
        }
    }
    Bn4 cbi;
    if (has_bn_in && cok) cbi = bn_load(bn_in, 0, g.C4, c4, blockIdx.x == 0 && p == 0);
    stage_tile<T, K, CQ, SW, ST>(g, src, bn_in, has_bn_in != 0, cbi, n, oh0, ow0, cq0, tile);
    stage_taps<K, CQ>(wt, g.C4, cq0, g.flip, taps);
    __syncthreads();

    constexpr int NIN = (SW - 1) * ST + K;
    f32x4 acc[SW];
#pragma unroll
    for (int i = 0; i < SW; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const f32x4* rowp = tile + ((row * ST + i) * L::PITCH + col0 * ST) * CQ + cq;
        f32x4 in[NIN], w[K];
#pragma unroll
        for (int j = 0; j < NIN; ++j) in[j] = rowp[j * CQ];
#pragma unroll
        for (int j = 0; j < K; ++j) w[j] = taps[(i * K + j) * CQ + cq];
#pragma unroll
        for (int o = 0; o < SW; ++o)
#pragma unroll
            for (int j = 0; j < K; ++j) acc[o] += in[o * ST + j] * w[j];
        // one window row at a time: pin the running sums here, or hipcc sinks every multiply-add below the LDS reads of
        // all K rows (K * (SW + K - 1) quads live at once) and spills
#pragma unroll
        for (int o = 0; o < SW; ++o) asm volatile("" : "+v"(acc[o]));
    }

    const Out4<T> o4{out};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cok && oh < g.Ho) {
        Bn4 cbo;
        float gs = 1.f;
        if (EPI == 2) {
            gs = gate_factor(gate_alpha, gate_mode);
            if (has_bn_out) cbo = bn_load(bn_out, 0, g.C4, c4, false);
        }
#pragma unroll
        for (int o = 0; o < SW; ++o) {
            const int ow = ow0 + col0 + o;
            if (ow >= g.Wo) continue;
            const long idx = obase + (long)ow * g.C4;
            f32x4 r = acc[o];
            if (EPI == 2) {
                r = r * gs;
                if (add) r += pre_add[o];
                if (has_bn_out) {
                    const f32x4 a = pre_x[o];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (a[e] - cbo.mu[e]) * cbo.is[e];
                        float d = r[e];
                        if (bn_out.act) d *= ud_act_grad_fast(cbo.ga[e] * xh + cbo.be[e], bn_out.act);
                        d = ud_rounded<T>(d);
                        r[e] = d;
                        v[e] += (double)d;
                        v[4 + e] += (double)d * (double)xh;
                    }
                }
            } else if (EPI == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = (double)ud_rounded<T>(r[e]);
                    v[e] += d;
                    v[4 + e] += d * d;
                }
            }
            o4.st(idx, r);
        }
    }
    if (EPI == 1 || (EPI == 2 && has_bn_out)) {
        // fold the pixel-threads of every channel quad: LDS (the tile is dead), then one fp64 partial per (tile, channel)
        __syncthreads();
        double* sm = reinterpret_cast<double*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[(e * PTH + p) * CQ + cq] = v[e];
        __syncthreads();
        if (threadIdx.x < 8 * CQ) {
            const int e = threadIdx.x / CQ, q = threadIdx.x % CQ;
            double s = 0.0;
            for (int k = 0; k < PTH; ++k) s += sm[(e * PTH + k) * CQ + q];
            if (cq0 + q < g.C4) {
                const long C = (long)g.C4 * 4, P = gridDim.x;
                if (part) {
                    // part[quantity][tile][C]: quantity 0 = first sum, 1 = second; tile = blockIdx.x
                    part[((long)(e / 4) * P + blockIdx.x) * C + (cq0 + q) * 4 + e % 4] = s;
                } else {
                    // few tiles per channel (the 8 x 8 / 16 x 16 maps): straight into the accumulator, no finalize launch
                    atomic_add_f64((e < 4 ? s1 : s2) + (cq0 + q) * 4 + e % 4, s);
                }
            }
        }
    }
}

// Weight gradient: acc[tap] += src_tile[.. + tap] * dy over the workgroup's tiles (images n0, n0 + nstep, ...), folded over
// its pixel-threads, one fp32 partial row [K*K][C] per workgroup for dw_tile_wgrad_finalize.
template <typename T, int K, int CQ, int SW, int ST = 1>
__global__ __launch_bounds__(NT, 2) void dw_tile_wgrad_kernel(TileGeom g, const T* __restrict__ src, ud_bn_ref bn_in,
                                                           int has_bn_in, const T* __restrict__ dy, int n_step,
                                                           float* __restrict__ part) {
    using L = Lds<K, CQ, SW, ST>;
    using S = TileShape<CQ, SW>;
    constexpr int NIN = (SW - 1) * ST + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* tile = reinterpret_cast<f32x4*>(smem);
    const int cq = threadIdx.x % CQ, p = threadIdx.x / CQ;
    const int cq0 = blockIdx.y * CQ;
    const int c4 = cq0 + cq;
    const bool cok = c4 < g.C4;
    const int tiles = g.tiles_h * g.tiles_w;
    const int tsp = blockIdx.x % tiles, nb = blockIdx.x / tiles;        // spatial tile, first image
    const int oh0 = (tsp / g.tiles_w) * S::TH_, ow0 = (tsp % g.tiles_w) * S::TW_;
    const int row = p % S::TH_, col0 = (p / S::TH_) * SW;
    const int oh = oh0 + row;
    const In4<T> dy4{dy};
    Bn4 cbi;
    if (has_bn_in && cok) cbi = bn_load(bn_in, 0, g.C4, c4, false);
    f32x4 acc[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int n = nb; n < g.N; n += n_step) {
        // this thread's dy strip (branch-free, zeroed when out of range), then the tile
        f32x4 gy[SW];
        const long gbase = (((long)n * g.Ho + (oh < g.Ho ? oh : 0)) * g.Wo) * g.C4 + (cok ? c4 : 0);
#pragma unroll
        for (int o = 0; o < SW; ++o) {
            const int ow = ow0 + col0 + o;
            gy[o] = dy4[gbase + (long)(ow < g.Wo ? ow : g.Wo - 1) * g.C4];
        }
        __syncthreads();                                       // previous tile fully consumed
        stage_tile<T, K, CQ, SW, ST>(g, src, bn_in, has_bn_in != 0, cbi, n, oh0, ow0, cq0, tile);
#pragma unroll
        for (int o = 0; o < SW; ++o)
            if (!(cok && oh < g.Ho && ow0 + col0 + o < g.Wo)) gy[o] = f32x4{0, 0, 0, 0};
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const f32x4* rowp = tile + ((row * ST + i) * L::PITCH + col0 * ST) * CQ + cq;
            f32x4 in[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) in[j] = rowp[j * CQ];
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int o = 0; o < SW; ++o) acc[i][j] += in[o * ST + j] * gy[o];
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("" : "+v"(acc[i][j]));      // one window row at a time (see dw_tile_kernel)
        }
    }
    // fold the pixel-threads of a wave (the lane bits above the channel quad), then the 4 waves through LDS
    __syncthreads();
    float* sm = reinterpret_cast<float*>(smem);                 // [4 waves][K*K][CQ][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            f32x4 a = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = a[e];
#pragma unroll
                for (int m = CQ; m < 64; m <<= 1) x += __shfl_xor(x, m, 64);
                a[e] = x;
            }
            if (lane < CQ) reinterpret_cast<f32x4*>(sm)[(wave * K * K + i * K + j) * CQ + lane] = a;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K * CQ * 4; i += NT) {
        const int e = i % 4, q = (i / 4) % CQ, tap = i / (4 * CQ);
        if (cq0 + q >= g.C4) continue;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += sm[((w * K * K + tap) * CQ + q) * 4 + e];
        part[((long)blockIdx.x * K * K + tap) * ((long)g.C4 * 4) + (cq0 + q) * 4 + e] = s;
    }
}

// Round 5: data gradient AND weight gradient of a stride-1 depthwise conv in ONE pass over its operands.  The two kernels
// above each stage a halo tile and stream the other operand: dy (+ halo) and x for the data gradient, act(bn(x)) (+ halo) and
// dy for the weight gradient — every element of both tensors crosses the memory system twice, on tensors of 13 - 300 MB.
// Here a workgroup stages BOTH halo tiles once per image — tileA = dy (zero outside the image), tileB = act(bn_in(x)) — and
//   phase 1  da = gate * conv_flipped(tileA) [+ add];  dz = da * act'(bn_in(x)); dz stored; BatchNorm backward sums of dz
//   phase 2  acc_w[i][j] += tileB(window + (i, j)) * tileA(centre)               (the dy strip comes out of the staged tile)
// over images n0, n0 + n_step, ...; one fp32 partial row [K*K][C] per workgroup for dw_tile_wgrad_finalize and one fp64
// partial (or atomics, <= 64 workgroups per channel) of the sums.  Tiles of (CQ, SW) = (8, 4): 16 x 8 outputs of 32 channels,
// two halo tiles = 67 KB at K = 5 (two workgroups per CU); maps up to 8 x 8: (16, 4), the whole map of 64 channels.
template <typename T, int K, int CQ, int SW>
__global__ __launch_bounds__(NT, 2) void dw_tile_bwd_kernel(TileGeom g, const T* __restrict__ dy, const T* __restrict__ x,
                                                         ud_bn_ref bn, int has_bn, const float* __restrict__ wt,
                                                         const float* __restrict__ gate_alpha, int gate_mode,
                                                         const T* __restrict__ add, T* __restrict__ dz, int n_step,
                                                         float* __restrict__ wpart, double* __restrict__ spart,
                                                         double* __restrict__ s1, double* __restrict__ s2) {
    using L = Lds<K, CQ, SW, 1>;
    using S = TileShape<CQ, SW>;
    constexpr int PTH = S::PTH;
    constexpr int NIN = SW - 1 + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* tileA = reinterpret_cast<f32x4*>(smem);
    f32x4* tileB = tileA + L::TILE_Q;
    f32x4* taps = tileB + L::TILE_Q;
    const int cq = threadIdx.x % CQ, p = threadIdx.x / CQ;
    const int cq0 = blockIdx.y * CQ;
    const int c4 = cq0 + cq;
    const bool cok = c4 < g.C4;
    const int tiles = g.tiles_h * g.tiles_w;
    const int tsp = blockIdx.x % tiles, nb = blockIdx.x / tiles;
    const int oh0 = (tsp / g.tiles_w) * S::TH_, ow0 = (tsp % g.tiles_w) * S::TW_;
    const int row = p % S::TH_, col0 = (p / S::TH_) * SW;
    const int oh = oh0 + row;
    const In4<T> add4{add}, x4{x};
    const Out4<T> o4{dz};
    Bn4 cb;
    if (has_bn && cok) cb = bn_load(bn, 0, g.C4, c4, false);
    const float gs = gate_factor(gate_alpha, gate_mode);
    // geometry of the two stagings: the data gradient reads dy through the flipped window (pads K-1 - P), the weight gradient
    // reads act(bn(x)) through the forward window (pads P)
    TileGeom ga = g, gb = g;
    ga.P_t = K - 1 - g.P_t;
    ga.P_l = K - 1 - g.P_l;
    ga.flip = 1;
    stage_taps<K, CQ>(wt, g.C4, cq0, 1, taps);
    // phase 2's decomposition: a pixel-thread owns ONE tap row ti and every GP-th tile row (K accumulator quads instead of the
    // K * K of dw_tile_wgrad_kernel: with the data gradient's window in the same kernel those 100 registers spilled)
    constexpr int GP = PTH / K;                                  // pixel-threads per tap row; PTH - GP * K of them sit phase 2 out
    constexpr int NRW = (S::TH_ + GP - 1) / GP;                  // tile rows per thread, at most
    const int ti = p / GP, pg = p % GP;
    const bool wact = p < GP * K;
    f32x4 accw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) accw[j] = f32x4{0, 0, 0, 0};
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ud_bn_ref none{};
    for (int n = nb; n < g.N; n += n_step) {
        const long obase = (((long)n * g.Ho + (oh < g.Ho ? oh : 0)) * g.Wo) * g.C4 + (cok ? c4 : 0);
        __syncthreads();                                       // the previous image's tiles are consumed
        stage_tile<T, K, CQ, SW, 1>(ga, dy, none, false, cb, n, oh0, ow0, cq0, tileA);
        stage_tile<T, K, CQ, SW, 1>(gb, x, bn, has_bn != 0, cb, n, oh0, ow0, cq0, tileB);
        __syncthreads();
        // ---- phase 1: data gradient of this thread's strip
        {
            f32x4 acc[SW];
#pragma unroll
            for (int o = 0; o < SW; ++o) acc[o] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const f32x4* rowp = tileA + ((row + i) * L::PITCH + col0) * CQ + cq;
                f32x4 in[NIN], w[K];
#pragma unroll
                for (int j = 0; j < NIN; ++j) in[j] = rowp[j * CQ];
#pragma unroll
                for (int j = 0; j < K; ++j) w[j] = taps[(i * K + j) * CQ + cq];
#pragma unroll
                for (int o = 0; o < SW; ++o)
#pragma unroll
                    for (int j = 0; j < K; ++j) acc[o] += in[o + j] * w[j];
#pragma unroll
                for (int o = 0; o < SW; ++o) asm volatile("" : "+v"(acc[o]));          // one window row at a time (see dw_tile_kernel)
            }
            // the epilogue's operands only now: fetched before the staging they would sit in 32 VGPRs across both stagings and
            // the window loop, next to the 25 tap accumulators of phase 2 (K = 5: 70 - 100 spilled registers); x was read by the
            // staging a moment ago, so these loads hit the L2
            f32x4 pre_add[SW], pre_x[SW];
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                int ow = ow0 + col0 + o;
                if (ow >= g.Wo) ow = g.Wo - 1;
                if (add) pre_add[o] = add4[obase + (long)ow * g.C4];
                if (has_bn) pre_x[o] = x4[obase + (long)ow * g.C4];
            }
            if (cok && oh < g.Ho) {
#pragma unroll
                for (int o = 0; o < SW; ++o) {
                    const int ow = ow0 + col0 + o;
                    if (ow >= g.Wo) continue;
                    f32x4 r = acc[o] * gs;
                    if (add) r += pre_add[o];
                    if (has_bn) {
                        const f32x4 a = pre_x[o];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xh = (a[e] - cb.mu[e]) * cb.is[e];
                            float d = r[e];
                            if (bn.act) d *= ud_act_grad_fast(cb.ga[e] * xh + cb.be[e], bn.act);
                            d = ud_rounded<T>(d);
                            r[e] = d;
                            v[e] += (double)d;
                            v[4 + e] += (double)d * (double)xh;
                        }
                    }
                    o4.st(obase + (long)ow * g.C4, r);
                }
            }
        }
        // ---- phase 2: weight gradient of tap row ti over tile rows pg, pg + GP, ...; dy is the centre of tileA (zero outside
        // the image), in strips of 4 columns
        if (wact) {
#pragma unroll
            for (int rr = 0; rr < NRW; ++rr) {
                const int r = pg + rr * GP;
                if (r >= S::TH_) break;
#pragma unroll
                for (int h = 0; h < S::TW_ / 4; ++h) {
                    const f32x4* cp = tileA + ((r + ga.P_t) * L::PITCH + h * 4 + ga.P_l) * CQ + cq;
                    const f32x4* rowp = tileB + ((r + ti) * L::PITCH + h * 4) * CQ + cq;
                    f32x4 gy[4], in[3 + K];
#pragma unroll
                    for (int o = 0; o < 4; ++o) gy[o] = cp[o * CQ];
#pragma unroll
                    for (int j = 0; j < 3 + K; ++j) in[j] = rowp[j * CQ];
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int o = 0; o < 4; ++o) accw[j] += in[o + j] * gy[o];
#pragma unroll
                    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(accw[j]));
                }
            }
        }
    }
    // ---- fold the weight-gradient partials of a tap over its GP pixel-threads through LDS: [p][j][cq] quads
    __syncthreads();
    {
        f32x4* sm = reinterpret_cast<f32x4*>(smem);
        if (wact) {
#pragma unroll
            for (int j = 0; j < K; ++j) sm[(p * K + j) * CQ + cq] = accw[j];
        }
        __syncthreads();
        const float* smf = reinterpret_cast<const float*>(smem);
        for (int i = threadIdx.x; i < K * K * CQ * 4; i += NT) {
            const int e = i % 4, q = (i / 4) % CQ, tap = i / (4 * CQ);
            if (cq0 + q >= g.C4) continue;
            const int ti_ = tap / K, j = tap % K;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < GP; ++k) t += smf[(((ti_ * GP + k) * K + j) * CQ + q) * 4 + e];
            wpart[((long)blockIdx.x * K * K + tap) * ((long)g.C4 * 4) + (cq0 + q) * 4 + e] = t;
        }
    }
    if (has_bn) {
        __syncthreads();
        double* sm = reinterpret_cast<double*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[(e * PTH + p) * CQ + cq] = v[e];
        __syncthreads();
        if (threadIdx.x < 8 * CQ) {
            const int e = threadIdx.x / CQ, q = threadIdx.x % CQ;
            double t = 0.0;
            for (int k = 0; k < PTH; ++k) t += sm[(e * PTH + k) * CQ + q];
            if (cq0 + q < g.C4) {
                const long C = (long)g.C4 * 4, P = gridDim.x;
                if (spart) spart[((long)(e / 4) * P + blockIdx.x) * C + (cq0 + q) * 4 + e % 4] = t;
                else atomic_add_f64((e < 4 ? s1 : s2) + (cq0 + q) * 4 + e % 4, t);
            }
        }
    }
}

// dw[c][tap] = gate * sum_p part[p][tap][c]   (fp64 accumulation; the parameter's own layout [C][K*K])
__global__ __launch_bounds__(NT) void dw_tile_wgrad_finalize(int nparts, int KK, int C, const float* __restrict__ part,
                                                             const float* __restrict__ gate_alpha, int gate_mode,
                                                             float* __restrict__ dw) {
    __shared__ double sm[16][16][4];
    const int KKC = KK * C;
    const int lane = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (blockIdx.x * 16 + lane) * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < KKC) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(part + i);
        const long step = (long)KKC / 4;
        for (int p = sl; p < nparts; p += 16) {
            const f32x4 v = p4[(long)p * step];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += (double)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[sl][lane][e] = a[e];
    __syncthreads();
    if (sl < 4 && i < KKC) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][lane][sl];
        const int o = i + sl, tap = o / C, c = o % C;
        dw[(long)c * KK + tap] = (float)(t * (double)gate_factor(gate_alpha, gate_mode));
    }
}

// The same fold for EVERY depthwise conv of a backward pass in one launch (ud_dwtile_wgrad_finalize_multi): the items travel by
// value in the kernel arguments (no device table: the launch sits inside a captured graph), a workgroup finds its item by a scan
// of the (uniform) block prefix.
constexpr int FOLD_MAX = 48;
struct FoldArgs {
    ud_wgrad_fold it[FOLD_MAX];
    int block0[FOLD_MAX + 1];
    int n;
};

__global__ __launch_bounds__(NT) void dw_tile_wgrad_finalize_multi(const FoldArgs a) {
    __shared__ double sm[16][16][4];
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.block0[j + 1]) ++j;
    const ud_wgrad_fold& f = a.it[j];
    const int KK = f.K * f.K, C = f.C, KKC = KK * C, nparts = f.nparts;
    const int lane = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (((int)blockIdx.x - a.block0[j]) * 16 + lane) * 4;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < KKC) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(f.part + i);
        const long step = (long)KKC / 4;
        for (int p = sl; p < nparts; p += 16) {
            const f32x4 w = p4[(long)p * step];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (double)w[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[sl][lane][e] = v[e];
    __syncthreads();
    if (sl < 4 && i < KKC) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][lane][sl];
        const int o = i + sl, tap = o / C, c = o % C;
        f.dwt[(long)c * KK + tap] = (float)(t * (double)gate_factor(f.gate_alpha, f.gate_mode));
    }
}

inline bool tile_args_ok(int N, int Hs, int Ws, int C, int Ho, int Wo, int K) {
    return N > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && C >= 4 && C % 4 == 0 && (K == 3 || K == 5);
}

// maps up to 8 x 8 with at least 64 channels: one 8 x 8 tile of 64 channels; everything else: 16 x 16 tiles of 32 channels
inline bool small_map(int Ho, int Wo, int C) { return Ho <= 8 && Wo <= 8 && C >= 64; }

template <int CQ, int SW> inline void tile_counts(int Ho, int Wo, int& th, int& tw) {
    th = ud_cdiv(Ho, TileShape<CQ, SW>::TH_);
    tw = ud_cdiv(Wo, TileShape<CQ, SW>::TW_);
}

template <typename T, int K, int CQ, int SW, int ST = 1>
int launch_tile(TileGeom g, const T* src, const ud_bn_ref* bn_in, const float* wt, T* out, const float* gate_alpha,
                int gate_mode, const T* add, const T* xbn, const ud_bn_ref* bn_out, int epi, double* s1, double* s2,
                double* ws, hipStream_t s) {
    using L = Lds<K, CQ, SW, ST>;
    tile_counts<CQ, SW>(g.Ho, g.Wo, g.tiles_h, g.tiles_w);
    const long nt = (long)g.N * g.tiles_h * g.tiles_w;
    if (nt > 0x7fffffffL) return UD_EINVAL;
    dim3 grid((unsigned)nt, (unsigned)ud_cdiv(g.C4, CQ));
    size_t lds = (size_t)(L::TILE_Q + L::W_Q) * 16;
    const size_t fold = (size_t)(8 * NT) * 8;
    if (lds < fold) lds = fold;
    ud_bn_ref none{};
    const ud_bn_ref& bi = bn_in ? *bn_in : none;
    const ud_bn_ref& bo = bn_out ? *bn_out : none;
    const bool sums = epi == 1 || (epi == 2 && bn_out);
    // at most 64 tiles add to one address: fp64 atomics (12 ns each, serialised per address) instead of partials + a
    // finalize launch — the rule of fused.hip's plan_reduce
    const bool use_part = sums && nt > 64;
#define UD_TILE(E)                                                                                                    \
    hipLaunchKernelGGL((dw_tile_kernel<T, K, CQ, SW, E, ST>), grid, dim3(NT), lds, s, g, src, bi, bn_in ? 1 : 0, wt,   \
                       out, gate_alpha, gate_mode, add, xbn, bo, bn_out ? 1 : 0, use_part ? ws : nullptr, s1, s2)
    if (epi == 0) UD_TILE(0);
    else if (epi == 1) UD_TILE(1);
    else if constexpr (ST == 1) UD_TILE(2);
    else return UD_EINVAL;
#undef UD_TILE
    UD_LAUNCH_CHECK();
    if (use_part) {
        const int C = g.C4 * 4;
        hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv(C, 8)), dim3(NT), 0, s, 2, 1, C, (int)nt, ws, s1, s2);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

template <typename T, int K, int CQ, int SW, int ST = 1>
int launch_wgrad(TileGeom g, const T* src, const ud_bn_ref* bn_in, const T* dy, const float* gate_alpha, int gate_mode,
                 float* part, long part_rows, float* dw, hipStream_t s) {
    using L = Lds<K, CQ, SW, ST>;
    tile_counts<CQ, SW>(g.Ho, g.Wo, g.tiles_h, g.tiles_w);
    const int tiles = g.tiles_h * g.tiles_w;
    const int cblocks = ud_cdiv(g.C4, CQ);
    // enough workgroups to fill the chip ~4 times over; every workgroup folds N / n_step images before it writes a partial
    int n_step = (int)((1024 + (long)tiles * cblocks - 1) / ((long)tiles * cblocks));
    if (n_step < 1) n_step = 1;
    if (n_step > g.N) n_step = g.N;
    const long nparts = (long)tiles * n_step;
    if (nparts > part_rows) return UD_EINVAL;
    size_t lds = (size_t)L::TILE_Q * 16;
    const size_t fold = (size_t)(NT / 64) * K * K * CQ * 16;
    if (lds < fold) lds = fold;
    ud_bn_ref none{};
    const ud_bn_ref& bi = bn_in ? *bn_in : none;
    dim3 grid((unsigned)nparts, (unsigned)cblocks);
    hipLaunchKernelGGL((dw_tile_wgrad_kernel<T, K, CQ, SW, ST>), grid, dim3(NT), lds, s, g, src, bi, bn_in ? 1 : 0, dy,
                       n_step, part);
    UD_LAUNCH_CHECK();
    if (!dw) return (int)nparts;          // the caller folds the partial rows later (ud_dwtile_wgrad_finalize_multi)
    const int KKC = K * K * g.C4 * 4;
    hipLaunchKernelGGL(dw_tile_wgrad_finalize, dim3(ud_cdiv(KKC, 64)), dim3(NT), 0, s, (int)nparts, K * K, g.C4 * 4, part,
                       gate_alpha, gate_mode, dw);
    UD_LAUNCH_CHECK();
    return 0;
}

template <typename T, int K, int CQ, int SW>
int launch_bwd(TileGeom g, const T* dy, const T* x, const ud_bn_ref* bn, const float* wt, const float* gate_alpha,
               int gate_mode, const T* add, T* dz, float* wpart, long part_rows, float* dwt, double* s1, double* s2,
               double* ws, hipStream_t s) {
    using L = Lds<K, CQ, SW, 1>;
    tile_counts<CQ, SW>(g.Ho, g.Wo, g.tiles_h, g.tiles_w);
    const int tiles = g.tiles_h * g.tiles_w;
    const int cblocks = ud_cdiv(g.C4, CQ);
    // ~1024 workgroups; every workgroup folds N / n_step images before it writes its partials
    int n_step = (int)((1024 + (long)tiles * cblocks - 1) / ((long)tiles * cblocks));
    if (n_step < 1) n_step = 1;
    if (n_step > g.N) n_step = g.N;
    const long nparts = (long)tiles * n_step;
    if (nparts > part_rows) return UD_EINVAL;
    size_t lds = (size_t)(2 * L::TILE_Q + L::W_Q) * 16;
    const size_t fold_w = (size_t)(NT / CQ) * K * CQ * 16, fold_s = (size_t)(8 * NT) * 8;
    if (lds < fold_w) lds = fold_w;
    if (lds < fold_s) lds = fold_s;
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_tile_bwd_kernel<T, K, CQ, SW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    ud_bn_ref none{};
    const bool use_part = bn && nparts > 64;          // the rule of fused.hip's plan_reduce
    dim3 grid((unsigned)nparts, (unsigned)cblocks);
    hipLaunchKernelGGL((dw_tile_bwd_kernel<T, K, CQ, SW>), grid, dim3(NT), lds, s, g, dy, x, bn ? *bn : none, bn ? 1 : 0, wt,
                       gate_alpha, gate_mode, add, dz, n_step, wpart, use_part ? ws : nullptr, s1, s2);
    UD_LAUNCH_CHECK();
    const int C = g.C4 * 4, KKC = K * K * C;
    if (dwt) {
        hipLaunchKernelGGL(dw_tile_wgrad_finalize, dim3(ud_cdiv(KKC, 64)), dim3(NT), 0, s, (int)nparts, K * K, C, wpart,
                           gate_alpha, gate_mode, dwt);
        UD_LAUNCH_CHECK();
    }
    if (use_part) {
        hipLaunchKernelGGL(partials_to_acc, dim3(ud_cdiv(C, 8)), dim3(NT), 0, s, 2, 1, C, (int)nparts, ws, s1, s2);
        UD_LAUNCH_CHECK();
    }
    return dwt ? 0 : (int)nparts;          // no dwt: the caller folds the partial rows later (ud_dwtile_wgrad_finalize_multi)
}

}  // namespace

extern "C" {

// doubles of scratch for the statistics partials of ud_dwtile (epi 1 / 2): 2 * tiles * C (bound: the smallest tile, 8 x 8)
long ud_dwtile_ws_doubles(int N, int Ho, int Wo, int C) {
    if (N < 1 || Ho < 1 || Wo < 1 || C < 4 || C % 4) return UD_EINVAL;
    return 2L * N * ud_cdiv(Ho, 8) * ud_cdiv(Wo, 8) * C;
}

// rows of K*K*C floats the weight-gradient partials may need (upper bound: one per (tile, image))
long ud_dwtile_wgrad_part_rows(int N, int Ho, int Wo) {
    if (N < 1 || Ho < 1 || Wo < 1) return UD_EINVAL;
    return (long)N * ud_cdiv(Ho, 8) * ud_cdiv(Wo, 8);
}

int ud_dwtile(const void* src, const ud_bn_ref* bn_in, const float* wt, void* out, int N, int Hs, int Ws, int C, int Ho,
              int Wo, int K, int P_t, int P_l, int flip, const float* gate_alpha, int gate_mode, const void* add,
              const void* xbn, const ud_bn_ref* bn_out, int epi, double* s1, double* s2, double* ws, int stride,
              int f16, ud_stream_t stream) {
    if (!tile_args_ok(N, Hs, Ws, C, Ho, Wo, K) || !src || !wt || !out || epi < 0 || epi > 2) return UD_EINVAL;
    if (stride != 1 && stride != 2) return UD_EINVAL;
    if (stride == 2 && ((flip != 0) != (epi == 2))) return UD_EINVAL;      // strided: forward (epi 0 / 1) or data gradient
    if (bn_in && bn_in->G != 1) return UD_EINVAL;
    if (epi != 2 && (add || xbn || bn_out || gate_mode)) return UD_EINVAL;
    if (bn_out && (!xbn || bn_out->G != 1)) return UD_EINVAL;
    if ((epi == 1 || bn_out) && (!s1 || !s2 || !ws)) return UD_EINVAL;
    // stride 2: the forward steps its window by 2 (8 x 8 output tiles); the data gradient reads dy through a zero-stuffed grid
    TileGeom g{N, Hs, Ws, C / 4, Ho, Wo, P_t, P_l, flip ? 1 : 0, 0, 0, (stride == 2 && flip) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    const bool sm = small_map(Ho, Wo, C);
#define UD_GO(KK, QQ, SS, ...)                                                                                        \
    UD_STORAGE_DISPATCH(f16, return (launch_tile<T, KK, QQ, SS, ##__VA_ARGS__>(g, (const T*)src, bn_in, wt, (T*)out,  \
                                                                gate_alpha, gate_mode, (const T*)add, (const T*)xbn,  \
                                                                bn_out, epi, s1, s2, ws, s)))
    if (stride == 2 && !flip) {
        if (K == 3) UD_GO(3, 8, 2, 2);
        UD_GO(5, 8, 2, 2);
    }
    // the data gradient (epi 2) keeps its epilogue operands in registers next to the window: strips of 4 (16 x 8 tiles)
    if (K == 3) { if (sm) UD_GO(3, 16, 4); if (epi == 2) UD_GO(3, 8, 4); UD_GO(3, 8, 8); }
    if (sm) UD_GO(5, 16, 4);
    if (epi == 2) UD_GO(5, 8, 4);
    UD_GO(5, 8, 8);
#undef UD_GO
}

int ud_dwtile_wgrad(const void* src, const ud_bn_ref* bn_in, const void* dy, const float* gate_alpha, int gate_mode,
                    float* dwt, float* part, long part_rows, int N, int Hs, int Ws, int C, int Ho, int Wo, int K, int P_t,
                    int P_l, int stride, int f16, ud_stream_t stream) {
    if (!tile_args_ok(N, Hs, Ws, C, Ho, Wo, K) || !src || !dy || !part || part_rows < 1) return UD_EINVAL;
    if (bn_in && bn_in->G != 1) return UD_EINVAL;
    if (stride != 1 && stride != 2) return UD_EINVAL;
    TileGeom g{N, Hs, Ws, C / 4, Ho, Wo, P_t, P_l, 0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const bool sm = small_map(Ho, Wo, C);
#define UD_WG(KK, QQ, SS, ...)                                                                                        \
    UD_STORAGE_DISPATCH(f16, return (launch_wgrad<T, KK, QQ, SS, ##__VA_ARGS__>(g, (const T*)src, bn_in, (const T*)dy, \
                                                                 gate_alpha, gate_mode, part, part_rows, dwt, s)))
    if (stride == 2) {
        if (K == 3) UD_WG(3, 8, 2, 2);
        UD_WG(5, 8, 2, 2);
    }
    // 5 x 5: 25 accumulator quads per thread — strips of 4 (16 x 8 tiles) keep the kernel clear of spills
    if (K == 3) { if (sm) UD_WG(3, 16, 4); UD_WG(3, 8, 8); }
    if (sm) UD_WG(5, 16, 4);
    UD_WG(5, 8, 4);
#undef UD_WG
}

int ud_dwtile_bwd(const void* dy, const void* x, const ud_bn_ref* bn, const float* wt, const float* gate_alpha, int gate_mode,
                  const void* add, void* dz, float* dwt, float* wpart, long part_rows, double* s1, double* s2, double* ws,
                  int N, int H, int W, int C, int K, int P_t, int P_l, int f16, ud_stream_t stream) {
    if (!tile_args_ok(N, H, W, C, H, W, K) || !dy || !x || !wt || !dz || !wpart || part_rows < 1) return UD_EINVAL;
    if (P_t < 0 || P_t > K - 1 || P_l < 0 || P_l > K - 1) return UD_EINVAL;
    if (bn && (bn->G != 1 || !s1 || !s2 || !ws)) return UD_EINVAL;
    TileGeom g{N, H, W, C / 4, H, W, P_t, P_l, 0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const bool sm = small_map(H, W, C);
#define UD_BW(KK, QQ, SS)                                                                                             \
    UD_STORAGE_DISPATCH(f16, return (launch_bwd<T, KK, QQ, SS>(g, (const T*)dy, (const T*)x, bn, wt, gate_alpha, gate_mode, \
                                                                (const T*)add, (T*)dz, wpart, part_rows, dwt, s1, s2, ws, s)))
    if (K == 3) { if (sm) UD_BW(3, 16, 4); UD_BW(3, 8, 4); }
    if (sm) UD_BW(5, 16, 4);
    UD_BW(5, 8, 4);
#undef UD_BW
}

int ud_dwtile_wgrad_finalize_multi(const ud_wgrad_fold* items, int n, ud_stream_t stream) {
    if (!items || n < 1) return UD_EINVAL;
    for (int i = 0; i < n; ++i) {
        const ud_wgrad_fold& f = items[i];
        if (!f.part || !f.dwt || f.nparts < 1 || (f.K != 3 && f.K != 5) || f.C < 4 || f.C % 4 || f.gate_mode < 0 || f.gate_mode > 2 ||
            (f.gate_mode != 0 && !f.gate_alpha))
            return UD_EINVAL;
    }
    for (int i0 = 0; i0 < n; i0 += FOLD_MAX) {
        FoldArgs a;
        a.n = n - i0 < FOLD_MAX ? n - i0 : FOLD_MAX;
        int b = 0;
        for (int j = 0; j < a.n; ++j) {
            a.it[j] = items[i0 + j];
            a.block0[j] = b;
            b += ud_cdiv(a.it[j].K * a.it[j].K * a.it[j].C, 64);
        }
        a.block0[a.n] = b;
        hipLaunchKernelGGL(dw_tile_wgrad_finalize_multi, dim3((unsigned)b), dim3(NT), 0, (hipStream_t)stream, a);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

int ud_dwtile_wgrad_finalize(const float* part, int nparts, int K, int C, const float* gate_alpha, int gate_mode, float* dwt,
                             ud_stream_t stream) {
    if (!part || nparts < 1 || (K != 3 && K != 5) || C < 4 || C % 4 || !dwt) return UD_EINVAL;
    hipLaunchKernelGGL(dw_tile_wgrad_finalize, dim3(ud_cdiv(K * K * C, 64)), dim3(NT), 0, (hipStream_t)stream, nparts, K * K, C,
                       part, gate_alpha, gate_mode, dwt);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
