// Backward of a THIN project 1x1 conv + the squeeze-excite gate in front of it WITHOUT materialising the conv's data gradient
// (round 6).  In the 64 x 64 blocks (model/efficientnet/model.py:113-126: BN1 + swish, SE gate, _project_conv 144 / 192 -> 32) the
// gradient dc = dp Wp of the gated tensor c = swish(bn1(d)) sigmoid(s) is M x CE (75-100 MB) and used to be written by one GEMM and
// read by two passes, next to a second GEMM reading c:
//     gemm_x3 (tn)   dWp = dp^T c                              reads c (stored by the forward), dp
//     gemm_x3 (nn)   dc = dp Wp                                writes dc
//     ud_coldot_bn   dgate[n][ch] = sum_hw dc swish(bn1(d))    reads dc, d
//     ud_se_scale_bwd_bn   dz = (dc gate + dpool / HW) swish'(bn1(d)), BatchNorm-1 backward sums     reads dc, d, writes dz
// dp is THIN (M x 32): a 32-row tile of dc is six MFMA k-steps away from it.  So:
//     pj_bwd_a_kernel: ONE pass over d: a = swish(bn1(d)) and c = a gate re-made on load (bit for bit the forward's), dWp from the
//                      c image (transposed reads, k = rows), the dc tile in registers, its dot with a -> dgate;
//     pj_bwd_b_kernel: ONE pass over d: the dc tile again, relaid through LDS to the row-major quads of d, dz written, sums taken.
// dc never exists in HBM and c is not read: 200 + 200 MB per block instead of 834.  Arithmetic of the products = gemm_x3's (exact
// three-way bf16 split, six piece products, fp32 accumulation); element-wise math = ud_se_scale_bn / ud_se_scale_bwd_bn's.
#include "pw_common.h"

#include <type_traits>

namespace {

using namespace pw;

constexpr int NTH = 256;
constexpr int R = 32;

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// CC: channels per CHUNK (CE % CC == 0): a wide tensor (336 channels) is walked as CE / CC column chunks — a workgroup stays inside one
// chunk, so its image, its fragments of w and its registers are those of a CC-channel problem with row stride CE.
template <int CE, int CO, int CC = CE> struct PjCfg {
    static_assert(CE % CC == 0 && CC % 8 == 0 && (CC % 16 == 0 || CC == CE) && CO % 8 == 0 && CO > 16 && CO <= 64, "thin project convs");
    static constexpr int CCP = (CC + 15) / 16 * 16;              // chunk width padded to whole 16-channel blocks (CE = 24: pad columns stay zero)
    static constexpr int NCH = CE / CC;
    static constexpr int QT = CE / 4;                            // quads per tensor row
    static constexpr int Q = CC / 4;                             // quads per row of the chunk
    static constexpr int NIT = (R * Q + NTH - 1) / NTH;          // float4 items of d per thread and tile
    static constexpr int RI = (NIT * NTH + Q - 1) / Q;           // image rows (items past row 31 land in rows nobody reads)
    static constexpr int NBC = CCP / 16;                         // 16-channel blocks of the chunk
    static constexpr int NBW = (NBC + 3) / 4;                    // ... per wave: wave w owns blocks w, w + 4, ...
    static constexpr int CS = CCP * 2 + 16;                      // bytes of a c-image row
    static constexpr int CP = RI * CS;
    static constexpr int AS = (CCP + 4) * 4;                     // bytes of an fp32 tile row (a / dc)
    static constexpr int AP = RI * AS;
    static constexpr int COP = (CO + 31) / 32 * 32;              // CO padded to whole MFMA k-steps (pad columns of the dp image stay zero)
    static constexpr int KS2 = COP / 32;
    static constexpr int NB2 = COP / 16;                         // 16-row blocks of the weight gradient
    static constexpr int DQ = CO / 4;
    static constexpr int NITP = (R * DQ + NTH - 1) / NTH;        // float4 items of dp per thread and tile
    static constexpr int DS = 2 * COP + 16;                      // bytes of a dp-image row
    static constexpr int DPL = ((NITP * NTH + DQ - 1) / DQ) * DS;
    static constexpr int COEF = 4 * CC * 4;                      // mu, invstd, gamma, beta of the chunk
    // kernel A: c image, a tile, dp image, coefficients, gate
    static constexpr int LDS_A = 3 * CP + AP + 3 * DPL + COEF + CC * 4;
    // kernel B: dc tile, dp image, coefficients, gate, dpool / HW, two fp64 column accumulators
    static constexpr int LDS_B = AP + 3 * DPL + COEF + 2 * CC * 4 + 2 * CC * 8;
};

struct PjArgs {
    const float* d;            // [N HW][CE] depthwise / SF output (pre-BatchNorm-1)
    const float* dp;           // [N HW][CO] gradient of the project conv's output
    const float* w;            // [CO][CE]
    const float* s;            // [N][CE] SE logits (gate = sigmoid)
    const float* dpool;        // [N][CE] (kernel B) gradient of the pooled mean
    float* dz;                 // [N HW][CE] (kernel B)
    float* part;               // [chunk][gc][CO][CC] (kernel A) weight-gradient partials
    double* dgate;             // [N][CE] (kernel A) += sum_hw dc a
    double* s1;                // [CE] (kernel B) += sum dz, sum dz xhat
    double* s2;
    const double* bsum;        // BatchNorm-1 statistics of d
    const double* bsumsq;
    const float* gamma;
    const float* beta;
    double inv_count;
    float eps;
    float inv_hw;
    int act;
    int HW;                    // rows per sample (HW % 32 == 0: a tile never straddles samples)
    long M;
    long tiles;
    int gc;                    // workgroups per chunk (grid = chunks * gc)
};

// act in {0, 1} without a branch per element (a branch on the runtime `act` made every element its own basic block)
__device__ __forceinline__ float act_sel(float z, bool swish) {
    const float sg = ud_sigmoid_fast(z);
    return swish ? z * sg : z;
}
__device__ __forceinline__ float act_grad_sel(float z, bool swish) {
    const float sg = ud_sigmoid_fast(z);
    return swish ? sg * (1.0f + z * (1.0f - sg)) : 1.f;
}

// coefficients of channels c0 .. c0 + CC - 1
__device__ __forceinline__ void load_coef(const PjArgs& a, float* coef, int c0, int CC, int tid) {
    for (int c = tid; c < CC; c += NTH) {
        const double m = a.bsum[c0 + c] * a.inv_count;
        double v = a.bsumsq[c0 + c] * a.inv_count - m * m;
        if (v < 0.0) v = 0.0;
        coef[c] = (float)m;
        coef[CC + c] = rsqrtf((float)(v + (double)a.eps));          // (bnref.h: bn_load's own form)
        coef[2 * CC + c] = a.gamma[c0 + c];
        coef[3 * CC + c] = a.beta[c0 + c];
    }
}

// Wp as the B operand of dc = dp Wp: lane holds Wp[k = 32 ks + 8 g + j][n = c0 + 16 nb + u] for its NBW channel blocks
template <int CE, int CO, int NBC, int NBW, int KS2>
__device__ __forceinline__ void load_wfrags(const float* w, int c0, int wave, int u, int g, u32x4 (&wf)[NBW][KS2][3]) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int n = c0 + 16 * min(wave + 4 * j, NBC - 1) + u;          // (n >= CE only where CE is not whole 16-channel blocks)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int k = 32 * ks + 8 * g + jj;
                const float wv = w[min(k, CO - 1) * CE + min(n, CE - 1)];
                v[jj] = (k < CO && n < CE) ? wv : 0.f;
            }
            split8(v, wf[j][ks]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel A: weight gradient + SE dot
// ---------------------------------------------------------------------------------------------------------------------
template <int CE, int CO, int CC>
__global__ __launch_bounds__(NTH, 2) void pj_bwd_a_kernel(PjArgs a) {
    using CF = PjCfg<CE, CO, CC>;
    extern __shared__ __attribute__((aligned(16))) char L[];
    char* cimg = L;
    char* abuf = L + 3 * CF::CP;
    char* dimg = abuf + CF::AP;
    float* coef = reinterpret_cast<float*>(dimg + 3 * CF::DPL);
    float* gate = coef + 4 * CC;
    const lds_char* Lp = (const lds_char*)L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4, q4 = u >> 2, pp = u & 3;
    const bool swish = a.act == 1;
    const int chunk = blockIdx.x / a.gc, bb = blockIdx.x - chunk * a.gc;          // (uniform)
    const int c0 = chunk * CC;

    // (no zero fill of the c image / a tile: every row and column a fragment read touches is rewritten per tile; the dp image's pad
    //  columns CO .. COP - 1 are zeroed once and never written)
    if constexpr (CF::COP != CO) {
        for (int i = tid; i < 3 * CF::DPL / 16; i += NTH) reinterpret_cast<u32x4*>(dimg)[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (CF::CCP != CC) {          // (pad channels of the c image and the a tile: zero once, never written)
        for (int i = tid; i < (3 * CF::CP + CF::AP) / 16; i += NTH) reinterpret_cast<u32x4*>(L)[i] = u32x4{0u, 0u, 0u, 0u};
    }

    f32x4 accw[CF::NBW][CF::NB2];
    double dot[CF::NBW];          // (per tile: 8 products per lane in fp32, then fp64 — a sample of 128 x 128 pixels is 512 tiles)
#pragma unroll
    for (int j = 0; j < CF::NBW; ++j) {
#pragma unroll
        for (int m = 0; m < CF::NB2; ++m) accw[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        dot[j] = 0.0;
    }

    const f32x4* d4 = reinterpret_cast<const f32x4*>(a.d) + chunk * CF::Q;
    const f32x4* dp4 = reinterpret_cast<const f32x4*>(a.dp);
    const long t0 = (long)bb * a.tiles / a.gc, t1 = (long)(bb + 1) * a.tiles / a.gc;
    f32x4 rd[CF::NIT], rp[CF::NITP];
    auto prefetch = [&](long tile_) {
        const long tile = min(tile_, a.tiles - 1);          // (uniform: a scalar base + the thread's fixed offsets; M % 32 == 0: no row tail)
        const f32x4* db = d4 + tile * (R * CF::QT);
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) {
            const int idx = tid + it * NTH;
            const int row = idx / CF::Q, q = idx - row * CF::Q;
            rd[it] = db[((it + 1) * NTH <= R * CF::Q || idx < R * CF::Q) ? row * CF::QT + q : 0];
        }
        const f32x4* pb = dp4 + tile * (R * CF::DQ);
#pragma unroll
        for (int it = 0; it < CF::NITP; ++it) {
            const int idx = tid + it * NTH;
            rp[it] = pb[((it + 1) * NTH <= R * CF::DQ || idx < R * CF::DQ) ? idx : 0];
        }
    };
    auto flush = [&](int n) {          // the dots of sample n (every lane of a column holds the same sum)
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j)
                if (wave + 4 * j < CF::NBC && 16 * (wave + 4 * j) + u < CC)
                    atomic_add_f64(a.dgate + (long)n * CE + c0 + 16 * (wave + 4 * j) + u, dot[j]);
        }
#pragma unroll
        for (int j = 0; j < CF::NBW; ++j) dot[j] = 0.0;
    };

    prefetch(t0);          // the first tile's loads, then everything else the prologue needs: one memory latency for all of it
    load_coef(a, coef, c0, CC, tid);
    u32x4 wf[CF::NBW][CF::KS2][3];
    load_wfrags<CE, CO, CF::NBC, CF::NBW, CF::KS2>(a.w, c0, wave, u, g, wf);
    int cur_n = -1;
    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gate);
    __syncthreads();
    for (long tile = t0; tile < t1; ++tile) {
        const int n = (int)(tile * R / a.HW);
        if (n != cur_n) {          // (uniform) the gate of the new sample; the finished sample's dots go out
            if (cur_n >= 0) flush(cur_n);
            for (int c = tid; c < CC; c += NTH) gate[c] = ud_sigmoid_fast(a.s[(long)n * CE + c0 + c]);
            cur_n = n;
            __syncthreads();
        }
        // ---- a = act(bn1(d)) (fp32 tile), c = a gate (split, image); dp (split, image)
        {
#pragma unroll
            for (int it = 0; it < CF::NIT; ++it) {
                const int idx = tid + it * NTH;
                const int row = idx / CF::Q, q = idx - row * CF::Q;
                const f32x4 mu = c4[q], is = c4[CF::Q + q], ga = c4[2 * CF::Q + q], be = c4[3 * CF::Q + q], gt = g4[q];
                f32x4 av, cv;          // (M % 32 == 0: every row of the tile exists; the items past row 31 go to rows nobody reads)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = ga[k] * ((rd[it][k] - mu[k]) * is[k]) + be[k];
                    av[k] = act_sel(z, swish);
                    cv[k] = av[k] * gt[k];
                }
                *reinterpret_cast<f32x4*>(abuf + row * CF::AS + q * 16) = av;
                store_split4(cimg + row * CF::CS + q * 8, CF::CP, cv);
            }
#pragma unroll
            for (int it = 0; it < CF::NITP; ++it) {
                const int idx = tid + it * NTH;
                const int row = idx / CF::DQ, q = idx - row * CF::DQ;
                store_split4(dimg + row * CF::DS + q * 8, CF::DPL, rp[it]);
            }
        }
        __syncthreads();
        prefetch(tile + 1);

        // ---- dc tile (registers) and its dot with a
        {
            const lds_char* drow = Lp + (3 * CF::CP + CF::AP) + u * CF::DS + g * 16;
            Frag3 a0[CF::KS2], a1[CF::KS2];
#pragma unroll
            for (int ks = 0; ks < CF::KS2; ++ks) {
                a0[ks] = read_rows(drow + ks * 64, CF::DPL);
                a1[ks] = read_rows(drow + 16 * CF::DS + ks * 64, CF::DPL);
            }
            const float* ab = reinterpret_cast<const float*>(abuf) + (4 * g) * (CF::AS / 4) + u;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);
                f32x4 dc0 = {0.f, 0.f, 0.f, 0.f}, dc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < CF::KS2; ++ks) {
                    const Frag3 b = frag_of(wf[j][ks]);
                    dc0 = mma6(a0[ks], b, dc0);
                    dc1 = mma6(a1[ks], b, dc1);
                }
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t += dc0[r] * ab[r * (CF::AS / 4) + 16 * nb];
                    t += dc1[r] * ab[(16 + r) * (CF::AS / 4) + 16 * nb];
                }
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                dot[j] += (double)t;
            }
        }
        // ---- weight gradient: dWp[co block][channel block] += dp^T c, k = the tile's rows (transposed reads of both images)
        {
            const lds_char* db = Lp + (3 * CF::CP + CF::AP) + (8 * g + q4) * CF::DS + 4 * pp * 2;
            Frag3 pm[CF::NB2];
#pragma unroll
            for (int m = 0; m < CF::NB2; ++m) pm[m] = read_cols(db + 32 * m, CF::DPL, CF::DS);
            const lds_char* cb = Lp + (8 * g + q4) * CF::CS + 4 * pp * 2;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);
                const Frag3 b = read_cols(cb + nb * 32, CF::CP, CF::CS);
#pragma unroll
                for (int m = 0; m < CF::NB2; ++m) accw[j][m] = mma6(pm[m], b, accw[j][m]);
            }
        }
        __syncthreads();
    }
    if (cur_n >= 0) flush(cur_n);
    {
        float* part = a.part + (long)blockIdx.x * (CO * CC);          // (blockIdx.x = chunk * gc + bb)
#pragma unroll
        for (int j = 0; j < CF::NBW; ++j) {
            if (wave + 4 * j < CF::NBC) {
                const int ch = 16 * (wave + 4 * j) + u;
#pragma unroll
                for (int m = 0; m < CF::NB2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = 16 * m + 4 * g + r;
                        if (co < CO && ch < CC) part[co * CC + ch] = accw[j][m][r];
                    }
            }
        }
    }
}

// dw[co][c0 + cc] = the gc partials of chunk c0 / CC summed in workgroup order (the form of pw_bwd_fold_kernel: 16 elements x 16
// lanes of partials per workgroup, the lane sums added in order through LDS)
__global__ __launch_bounds__(256) void pj_fold_kernel(const float* __restrict__ part, int gc, int CO, int CE, int CC, float* __restrict__ dw) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el, numel = CO * CE;
    float s0 = 0.f, s1 = 0.f;
    if (i < numel) {
        const int co = i / CE, c = i - co * CE, chunk = c / CC, cc = c - chunk * CC;
        const float* src = part + (long)chunk * gc * (CO * CC) + co * CC + cc;
        const long st = (long)CO * CC;
        int p = grp;
        for (; p + 16 < gc; p += 32) {
            s0 += src[p * st];
            s1 += src[(p + 16) * st];
        }
        if (p < gc) s0 += src[p * st];
    }
    red[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0 && i < numel) {
        float t = red[0][el];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][el];
        dw[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel B: dz = (dc gate + dpool / HW) act'(bn1(d)) with dc re-made per tile; BatchNorm-1 backward sums
// ---------------------------------------------------------------------------------------------------------------------
template <int CE, int CO, int CC>
__global__ __launch_bounds__(NTH, 2) void pj_bwd_b_kernel(PjArgs a) {
    using CF = PjCfg<CE, CO, CC>;
    extern __shared__ __attribute__((aligned(16))) char L[];
    char* dcbuf = L;
    char* dimg = L + CF::AP;
    float* coef = reinterpret_cast<float*>(dimg + 3 * CF::DPL);
    float* gate = coef + 4 * CC;
    float* dpl = gate + CC;
    double* lacc = reinterpret_cast<double*>(dpl + CC);          // [2][CC]
    const lds_char* Lp = (const lds_char*)L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4;
    const bool swish = a.act == 1;
    const int chunk = blockIdx.x / a.gc, bb = blockIdx.x - chunk * a.gc;          // (uniform)
    const int c0 = chunk * CC;

    if constexpr (CF::COP != CO) {
        for (int i = tid; i < 3 * CF::DPL / 16; i += NTH) reinterpret_cast<u32x4*>(dimg)[i] = u32x4{0u, 0u, 0u, 0u};
    }
    for (int c = tid; c < 2 * CC; c += NTH) lacc[c] = 0.0;

    // a thread's column sums over its tiles: fp64 where that costs few registers (the narrow tensors: up to 64 tiles per workgroup),
    // fp32 over <= 16 tiles otherwise (then fp64 through LDS and the atomics)
    using ST = typename std::conditional<(CF::NIT <= 2), double, float>::type;
    ST sa[CF::NIT][4], sb[CF::NIT][4];
#pragma unroll
    for (int it = 0; it < CF::NIT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) sa[it][k] = sb[it][k] = (ST)0;

    const f32x4* d4 = reinterpret_cast<const f32x4*>(a.d) + chunk * CF::Q;
    const f32x4* dp4 = reinterpret_cast<const f32x4*>(a.dp);
    f32x4* dz4 = reinterpret_cast<f32x4*>(a.dz) + chunk * CF::Q;
    const long t0 = (long)bb * a.tiles / a.gc, t1 = (long)(bb + 1) * a.tiles / a.gc;
    f32x4 rd[CF::NIT], rp[CF::NITP];
    auto prefetch = [&](long tile_) {
        const long tile = min(tile_, a.tiles - 1);          // (uniform: a scalar base + the thread's fixed offsets; M % 32 == 0: no row tail)
        const f32x4* db = d4 + tile * (R * CF::QT);
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) {
            const int idx = tid + it * NTH;
            const int row = idx / CF::Q, q = idx - row * CF::Q;
            rd[it] = db[((it + 1) * NTH <= R * CF::Q || idx < R * CF::Q) ? row * CF::QT + q : 0];
        }
        const f32x4* pb = dp4 + tile * (R * CF::DQ);
#pragma unroll
        for (int it = 0; it < CF::NITP; ++it) {
            const int idx = tid + it * NTH;
            rp[it] = pb[((it + 1) * NTH <= R * CF::DQ || idx < R * CF::DQ) ? idx : 0];
        }
    };

    prefetch(t0);
    load_coef(a, coef, c0, CC, tid);
    u32x4 wf[CF::NBW][CF::KS2][3];
    load_wfrags<CE, CO, CF::NBC, CF::NBW, CF::KS2>(a.w, c0, wave, u, g, wf);
    int cur_n = -1;
    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gate);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(dpl);
    __syncthreads();
    for (long tile = t0; tile < t1; ++tile) {
        const int n = (int)(tile * R / a.HW);
        if (n != cur_n) {          // (uniform) the previous tile's element-wise phase still reads gate / dpl: wait for it
            __syncthreads();
            for (int c = tid; c < CC; c += NTH) {
                gate[c] = ud_sigmoid_fast(a.s[(long)n * CE + c0 + c]);
                dpl[c] = a.dpool[(long)n * CE + c0 + c] * a.inv_hw;
            }
            cur_n = n;
        }
#pragma unroll
        for (int it = 0; it < CF::NITP; ++it) {
            const int idx = tid + it * NTH;
            const int row = idx / CF::DQ, q = idx - row * CF::DQ;
            store_split4(dimg + row * CF::DS + q * 8, CF::DPL, rp[it]);
        }
        // this tile's d quads move out of the ring before the next tile's loads are issued into it
        f32x4 dq[CF::NIT];
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) dq[it] = rd[it];
        __syncthreads();
        prefetch(tile + 1);
        // ---- dc tile -> fp32 tile in LDS (from the MFMA layout: lane = column 16 nb + u, rows 16 mb + 4 g + r)
        {
            const lds_char* drow = Lp + CF::AP + u * CF::DS + g * 16;
            Frag3 a0[CF::KS2], a1[CF::KS2];
#pragma unroll
            for (int ks = 0; ks < CF::KS2; ++ks) {
                a0[ks] = read_rows(drow + ks * 64, CF::DPL);
                a1[ks] = read_rows(drow + 16 * CF::DS + ks * 64, CF::DPL);
            }
            float* ob = reinterpret_cast<float*>(dcbuf) + (4 * g) * (CF::AS / 4) + u;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);          // (a wave's block past the last one rewrites the last block's values)
                f32x4 dc0 = {0.f, 0.f, 0.f, 0.f}, dc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < CF::KS2; ++ks) {
                    const Frag3 b = frag_of(wf[j][ks]);
                    dc0 = mma6(a0[ks], b, dc0);
                    dc1 = mma6(a1[ks], b, dc1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ob[r * (CF::AS / 4) + 16 * nb] = dc0[r];
                    ob[(16 + r) * (CF::AS / 4) + 16 * nb] = dc1[r];
                }
            }
        }
        __syncthreads();
        // ---- element-wise over the row-major quads
        {
            f32x4* ob4 = dz4 + tile * (R * CF::QT);
#pragma unroll
            for (int it = 0; it < CF::NIT; ++it) {
                const int idx = tid + it * NTH;
                const int row = idx / CF::Q, q = idx - row * CF::Q;
                const bool ok = (it + 1) * NTH <= R * CF::Q || idx < R * CF::Q;          // (compile-time true but for a last partial item)
                const f32x4 mu = c4[q], is = c4[CF::Q + q], ga = c4[2 * CF::Q + q], be = c4[3 * CF::Q + q], gt = g4[q], pl = p4[q];
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dcbuf + row * CF::AS + q * 16);
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (dq[it][k] - mu[k]) * is[k];
                    float gg = (dc[k] * gt[k] + pl[k]) * act_grad_sel(ga[k] * xh + be[k], swish);
                    gg = ok ? gg : 0.f;
                    o[k] = gg;
                    sa[it][k] += (ST)gg;
                    sb[it][k] += (ST)(gg * xh);
                }
                if (ok) ob4[row * CF::QT + q] = o;
            }
        }
        // (the next tile's dp image store and dc tile stores come after its own barriers; dcbuf readers finish before the next
        //  tile's MFMA phase because of the barrier that follows the dp image store)
    }
    // ---- column sums: thread partials -> LDS fp64 -> one fp64 atomic per channel and workgroup
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CF::NIT; ++it) {
        const int idx = tid + it * NTH;
        if (idx < R * CF::Q) {
            const int q = idx % CF::Q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&lacc[4 * q + k], (double)sa[it][k]);
                atomicAdd(&lacc[CC + 4 * q + k], (double)sb[it][k]);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < CC; c += NTH) {
        atomic_add_f64(a.s1 + c0 + c, lacc[c]);
        atomic_add_f64(a.s2 + c0 + c, lacc[CC + c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: p = (act(bn1(d)) sigmoid(s)) Wp^T in ONE pass over d — the gated tensor c is made on load (ud_se_scale_bn's values), laid
// into the image as pieces and multiplied from there; p's BatchNorm-2 statistics leave with it.  c is never written: the backward
// above re-makes it as well.  (ud_se_scale_bn + gemm_x3 nt + the statistics fold: 34 + 33 + 5 us per 64 x 64 block before.)
// ---------------------------------------------------------------------------------------------------------------------
template <int CE, int CO> struct PjFwdCfg {
    using B = PjCfg<CE, CO>;
    static constexpr int KD = (CE + 31) / 32 * 32;          // k padded to whole MFMA steps (pad columns of the image stay zero)
    static constexpr int KS = KD / 32;
    static constexpr int CS = KD * 2 + 16;
    static constexpr int CP = B::RI * CS;
    static constexpr int LDS = 3 * CP + B::COEF + CE * 4 + 2 * 4 * 32 * 8;          // image, coefficients, gate, the waves' column sums
};

struct PjFwdArgs {
    const float* d;
    const float* w;            // [CO][CE]
    const float* s;            // [N][CE]
    float* p;                  // [N HW][CO]
    double* sum;               // [CO] += sum p, [CO] += sum p^2 (may be NULL together)
    double* sumsq;
    const double* bsum;
    const double* bsumsq;
    const float* gamma;
    const float* beta;
    double inv_count;
    float eps;
    int act;
    int HW;
    long M;
    long tiles;
};

template <int CE, int CO>
__global__ __launch_bounds__(NTH, 2) void pj_fwd_kernel(PjFwdArgs a) {
    using CB = PjCfg<CE, CO>;
    using CF = PjFwdCfg<CE, CO>;
    static_assert(CO > 16 && CO <= 32, "two 16-column blocks: one (row block, column block) per wave");
    extern __shared__ __attribute__((aligned(16))) char L[];
    char* cimg = L;
    float* coef = reinterpret_cast<float*>(L + 3 * CF::CP);
    float* gate = coef + 4 * CE;
    double* wsum = reinterpret_cast<double*>(gate + CE);          // [2][4 waves][32 columns... 16 used per wave]
    const lds_char* Lp = (const lds_char*)L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4;
    const int mb = wave >> 1, nb = wave & 1;
    const bool swish = a.act == 1;

    if constexpr (CF::KD != CE) {
        for (int i = tid; i < 3 * CF::CP / 16; i += NTH) reinterpret_cast<u32x4*>(L)[i] = u32x4{0u, 0u, 0u, 0u};
    }
    const f32x4* d4 = reinterpret_cast<const f32x4*>(a.d);
    const long t0 = (long)blockIdx.x * a.tiles / gridDim.x, t1 = (long)(blockIdx.x + 1) * a.tiles / gridDim.x;
    f32x4 rd[CB::NIT];
    auto prefetch = [&](long tile_) {
        const long tile = min(tile_, a.tiles - 1);
        const f32x4* db = d4 + tile * (R * CB::Q);
#pragma unroll
        for (int it = 0; it < CB::NIT; ++it) {
            const int idx = tid + it * NTH;
            rd[it] = db[((it + 1) * NTH <= R * CB::Q || idx < R * CB::Q) ? idx : 0];
        }
    };
    prefetch(t0);
    {          // (CE > NTH never happens here: one pass of the loop)
        PjArgs ca;
        ca.bsum = a.bsum; ca.bsumsq = a.bsumsq; ca.gamma = a.gamma; ca.beta = a.beta; ca.inv_count = a.inv_count; ca.eps = a.eps;
        load_coef(ca, coef, 0, CE, tid);
    }
    // Wp as the B operand: lane holds Wp[n = 16 nb + u][k = 32 ks + 8 g + j] (k-contiguous in memory)
    u32x4 wf[CF::KS][3];
#pragma unroll
    for (int ks = 0; ks < CF::KS; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 8 * g + j;
            const float wv = a.w[min(16 * nb + u, CO - 1) * CE + min(k, CE - 1)];
            v[j] = (k < CE && 16 * nb + u < CO) ? wv : 0.f;
        }
        split8(v, wf[ks]);
    }
    double s1 = 0.0, s2 = 0.0;
    int cur_n = -1;
    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gate);
    __syncthreads();
    for (long tile = t0; tile < t1; ++tile) {
        const int n = (int)(tile * R / a.HW);
        if (n != cur_n) {          // (uniform; the previous tile's closing barrier is behind every wave)
            for (int c = tid; c < CE; c += NTH) gate[c] = ud_sigmoid_fast(a.s[(long)n * CE + c]);
            cur_n = n;
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < CB::NIT; ++it) {
            const int idx = tid + it * NTH;
            const int row = idx / CB::Q, q = idx - row * CB::Q;
            const f32x4 mu = c4[q], is = c4[CB::Q + q], ga = c4[2 * CB::Q + q], be = c4[3 * CB::Q + q], gt = g4[q];
            f32x4 cv;
#pragma unroll
            for (int k = 0; k < 4; ++k) cv[k] = act_sel(ga[k] * ((rd[it][k] - mu[k]) * is[k]) + be[k], swish) * gt[k];
            store_split4(cimg + row * CF::CS + q * 8, CF::CP, cv);
        }
        __syncthreads();
        prefetch(tile + 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const lds_char* arow = Lp + (16 * mb + u) * CF::CS + g * 16;
#pragma unroll
        for (int ks = 0; ks < CF::KS; ++ks) acc = mma6(read_rows(arow + ks * 64, CF::CP), frag_of(wf[ks]), acc);
        float* po = a.p + (tile * R + 16 * mb + 4 * g) * CO + 16 * nb + u;
        float t1_ = 0.f, t2_ = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (CO == 32 || 16 * nb + u < CO) po[r * CO] = acc[r];          // (a pad column's accumulator is zero: its sums add nothing)
            t1_ += acc[r];
            t2_ += acc[r] * acc[r];
        }
        s1 += (double)t1_;
        s2 += (double)t2_;
        __syncthreads();
    }
    if (a.sum) {          // (uniform) column sums: the four row groups of a wave, then the two waves of a column block, then one atomic
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) { wsum[wave * 16 + u] = s1; wsum[64 + wave * 16 + u] = s2; }
        __syncthreads();
        if (tid < CO) {
            const int nbb = tid >> 4, uu = tid & 15;
            atomic_add_f64(a.sum + tid, wsum[nbb * 16 + uu] + wsum[(2 + nbb) * 16 + uu]);
            atomic_add_f64(a.sumsq + tid, wsum[64 + nbb * 16 + uu] + wsum[64 + (2 + nbb) * 16 + uu]);
        }
    }
}

template <int CE, int CO>
int launch_f(const PjFwdArgs& a, int grid, hipStream_t s) {
    using CF = PjFwdCfg<CE, CO>;
    static_assert(CF::LDS <= 64 * 1024, "dynamic LDS within the default limit");
    hipLaunchKernelGGL((pj_fwd_kernel<CE, CO>), dim3((unsigned)grid), dim3(NTH), CF::LDS, s, a);
    return 0;
}

template <int CE, int CO, int CC>
int launch_a(const PjArgs& a, int grid, hipStream_t s) {
    using CF = PjCfg<CE, CO, CC>;
    static_assert(CF::LDS_A <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;          // (> 64 KiB of dynamic LDS; set on the first launch, which is never inside a capture: eager warm-up)
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pj_bwd_a_kernel<CE, CO, CC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                CF::LDS_A) != hipSuccess)
            return UD_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL((pj_bwd_a_kernel<CE, CO, CC>), dim3((unsigned)grid), dim3(NTH), CF::LDS_A, s, a);
    return 0;
}
template <int CE, int CO, int CC>
int launch_b(const PjArgs& a, int grid, hipStream_t s) {
    using CF = PjCfg<CE, CO, CC>;
    static_assert(CF::LDS_B <= 64 * 1024, "dynamic LDS within the default limit");
    hipLaunchKernelGGL((pj_bwd_b_kernel<CE, CO, CC>), dim3((unsigned)grid), dim3(NTH), CF::LDS_B, s, a);
    return 0;
}

// the (CE, CO) pairs built, and the chunk width each is walked in
int chunk_of(int CE, int CO) {
    if (CO == 32 && (CE == 144 || CE == 192)) return CE;
    if (CO == 24 && (CE == 48 || CE == 24)) return CE;
    if (CO == 56 && CE == 336) return 112;
    if (CO == 56 && CE == 192) return 96;
    return 0;
}

bool args_ok(const void* d, const void* dp, const ud_bn_ref* bn, const void* s, const void* w, int N, int HW, int CE, int CO) {
    return d && dp && bn && s && w && N >= 1 && HW >= R && ud_pj_bwd_fused_ok(CE, CO, HW) && bn->G == 1 && bn->gamma && bn->beta;
}

void fill(PjArgs& a, const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO) {
    a.d = d; a.dp = dp; a.w = w; a.s = s;
    a.bsum = bn->sum; a.bsumsq = bn->sumsq; a.gamma = bn->gamma; a.beta = bn->beta;
    a.inv_count = bn->inv_count; a.eps = bn->eps; a.act = bn->act; a.HW = HW;
    a.M = (long)N * HW; a.tiles = a.M / R;
    a.gc = (int)(ud_pj_bwd_fused_grid(N, HW, CE, CO) / (CE / chunk_of(CE, CO)));
    a.dpool = nullptr; a.dz = nullptr; a.part = nullptr; a.dgate = nullptr; a.s1 = a.s2 = nullptr; a.inv_hw = 0.f;
}

}  // namespace

extern "C" {

int ud_pj_bwd_fused_ok(int CE, int CO, int HW) { return (chunk_of(CE, CO) && HW >= R && HW % R == 0) ? 1 : 0; }
int ud_pj_fwd_fused_ok(int CE, int CO, int HW) { return (CO <= 32 && chunk_of(CE, CO) == CE && HW >= R && HW % R == 0) ? 1 : 0; }

long ud_pj_bwd_fused_grid(int N, int HW, int CE, int CO) {
    const int cc = chunk_of(CE, CO);
    if (N < 1 || HW < R || HW % R || !cc) return UD_EINVAL;
    const long tiles = (long)N * HW / R, nch = CE / cc, per = 512 / nch;
    return nch * (tiles < per ? tiles : per);
}

int ud_pj_fwd_fused(const float* d, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO, float* p,
                     double* sum, double* sumsq, ud_stream_t stream) {
    if (!d || !bn || !s || !w || !p || N < 1 || !ud_pj_fwd_fused_ok(CE, CO, HW) || bn->G != 1 || !bn->gamma || !bn->beta || (!sum != !sumsq))
        return UD_EINVAL;
    PjFwdArgs a;
    a.d = d; a.w = w; a.s = s; a.p = p; a.sum = sum; a.sumsq = sumsq;
    a.bsum = bn->sum; a.bsumsq = bn->sumsq; a.gamma = bn->gamma; a.beta = bn->beta;
    a.inv_count = bn->inv_count; a.eps = bn->eps; a.act = bn->act; a.HW = HW;
    a.M = (long)N * HW; a.tiles = a.M / R;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW, CE, CO);
    hipStream_t st = (hipStream_t)stream;
    const int rc = CO == 32 ? (CE == 144 ? launch_f<144, 32>(a, grid, st) : launch_f<192, 32>(a, grid, st))
                            : (CE == 48 ? launch_f<48, 24>(a, grid, st) : launch_f<24, 24>(a, grid, st));
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_pj_bwd_fused_a(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO,
                      float* dw, double* dgate, float* part, ud_stream_t stream) {
    if (!args_ok(d, dp, bn, s, w, N, HW, CE, CO) || !dw || !dgate || !part) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PjArgs a;
    fill(a, d, dp, bn, s, w, N, HW, CE, CO);
    a.part = part; a.dgate = dgate;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW, CE, CO);
    const int rc = CO == 32 ? (CE == 144 ? launch_a<144, 32, 144>(a, grid, st) : launch_a<192, 32, 192>(a, grid, st))
                 : CO == 24 ? (CE == 48 ? launch_a<48, 24, 48>(a, grid, st) : launch_a<24, 24, 24>(a, grid, st))
                            : (CE == 336 ? launch_a<336, 56, 112>(a, grid, st) : launch_a<192, 56, 96>(a, grid, st));
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(pj_fold_kernel, dim3((unsigned)ud_cdiv((long)CO * CE, 16)), dim3(256), 0, st, part, a.gc, CO, CE, chunk_of(CE, CO), dw);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_pj_bwd_fused_b(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* dpool, float inv_hw,
                      const float* w, int N, int HW, int CE, int CO, float* dz, double* s1, double* s2, ud_stream_t stream) {
    if (!args_ok(d, dp, bn, s, w, N, HW, CE, CO) || !dpool || !dz || !s1 || !s2) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PjArgs a;
    fill(a, d, dp, bn, s, w, N, HW, CE, CO);
    a.dpool = dpool; a.inv_hw = inv_hw; a.dz = dz; a.s1 = s1; a.s2 = s2;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW, CE, CO);
    const int rc = CO == 32 ? (CE == 144 ? launch_b<144, 32, 144>(a, grid, st) : launch_b<192, 32, 192>(a, grid, st))
                 : CO == 24 ? (CE == 48 ? launch_b<48, 24, 48>(a, grid, st) : launch_b<24, 24, 24>(a, grid, st))
                            : (CE == 336 ? launch_b<336, 56, 112>(a, grid, st) : launch_b<192, 56, 96>(a, grid, st));
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
