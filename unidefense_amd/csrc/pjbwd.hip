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

namespace {

using namespace pw;

constexpr int NTH = 256;
constexpr int R = 32;

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

template <int CE, int CO> struct PjCfg {
    static_assert(CE % 16 == 0 && CO % 8 == 0 && CO > 16 && CO <= 32, "thin project convs: CE % 16 == 0, 16 < CO <= 32");
    static constexpr int Q = CE / 4;
    static constexpr int NIT = (R * Q + NTH - 1) / NTH;          // float4 items of d per thread and tile
    static constexpr int RI = (NIT * NTH + Q - 1) / Q;           // image rows (items past row 31 land in rows nobody reads)
    static constexpr int NBC = CE / 16;                          // 16-channel blocks
    static constexpr int NBW = (NBC + 3) / 4;                    // ... per wave: wave w owns blocks w, w + 4, ...
    static constexpr int CS = CE * 2 + 16;                       // bytes of a c-image row
    static constexpr int CP = RI * CS;
    static constexpr int AS = (CE + 4) * 4;                      // bytes of an fp32 tile row (a / dc)
    static constexpr int AP = RI * AS;
    static constexpr int DQ = CO / 4;
    static constexpr int DS = 2 * 32 + 16;                       // bytes of a dp-image row (CO padded to 32: one MFMA k-step)
    static constexpr int DPL = (NTH / DQ + 1) * DS;
    static constexpr int COEF = 4 * CE * 4;                      // mu, invstd, gamma, beta
    // kernel A: c image, a tile, dp image, coefficients, gate
    static constexpr int LDS_A = 3 * CP + AP + 3 * DPL + COEF + CE * 4;
    // kernel B: dc tile, dp image, coefficients, gate, dpool / HW, two fp64 column accumulators
    static constexpr int LDS_B = AP + 3 * DPL + COEF + 2 * CE * 4 + 2 * CE * 8;
};

struct PjArgs {
    const float* d;            // [N HW][CE] depthwise / SF output (pre-BatchNorm-1)
    const float* dp;           // [N HW][CO] gradient of the project conv's output
    const float* w;            // [CO][CE]
    const float* s;            // [N][CE] SE logits (gate = sigmoid)
    const float* dpool;        // [N][CE] (kernel B) gradient of the pooled mean
    float* dz;                 // [N HW][CE] (kernel B)
    float* part;               // [grid][CO][CE] (kernel A) weight-gradient partials
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

__device__ __forceinline__ void load_coef(const PjArgs& a, float* coef, int CE, int tid) {
    for (int c = tid; c < CE; c += NTH) {
        const double m = a.bsum[c] * a.inv_count;
        double v = a.bsumsq[c] * a.inv_count - m * m;
        if (v < 0.0) v = 0.0;
        coef[c] = (float)m;
        coef[CE + c] = rsqrtf((float)(v + (double)a.eps));          // (bnref.h: bn_load's own form)
        coef[2 * CE + c] = a.gamma[c];
        coef[3 * CE + c] = a.beta[c];
    }
}

// Wp as the B operand of dc = dp Wp: lane holds Wp[k = 8 g + j][n = 16 nb + u] for its NBW channel blocks
template <int CE, int CO, int NBW>
__device__ __forceinline__ void load_wfrags(const float* w, int wave, int u, int g, u32x4 (&wf)[NBW][3]) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int n = 16 * min(wave + 4 * j, CE / 16 - 1) + u;
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = 8 * g + jj;
            const float wv = w[min(k, CO - 1) * CE + n];
            v[jj] = k < CO ? wv : 0.f;
        }
        split8(v, wf[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel A: weight gradient + SE dot
// ---------------------------------------------------------------------------------------------------------------------
template <int CE, int CO>
__global__ __launch_bounds__(NTH, 2) void pj_bwd_a_kernel(PjArgs a) {
    using CF = PjCfg<CE, CO>;
    extern __shared__ __attribute__((aligned(16))) char L[];
    char* cimg = L;
    char* abuf = L + 3 * CF::CP;
    char* dimg = abuf + CF::AP;
    float* coef = reinterpret_cast<float*>(dimg + 3 * CF::DPL);
    float* gate = coef + 4 * CE;
    const lds_char* Lp = (const lds_char*)L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4, q4 = u >> 2, pp = u & 3;
    const bool swish = a.act == 1;

    // (no zero fill: CO == 32 and CE % 16 == 0 leave no pad column, and every image row a fragment read touches is rewritten per tile)
    static_assert(CO == 32, "a narrower CO needs the dp image's pad columns zeroed once (and R * DQ == NTH is assumed by the dp loads)");

    f32x4 accw[CF::NBW][2];
    float dot[CF::NBW];
#pragma unroll
    for (int j = 0; j < CF::NBW; ++j) {
        accw[j][0] = accw[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        dot[j] = 0.f;
    }

    const f32x4* d4 = reinterpret_cast<const f32x4*>(a.d);
    const f32x4* dp4 = reinterpret_cast<const f32x4*>(a.dp);

    const long t0 = (long)blockIdx.x * a.tiles / gridDim.x, t1 = (long)(blockIdx.x + 1) * a.tiles / gridDim.x;
    f32x4 rd[CF::NIT], rp;
    auto prefetch = [&](long tile_) {
        const long tile = min(tile_, a.tiles - 1);          // (uniform: a scalar base + the thread's fixed offset; M % 32 == 0: no row tail)
        const f32x4* db = d4 + tile * (R * CF::Q);
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) {
            const int idx = tid + it * NTH;
            rd[it] = db[((it + 1) * NTH <= R * CF::Q || idx < R * CF::Q) ? idx : 0];
        }
        rp = dp4[tile * (R * CF::DQ) + tid];
    };
    auto flush = [&](int n) {          // the dots of sample n (every lane of a column holds the same sum)
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j)
                if (wave + 4 * j < CF::NBC) atomic_add_f64(a.dgate + (long)n * CE + 16 * (wave + 4 * j) + u, (double)dot[j]);
        }
#pragma unroll
        for (int j = 0; j < CF::NBW; ++j) dot[j] = 0.f;
    };

    prefetch(t0);          // the first tile's loads, then everything else the prologue needs: one memory latency for all of it
    load_coef(a, coef, CE, tid);
    u32x4 wf[CF::NBW][3];
    load_wfrags<CE, CO, CF::NBW>(a.w, wave, u, g, wf);
    int cur_n = -1;
    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gate);
    __syncthreads();
    for (long tile = t0; tile < t1; ++tile) {
        const int n = (int)(tile * R / a.HW);
        if (n != cur_n) {          // (uniform) the gate of the new sample; the finished sample's dots go out
            if (cur_n >= 0) flush(cur_n);
            for (int c = tid; c < CE; c += NTH) gate[c] = ud_sigmoid_fast(a.s[(long)n * CE + c]);
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
            {
                const int row = tid / CF::DQ, q = tid - row * CF::DQ;
                store_split4(dimg + row * CF::DS + q * 8, CF::DPL, rp);
            }
        }
        __syncthreads();
        prefetch(tile + 1);

        // ---- dc tile (registers) and its dot with a
        {
            const lds_char* drow = Lp + (3 * CF::CP + CF::AP) + u * CF::DS + g * 16;
            const Frag3 a0 = read_rows(drow, CF::DPL), a1 = read_rows(drow + 16 * CF::DS, CF::DPL);
            const float* ab = reinterpret_cast<const float*>(abuf) + (4 * g) * (CF::AS / 4) + u;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);
                const Frag3 b = frag_of(wf[j]);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 dc0 = mma6(a0, b, z), dc1 = mma6(a1, b, z);
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t += dc0[r] * ab[r * (CF::AS / 4) + 16 * nb];
                    t += dc1[r] * ab[(16 + r) * (CF::AS / 4) + 16 * nb];
                }
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                dot[j] += t;
            }
        }
        // ---- weight gradient: dWp[co block][channel block] += dp^T c, k = the tile's rows (transposed reads of both images)
        {
            const lds_char* db = Lp + (3 * CF::CP + CF::AP) + (8 * g + q4) * CF::DS + 4 * pp * 2;
            const Frag3 p0 = read_cols(db, CF::DPL, CF::DS), p1 = read_cols(db + 32, CF::DPL, CF::DS);
            const lds_char* cb = Lp + (8 * g + q4) * CF::CS + 4 * pp * 2;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);
                const Frag3 b = read_cols(cb + nb * 32, CF::CP, CF::CS);
                accw[j][0] = mma6(p0, b, accw[j][0]);
                accw[j][1] = mma6(p1, b, accw[j][1]);
            }
        }
        __syncthreads();
    }
    if (cur_n >= 0) flush(cur_n);
    {
        float* part = a.part + (long)blockIdx.x * (CO * CE);
#pragma unroll
        for (int j = 0; j < CF::NBW; ++j) {
            if (wave + 4 * j < CF::NBC) {
                const int ch = 16 * (wave + 4 * j) + u;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = 16 * mb + 4 * g + r;
                        if (co < CO) part[co * CE + ch] = accw[j][mb][r];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel B: dz = (dc gate + dpool / HW) act'(bn1(d)) with dc re-made per tile; BatchNorm-1 backward sums
// ---------------------------------------------------------------------------------------------------------------------
template <int CE, int CO>
__global__ __launch_bounds__(NTH, 2) void pj_bwd_b_kernel(PjArgs a) {
    using CF = PjCfg<CE, CO>;
    extern __shared__ __attribute__((aligned(16))) char L[];
    char* dcbuf = L;
    char* dimg = L + CF::AP;
    float* coef = reinterpret_cast<float*>(dimg + 3 * CF::DPL);
    float* gate = coef + 4 * CE;
    float* dpl = gate + CE;
    double* lacc = reinterpret_cast<double*>(dpl + CE);          // [2][CE]
    const lds_char* Lp = (const lds_char*)L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = lane & 15, g = lane >> 4;
    const bool swish = a.act == 1;

    static_assert(CO == 32, "a narrower CO needs the dp image's pad columns zeroed once (and R * DQ == NTH is assumed by the dp loads)");
    for (int c = tid; c < 2 * CE; c += NTH) lacc[c] = 0.0;

    float sa[CF::NIT][4], sb[CF::NIT][4];
#pragma unroll
    for (int it = 0; it < CF::NIT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) sa[it][k] = sb[it][k] = 0.f;

    const f32x4* d4 = reinterpret_cast<const f32x4*>(a.d);
    const f32x4* dp4 = reinterpret_cast<const f32x4*>(a.dp);
    f32x4* dz4 = reinterpret_cast<f32x4*>(a.dz);

    const long t0 = (long)blockIdx.x * a.tiles / gridDim.x, t1 = (long)(blockIdx.x + 1) * a.tiles / gridDim.x;
    f32x4 rd[CF::NIT], rp;
    auto prefetch = [&](long tile_) {
        const long tile = min(tile_, a.tiles - 1);          // (uniform: a scalar base + the thread's fixed offset; M % 32 == 0: no row tail)
        const f32x4* db = d4 + tile * (R * CF::Q);
#pragma unroll
        for (int it = 0; it < CF::NIT; ++it) {
            const int idx = tid + it * NTH;
            rd[it] = db[((it + 1) * NTH <= R * CF::Q || idx < R * CF::Q) ? idx : 0];
        }
        rp = dp4[tile * (R * CF::DQ) + tid];
    };

    prefetch(t0);
    load_coef(a, coef, CE, tid);
    u32x4 wf[CF::NBW][3];
    load_wfrags<CE, CO, CF::NBW>(a.w, wave, u, g, wf);
    int cur_n = -1;
    const f32x4* c4 = reinterpret_cast<const f32x4*>(coef);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gate);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(dpl);
    __syncthreads();
    for (long tile = t0; tile < t1; ++tile) {
        const int n = (int)(tile * R / a.HW);
        if (n != cur_n) {          // (uniform) the previous tile's element-wise phase still reads gate / dpl: wait for it
            __syncthreads();
            for (int c = tid; c < CE; c += NTH) {
                gate[c] = ud_sigmoid_fast(a.s[(long)n * CE + c]);
                dpl[c] = a.dpool[(long)n * CE + c] * a.inv_hw;
            }
            cur_n = n;
        }
        {
            const int row = tid / CF::DQ, q = tid - row * CF::DQ;
            store_split4(dimg + row * CF::DS + q * 8, CF::DPL, rp);
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
            const Frag3 a0 = read_rows(drow, CF::DPL), a1 = read_rows(drow + 16 * CF::DS, CF::DPL);
            float* ob = reinterpret_cast<float*>(dcbuf) + (4 * g) * (CF::AS / 4) + u;
#pragma unroll
            for (int j = 0; j < CF::NBW; ++j) {
                const int nb = min(wave + 4 * j, CF::NBC - 1);          // (a wave's block past the last one rewrites the last block's values)
                const Frag3 b = frag_of(wf[j]);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 dc0 = mma6(a0, b, z), dc1 = mma6(a1, b, z);
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
            const long b4 = tile * (R * CF::Q);
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
                    sa[it][k] += gg;
                    sb[it][k] += gg * xh;
                }
                if (ok) dz4[b4 + idx] = o;
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
                atomicAdd(&lacc[CE + 4 * q + k], (double)sb[it][k]);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < CE; c += NTH) {
        atomic_add_f64(a.s1 + c, lacc[c]);
        atomic_add_f64(a.s2 + c, lacc[CE + c]);
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
    static_assert(CO == 32, "two 16-column blocks: one (row block, column block) per wave");
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
        load_coef(ca, coef, CE, tid);
    }
    // Wp as the B operand: lane holds Wp[n = 16 nb + u][k = 32 ks + 8 g + j] (k-contiguous in memory)
    u32x4 wf[CF::KS][3];
#pragma unroll
    for (int ks = 0; ks < CF::KS; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 8 * g + j;
            const float wv = a.w[(16 * nb + u) * CE + min(k, CE - 1)];
            v[j] = k < CE ? wv : 0.f;
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
            po[r * CO] = acc[r];
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
        if (tid < 32) {
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

template <int CE, int CO>
int launch_a(const PjArgs& a, int grid, hipStream_t s) {
    using CF = PjCfg<CE, CO>;
    static_assert(CF::LDS_A <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;          // (> 64 KiB of dynamic LDS; set on the first launch, which is never inside a capture: eager warm-up)
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pj_bwd_a_kernel<CE, CO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                CF::LDS_A) != hipSuccess)
            return UD_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL((pj_bwd_a_kernel<CE, CO>), dim3((unsigned)grid), dim3(NTH), CF::LDS_A, s, a);
    return 0;
}
template <int CE, int CO>
int launch_b(const PjArgs& a, int grid, hipStream_t s) {
    using CF = PjCfg<CE, CO>;
    static_assert(CF::LDS_B <= 64 * 1024, "dynamic LDS within the default limit");
    hipLaunchKernelGGL((pj_bwd_b_kernel<CE, CO>), dim3((unsigned)grid), dim3(NTH), CF::LDS_B, s, a);
    return 0;
}

bool args_ok(const void* d, const void* dp, const ud_bn_ref* bn, const void* s, const void* w, int N, int HW, int CE, int CO) {
    return d && dp && bn && s && w && N >= 1 && HW >= R && ud_pj_bwd_fused_ok(CE, CO, HW) && bn->G == 1 && bn->gamma && bn->beta;
}

void fill(PjArgs& a, const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW) {
    a.d = d; a.dp = dp; a.w = w; a.s = s;
    a.bsum = bn->sum; a.bsumsq = bn->sumsq; a.gamma = bn->gamma; a.beta = bn->beta;
    a.inv_count = bn->inv_count; a.eps = bn->eps; a.act = bn->act; a.HW = HW;
    a.M = (long)N * HW; a.tiles = a.M / R;
    a.dpool = nullptr; a.dz = nullptr; a.part = nullptr; a.dgate = nullptr; a.s1 = a.s2 = nullptr; a.inv_hw = 0.f;
}

}  // namespace

extern "C" {

int ud_pj_bwd_fused_ok(int CE, int CO, int HW) { return ((CE == 144 || CE == 192) && CO == 32 && HW >= R && HW % R == 0) ? 1 : 0; }

long ud_pj_bwd_fused_grid(int N, int HW) {
    if (N < 1 || HW < R || HW % R) return UD_EINVAL;
    const long tiles = (long)N * HW / R;
    return tiles < 512 ? tiles : 512;
}

int ud_pj_fwd_fused(const float* d, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO, float* p,
                     double* sum, double* sumsq, ud_stream_t stream) {
    if (!d || !bn || !s || !w || !p || N < 1 || !ud_pj_bwd_fused_ok(CE, CO, HW) || bn->G != 1 || !bn->gamma || !bn->beta || (!sum != !sumsq))
        return UD_EINVAL;
    PjFwdArgs a;
    a.d = d; a.w = w; a.s = s; a.p = p; a.sum = sum; a.sumsq = sumsq;
    a.bsum = bn->sum; a.bsumsq = bn->sumsq; a.gamma = bn->gamma; a.beta = bn->beta;
    a.inv_count = bn->inv_count; a.eps = bn->eps; a.act = bn->act; a.HW = HW;
    a.M = (long)N * HW; a.tiles = a.M / R;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW);
    const int rc = CE == 144 ? launch_f<144, 32>(a, grid, (hipStream_t)stream) : launch_f<192, 32>(a, grid, (hipStream_t)stream);
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_pj_bwd_fused_a(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO,
                      float* dw, double* dgate, float* part, ud_stream_t stream) {
    if (!args_ok(d, dp, bn, s, w, N, HW, CE, CO) || !dw || !dgate || !part) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PjArgs a;
    fill(a, d, dp, bn, s, w, N, HW);
    a.part = part; a.dgate = dgate;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW);
    const int rc = CE == 144 ? launch_a<144, 32>(a, grid, st) : launch_a<192, 32>(a, grid, st);
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return pw::fold_launch(part, grid, CO * CE, dw, nullptr, nullptr, 0, nullptr, nullptr, st);
}

int ud_pj_bwd_fused_b(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* dpool, float inv_hw,
                      const float* w, int N, int HW, int CE, int CO, float* dz, double* s1, double* s2, ud_stream_t stream) {
    if (!args_ok(d, dp, bn, s, w, N, HW, CE, CO) || !dpool || !dz || !s1 || !s2) return UD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PjArgs a;
    fill(a, d, dp, bn, s, w, N, HW);
    a.dpool = dpool; a.inv_hw = inv_hw; a.dz = dz; a.s1 = s1; a.s2 = s2;
    const int grid = (int)ud_pj_bwd_fused_grid(N, HW);
    const int rc = CE == 144 ? launch_b<144, 32>(a, grid, st) : launch_b<192, 32>(a, grid, st);
    if (rc) return rc;
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
