// Pass-2 input perturbations of the train step: no-grad preprocessing of the [N,3,H,W]
// input batch between the two passes of AbstractEngine.train_unidefense_model.  All HBM-bound, NCHW planes (the
// network input layout), fp32.
//
// Reference:  model/unidefense.py:177-198 (branching), model/modules.py:7-21 (noise / blur / downscale),
//             model/modules.py:35-55 (frequency amplitude transfer), :58-76 (exact feature-distribution matching),
//             utils/operation.py:7-45 (CORAL colour transfer).
// The 2-D FFTs of the amplitude transfer run as DFT-matrix GEMMs on ud_gemm (kernels.dft_rfft2_planes); this file
// holds the spectrum mixing between them.  The per-(sample, channel) sorts of the distribution matching are a SEGMENTED
// radix sort written here (seg_sort_kernel: one workgroup per row and per tensor — content and style in ONE launch —
// four stable 8-bit passes through a ping-pong workspace); the rank gather of the reference (argsort of argsort + gather)
// is folded into one scatter pass.
#include <cstring>

#include "ud_common.h"

namespace {

constexpr int NT = 256;

inline int blocks_for(long total, int cap = 16384) {
    long b = (total + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// out[p][y][x] = in[p][iy[y]][ix[x]]   — F.interpolate(nearest) x0.75 and back, composed into one gather
__global__ __launch_bounds__(NT) void gather2d(const float* __restrict__ in, float* __restrict__ out,
                                               const int* __restrict__ iy, const int* __restrict__ ix, long total,
                                               int H, int W) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int x = (int)(e % W);
        const long t = e / W;
        const int y = (int)(t % H);
        const long p = t / H;
        out[e] = in[(p * H + iy[y]) * W + ix[x]];
    }
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// 5x5 separable Gaussian with reflect padding, evaluated as rows-then-columns like torchvision's conv on the padded
// image (the 25 products are summed row-major there; the difference is rounding only)
__global__ __launch_bounds__(NT) void blur5(const float* __restrict__ in, float* __restrict__ out, long total, int H,
                                            int W, float k0, float k1, float k2) {
    const float k[5] = {k0, k1, k2, k1, k0};
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int x = (int)(e % W);
        const long t = e / W;
        const int y = (int)(t % H);
        const float* pl = in + (t / H) * (long)H * W;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const float* row = pl + (long)reflect(y + dy - 2, H) * W;
            float r = 0.f;
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) r += k[dx] * row[reflect(x + dx - 2, W)];
            acc += k[dy] * r;
        }
        out[e] = acc;
    }
}

// Spectra as kernels.dft_rfft2_planes lays them out: Y[p][2S][Whp], rows [0,S) = Re(ky), [S,2S) = Im(ky), columns
// [0, S/2] valid.  out = w(kx) * (l |A| + (1-l) |B|) * A/|A|   (|A| = 0: phase 0, like torch.angle), w = 2 on the
// interior columns so that the ADJOINT of the forward transform (dft_rfft2_planes_adjoint) is irfft2.
__global__ __launch_bounds__(NT) void amp_mix(const float* __restrict__ A, const float* __restrict__ B,
                                              const float* __restrict__ lmda, float* __restrict__ out, long total,
                                              int S, int Whp, int planes_per_sample) {
    const int Wh = S / 2 + 1;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int kx = (int)(e % Whp);
        const long t = e / Whp;
        const int ky = (int)(t % S);
        const long p = t / S;
        const long re_i = (p * 2 * S + ky) * Whp + kx, im_i = re_i + (long)S * Whp;
        if (kx >= Wh) {
            out[re_i] = 0.f;
            out[im_i] = 0.f;
            continue;
        }
        const float l = lmda[p / planes_per_sample];
        const float ar = A[re_i], ai = A[im_i], br = B[re_i], bi = B[im_i];
        const float ma = hypotf(ar, ai), mb = hypotf(br, bi);
        const float amp = l * ma + (1.0f - l) * mb;
        const float w = (kx == 0 || kx == S / 2) ? 1.f : 2.f;
        const float c = ma > 0.f ? ar / ma : 1.f, s = ma > 0.f ? ai / ma : 0.f;
        out[re_i] = w * amp * c;
        out[im_i] = w * amp * s;
    }
}

// Order-preserving map float -> uint32 (negative: all bits flipped, else sign bit set) and back.
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- segmented LSD radix sort: rows x L floats, ascending, STABLE (ties keep their index order — what torch.sort gives the
// reference on the CPU, model/modules.py:67-70), payload = the element's index in its row.
// One workgroup of 1024 threads per (row, tensor): 96 content rows + 96 style rows = 192 workgroups on 256 CUs, one launch.
// A pre-pass builds the four 256-bin digit histograms of the row (the multiset of keys does not change between passes); each
// pass then walks the row in rounds of 1024 consecutive elements: a lane finds the lanes of its wave holding the same digit with
// eight ballots (rank inside the wave = population count below it), the 16 waves' per-digit counts are scanned in LDS, and
// the element goes to  bucket base + elements of that digit in earlier rounds + in earlier waves + rank.  A round's 4096
// elements are first put in digit order in LDS and written out from there, so that the stores are runs (16 elements per digit
// on average) instead of 4-byte scatters over 256 buckets (which bound the first version: 17 us per round).  The row ping-pongs
// between two workspace buffers (1 MB per row at 256 x 256: L2 / Infinity Cache resident).
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, which would wait for
// the round's scattered stores and the prefetched loads of the next round at every one of the six barriers of a round
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int SORT_NT = 1024, SORT_E = 4, SORT_ROUND = SORT_NT * SORT_E, SORT_VW = SORT_E * SORT_NT / 64;   // virtual waves per round

__global__ __launch_bounds__(SORT_NT) void seg_sort_kernel(const float* __restrict__ content, const float* __restrict__ style,
                                                           unsigned* __restrict__ ws, int rows, int L) {
    __shared__ unsigned hist[4][256];
    __shared__ unsigned base[256], running[256], tot[256];
    __shared__ unsigned wcount[SORT_VW][256];          // 64 KB: per (sub-round, wave) and digit
    __shared__ unsigned lstart[256];                   // first round-local position of a digit
    __shared__ unsigned psum[4][256];
    __shared__ unsigned kbuf[SORT_ROUND], ibuf[SORT_ROUND];          // the round in digit order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, which = blockIdx.y;
    const long n = (long)rows * L;
    const float* src = (which ? style : content) + (long)row * L;
    // workspace: [which][keysA | idxA | keysB | idxB][rows][L]
    unsigned* keysA = ws + ((long)which * 4 + 0) * n + (long)row * L;
    unsigned* idxA = ws + ((long)which * 4 + 1) * n + (long)row * L;
    unsigned* keysB = ws + ((long)which * 4 + 2) * n + (long)row * L;
    unsigned* idxB = ws + ((long)which * 4 + 3) * n + (long)row * L;

    for (int i = tid; i < 4 * 256; i += SORT_NT) (&hist[0][0])[i] = 0;
    __syncthreads();
    for (int i = tid; i < L; i += SORT_NT) {
        const unsigned k = f2key(src[i]);
        atomicAdd(&hist[0][k & 255], 1u);
        atomicAdd(&hist[1][(k >> 8) & 255], 1u);
        atomicAdd(&hist[2][(k >> 16) & 255], 1u);
        atomicAdd(&hist[3][k >> 24], 1u);
    }
    __syncthreads();

    const int rounds = (L + SORT_ROUND - 1) / SORT_ROUND;
    for (int pass = 0; pass < 4; ++pass) {
        const unsigned* kin = (pass & 1) ? keysA : keysB;          // pass 0 reads the floats; 1: A -> B; 2: B -> A; 3: A -> B
        const unsigned* iin = (pass & 1) ? idxA : idxB;
        unsigned* kout = (pass & 1) ? keysB : keysA;
        unsigned* iout = (pass & 1) ? idxB : idxA;
        // exclusive scan of this digit's histogram (256 threads, Hillis-Steele through `tot`)
        if (tid < 256) tot[tid] = hist[pass][tid];
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            unsigned v = 0;
            if (tid < 256 && tid >= o) v = tot[tid - o];
            __syncthreads();
            if (tid < 256) tot[tid] += v;
            __syncthreads();
        }
        if (tid < 256) {
            base[tid] = tot[tid] - hist[pass][tid];
            running[tid] = 0;
        }
        // the next round's elements are loaded while this round is ranked and written (a round would otherwise start with
        // a dependent L2 / Infinity Cache round trip); the style rows carry no payload
        unsigned nkey[SORT_E], nidx[SORT_E];
        auto fetch = [&](int r) {
#pragma unroll
            for (int e = 0; e < SORT_E; ++e) {
                const int i = r * SORT_ROUND + e * SORT_NT + tid;
                nkey[e] = 0;
                nidx[e] = 0;
                if (i < L) {
                    nkey[e] = pass == 0 ? f2key(src[i]) : kin[i];
                    if (which == 0) nidx[e] = pass == 0 ? (unsigned)i : iin[i];
                }
            }
        };
        fetch(0);
        for (int r = 0; r < rounds; ++r) {
            for (int i = tid; i < SORT_VW * 256; i += SORT_NT) (&wcount[0][0])[i] = 0;
            lds_barrier();
            // SORT_E consecutive runs of 1024 elements: element (sub-round e, thread t) = r * 4096 + e * 1024 + t, so the order
            // (sub-round, wave, lane) is the input order
            unsigned key[SORT_E], idx[SORT_E], rank[SORT_E];
            bool valid[SORT_E];
#pragma unroll
            for (int e = 0; e < SORT_E; ++e) {
                valid[e] = r * SORT_ROUND + e * SORT_NT + tid < L;
                key[e] = nkey[e];
                idx[e] = nidx[e];
            }
            if (r + 1 < rounds) fetch(r + 1);
#pragma unroll
            for (int e = 0; e < SORT_E; ++e) {
                const unsigned digit = (key[e] >> (8 * pass)) & 255u;
                unsigned long long same = __ballot(valid[e]);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const unsigned long long bal = __ballot((digit >> b) & 1u);
                    same &= ((digit >> b) & 1u) ? bal : ~bal;
                }
                rank[e] = __popcll(same & ((1ull << lane) - 1ull));
                if (valid[e] && rank[e] == 0) wcount[e * (SORT_NT / 64) + wave][digit] = __popcll(same);
            }
            lds_barrier();
            {
                // scan of the 64 (sub-round, wave) counts of every digit by all 1024 threads: thread (digit d, quarter q) reads its
                // 16 counts into registers FIRST (a serial read-add-write chain over 64 entries by 256 threads cost ~5 us per
                // round: every step waited for an LDS round trip), the four quarters meet through psum
                const int d = tid & 255, q = tid >> 8;
                unsigned c[SORT_VW / 4];
#pragma unroll
                for (int i = 0; i < SORT_VW / 4; ++i) c[i] = wcount[q * (SORT_VW / 4) + i][d];
                unsigned sum = 0;
#pragma unroll
                for (int i = 0; i < SORT_VW / 4; ++i) { const unsigned v = c[i]; c[i] = sum; sum += v; }
                psum[q][d] = sum;
                lds_barrier();
                unsigned off = 0;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) off += (qq < q) ? psum[qq][d] : 0u;
#pragma unroll
                for (int i = 0; i < SORT_VW / 4; ++i) wcount[q * (SORT_VW / 4) + i][d] = off + c[i];
                if (q == 3) tot[d] = off + sum;
            }
            lds_barrier();
            if (wave == 0) {          // exclusive scan of tot[] over the digits: 4 per lane + a wave scan
                unsigned t4[4], sum = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { t4[q] = tot[lane * 4 + q]; sum += t4[q]; }
                unsigned inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned v = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += v;
                }
                unsigned ex = inc - sum;
#pragma unroll
                for (int q = 0; q < 4; ++q) { lstart[lane * 4 + q] = ex; ex += t4[q]; }
            }
            lds_barrier();
#pragma unroll
            for (int e = 0; e < SORT_E; ++e) {
                if (valid[e]) {
                    const unsigned digit = (key[e] >> (8 * pass)) & 255u;
                    const unsigned lp = lstart[digit] + wcount[e * (SORT_NT / 64) + wave][digit] + rank[e];
                    kbuf[lp] = key[e];
                    if (which == 0) ibuf[lp] = idx[e];
                }
            }
            lds_barrier();
            const int in_round = min(SORT_ROUND, L - r * SORT_ROUND);
            for (int j = tid; j < in_round; j += SORT_NT) {
                const unsigned k = kbuf[j], d = (k >> (8 * pass)) & 255u;
                const unsigned pos = base[d] + running[d] + ((unsigned)j - lstart[d]);
                kout[pos] = k;
                if (which == 0) iout[pos] = ibuf[j];
            }
            lds_barrier();
            if (tid < 256) running[tid] += tot[tid];
        }
        // the next pass reads what other waves of THIS workgroup just wrote: same CU, same (write-through) L1 — workgroup scope.
        // __syncthreads() waits for the stores (vmcnt(0)) and meets; an agent-scope fence here would write back this XCD's whole
        // L2 once per pass and workgroup.
        __syncthreads();
    }
}

// sorted position j of row r holds content index i = sidx[r][j] and the style value of the same rank:
// out[r][i] = (c + (1-l) * sv) - (1-l) * c      (model/modules.py:70-73, same operation order, no contraction)
__global__ __launch_bounds__(NT) void efdm_scatter(const float* __restrict__ content, const unsigned* __restrict__ sidx,
                                                   const unsigned* __restrict__ sk_sorted,
                                                   const float* __restrict__ lmda, float* __restrict__ out, long total,
                                                   unsigned L, int rows_per_sample) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const long r = e / L;
        const long i = r * L + sidx[e];
        const float om = __fsub_rn(1.0f, lmda[r / rows_per_sample]);
        const float c = content[i], sv = key2f(sk_sorted[e]);
        out[i] = __fsub_rn(__fadd_rn(c, __fmul_rn(om, sv)), __fmul_rn(om, c));
    }
}

// per (sample, chunk): sum x_c (3) and sum x_c x_d (6: 00 01 02 11 12 22) over the chunk's pixels, fp64
__global__ __launch_bounds__(NT) void coral_moments(const float* __restrict__ x, double* __restrict__ part, int HW,
                                                    int chunks) {
    const int n = blockIdx.y, ch = blockIdx.x;
    const float* p0 = x + (long)n * 3 * HW;
    const int per = (HW + chunks - 1) / chunks, lo = ch * per, hi = min(HW, lo + per);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lo + threadIdx.x; i < hi; i += NT) {
        const double a = p0[i], b = p0[HW + i], c = p0[2 * HW + i];
        acc[0] += a; acc[1] += b; acc[2] += c;
        acc[3] += a * a; acc[4] += a * b; acc[5] += a * c;
        acc[6] += b * b; acc[7] += b * c; acc[8] += c * c;
    }
    __shared__ double sm[NT / 64][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double v = ud_wave_sum_d(acc[k]);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double t = 0;
        for (int w = 0; w < NT / 64; ++w) t += sm[w][threadIdx.x];
        part[((long)n * chunks + ch) * 9 + threadIdx.x] = t;
    }
}

// out[n][c][i] = sum_k M[n][c][k] * x[n][k][i] + M[n][c][3]
__global__ __launch_bounds__(NT) void affine3(const float* __restrict__ x, const float* __restrict__ M,
                                              float* __restrict__ out, int HW) {
    const int n = blockIdx.y;
    const float* m = M + n * 12;
    const float* p0 = x + (long)n * 3 * HW;
    float* o0 = out + (long)n * 3 * HW;
    for (int i = blockIdx.x * NT + threadIdx.x; i < HW; i += gridDim.x * NT) {
        const float a = p0[i], b = p0[HW + i], c = p0[2 * HW + i];
#pragma unroll
        for (int r = 0; r < 3; ++r) o0[r * HW + i] = m[r * 4] * a + m[r * 4 + 1] * b + m[r * 4 + 2] * c + m[r * 4 + 3];
    }
}

}  // namespace

extern "C" {

int ud_gather2d(const float* in, float* out, const int* iy, const int* ix, long planes, int H, int W,
                ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    const long total = planes * H * W;
    if (total <= 0) return 0;
    gather2d<<<blocks_for(total), NT, 0, stream>>>(in, out, iy, ix, total, H, W);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_blur5_reflect(const float* in, float* out, long planes, int H, int W, float k0, float k1, float k2,
                     ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    if (H < 3 || W < 3) return UD_EINVAL;          // reflect padding of 2 needs at least 3 pixels
    const long total = planes * H * W;
    if (total <= 0) return 0;
    blur5<<<blocks_for(total), NT, 0, stream>>>(in, out, total, H, W, k0, k1, k2);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_amp_mix(const float* A, const float* B, const float* lmda, float* out, long planes, int S, int Whp,
               int planes_per_sample, ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    if (Whp < S / 2 + 1 || planes_per_sample < 1) return UD_EINVAL;
    const long total = planes * S * Whp;
    if (total <= 0) return 0;
    amp_mix<<<blocks_for(total), NT, 0, stream>>>(A, B, lmda, out, total, S, Whp, planes_per_sample);
    UD_LAUNCH_CHECK();
    return 0;
}

long ud_efdm_ws_bytes(int rows, int L) {
    if (rows <= 0 || L <= 0 || (long)rows * L >= (1L << 31)) return UD_EINVAL;
    return 8L * rows * L * (long)sizeof(unsigned);          // content and style: keys + indices, ping and pong
}

int ud_efdm(const float* content, const float* style, const float* lmda, float* out, int rows, int L,
            int rows_per_sample, void* ws, long ws_bytes, ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    const long need = ud_efdm_ws_bytes(rows, L);
    if (need < 0 || ws_bytes < need || rows_per_sample < 1 || !content || !style || !lmda || !out || !ws) return UD_EINVAL;
    const long n = (long)rows * L;
    unsigned* w = (unsigned*)ws;
    seg_sort_kernel<<<dim3((unsigned)rows, 2), SORT_NT, 0, stream>>>(content, style, w, rows, L);
    UD_LAUNCH_CHECK();
    // after the fourth pass: content's sorted indices in its idxB, style's sorted keys in its keysB
    efdm_scatter<<<blocks_for(n), NT, 0, stream>>>(content, w + 3 * n, w + 6 * n, lmda, out, n, (unsigned)L, rows_per_sample);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_coral_moments(const float* x, double* part, int N, int HW, int chunks, ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    if (N <= 0) return 0;
    if (chunks < 1 || HW < 1) return UD_EINVAL;
    coral_moments<<<dim3(chunks, N), NT, 0, stream>>>(x, part, HW, chunks);
    UD_LAUNCH_CHECK();
    return 0;
}

int ud_affine3(const float* x, const float* M, float* out, int N, int HW, ud_stream_t sh) {
    hipStream_t stream = (hipStream_t)sh;
    if (N <= 0 || HW <= 0) return 0;
    int bx = (HW + NT - 1) / NT;
    if (bx > 64) bx = 64;
    affine3<<<dim3(bx, N), NT, 0, stream>>>(x, M, out, HW);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
