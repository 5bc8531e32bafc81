// Pass-2 input perturbations of the train step: no-grad preprocessing of the [N,3,H,W]
// input batch between the two passes of AbstractEngine.train_unidefense_model.  All HBM-bound, NCHW planes (the
// network input layout), fp32.
//
// Reference:  model/unidefense.py:177-198 (branching), model/modules.py:7-21 (noise / blur / downscale),
//             model/modules.py:35-55 (frequency amplitude transfer), :58-76 (exact feature-distribution matching),
//             utils/operation.py:7-45 (CORAL colour transfer).
// The 2-D FFTs of the amplitude transfer run as DFT-matrix GEMMs on ud_gemm (kernels.dft_rfft2_planes); this file
// holds the spectrum mixing between them.  The per-(sample, channel) sorts of the distribution matching are ONE
// device-wide rocPRIM radix sort over composite (row, value) keys (ROCm's primitive — there is nothing
// UniDefense-specific to gain by re-writing it); the rank gather of the reference (argsort of argsort + gather) is
// folded into one scatter pass.
#include <cstring>
#include <rocprim/rocprim.hpp>

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

// 96 rows of 65 536 values: a segmented sort keeps one workgroup per row (96 of 256 CUs busy).  Instead ONE
// device-wide radix sort over composite 64-bit keys (row << 32 | ordered value bits), 32 + ceil(log2 rows) bits.
__global__ __launch_bounds__(NT) void efdm_keys(const float* __restrict__ content, const float* __restrict__ style,
                                                unsigned long long* __restrict__ ck, unsigned long long* __restrict__ sk,
                                                unsigned* __restrict__ idx, long total, unsigned L) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const unsigned long long r = (unsigned long long)(e / L) << 32;
        ck[e] = r | f2key(content[e]);
        sk[e] = r | f2key(style[e]);
        idx[e] = (unsigned)(e % L);
    }
}

// sorted position j of row r holds content index i = sidx[r][j] and the style value of the same rank:
// out[r][i] = (c + (1-l) * sv) - (1-l) * c      (model/modules.py:70-73, same operation order, no contraction)
__global__ __launch_bounds__(NT) void efdm_scatter(const float* __restrict__ content, const unsigned* __restrict__ sidx,
                                                   const unsigned long long* __restrict__ sk_sorted,
                                                   const float* __restrict__ lmda, float* __restrict__ out, long total,
                                                   unsigned L, int rows_per_sample) {
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const long r = e / L;
        const long i = r * L + sidx[e];
        const float om = __fsub_rn(1.0f, lmda[r / rows_per_sample]);
        const float c = content[i], sv = key2f((unsigned)sk_sorted[e]);
        out[i] = __fsub_rn(__fadd_rn(c, __fmul_rn(om, sv)), __fmul_rn(om, c));
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline unsigned key_bits(unsigned rows) {
    unsigned b = 0;
    while ((1u << b) < rows) ++b;
    return 32 + b;
}

hipError_t sort_temp_bytes(unsigned rows, unsigned L, size_t* pairs, size_t* keys) {
    typedef unsigned long long u64;
    const unsigned bits = key_bits(rows);
    hipError_t err = rocprim::radix_sort_pairs(nullptr, *pairs, (const u64*)nullptr, (u64*)nullptr,
                                               (const unsigned*)nullptr, (unsigned*)nullptr, (size_t)rows * L, 0, bits,
                                               (hipStream_t)0);
    if (err != hipSuccess) return err;
    return rocprim::radix_sort_keys(nullptr, *keys, (const u64*)nullptr, (u64*)nullptr, (size_t)rows * L, 0, bits,
                                    (hipStream_t)0);
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
    size_t tp = 0, tk = 0;
    if (sort_temp_bytes((unsigned)rows, (unsigned)L, &tp, &tk) != hipSuccess) return UD_EINVAL - 1;
    const size_t n = (size_t)rows * L;
    // content keys, style keys, one sorted-key buffer (reused), index in / out, rocPRIM's temporary storage
    return (long)(3 * align256(n * 8) + 2 * align256(n * 4) + align256(tp > tk ? tp : tk));
}

int ud_efdm(const float* content, const float* style, const float* lmda, float* out, int rows, int L,
            int rows_per_sample, void* ws, long ws_bytes, ud_stream_t sh) {
    typedef unsigned long long u64;
    hipStream_t stream = (hipStream_t)sh;
    const long need = ud_efdm_ws_bytes(rows, L);
    if (need < 0 || ws_bytes < need || rows_per_sample < 1) return UD_EINVAL;
    const size_t n = (size_t)rows * L, s8 = align256(n * 8), s4 = align256(n * 4);
    char* base = (char*)ws;
    u64* ck = (u64*)base;
    u64* sk = (u64*)(base + s8);
    u64* sorted = (u64*)(base + 2 * s8);
    unsigned* idx = (unsigned*)(base + 3 * s8);
    unsigned* sidx = (unsigned*)(base + 3 * s8 + s4);
    void* temp = base + 3 * s8 + 2 * s4;
    size_t temp_bytes = (size_t)ws_bytes - (3 * s8 + 2 * s4);
    const unsigned bits = key_bits((unsigned)rows);
    efdm_keys<<<blocks_for((long)n), NT, 0, stream>>>(content, style, ck, sk, idx, (long)n, (unsigned)L);
    UD_LAUNCH_CHECK();
    hipError_t err = rocprim::radix_sort_pairs(temp, temp_bytes, (const u64*)ck, sorted, (const unsigned*)idx, sidx, n,
                                               0, bits, stream);
    if (err != hipSuccess) return -(int)err;
    err = rocprim::radix_sort_keys(temp, temp_bytes, (const u64*)sk, sorted, n, 0, bits, stream);   // content's sorted keys are dead
    if (err != hipSuccess) return -(int)err;
    efdm_scatter<<<blocks_for((long)n), NT, 0, stream>>>(content, sidx, sorted, lmda, out, (long)n, (unsigned)L,
                                                         rows_per_sample);
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
