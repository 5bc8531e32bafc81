// Real 2-D FFT of [P planes][S][S] fp32 (NCHW image planes), S in {128, 256, 320}, and its adjoint.
//
// Serves the frequency reconstruction loss rfft2(rec) - rfft2(x) on 256 x 256 / 320 x 320 / 128 x 128 images
// (model/unidefense.py:244-253 and the ResNet variants :421-431, :615-625) and the amplitude transfer of the pass-2
// perturbation (model/modules.py:35-55), forward and backward.  Replaces the DFT-matrix GEMM stand-in of round 1.
//
// Two launches per transform, both streaming:
//   rows    : one 64-lane group per image row: Stockham autosort FFT (radix 4 / 2 / 5 passes, ping-pong in LDS, twiddles
//             from an LDS table built with sincospi), the half spectrum kx <= S/2 goes to the scratch T[P][S][Whp] (float2)
//   columns : one workgroup per (plane, strip of 8 kx): the strip's S x 8 complex values staged in LDS, four 64-lane groups
//             transform two columns each along ky, result written as Y[P][2S][Whp]: rows [0,S) = Re, rows [S,2S) = Im
//             (Whp = ceil4(S/2+1); the padding columns are written as zeros) — the layout ud_amp_mix and the L1 loss read.
// The adjoint runs the conjugate transforms in reverse order (columns, then rows with the half spectrum zero-extended and
// the real part kept).  HBM traffic: 4 S^2 (read) + 2 x 8 S Whp (scratch) + 8 S Whp (write) bytes per plane.
#include "ud_common.h"

namespace {

constexpr int NT = 256;
constexpr int TPR = 64;          // lanes per transform
constexpr int RPW = NT / TPR;    // transforms in flight per workgroup
constexpr int CB = 8;            // kx columns per workgroup in the column pass

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
// multiplication by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot(float2 a) { return INV ? float2{-a.y, a.x} : float2{a.y, -a.x}; }

// One Stockham autosort pass of radix R over n = N complex values: sub-transform size NS (product of the previous
// radices; a power of two for every plan here), executed by the TPR lanes of ONE wave — LDS operations of a wave
// complete in issue order, so consecutive passes of the same wave need no barrier, only a scheduling fence.
template <int N, int R, int NS, bool INV>
__device__ __forceinline__ void fft_pass(const float2* __restrict__ a, float2* __restrict__ b,
                                         const float2* __restrict__ tw, int lane) {
    constexpr int NB = N / R;
    constexpr int TS = N / (NS * R);           // twiddle stride
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += TPR) {
        const int j = j0 + lane;
        if (NB % TPR != 0 && j >= NB) break;
        const int jm = j % NS;
        float2 v[R];
        v[0] = a[j];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            float2 w = tw[jm * r * TS];
            if (INV) w.y = -w.y;
            v[r] = cmul(a[j + r * NB], w);
        }
        float2* o = b + (j / NS) * NS * R + jm;
        if constexpr (R == 2) {
            o[0] = cadd(v[0], v[1]);
            o[NS] = csub(v[0], v[1]);
        } else if constexpr (R == 4) {
            const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
            const float2 s13 = cadd(v[1], v[3]), d13 = rot<INV>(csub(v[1], v[3]));
            o[0] = cadd(s02, s13);
            o[NS] = cadd(d02, d13);
            o[2 * NS] = csub(s02, s13);
            o[3 * NS] = csub(d02, d13);
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                float2 acc = v[0];
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    float2 w = tw[((q * r) % R) * (N / R)];
                    if (INV) w.y = -w.y;
                    acc = cadd(acc, cmul(v[r], w));
                }
                o[q * NS] = acc;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// FFT of N complex values in LDS buffer a (b: scratch of the same size) by one wave; returns the result buffer.
template <int N, bool INV>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const float2* __restrict__ tw, int lane) {
    constexpr int RL = (N == 128) ? 2 : (N == 320 ? 5 : 4);       // last radix: 128 = 4^3 * 2, 256 = 4^4, 320 = 4^3 * 5
    fft_pass<N, 4, 1, INV>(a, b, tw, lane);
    fft_pass<N, 4, 4, INV>(b, a, tw, lane);
    fft_pass<N, 4, 16, INV>(a, b, tw, lane);
    fft_pass<N, RL, 64, INV>(b, a, tw, lane);
    return a;
}

__device__ __forceinline__ void build_twiddles(float2* tw, int n) {
    for (int k = threadIdx.x; k < n; k += NT) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)n, &s, &c);
        tw[k] = float2{c, s};
    }
}

// rows: x[P][S][S] -> T[P][S][Whp] (float2), kx <= S/2 valid, padding zero
template <int S>
__global__ __launch_bounds__(NT) void rows_fwd(const float* __restrict__ x, float2* __restrict__ T, int Whp,
                                              long nrows, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2* tw = lds;
    float2* bufs = lds + S;
    const int grp = threadIdx.x / TPR, lane = threadIdx.x % TPR;
    build_twiddles(tw, S);
    const long row = (long)blockIdx.x * RPW + grp;
    const bool ok = row < nrows;
    float2* a = bufs + (size_t)grp * 2 * S;
    float2* b = a + S;
    if (ok) {
        const float* src = x + row * S;
        for (int w = lane; w < S; w += TPR) a[w] = float2{src[w], 0.f};
    }
    __syncthreads();
    float2* r = fft_lds<S, false>(a, b, tw, lane);
    if (ok) {
        float2* dst = T + row * Whp;
        for (int k = lane; k < Whp; k += TPR) dst[k] = (k <= S / 2) ? float2{r[k].x * scale, r[k].y * scale} : float2{0.f, 0.f};
    }
}

// adjoint of rows_fwd: dT[P][S][Whp] -> dx[P][S][S] = scale * Re(IFFT_unnormalised(zero-extended dT row))
template <int S>
__global__ __launch_bounds__(NT) void rows_adj(const float2* __restrict__ dT, float* __restrict__ dx, int Whp,
                                              long nrows, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2* tw = lds;
    float2* bufs = lds + S;
    const int grp = threadIdx.x / TPR, lane = threadIdx.x % TPR;
    build_twiddles(tw, S);
    const long row = (long)blockIdx.x * RPW + grp;
    const bool ok = row < nrows;
    float2* a = bufs + (size_t)grp * 2 * S;
    float2* b = a + S;
    if (ok) {
        const float2* src = dT + row * Whp;
        for (int k = lane; k < S; k += TPR) a[k] = (k <= S / 2) ? src[k] : float2{0.f, 0.f};
    }
    __syncthreads();
    float2* r = fft_lds<S, true>(a, b, tw, lane);
    if (ok) {
        float* dst = dx + row * S;
        for (int w = lane; w < S; w += TPR) dst[w] = r[w].x * scale;
    }
}

// columns.  FWD: T[P][S][Whp] -> Y[P][2S][Whp] (Re rows | Im rows), forward transform along ky.
//           !FWD (adjoint): dY[P][2S][Whp] -> dT[P][S][Whp], conjugate transform along ky.
template <int S, bool FWD>
__global__ __launch_bounds__(NT) void cols_pass(const float* __restrict__ in, float* __restrict__ out, int Whp,
                                                float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2* tw = lds;
    float2* bufs = lds + S;                  // [CB columns][2][S]
    const int grp = threadIdx.x / TPR, lane = threadIdx.x % TPR;
    build_twiddles(tw, S);
    const int p = blockIdx.y;
    const int kx0 = blockIdx.x * CB;
    // stage the strip: element (ky, c) of the strip -> column buffer c
    for (int e = threadIdx.x; e < S * CB; e += NT) {
        const int ky = e / CB, c = e % CB;
        const int kx = kx0 + c;
        float2 v{0.f, 0.f};
        if (kx < Whp) {
            if (FWD) {
                v = reinterpret_cast<const float2*>(in)[((long)p * S + ky) * Whp + kx];
            } else {
                v.x = in[((long)p * 2 * S + ky) * Whp + kx];
                v.y = in[((long)p * 2 * S + S + ky) * Whp + kx];
            }
        }
        bufs[(size_t)c * 2 * S + ky] = v;
    }
    __syncthreads();
    // CB = 2 * RPW columns: two rounds of RPW concurrent transforms (results land back in the first half of each buffer)
#pragma unroll
    for (int rd = 0; rd < CB / RPW; ++rd) {
        float2* a = bufs + (size_t)(rd * RPW + grp) * 2 * S;
        fft_lds<S, !FWD>(a, a + S, tw, lane);
    }
    __syncthreads();
    constexpr size_t roff = 0;
    for (int e = threadIdx.x; e < S * CB; e += NT) {
        const int ky = e / CB, c = e % CB;
        const int kx = kx0 + c;
        if (kx >= Whp) continue;
        const float2 v = bufs[(size_t)c * 2 * S + roff + ky];
        if (FWD) {
            const bool valid = kx <= S / 2;
            out[((long)p * 2 * S + ky) * Whp + kx] = valid ? v.x * scale : 0.f;
            out[((long)p * 2 * S + S + ky) * Whp + kx] = valid ? v.y * scale : 0.f;
        } else {
            reinterpret_cast<float2*>(out)[((long)p * S + ky) * Whp + kx] = float2{v.x * scale, v.y * scale};
        }
    }
}

inline bool size_ok(int S) { return S == 128 || S == 256 || S == 320; }
inline int whp_of(int S) { return (S / 2 + 1 + 3) / 4 * 4; }
inline size_t rows_lds(int S) { return (size_t)(S + RPW * 2 * S) * sizeof(float2); }
inline size_t cols_lds(int S) { return (size_t)(S + CB * 2 * S) * sizeof(float2); }

template <int S>
int launch_fwd(const float* x, float* Y, float* ws, long P, float scale, hipStream_t s) {
    const int Whp = whp_of(S);
    const long nrows = P * S;
    hipLaunchKernelGGL(rows_fwd<S>, dim3((unsigned)ud_cdiv(nrows, RPW)), dim3(NT), rows_lds(S), s, x,
                       reinterpret_cast<float2*>(ws), Whp, nrows, 1.f);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL((cols_pass<S, true>), dim3((unsigned)ud_cdiv(Whp, CB), (unsigned)P), dim3(NT), cols_lds(S), s, ws,
                       Y, Whp, scale);
    UD_LAUNCH_CHECK();
    return 0;
}

template <int S>
int launch_adj(const float* dY, float* dx, float* ws, long P, float scale, hipStream_t s) {
    const int Whp = whp_of(S);
    const long nrows = P * S;
    hipLaunchKernelGGL((cols_pass<S, false>), dim3((unsigned)ud_cdiv(Whp, CB), (unsigned)P), dim3(NT), cols_lds(S), s, dY,
                       ws, Whp, 1.f);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_adj<S>, dim3((unsigned)ud_cdiv(nrows, RPW)), dim3(NT), rows_lds(S), s,
                       reinterpret_cast<const float2*>(ws), dx, Whp, nrows, scale);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

// floats of scratch (the row-transformed half spectrum) for P planes of S x S
long ud_rfft2_planes_ws_floats(long P, int S) {
    if (P < 1 || !size_ok(S)) return UD_EINVAL;
    return 2L * P * S * whp_of(S);
}

// Y[P][2S][Whp] = scale * rfft2(x[P][S][S]) (rows [0,S): Re, [S,2S): Im; columns > S/2 zero)
int ud_rfft2_planes(const float* x, float* Y, float* ws, long P, int S, float scale, ud_stream_t stream) {
    if (P < 1 || P > 0x7fffffffL / 512 || !size_ok(S) || !x || !Y || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (S) {
        case 128: return launch_fwd<128>(x, Y, ws, P, scale, s);
        case 256: return launch_fwd<256>(x, Y, ws, P, scale, s);
        default: return launch_fwd<320>(x, Y, ws, P, scale, s);
    }
}

// dx[P][S][S] = adjoint of ud_rfft2_planes applied to dY[P][2S][Whp] (same scale)
int ud_rfft2_planes_adjoint(const float* dY, float* dx, float* ws, long P, int S, float scale, ud_stream_t stream) {
    if (P < 1 || P > 0x7fffffffL / 512 || !size_ok(S) || !dY || !dx || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (S) {
        case 128: return launch_adj<128>(dY, dx, ws, P, scale, s);
        case 256: return launch_adj<256>(dY, dx, ws, P, scale, s);
        default: return launch_adj<320>(dY, dx, ws, P, scale, s);
    }
}

}  // extern "C"
