// Shared helpers for the gfx950 kernels of libunidefense_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unidefense_hip.h"

#define UD_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return -(int)e__;            \
    } while (0)

static inline int ud_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Storage type of the trunk's activation tensors (BASELINE configs[4]): float, or _Float16 with every kernel computing
// in fp32 registers — a channel quad is one 16-byte (float) or 8-byte (half) access, converted on load / rounded to
// nearest even on store.  Kernels are templates over T; entry points take `f16` (0 / 1) and dispatch.
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p, long i) { return reinterpret_cast<const f32x4*>(p)[i]; }
    static __device__ __forceinline__ void st(float* p, long i, const f32x4& v) { reinterpret_cast<f32x4*>(p)[i] = v; }
};
template <> struct Quad<_Float16> {
    static __device__ __forceinline__ f32x4 ld(const _Float16* p, long i) {
        return __builtin_convertvector(reinterpret_cast<const f16x4*>(p)[i], f32x4);
    }
    static __device__ __forceinline__ void st(_Float16* p, long i, const f32x4& v) {
        reinterpret_cast<f16x4*>(p)[i] = __builtin_convertvector(v, f16x4);
    }
};
// read-only / writable views indexed in channel quads
template <typename T> struct In4 {
    const T* p;
    __device__ __forceinline__ f32x4 operator[](long i) const { return Quad<T>::ld(p, i); }
    __device__ __forceinline__ In4 operator+(long i) const { return In4{p + 4 * i}; }
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};
template <typename T> struct Out4 {
    T* p;
    __device__ __forceinline__ void st(long i, const f32x4& v) const { Quad<T>::st(p, i, v); }
    __device__ __forceinline__ Out4 operator+(long i) const { return Out4{p + 4 * i}; }
};
// value actually stored for v (statistics are taken of what consumers will read back)
template <typename T> __device__ __forceinline__ float ud_rounded(float v) { return (float)(T)v; }

#define UD_STORAGE_DISPATCH(f16, ...)            \
    do {                                         \
        if (f16) {                               \
            using T = _Float16;                  \
            __VA_ARGS__;                         \
        } else {                                 \
            using T = float;                     \
            __VA_ARGS__;                         \
        }                                        \
    } while (0)

// accurate expf (not the __expf fast intrinsic): these kernels are bandwidth bound, the ALU work is free
__device__ __forceinline__ float ud_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ud_swish(float x) { return x * ud_sigmoid(x); }
// d/dx [x*sigmoid(x)] = s*(1 + x*(1-s))      (model/efficientnet/utils.py:73-77)
__device__ __forceinline__ float ud_swish_grad(float x) {
    float s = ud_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}

// Fast forms for the fused kernels, where the activation is RE-evaluated by every consumer of a deferred BatchNorm and
// the ALU work is no longer free: v_exp_f32 + v_rcp_f32 (about 1e-6 relative) instead of expf + an IEEE division.
__device__ __forceinline__ float ud_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
__device__ __forceinline__ float ud_act_fast(float z, int act) {
    return act == 1 ? z * ud_sigmoid_fast(z) : (act == 2 ? fmaxf(z, 0.f) : z);
}
__device__ __forceinline__ float ud_act_grad_fast(float z, int act) {
    if (act == 1) {
        const float s = ud_sigmoid_fast(z);
        return s * (1.0f + z * (1.0f - s));
    }
    return act == 2 ? (z > 0.f ? 1.f : 0.f) : 1.f;
}

// act: 0 identity, 1 swish, 2 ReLU
__device__ __forceinline__ float ud_act(float z, int act) {
    return act == 1 ? ud_swish(z) : (act == 2 ? fmaxf(z, 0.f) : z);
}
__device__ __forceinline__ float ud_act_grad(float z, int act) {
    return act == 1 ? ud_swish_grad(z) : (act == 2 ? (z > 0.f ? 1.f : 0.f) : 1.f);
}

// wave64 all-reduce (sum) via DPP-free shuffles
__device__ __forceinline__ float ud_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double ud_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ud_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Tensor |x|max as a SIDE OUTPUT of the kernel that produces the tensor (the scale of ud_split_planes_h2t): every thread of the
// workgroup calls this once with the largest |value| it stored (0 if none; `slots` uniform, NULL = not wanted): wave maximum,
// workgroup maximum through LDS, ONE atomic maximum per workgroup onto one of the 256 slots (bit patterns of non-negative
// floats order like unsigned integers).  The caller zeroes the slots.
__device__ __forceinline__ void ud_absmax_commit(float m, uint32_t* __restrict__ slots) {
    if (!slots) return;
    __shared__ unsigned ud_sm_absmax;
    if (threadIdx.x == 0 && threadIdx.y == 0) ud_sm_absmax = 0u;
    __syncthreads();
    unsigned bits = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));          // (inactive lanes read as 0 or own)
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(&ud_sm_absmax, bits);
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && ud_sm_absmax)
        atomicMax(slots + ((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) & 255u), ud_sm_absmax);
}

// ---- fp16 x 2 planes of ud_gemm_p3 prec 2 (csrc/gemm_p3.hip), for producers that write their result as planes themselves ----
// scale of a tensor whose |x|max (or an UPPER BOUND of it) has the bit pattern `bits`: the power of two taking it into
// [2^14, 2^15) (exponent field 268 - e, kept inside the normal range; all zeros: 1); inv = 1 / scale.
__device__ __forceinline__ void ud_h2_scale(uint32_t bits, float& s, float& inv) {
    const int e = (int)(bits >> 23) & 0xff;
    int fs = bits ? 268 - e : 127;
    fs = fs < 1 ? 1 : fs > 254 ? 254 : fs;
    s = __uint_as_float((uint32_t)fs << 23);
    inv = __uint_as_float((uint32_t)(254 - fs) << 23);
}
// the two pieces of the SCALED value xs: h0 = fp16(xs), h1 = fp16(2^11 (xs - h0))  (exact residual; gemm_p3.hip: split2h)
__device__ __forceinline__ void ud_split_h2(float xs, uint16_t& h0, uint16_t& h1) {
    const _Float16 a = (_Float16)xs;
    const _Float16 b = (_Float16)((xs - (float)a) * 2048.f);
    h0 = __builtin_bit_cast(uint16_t, a);
    h1 = __builtin_bit_cast(uint16_t, b);
}
