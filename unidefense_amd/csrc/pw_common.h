// Shared pieces of the one-pass thin-conv backward kernels (pwbwd.hip, pjbwd.hip): gemm_x3's exact three-way bf16 split, the
// six-product MFMA step on v_mfma_f32_16x16x32_bf16, and the fragment reads from a row-major bf16 LDS image (rows: ds_read_b128,
// columns: ds_read_b64_tr_b16).
#pragma once
#include "ud_common.h"

namespace pw {

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    bf16x2v v = {(__bf16)x, (__bf16)y};
    return __builtin_bit_cast(uint32_t, v);
}
// exact three-way split of two floats (gemm_x3.hip: split2); p[i] packs piece i of (x, y)
__device__ __forceinline__ void split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(sx, sy);
}
// a quad of floats -> three 8-byte pieces at dst, dst + plane, dst + 2 plane
__device__ __forceinline__ void store_split4(char* dst, int plane, const f32x4& o) {
    uint32_t p0[2], p1[2], p2[2];
    split2(o[0], o[1], p0[0], p1[0], p2[0]);
    split2(o[2], o[3], p0[1], p1[1], p2[1]);
    *reinterpret_cast<u32x2*>(dst) = u32x2{p0[0], p0[1]};
    *reinterpret_cast<u32x2*>(dst + plane) = u32x2{p1[0], p1[1]};
    *reinterpret_cast<u32x2*>(dst + 2 * plane) = u32x2{p2[0], p2[1]};
}
// 8 floats (k = 0..7 of one lane's operand) -> the three pieces as MFMA fragments
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&f)[3]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        uint32_t p0, p1, p2;
        split2(v[2 * h], v[2 * h + 1], p0, p1, p2);
        f[0][h] = p0; f[1][h] = p1; f[2][h] = p2;
    }
}

struct Frag3 { bf16x8 p[3]; };

// row read: the lane's 16 bytes at s, s + plane, s + 2 plane
__device__ __forceinline__ Frag3 read_rows(const lds_char* s, int plane) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        f.p[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(s + p * plane));
    return f;
}
// transposed read: s = the lane's address in the block of image rows 8 g .. 8 g + 3 (row q = (lane & 15) >> 2, 4 columns from
// 4 (lane & 3)); the block 4 rows below completes k = 8 g .. 8 g + 7
__device__ __forceinline__ Frag3 read_cols(const lds_char* s, int plane, int row_stride) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + p * plane));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + p * plane + 4 * row_stride));
        f.p[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    return f;
}
__device__ __forceinline__ Frag3 frag_of(const u32x4 (&w)[3]) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = __builtin_bit_cast(bf16x8, w[p]);
    return f;
}
// the six piece products of one k-step, smallest first (gemm_x3.hip's order)
__device__ __forceinline__ f32x4 mma6(const Frag3& a, const Frag3& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    return c;
}


// dw[numel] = the nparts per-workgroup partials summed in a fixed order (pwbwd.hip); optionally dgamma / dbeta [CE] from the
// fp64 sums s2l / s1l (ud_normbwd_apply's side outputs)
int fold_launch(const float* part, int nparts, int numel, float* dw, const double* s1l, const double* s2l, int CE, float* dgamma,
                float* dbeta, hipStream_t s);

}  // namespace pw
