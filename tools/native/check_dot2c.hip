// Is  v_dot2c_f32_bf16(acc = x, piece, {-1, 0})  exactly  x - float(bf16(x)) ?  (the residual step of gemm_x3.hip's split)
// build: hipcc -O2 --offload-arch=gfx950 tools/native/check_dot2c.hip -o tools/native/check_dot2c ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x2 opaque(uint32_t bits) {
    asm volatile("" : "+s"(bits));
    return __builtin_bit_cast(bf16x2, bits);
}
__global__ void k(const float* x, float* r_dot, float* r_ref, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    bf16x2 p = {(__bf16)a, (__bf16)b};
    const bf16x2 k10 = opaque(0x0000bf80u), k01 = opaque(0xbf800000u);
    r_dot[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(p, k10, a, false);
    r_dot[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(p, k01, b, false);
    uint32_t w = __builtin_bit_cast(uint32_t, p);
    r_ref[2 * i] = a - __uint_as_float(w << 16);
    r_ref[2 * i + 1] = b - __uint_as_float(w & 0xffff0000u);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t bits = (uint32_t)(s >> 16);
        // every exponent incl. subnormals; no inf / nan
        if (((bits >> 23) & 0xff) == 0xff) bits &= ~(1u << 30);
        float f; memcpy(&f, &bits, 4);
        h[i] = f;
    }
    float *x, *a, *b;
    hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, x, a, b, n);
    std::vector<float> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] == 0.f && hb[i] == 0.f)) {
            ++bad;
            uint32_t xb; memcpy(&xb, &h[i], 4);
            const int e = (xb >> 23) & 0xff;
            if (e > 24) ++bad_normal;       // residual of a value this large is itself a normal number
            if (bad <= 8) printf("x %.9g (exp %d): dot2c %.9g  ref %.9g\n", h[i], e, ha[i], hb[i]);
        }
    }
    printf("mismatches %ld of %d (with a normal residual: %ld)\n", bad, n, bad_normal);
    return bad_normal ? 1 : 0;
}
