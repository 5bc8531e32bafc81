// Bare MFMA loops on random bf16 operands held in registers: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 (and the f16 forms),
// one wave per SIMD (256 threads, 1 workgroup per CU).  Prints TFLOP/s of each.  hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const s16x8* in, float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    s16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = in[(tid * 12 + i) & 0xffff]; b[i] = in[(tid * 12 + 6 + i) & 0xffff]; }
    float s = 0.f;
    if constexpr (MODE == 0 || MODE == 2) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                if constexpr (MODE == 0)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m % 6]), __builtin_bit_cast(bf16x8, b[(m / 4) % 6]), acc[m & 3], 0, 0, 0);
                else
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m % 6]), __builtin_bit_cast(f16x8, b[(m / 4) % 6]), acc[m & 3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[16] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                if constexpr (MODE == 1)
                    acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m % 6]), __builtin_bit_cast(bf16x8, b[(m / 8) % 6]), acc[m & 15], 0, 0, 0);
                else
                    acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m % 6]), __builtin_bit_cast(f16x8, b[(m / 8) % 6]), acc[m & 15], 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[tid] = s;
}

int main() {
    const int n = 65536;
    s16x8* in; float* out;
    hipMalloc(&in, n * sizeof(s16x8)); hipMalloc(&out, 256 * 256 * 4);
    short* h = (short*)malloc(n * 16);
    srand(1);
    for (int i = 0; i < n * 8; ++i) {          // random bf16 / fp16 in (-2, 2): exponent field near the bias
        const int mant = rand() & 0x7f, sign = rand() & 1, e = 126 + (rand() & 1);
        h[i] = (short)((sign << 15) | (e << 7) | mant);
    }
    hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[4] = {"32x32x16 bf16", "16x16x32 bf16", "32x32x16 f16 ", "16x16x32 f16 "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            auto run = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, in, out, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, in, out, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, in, out, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, in, out, iters);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); run(); run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (mode & 1) ? 48.0 * 16 * 16 * 32 * 2 : 24.0 * 32 * 32 * 16 * 2;
            const double fl = 2.0 * 256 * 4 * (double)iters * mf;
            printf("%s  %.1f ms  %.0f TFLOP/s\n", names[mode], ms / 2, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
