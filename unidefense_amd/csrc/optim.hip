// Multi-tensor AdamW (optionally amsgrad) for the whole parameter set in ONE launch.
//
// Serves the optimizer step of the reference's engines (engine/forgery_engine.py:149-156: timm's two weight-decay
// groups -> torch.optim.AdamW(amsgrad=True); engine/abstract_engine.py:281-283,374-378: GradScaler.step / update),
// twice per train step.  HBM-bound: 36 bytes per parameter (read p, g, m, v, vmax; write p, m, v, vmax) over
// 128.3 M parameters = 4.6 GB per call.
//
// Decomposition: a device-resident table describes every tensor (pointers, length, group), a chunk map assigns each
// workgroup one chunk of CHUNK elements of one tensor (so the 3264 x 3264 spectral weights spread over ~160
// workgroups and the 25 scalar gates cost one workgroup each), 16-byte accesses wherever the tensor's base pointers
// allow.  Folded into the pass: the GradScaler unscale (grad * 1/scale, read from the device), the skip on a
// non-finite gradient (found_inf, read from the device), per-group lr / weight decay, the bias corrections (from the
// device-side step counter: a skipped step does not advance it), decoupled weight decay.
// Arithmetic order = ATen's fused Adam functor (fused_adam_utils.cuh: adam_math) so that parameter updates agree with
// torch.optim.AdamW to rounding.
#include "ud_common.h"

namespace {

constexpr int NT = 256;
constexpr int CHUNK = 65536;              // elements per workgroup: 64 float4 per thread, 4 in flight

struct TensorEntry {                       // 8 x 8 bytes
    float* p;
    const float* g;
    float* m;
    float* v;
    float* vmax;                           // NULL without amsgrad
    long numel;
    long group;
    long pad;
};
static_assert(sizeof(TensorEntry) == 64, "table entry layout");

struct Hyper {
    float lr[8], wd[8];
    float beta1, beta2, eps;
    float omb1, omb2;                      // 1 - beta, rounded from the double difference (as ATen computes it)
    int amsgrad, maximize;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float& vm, float lr, float wd,
                                          const Hyper& h, float step_size, float bc2_sqrt, bool ams) {
    if (h.maximize) g = -g;
    p -= lr * wd * p;                                     // decoupled weight decay (AdamW)
    m = m + (g - m) * h.omb1;                    // lerp
    v = h.beta2 * v + h.omb2 * g * g;
    float denom;
    if (ams) {
        vm = fmaxf(vm, v);
        denom = sqrtf(vm) / bc2_sqrt + h.eps;
    } else {
        denom = sqrtf(v) / bc2_sqrt + h.eps;
    }
    p -= step_size * m / denom;
}

// chunk_map[b] = (tensor index, chunk index) of workgroup b
__global__ __launch_bounds__(NT) void adamw_multi(const TensorEntry* __restrict__ table, const int2* __restrict__ chunk_map,
                                                  Hyper h, const float* __restrict__ grad_scale,
                                                  const float* __restrict__ found_inf, const int* __restrict__ step_in,
                                                  int* __restrict__ step_out) {
    const bool skip = found_inf && found_inf[0] != 0.f;
    const int step = step_in[0] + (skip ? 0 : 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) step_out[0] = step;
    if (skip) return;
    const int2 cm = chunk_map[blockIdx.x];
    const TensorEntry t = table[cm.x];
    const float inv_scale = grad_scale ? 1.f / grad_scale[0] : 1.f;
    const float lr = h.lr[t.group], wd = h.wd[t.group];
    const double bc1 = 1.0 - pow((double)h.beta1, (double)step);
    const double bc2 = 1.0 - pow((double)h.beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const bool ams = h.amsgrad && t.vmax;
    const long begin = (long)cm.y * CHUNK;
    long end = begin + CHUNK;
    if (end > t.numel) end = t.numel;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v) |
                       reinterpret_cast<uintptr_t>(t.vmax)) & 15) == 0;
    long i = begin + (long)threadIdx.x * 4;
    if (vec) {
        f32x4* p4 = reinterpret_cast<f32x4*>(t.p);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(t.g);
        f32x4* m4 = reinterpret_cast<f32x4*>(t.m);
        f32x4* v4 = reinterpret_cast<f32x4*>(t.v);
        f32x4* x4 = reinterpret_cast<f32x4*>(t.vmax);
#pragma unroll 4
        for (; i + 3 < end; i += NT * 4) {
            const long q = i >> 2;
            f32x4 p = p4[q], g = g4[q] * inv_scale, m = m4[q], v = v4[q], vm = ams ? x4[q] : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e], xe = vm[e];
                adam_elem(pe, g[e], me, ve, xe, lr, wd, h, step_size, bc2_sqrt, ams);
                p[e] = pe; m[e] = me; v[e] = ve; vm[e] = xe;
            }
            p4[q] = p; m4[q] = m; v4[q] = v;
            if (ams) x4[q] = vm;
        }
        // ragged tail of the tensor (numel % 4): the thread whose float4 straddles the end
        if (i < end) {
            for (long j = i; j < end; ++j) {
                float p = t.p[j], m = t.m[j], v = t.v[j], vm = ams ? t.vmax[j] : 0.f;
                adam_elem(p, t.g[j] * inv_scale, m, v, vm, lr, wd, h, step_size, bc2_sqrt, ams);
                t.p[j] = p; t.m[j] = m; t.v[j] = v;
                if (ams) t.vmax[j] = vm;
            }
        }
    } else {
        for (long j = begin + threadIdx.x; j < end; j += NT) {
            float p = t.p[j], m = t.m[j], v = t.v[j], vm = ams ? t.vmax[j] : 0.f;
            adam_elem(p, t.g[j] * inv_scale, m, v, vm, lr, wd, h, step_size, bc2_sqrt, ams);
            t.p[j] = p; t.m[j] = m; t.v[j] = v;
            if (ams) t.vmax[j] = vm;
        }
    }
}

// ---- gradient accumulation over the whole parameter set --------------------------------------------------------------------
// The reference's train step calls backward() twice per zero_grad() (engine/abstract_engine.py:281, 374): the second backward
// ADDS its gradients to the first's — in torch one AccumulateGrad `grad += new` launch per parameter tensor, 504 launches of a few
// microseconds for the EfficientNet-b4 model (2.7 ms of a 64 ms train step).  Here: the (dst, src, numel) triples travel BY VALUE
// in the kernel arguments (no device table: the sources are fresh allocations of each backward, and a captured step must not
// stage a host table), MA_ITEMS per launch, one workgroup per MA_CHUNK elements of one tensor.
constexpr int MA_ITEMS = 120;
constexpr int MA_CHUNK = 16384;
struct MultiAddArgs {
    float* dst[MA_ITEMS];
    const float* src[MA_ITEMS];
    int numel[MA_ITEMS];
    int block0[MA_ITEMS + 1];              // first workgroup of every item; block0[n] = grid
    int n;
};
static_assert(sizeof(MultiAddArgs) <= 3968, "kernel arguments: 4 KiB in all");

__global__ __launch_bounds__(NT) void multi_add(const MultiAddArgs a) {
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {                                          // uniform: scalar loads from the argument segment
        const int mid = (lo + hi + 1) >> 1;
        if (a.block0[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    float* __restrict__ d = a.dst[lo];
    const float* __restrict__ s = a.src[lo];
    const int n = a.numel[lo];
    const int begin = ((int)blockIdx.x - a.block0[lo]) * MA_CHUNK, end = min(n, begin + MA_CHUNK);
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        const int e4 = begin + ((end - begin) & ~3);
        for (int j = begin + 4 * threadIdx.x; j < e4; j += 4 * NT) {
            f32x4 x = *reinterpret_cast<const f32x4*>(d + j);
            x += *reinterpret_cast<const f32x4*>(s + j);
            *reinterpret_cast<f32x4*>(d + j) = x;
        }
        for (int j = e4 + threadIdx.x; j < end; j += NT) d[j] += s[j];
    } else {
        for (int j = begin + threadIdx.x; j < end; j += NT) d[j] += s[j];
    }
}

}  // namespace

extern "C" {

int ud_adamw_chunk_elems(void) { return CHUNK; }

// table: n_tensors entries of 8 int64 on the device (p, g, m, v, vmax pointers, numel, group, 0);
// chunk_map: n_chunks int32 pairs (tensor, chunk) on the device; lr / wd: per group (<= 8 groups);
// grad_scale / found_inf: device scalars or NULL; step_in / step_out: device int32 (ping-pong: out = in + 1 unless
// found_inf).
int ud_adamw_multi(const void* table, const void* chunk_map, int n_chunks, const float* lr, const float* wd, int n_groups,
                   double beta1, double beta2, double eps, int amsgrad, int maximize, const float* grad_scale,
                   const float* found_inf, const int* step_in, int* step_out, ud_stream_t stream) {
    if (!table || !chunk_map || n_chunks < 1 || !lr || !wd || n_groups < 1 || n_groups > 8 || !step_in || !step_out)
        return UD_EINVAL;
    Hyper h{};
    for (int i = 0; i < n_groups; ++i) { h.lr[i] = lr[i]; h.wd[i] = wd[i]; }
    h.beta1 = (float)beta1; h.beta2 = (float)beta2; h.eps = (float)eps;
    h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2);
    h.amsgrad = amsgrad; h.maximize = maximize;
    hipLaunchKernelGGL(adamw_multi, dim3((unsigned)n_chunks), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const TensorEntry*>(table), reinterpret_cast<const int2*>(chunk_map), h,
                       grad_scale, found_inf, step_in, step_out);
    UD_LAUNCH_CHECK();
    return 0;
}

// dst[i][0..numel[i]) += src[i][0..numel[i]) for n fp32 tensors (host arrays of DEVICE pointers; every numel < 2^31)
int ud_multi_add(void* const* dst, const void* const* src, const long* numel, int n, ud_stream_t stream) {
    if (n < 0 || (n > 0 && (!dst || !src || !numel))) return UD_EINVAL;
    for (int i = 0; i < n; ++i)
        if (numel[i] < 0 || numel[i] > 0x7fffffffL || (numel[i] > 0 && (!dst[i] || !src[i]))) return UD_EINVAL;
    for (int i0 = 0; i0 < n;) {
        MultiAddArgs a{};
        int k = 0, blocks = 0;
        for (; i0 < n && k < MA_ITEMS; ++i0) {
            if (numel[i0] == 0) continue;
            a.dst[k] = (float*)dst[i0];
            a.src[k] = (const float*)src[i0];
            a.numel[k] = (int)numel[i0];
            a.block0[k] = blocks;
            blocks += (int)((numel[i0] + MA_CHUNK - 1) / MA_CHUNK);
            ++k;
        }
        if (k == 0) break;
        a.block0[k] = blocks;
        a.n = k;
        hipLaunchKernelGGL(multi_add, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
        UD_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
