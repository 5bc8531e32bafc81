// One-shot SyncBatchNorm exchange over xGMI (reference: nn.SyncBatchNorm's all_gather of per-rank statistics and the
// all_reduce in its backward, engine/forgery_engine.py:142; torch/nn/modules/_functions.py SyncBatchNorm).
//
// The fused MBConv path sums 2C fp64 accumulators over the ranks ~200 times per step (tape.DataParallelCtx.reduce);
// each is at most 52 KB, so a library collective is pure latency (launch + protocol, ~15-25 us at 8 ranks).  Here every
// rank owns a MAILBOX in fine-grained device memory, mapped into all peers through HIP IPC, and an exchange is ONE
// small kernel (one thread per double) speaking a flag-in-data protocol (no fences, no cache flushes, no separate flag round trip):
//   * a double travels as two 8-byte words {low half | seq, high half | seq}, each an atomic system-scope store into
//     slot (seq % slots), row `rank` of EVERY rank's mailbox (peer writes over xGMI);
//   * the reader polls the words of all `world` rows of its OWN mailbox until they carry this exchange's sequence tag,
//     and adds the rows in rank order (the same order on every rank: bit-identical results) into acc.
// The sequence number lives in device memory and is advanced by the kernel, so a hipGraph replay of the step keeps
// counting.  Slots: a rank can be at most one exchange ahead of the slowest (exchange k+1 cannot complete before
// everyone has entered it, i.e. left exchange k), so two slots suffice; four are used.  The wait is bounded by WALL-CLOCK
// time (wall_clock64, minutes by default — ranks legitimately drift apart by seconds: a data-loader stall, rank 0 writing a
// checkpoint, first-step GEMM tuning): a peer that never arrives raises *err instead of hanging the GPU, and the sums of
// that exchange are POISONED with NaN, never left as the partial sum of the rows that did arrive.  (A first version with plain stores + system-scope
// release / acquire fences cost 9.5 us per exchange on one device: the fences write back and invalidate the L2.)
#include <string.h>

#include "ud_common.h"

namespace {

constexpr int NT = 256;

typedef unsigned long long u64;

// words[slots][world][max_doubles][2]
__host__ __device__ inline u64* mailbox_row(void* base, int world, int max_doubles, int slot, int r) {
    return reinterpret_cast<u64*>(base) + ((long)slot * world + r) * max_doubles * 2;
}
inline size_t mailbox_bytes(int world, int max_doubles, int slots) { return (size_t)slots * world * max_doubles * 16; }

constexpr int MAXW = 16;          // ranks of one node

// grid: ceil(n / NT) workgroups, one element per thread — a single poll round trip whatever n is.  The workgroups are
// independent (the protocol needs no cross-workgroup step); counters[0] = exchanges completed, counters[1] = workgroups
// of the running exchange that have read counters[0]: the last of them advances the sequence number.
__global__ __launch_bounds__(NT) void xchg_allreduce_kernel(double* __restrict__ acc, int n, void* const* __restrict__ peers,
                                                            int rank, int world, int max_doubles, int slots,
                                                            u64* __restrict__ counters, int* __restrict__ err,
                                                            long long timeout_ticks, double* __restrict__ local_out) {
    const int i = blockIdx.x * NT + threadIdx.x;
    const u64 seq = __hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    asm volatile("" ::"v"(seq));                                // the loaded value has ARRIVED in this wave before it reaches the barrier
    __syncthreads();                                            // every wave of this workgroup has read it (the barrier's
    if (threadIdx.x == 0) {                                     //  own fence does not wait for loads: hence the line above)
        // (rounds 2-5 had a __threadfence() here: an agent-scope release + acquire, i.e. an L2 write-back and invalidate per
        //  exchange, for an ordering the returned atomic load above and this returning atomic already give)
        const u64 arrived = __hip_atomic_fetch_add(counters + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (arrived == gridDim.x) {                             // all workgroups have read counters[0]
            __hip_atomic_store(counters + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(counters, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (i >= n) return;
    const int slot = (int)(seq % (u64)slots);
    u64 tag = seq & 0xffffffffull;
    if (tag == 0) tag = 0x80000000ull;                         // 0 is what a fresh mailbox holds
    tag <<= 32;
    // 1. publish
    const double mine_v = acc[i];
    if (local_out) local_out[i] = mine_v;                      // this rank's own sums (the BatchNorm's dgamma / dbeta): no clone launch
    const u64 bits = (u64)__double_as_longlong(mine_v);
    const u64 w0 = (bits & 0xffffffffull) | tag, w1 = (bits >> 32) | tag;
    for (int r = 0; r < world; ++r) {
        u64* dst = mailbox_row(peers[r], world, max_doubles, slot, rank) + 2 * i;
        __hip_atomic_store(dst, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. collect: poll the words of every row of the own mailbox until they carry this exchange's tag; rank-ordered sum
    void* mine = peers[rank];
    u64 lo[MAXW], hi[MAXW];
    unsigned pending = (1u << world) - 1u;
    long spins = 0;
    const long long t0 = wall_clock64();
    bool timed_out = false;
    while (pending) {
#pragma unroll
        for (int r = 0; r < MAXW; ++r) {
            if (r < world && ((pending >> r) & 1u)) {
                const u64* src = mailbox_row(mine, world, max_doubles, slot, r) + 2 * i;
                lo[r] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                hi[r] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
#pragma unroll
        for (int r = 0; r < MAXW; ++r)
            if (r < world && (lo[r] & 0xffffffff00000000ull) == tag && (hi[r] & 0xffffffff00000000ull) == tag)
                pending &= ~(1u << r);
        if (pending && (++spins & 1023) == 0 && wall_clock64() - t0 > timeout_ticks) {
            atomicExch(err, __ffs(pending));                    // 1 + the lowest rank that never arrived
            timed_out = true;
            break;
        }
    }
    if (timed_out) {                                            // never a partial sum: the step's statistics are void
        acc[i] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < MAXW; ++r)
        if (r < world) s += __longlong_as_double((long long)((lo[r] & 0xffffffffull) | (hi[r] << 32)));
    acc[i] = s;
}

}  // namespace

extern "C" {

long ud_xchg_bytes(int world, int max_doubles, int slots) {
    if (world < 1 || max_doubles < 1 || slots < 2) return UD_EINVAL;
    return (long)mailbox_bytes(world, max_doubles, slots);
}

// Allocates this rank's mailbox (fine-grained device memory, zeroed) and exports it: handle receives the 64 bytes of a
// hipIpcMemHandle_t for the peers' ud_xchg_open.
int ud_xchg_create(int world, int max_doubles, int slots, void** base, char* handle) {
    if (world < 1 || world > MAXW || max_doubles < 1 || slots < 2 || !base || !handle) return UD_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    void* p = nullptr;
    const size_t bytes = mailbox_bytes(world, max_doubles, slots);
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return -(int)e;
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return -(int)e;
    }
    memcpy(handle, &h, 64);
    *base = p;
    return 0;
}

int ud_xchg_open(const char* handle, void** ptr) {
    if (!handle || !ptr) return UD_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : -(int)e;
}

int ud_xchg_close(void* ptr) {
    if (!ptr) return UD_EINVAL;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? 0 : -(int)e;
}

int ud_xchg_destroy(void* base) {
    if (!base) return UD_EINVAL;
    hipError_t e = hipFree(base);
    return e == hipSuccess ? 0 : -(int)e;
}

// acc[0..n) <- sum over the ranks, in place.  peers: DEVICE array of `world` mailbox pointers in rank order (the own
// one at [rank]); seq_counter: TWO device words (exchanges completed, workgroups arrived) and err: one, owned by the
// caller, zero-initialised.  timeout_ms: wall-clock wait for the slowest peer before it is reported missing through *err
// (and the affected sums set to NaN).
int ud_xchg_allreduce(double* acc, int n, void* const* peers, int rank, int world, int max_doubles, int slots,
                      unsigned long long* seq_counter, int* err, long timeout_ms, double* local_out, ud_stream_t stream) {
    if (!acc || n < 1 || n > max_doubles || !peers || rank < 0 || rank >= world || world > MAXW || slots < 2 ||
        !seq_counter || !err || timeout_ms < 1)
        return UD_EINVAL;
    static const long long khz = [] {                           // wall_clock64() ticks per millisecond
        int dev = 0, rate = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate <= 0)
            rate = 100000;                                      // 100 MHz: the constant counter of gfx9
        return (long long)rate;
    }();
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(ud_cdiv(n, NT)), dim3(NT), 0, (hipStream_t)stream, acc, n, peers,
                       rank, world, max_doubles, slots, seq_counter, err, khz * (long long)timeout_ms, local_out);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
