// One-shot SyncBatchNorm exchange over xGMI (reference: nn.SyncBatchNorm's all_gather of per-rank statistics and the
// all_reduce in its backward, engine/forgery_engine.py:142; torch/nn/modules/_functions.py SyncBatchNorm).
//
// The fused MBConv path sums 2C fp64 accumulators over the ranks ~200 times per step (tape.DataParallelCtx.reduce);
// each is at most 52 KB, so a library collective is pure latency (launch + protocol, ~15-25 us at 8 ranks).  Here every
// rank owns a MAILBOX in fine-grained device memory, mapped into all peers through HIP IPC.  One kernel per exchange,
// one workgroup:
//   1. write this rank's n doubles into slot (seq % slots), row `rank` of EVERY rank's mailbox (peer writes over xGMI),
//      fence at system scope, then release-store the sequence number into the matching flag of every mailbox;
//   2. wait until all `world` flags of the own mailbox show this sequence number;
//   3. sum the `world` rows in rank order (the same order on every rank: bit-identical results) into acc.
// The sequence number lives in device memory and is advanced by the kernel, so a hipGraph replay of the step keeps
// counting.  Slots: a rank can be at most one exchange ahead of the slowest (exchange k+1 cannot complete before
// everyone has entered it, i.e. left exchange k), so two slots suffice; four are used.  The wait is bounded: a peer
// that never arrives raises *err instead of hanging the GPU.
#include <string.h>

#include "ud_common.h"

namespace {

constexpr int NT = 256;

struct Mailbox {
    double* data;                  // [slots][world][max_doubles]
    unsigned long long* flags;     // [slots][world]
};
__host__ __device__ inline Mailbox mailbox_of(void* base, int world, int max_doubles, int slots) {
    Mailbox m;
    m.data = reinterpret_cast<double*>(base);
    m.flags = reinterpret_cast<unsigned long long*>(m.data + (long)slots * world * max_doubles);
    return m;
}
inline size_t mailbox_bytes(int world, int max_doubles, int slots) {
    return ((size_t)slots * world * max_doubles + (size_t)slots * world) * 8;
}

__global__ __launch_bounds__(NT) void xchg_allreduce_kernel(double* __restrict__ acc, int n, void* const* __restrict__ peers,
                                                            int rank, int world, int max_doubles, int slots,
                                                            unsigned long long* __restrict__ seq_counter,
                                                            int* __restrict__ err, long spin_limit) {
    const int tid = threadIdx.x;
    const unsigned long long seq = *seq_counter + 1;
    const int slot = (int)(seq % (unsigned long long)slots);
    // 1. publish
    for (int r = 0; r < world; ++r) {
        double* dst = mailbox_of(peers[r], world, max_doubles, slots).data + ((long)slot * world + rank) * max_doubles;
        for (int i = tid; i < n; i += NT) __hip_atomic_store(dst + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world) {
        unsigned long long* f = mailbox_of(peers[tid], world, max_doubles, slots).flags + (long)slot * world + rank;
        __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. wait for every rank's row of the own mailbox
    const Mailbox mine = mailbox_of(peers[rank], world, max_doubles, slots);
    if (tid < world) {
        const unsigned long long* f = mine.flags + (long)slot * world + tid;
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) {
                atomicExch(err, 1 + tid);          // rank `tid` never arrived: results below are garbage, the host checks
                break;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // 3. sum in rank order
    const double* rows = mine.data + (long)slot * world * max_doubles;
    for (int i = tid; i < n; i += NT) {
        double s = 0.0;
        for (int r = 0; r < world; ++r)
            s += __hip_atomic_load(rows + (long)r * max_doubles + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc[i] = s;
    }
    if (tid == 0) *seq_counter = seq;
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
    if (world < 1 || max_doubles < 1 || slots < 2 || !base || !handle) return UD_EINVAL;
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
// one at [rank]); seq_counter / err: device words owned by the caller (zero-initialised).  spin_limit: polls (each
// ~64 cycles apart) before a missing peer is reported through *err.
int ud_xchg_allreduce(double* acc, int n, void* const* peers, int rank, int world, int max_doubles, int slots,
                      unsigned long long* seq_counter, int* err, long spin_limit, ud_stream_t stream) {
    if (!acc || n < 1 || n > max_doubles || !peers || rank < 0 || rank >= world || world > NT || slots < 2 ||
        !seq_counter || !err || spin_limit < 1)
        return UD_EINVAL;
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, acc, n, peers, rank, world,
                       max_doubles, slots, seq_counter, err, spin_limit);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
