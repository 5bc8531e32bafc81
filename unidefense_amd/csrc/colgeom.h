// Column-owning decomposition shared by the per-channel reduction kernels (norm.hip) and the fused deferred-
// normalisation kernels (fused.hip) on pixel-major [G groups][R rows][C] fp32.
//
// grid = (P row-chunks, column groups, G); a 256-thread workgroup owns CW float4 columns (CW * 16 contiguous bytes of
// every row) and one row-chunk; thread (ri, c4) walks rows ri, ri + rpi, ... of the chunk with its channel quad c4
// fixed — so per-channel coefficients are computed once per thread and per-channel sums fold through LDS.
#pragma once
#include <stdlib.h>

#include "ud_common.h"

namespace {

constexpr int UD_COL_NT = 256;

struct RedGeom {
    int G, R, C4, P;      // groups, rows per group, float4 channels, row-chunks per group
    int CW;               // float4 columns per workgroup (<= 16 by default)
    int rpi;              // rows per block iteration = NT / CW
    int rows_per_chunk;
};

__device__ __forceinline__ bool thread_coords(const RedGeom& q, int& ri, int& c4) {
    int t = threadIdx.x;
    ri = t / q.CW;
    c4 = blockIdx.y * q.CW + t % q.CW;
    return ri < q.rpi && c4 < q.C4;
}

// fold the row-lanes of a block: v[NQ] per thread -> thread (ri == 0) holds the block total
template <int NQ>
__device__ __forceinline__ void block_fold(const RedGeom& q, int ri, bool active, double (&v)[8]) {
    if (q.rpi == 1) return;
    __shared__ double sm[UD_COL_NT * NQ];
    if (active) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) sm[threadIdx.x * NQ + i] = v[i];
    }
    __syncthreads();
    if (active && ri == 0) {
        for (int r = 1; r < q.rpi; ++r) {
            int t = r * q.CW + (int)threadIdx.x;          // ri == 0: threadIdx.x is the column lane
#pragma unroll
            for (int i = 0; i < NQ; ++i) v[i] += sm[t * NQ + i];
        }
    }
}

// Decomposition: a workgroup owns CW float4 columns (<= 64 channels) and one row-chunk; ~target workgroups,
// >= 8 rows per thread where the tensor allows, at most max_p chunks per group.
inline RedGeom make_geom_ex(int G, int R, int C, long target_blocks, long max_p, int min_rows = 8, int cw_limit = 16) {
    RedGeom q;
    q.G = G; q.R = R; q.C4 = C / 4;
    // float4 columns per workgroup (cw_limit): 16 (256-byte row segments, 16 rows per iteration) measured best for the REDUCTIONS
    // (8: +0.4 %, 32: +0.4 %, 64: +0.5 %, 128: +1.1 % step time).  The columns are spread
    // evenly over the groups (C4 = 36 -> 3 groups of 12, not 16 + 16 + 4 with a quarter-filled last workgroup).
    const int cw_max = cw_limit;
    const int ngroups = (q.C4 + cw_max - 1) / cw_max;
    q.CW = (q.C4 + ngroups - 1) / ngroups;
    q.rpi = UD_COL_NT / q.CW;
    const int cgroups = (q.C4 + q.CW - 1) / q.CW;
    long want = target_blocks / ((long)G * cgroups);
    if (want < 1) want = 1;
    long maxp = (R + (long)q.rpi * min_rows - 1) / ((long)q.rpi * min_rows);
    if (maxp < 1) maxp = 1;
    long P = want < maxp ? want : maxp;
    if (P > max_p) P = max_p;
    q.P = (int)P;
    q.rows_per_chunk = (R + q.P - 1) / q.P;
    return q;
}

// two-launch reductions (norm.hip): the finalize folds P partials per channel, keep its chain short
inline RedGeom make_geom(int G, int R, int C) { return make_geom_ex(G, R, C, 2048, 512); }

inline dim3 red_grid(const RedGeom& q) { return dim3((unsigned)q.P, (unsigned)((q.C4 + q.CW - 1) / q.CW), (unsigned)q.G); }

}  // namespace
