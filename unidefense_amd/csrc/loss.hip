// Asymmetrical weighted triplet loss on [N][D] feature vectors, forward value AND d loss / d feat in two launches.
//
// Reference: loss/triplet_loss.py:16-82 (AsymmetricalWeightedTripletLoss; pairwise distances :16-30).  Anchors are
// the first n_real rows (label 0 = real, the batch is ordered [real...; fake...], triplet_loss.py:46-53):
//   d_ij   = sqrt(clamp(|x_i|^2 + |x_j|^2 - 2 x_i.x_j, 1e-12))
//   wp_ij  = exp(+d_ij) [j real, j != i] / (sum + 1e-12),   wn_ij = exp(-d_ij) [j fake] / (sum + 1e-12)
//   margin_i = sum_j wn_ij d_ij - sum_j wp_ij d_ij,          loss = mean_i log(1 + exp(-margin_i))
// As torch ops this is ~50 tiny kernels forward and ~40 backward per feature (x3 features per pass): pure launch
// latency.  Here: one workgroup per anchor builds its row of distances, weights, the row loss and the
// coefficients c_ij = (dloss/dd_ij)/d_ij; a second launch turns c into dloss/dx and sums the row losses.
#include "ud_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = ud_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += sm[w];
    return t;
}

// grid = n_real blocks.  coef[i][j], row_loss[i]
__global__ __launch_bounds__(NT) void triplet_rows(const float* __restrict__ x, int N, int D, int R,
                                                   float* __restrict__ coef, float* __restrict__ row_loss) {
    extern __shared__ float sh[];          // dist[N]
    __shared__ float red[NT / 64];
    const int i = blockIdx.x;
    const float* xi = x + (long)i * D;
    float sq_i = 0.f;
    for (int k = threadIdx.x; k < D; k += NT) sq_i += xi[k] * xi[k];
    sq_i = block_sum(sq_i, red);
    // distances: one wave per column j at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < N; j += NT / 64) {
        const float* xj = x + (long)j * D;
        float dot = 0.f, sq_j = 0.f;
        for (int k = lane; k < D; k += 64) {
            float a = xj[k];
            dot += xi[k] * a;
            sq_j += a * a;
        }
        dot = ud_wave_sum(dot);
        sq_j = ud_wave_sum(sq_j);
        if (lane == 0) sh[j] = sq_i + sq_j - 2.f * dot;        // d^2 before the clamp
    }
    __syncthreads();
    // per-row statistics (N <= a few hundred: thread-strided loops)
    float s_ap = 0.f, s_an = 0.f, t_ap = 0.f, t_an = 0.f;
    for (int j = threadIdx.x; j < N; j += NT) {
        const float d = sqrtf(fmaxf(sh[j], 1e-12f));
        if (j < R) {
            if (j != i) { float e = expf(d); s_ap += e; t_ap += e * d; }
        } else {
            float e = expf(-d); s_an += e; t_an += e * d;
        }
    }
    s_ap = block_sum(s_ap, red);
    s_an = block_sum(s_an, red);
    t_ap = block_sum(t_ap, red);
    t_an = block_sum(t_an, red);
    const float ip = 1.f / (s_ap + 1e-12f), in = 1.f / (s_an + 1e-12f);
    const float fp = t_ap * ip, fn = t_an * in;
    const float margin = fn - fp;
    // soft margin with target +1: log(1 + exp(-margin));  d/dmargin = -sigmoid(-margin)
    const float dmargin = -1.f / (1.f + expf(margin)) / (float)R;
    if (threadIdx.x == 0) row_loss[i] = log1pf(expf(-margin));
    for (int j = threadIdx.x; j < N; j += NT) {
        const float d2 = sh[j];
        float c = 0.f;
        if (d2 > 1e-12f) {
            const float d = sqrtf(d2);
            float dd;      // d margin_i / d d_ij
            if (j < R) dd = (j != i) ? -(expf(d) * ip) * (1.f + d - fp) : 0.f;
            else dd = (expf(-d) * in) * (1.f - d + fn);
            c = dmargin * dd / d;
        }
        coef[(long)i * N + j] = c;
    }
}

// dx[n][k] = [n < R] sum_j c[n][j] (x[n][k] - x[j][k])  -  sum_{i<R} c[i][n] (x[i][k] - x[n][k]);  loss = mean row_loss
__global__ __launch_bounds__(NT) void triplet_grad(const float* __restrict__ x, int N, int D, int R,
                                                   const float* __restrict__ coef, const float* __restrict__ row_loss,
                                                   float* __restrict__ dx, float* __restrict__ loss) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * NT + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < R; ++i) s += row_loss[i];
        *loss = s / (float)R;
    }
    if (k >= D) return;
    const float xn = x[(long)n * D + k];
    float acc = 0.f;
    if (n < R)
        for (int j = 0; j < N; ++j) acc += coef[(long)n * N + j] * (xn - x[(long)j * D + k]);
    for (int i = 0; i < R; ++i) acc -= coef[(long)i * N + n] * (x[(long)i * D + k] - xn);
    dx[(long)n * D + k] = acc;
}

}  // namespace

extern "C" {

// ws: n_real * (N + 1) floats of scratch
int ud_aw_triplet(const float* feat, int N, int D, int n_real, float* loss, float* dfeat, float* ws,
                  ud_stream_t stream) {
    if (N < 2 || D < 1 || n_real < 1 || n_real >= N || N > 8192 || !ws) return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* coef = ws;
    float* row_loss = ws + (long)n_real * N;
    hipLaunchKernelGGL(triplet_rows, dim3(n_real), dim3(NT), N * sizeof(float), s, feat, N, D, n_real, coef, row_loss);
    UD_LAUNCH_CHECK();
    hipLaunchKernelGGL(triplet_grad, dim3(ud_cdiv(D, NT), N), dim3(NT), 0, s, feat, N, D, n_real, coef, row_loss, dfeat,
                       loss);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
