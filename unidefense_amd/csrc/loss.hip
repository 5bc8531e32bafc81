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

// one block per anchor i.  coef[i][j], row_loss[i]
__device__ __forceinline__ void triplet_rows_body(const float* __restrict__ x, int N, int D, int R, float* __restrict__ coef,
                                                  float* __restrict__ row_loss, float* sh, float* red) {
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

__global__ __launch_bounds__(NT) void triplet_rows(const float* __restrict__ x, int N, int D, int R,
                                                   float* __restrict__ coef, float* __restrict__ row_loss) {
    extern __shared__ float sh[];          // dist[N]
    __shared__ float red[NT / 64];
    triplet_rows_body(x, N, D, R, coef, row_loss, sh, red);
}

// dx[n][k] = w * ([n < R] sum_j c[n][j] (x[n][k] - x[j][k])  -  sum_{i<R} c[i][n] (x[i][k] - x[n][k]))
__device__ __forceinline__ void triplet_grad_body(const float* __restrict__ x, int N, int D, int R, const float* __restrict__ coef,
                                                  float* __restrict__ dx, int n, int k, float w) {
    if (k >= D) return;
    const float xn = x[(long)n * D + k];
    float acc = 0.f;
    if (n < R)
        for (int j = 0; j < N; ++j) acc += coef[(long)n * N + j] * (xn - x[(long)j * D + k]);
    for (int i = 0; i < R; ++i) acc -= coef[(long)i * N + n] * (x[(long)i * D + k] - xn);
    dx[(long)n * D + k] = w * acc;
}

// ... and loss = mean row_loss
__global__ __launch_bounds__(NT) void triplet_grad(const float* __restrict__ x, int N, int D, int R,
                                                   const float* __restrict__ coef, const float* __restrict__ row_loss,
                                                   float* __restrict__ dx, float* __restrict__ loss) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < R; ++i) s += row_loss[i];
        *loss = s / (float)R;
    }
    triplet_grad_body(x, N, D, R, coef, dx, blockIdx.y, blockIdx.x * NT + threadIdx.x, 1.f);
}

// ---------------------------------------------------------------------------------------------------------
// The scalar tail of a pass's loss (engine/abstract_engine.py:241-281 of the reference: cross entropy on cls_out, the two mask
// means, the triplet terms of up to three features, the real / fake means of the per-sample reconstruction and frequency
// terms, their weighted sum) — as torch ops ~45 launches of 4-5 us between the forward and the backward of every pass.
// Launch A: the triplet rows of all features (grid y = feature).  Launch B: the triplet gradients of all features + ONE block
// that evaluates every scalar term, the weighted total, and the gradients of cls_out / masks / per-sample terms.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void tail_rows(ud_loss_tail t) {
    extern __shared__ float sh[];
    __shared__ float red[NT / 64];
    const int f = blockIdx.y;
    float* ws = t.ws + (long)f * t.R * (t.N + 1);
    triplet_rows_body(t.feat[f], t.N, t.D[f], t.R, ws, ws + (long)t.R * t.N, sh, red);
}

__device__ __forceinline__ float block_sum_all(float v, float* sm) { return block_sum(v, sm); }

__global__ __launch_bounds__(NT) void tail_finish(ud_loss_tail t, int b0, int b1, int b2) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const int N = t.N, R = t.R;
    if (b < b2) {                                   // triplet gradient blocks: feature f, sample n, column chunk
        const int f = b < b0 ? 0 : (b < b1 ? 1 : 2);
        const int lb = b - (f == 0 ? 0 : (f == 1 ? b0 : b1));
        const int chunks = (t.D[f] + NT - 1) / NT;
        const float* ws = t.ws + (long)f * R * (N + 1);
        triplet_grad_body(t.feat[f], N, t.D[f], R, ws, t.dfeat[f], lb / chunks, (lb % chunks) * NT + threadIdx.x, t.w_trip);
        return;
    }
    // ---- the scalar block
    const int tid = threadIdx.x;
    // cross entropy (mean over the batch) and its gradient (softmax - onehot) / N
    float ce = 0.f;
    for (int n = tid; n < N; n += NT) {
        const float* l = t.cls + (long)n * t.C;
        float m = l[0];
        for (int c = 1; c < t.C; ++c) m = fmaxf(m, l[c]);
        float se = 0.f;
        for (int c = 0; c < t.C; ++c) se += expf(l[c] - m);
        const int y = (int)t.tgt[n];
        ce += (logf(se) + m) - l[y];
        const float inv = t.w_cls / (se * (float)N);
        for (int c = 0; c < t.C; ++c) t.dcls[(long)n * t.C + c] = expf(l[c] - m) * inv - (c == y ? t.w_cls / (float)N : 0.f);
    }
    ce = block_sum_all(ce, red) / (float)N;
    // mask means (gradient: a constant)
    float fm = 0.f, sm = 0.f;
    if (t.fm) {
        for (int i = tid; i < t.nfm; i += NT) { fm += t.fm[i]; t.dfm[i] = t.w_fm / (float)t.nfm; }
        fm = block_sum_all(fm, red) / (float)t.nfm;
    }
    if (t.sm) {
        for (int i = tid; i < t.nsm; i += NT) { sm += t.sm[i]; t.dsm[i] = t.w_sm / (float)t.nsm; }
        sm = block_sum_all(sm, red) / (float)t.nsm;
    }
    // per-sample reconstruction / frequency terms: mean over the real samples enters the loss, the fake mean is reported
    float rr = 0.f, fr = 0.f, rq = 0.f, fq = 0.f;
    const int Fk = t.F;
    for (int n = tid; n < R + Fk; n += NT) {
        if (t.spatial) {
            const float v = t.spatial[n];
            if (n < R) rr += v; else fr += v;
            t.dspatial[n] = n < R ? t.w_rec / (float)R : 0.f;
        }
        if (t.freq) {
            const float v = t.freq[n];
            if (n < R) rq += v; else fq += v;
            t.dfreq[n] = n < R ? t.w_freq / (float)R : 0.f;
        }
    }
    for (int n = R + Fk + tid; n < N; n += NT) {          // (samples beyond real + fake, if any, carry no gradient)
        if (t.spatial) t.dspatial[n] = 0.f;
        if (t.freq) t.dfreq[n] = 0.f;
    }
    rr = block_sum_all(rr, red) / (float)R;
    fr = block_sum_all(fr, red) / (float)(Fk > 0 ? Fk : 1);
    rq = block_sum_all(rq, red) / (float)R;
    fq = block_sum_all(fq, red) / (float)(Fk > 0 ? Fk : 1);
    // triplet terms: mean row loss of every feature, summed
    float tr = 0.f;
    for (int f = 0; f < t.nfeat; ++f) {
        const float* rl = t.ws + (long)f * R * (N + 1) + (long)R * N;
        float s = 0.f;
        for (int i = tid; i < R; i += NT) s += rl[i];
        tr += block_sum_all(s, red) / (float)R;
    }
    if (tid == 0) {
        t.vals[0] = t.w_cls * ce + t.w_fm * fm + t.w_sm * sm + t.w_trip * tr + t.w_rec * rr + t.w_freq * rq;
        t.vals[1] = ce; t.vals[2] = tr; t.vals[3] = rr; t.vals[4] = fr; t.vals[5] = rq; t.vals[6] = fq;
        t.vals[7] = fm; t.vals[8] = sm;
    }
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


int ud_loss_tail_ws_floats(int N, int n_real, int nfeat) { return nfeat * n_real * (N + 1); }

int ud_loss_tail_run(const ud_loss_tail* t, ud_stream_t stream) {
    if (!t || t->N < 2 || t->R < 1 || t->R >= t->N || t->F < 0 || t->R + t->F > t->N || t->N > 8192 || t->C < 2 || t->C > 64 ||
        t->nfeat < 0 || t->nfeat > 3 || !t->cls || !t->tgt || !t->dcls || !t->vals || (t->nfeat && !t->ws))
        return UD_EINVAL;
    for (int f = 0; f < t->nfeat; ++f)
        if (!t->feat[f] || !t->dfeat[f] || t->D[f] < 1) return UD_EINVAL;
    if ((t->fm && (!t->dfm || t->nfm < 1)) || (t->sm && (!t->dsm || t->nsm < 1)) || (t->spatial && !t->dspatial) ||
        (t->freq && !t->dfreq))
        return UD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (t->nfeat) {
        hipLaunchKernelGGL(tail_rows, dim3(t->R, t->nfeat), dim3(NT), t->N * sizeof(float), s, *t);
        UD_LAUNCH_CHECK();
    }
    int b[3] = {0, 0, 0}, acc = 0;
    for (int f = 0; f < 3; ++f) {
        if (f < t->nfeat) acc += ud_cdiv(t->D[f], NT) * t->N;
        b[f] = acc;
    }
    hipLaunchKernelGGL(tail_finish, dim3(acc + 1), dim3(NT), 0, s, *t, b[0], b[1], b[2]);
    UD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
