/*
 * unidefense_hip.h — C ABI of libunidefense_hip.so (gfx950 / MI355X).
 *
 * The reference (VISION-SJTU/UniDefense) has no native code and no FFI: its hot path reaches the
 * GPU through torch.nn / torch.fft (ATen -> vendor libraries).  Each entry point below replaces
 * one of those implicit kernels; the comment on each names the reference call site it serves
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every tensor is fp32, contiguous, "pixel-major" (NHWC): [N][H][W][C], i.e. a row-major
 *     matrix [M = N*H*W][C]; image-domain tensors with C = 3 are planes [N*C][H][W] (NCHW).
 *     "[G][R][C]" below means G groups of R rows (G = 1: batch norm; G = N, R = H*W: per sample).
 *   - plain pointers + sizes; the caller owns every buffer, including the scratch buffers
 *     ("part*") whose sizes the *_chunks / *_parts / *_blocks helpers return.  No allocation, no
 *     host synchronisation inside: every function only enqueues kernels on `stream` (a hipStream_t
 *     passed as void*) and is safe to capture into a hipGraph.
 *   - return 0 on success, a negative hipError_t on launch failure, UD_EINVAL on invalid arguments
 *     (the helpers return a non-negative count).
 *   - act / act_in: 0 = identity, 1 = swish x*sigmoid(x) (model/efficientnet/utils.py:66-82).
 */
#ifndef UNIDEFENSE_HIP_H
#define UNIDEFENSE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UD_EINVAL (-1000)

typedef void* ud_stream_t;

/* ---- conv geometry used by the gather modes of ud_gemm ------------------------------------
 * rows r = (n, oh, ow) over [N][Hout][Wout]; taps (kh, kw); source pixel of (r, tap):
 *   transposed == 0 :  ih = oh*stride - pad_t + kh                       (F.conv2d)
 *   transposed == 1 :  t = oh + pad_t - kh; valid iff t % stride == 0;  ih = t / stride
 *                                                                          (F.conv_transpose2d)
 * out-of-range sources read as 0. */
typedef struct {
    int N, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_t, pad_l, transposed;
} ud_conv_geom;

/* ---- ud_gemm: C[M][N] (+)= A . B on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) ---------
 * a_mode 0: A[m][k] at A + m*lda + k          (activations [pixels][Cin])
 *        1: A[k][m] at A + k*lda + m          (dY as the A operand of a weight gradient)
 *        2: conv gather: m = (n,oh,ow), k = (kh*KW+kw)*Cin + ci -> In[n][ih][iw][ci]  (geom g)
 * b_mode 0: B[n][k] at B + n*ldb + k          (weights [Cout][K])
 *        1: B[k][n] at B + k*ldb + n          (weights for the data gradient, X for the weight gradient)
 *        2: conv gather with k = (n,oh,ow) and n = (kh*KW+kw)*Cin + ci        (weight gradient of a conv)
 * supported (a_mode, b_mode): (0,0) (0,1) (1,1) (2,0) (1,2).
 * out_mode 0: store, 1: C += result, 2: atomicAdd (split_k > 1; C pre-zeroed or holding a term to add to),
 *          3: split s STORES its partial product into its own slice C + s * slice_stride (split_k >= 1; fp32, batch 1):
 *             the deterministic form of split-K — ud_sum_slices then adds the slices in ascending order
 * Serves: F.conv2d 1x1 in model/efficientnet/model.py:108,125 and exp.py:57 (freq_conv),
 *         nn.Conv2d 3x3 / nn.ConvTranspose2d in model/unidefense.py:59-102, model/modules.py:82,111,
 *         the stem conv model/efficientnet/model.py:185, torch.fft.rfft2 at 256x256 (as DFT-matrix
 *         GEMMs, model/unidefense.py:246-249), and their autograd backward (convolution_backward). */
typedef struct {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    int a_mode, b_mode, out_mode, split_k;
    int batch; long strideA, strideB, strideC;   /* batch >= 1; element strides between batches */
    ud_conv_geom g;
    /* Optional epilogue statistics of the RESULT (training-mode BatchNorm behind a 1x1 conv, model.py:108-109,125-126):
     * stat_sum[n] += sum_m C[m][n], stat_sumsq[n] += sum_m C[m][n]^2, fp64 atomic adds — the BatchNorm's statistics pass
     * disappears.  Honoured only where ud_gemm_stats_slots() returns > 0 (split-bf16 kernel, out_mode 0, split_k 1,
     * batch 1); with more than 64 row tiles the adds go to slot (row tile % 64) of [64][N] arrays instead (the caller
     * passes zeroed slot arrays as stat_sum / stat_sumsq and folds them, e.g. with ud_stat_slots_fold). */
    double* stat_sum; double* stat_sumsq;
    /* Half storage (BASELINE configs[4]): bit 0 / 1 / 2 set = A / B / C point to _Float16 instead of float (element
     * strides unchanged).  Such descriptors always run on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulation):
     * plain modes only — (0,0) and (0,1) with A half (activations x fp32 weights), (1,1) with A and B half (the weight
     * gradient), any of them with fp32 operands; batch 1; a half C takes out_mode 0 / 1 (no atomics: split_k 1). */
    int half_mask;
    /* Tile of the BF16-pipe kernel: 0 = chosen by the library's cost model; 1..4 = 128x128, 128x64, 64x128, 64x64 —
     * for callers that tune per shape by measurement (unidefense_amd/kernels.py does, on the thin expand / project
     * GEMMs whose few tiles leave the k-loop latency exposed).
     * bit 8 (0x100): each XCD takes a contiguous range of the tile order (measured neutral; off in the shipped plans). */
    int tile_cfg;
    long slice_stride;       /* out_mode 3: elements between the splits' slices (>= M * ldc) */
} ud_gemm_desc;
int ud_gemm(const ud_gemm_desc* d, ud_stream_t stream);
/* out[i] (+)= sum_s ws[s * slice_stride + i], s = 0 .. slices-1 in this order (total, slice_stride multiples of 4) */
int ud_sum_slices(const float* ws, float* out, int slices, long total, long slice_stride, int accumulate,
                  ud_stream_t stream);
/* 0: ud_gemm would ignore stat_sum / stat_sumsq for this descriptor (the caller runs ud_colstats on the result);
 * 1: the epilogue adds straight into [N] accumulators; 64: it adds into 64 slots of [64][N] (see ud_gemm_desc). */
int ud_gemm_stats_slots(const ud_gemm_desc* d);
/* acc_sum[n] += sum_s slot_sum[s][n], likewise sumsq   (slots: the 64 above) */
int ud_stat_slots_fold(const double* slot_sum, const double* slot_sumsq, int slots, int N, double* acc_sum,
                       double* acc_sumsq, ud_stream_t stream);
/* Arithmetic path of ud_gemm (process-wide; initial value from env UD_GEMM_PATH):
 *   0 auto (default): plain GEMMs (a_mode, b_mode in {0,1}, 16-byte aligned, M,N,K >= 16) run on the BF16
 *     matrix pipe with every fp32 operand split exactly into three bf16 pieces and six piece products
 *     accumulated in fp32 (csrc/gemm_x3.hip: fp32-GEMM accuracy, error terms < 2^-26 |a||b|); all other shapes
 *     and the gather modes run on v_mfma_f32_32x32x2_f32;
 *   1 fp32 MFMA only;   2 split-bf16 wherever eligible;
 *   3 mixed precision (BASELINE configs[4]): wherever the split-bf16 kernel would run, the operands are rounded to fp16
 *     and multiplied by ONE v_mfma_f32_32x32x16_f16 per product tile with fp32 accumulation (fp32 storage stays). */
int ud_gemm_set_path(int path);
int ud_gemm_get_path(void);
/* 2 if ud_gemm would run this descriptor on the BF16 matrix pipe (split-bf16 kernel), 3 for its fp16 mixed-precision
 * mode (path 3), 1 for the fp32 pipe */
int ud_gemm_query_path(const ud_gemm_desc* d);

/* ---- ud_gemm_p3: the same fp32-accurate product from PRE-SPLIT operands (round 4) ----------------------------------
 * Serves the spectral 1x1 conv of the SF blocks, F.conv2d(x_freq, freq_conv.weight) in model/efficientnet/exp.py:57 (and
 * model/resnet/exp.py's copy), its data gradient and its weight gradient — the large GEMMs of the step.
 * An operand is a matrix X[R][Cx] stored as 16-bit planes in the "P32" panel layout:
 *     piece p of X[r][c]   at   X + p * plane + (c / 32) * panel + r * 32 + (c % 32)        (16-bit elements)
 * Each panel must be backed by rows up to the next multiple of 128 (panel >= 32 * roundup(R, 128); slack rows are read,
 * their products discarded).
 *   prec 3: three bf16 planes x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): the exact split
 *           ud_gemm performs inside its k-loop), six piece products — bitwise ud_gemm's result.  Written by ud_split_planes.
 *   prec 2: two fp16 planes s * x = h0 + 2^-11 h1 with a power-of-two scale s that takes the largest |x| it covers into
 *           [2^14, 2^15) — ONE scale for the tensor (ud_absmax + ud_split_planes_h2t; any mode) or one per row of X
 *           (ud_split_planes_h2; mode 0 only: the scale must be constant along k).  Three piece products on the fp16 matrix
 *           pipe into two fp32 accumulators (a0b0, and a1b0 + a0b1), the result (hi + 2^-11 lo) * a_inv_scale[m * a_scale_stride]
 *           * b_inv_scale[n * b_scale_stride] (= 1 / s; stride 0: a scalar, 1: a vector [M] / [N]).  22 significand bits of every
 *           element within 2^-18 of the scale's maximum (smaller elements: an absolute error of 2^-40 of that maximum — so with ONE
 *           scale per tensor a GEMM row lying 2^-24 below the largest keeps ~16 bits; with a scale per row every row keeps 22), the
 *           dropped a1b1 < 2^-24 |ab|, fp32 accumulation: the accuracy of an
 *           fp32 GEMM (measured at or below ud_gemm's error on every operand distribution tried) at half the matrix work.
 *   prec 1: ONE fp16 plane per operand — the first plane of prec-2 operands (or planes written by ud_planes_from_half, scale 1):
 *           one product per tile on the fp16 matrix pipe, fp32 accumulation, the result times the two inverse scales — the
 *           mixed-precision mode (BASELINE configs[4]: fp16 MFMA operands), 530-770 TFLOP/s on the spectral convs' shapes against
 *           290-530 of ud_gemm's path 3 (tools/bench_p3_prec1.py).  (a_mode, b_mode) as prec 2; c_half: a half-stored result.
 *   mode 0: GEMM row = row of X, k = column of X     (activations [pixels][C] as A; weights [Cout][Cin] as B)
 *   mode 1: GEMM row = column of X, k = row of X     (dY / X of a weight gradient; weights of a data gradient)
 * (a_mode, b_mode): prec 3 any; prec 2 (0,0) (0,1) (1,1).  K % 32 == 0.  *_npanel: panels reachable from the pointer (mode 1 clamps
 * its tile to them).  out_mode / split_k / slice_stride / stat_sum / stat_sumsq as in ud_gemm_desc (statistics: out_mode 0,
 * split_k 1; one slot array [N] while M <= 64 * 128, else 64 slots).
 * tile_cfg bit 8: each XCD takes a contiguous range of the tile order; bit 9 (0x200): XCD-aware grouped raster — XCD x takes a
 * contiguous range of an order that walks groups of GM = bits 12-15 (0: 4) tile rows column by column, so the workgroups of one L2
 * share GM A panels and 32 / GM B panels; bit 11 (0x800): stream-K — one workgroup per CU, the
 * (tile, K-tile) units dealt evenly; out_mode 0 (C zeroed by the caller: whole-tile segments store, partial ones add
 * atomically) or 1 (C holds a term to add to); split_k 1, no statistics. */
typedef struct {
    const uint16_t* A; const uint16_t* B; float* C;
    int M, N, K;
    long a_panel, a_plane, b_panel, b_plane;   /* 16-bit elements between panels / between the planes */
    int a_npanel, b_npanel;
    long ldc;
    int a_mode, b_mode, out_mode, split_k;
    double* stat_sum; double* stat_sumsq;
    int tile_cfg;
    long slice_stride;
    int prec;                                   /* 3, 2 or 1 */
    const float* a_inv_scale; const float* b_inv_scale;   /* prec 2 / 1 */
    int a_scale_stride, b_scale_stride;         /* 0: one scale for the operand; 1: one per GEMM row */
    int c_half;                                 /* prec 1: C is _Float16 (out_mode 0 / 1, split_k 1, no stream-K) */
} ud_gemm_p3_desc;
int ud_gemm_p3(const ud_gemm_p3_desc* d, ud_stream_t stream);
/* TWO products in one launch: the data gradient (nn: a_mode 0, b_mode 1) and the weight gradient (tn: a_mode 1, b_mode 1) of one
 * 1x1 conv — prec 2, out_mode 0 / 1 / 2, split_k >= 1, no stream-K form, no epilogue statistics.  The weight gradient's workgroups
 * follow the data gradient's in the same grid, so they start on the CUs the data gradient's last round of tiles leaves idle
 * (540 + 225 tiles on 256 CUs: 3 rounds instead of 3 + 1); tn->tile_cfg bit 16 (0x10000): the weight gradient's workgroups lead
 * the grid instead (few long tiles: the pair's critical path).  Results are those of two ud_gemm_p3 calls.
 * Round 6, the TAIL PAIR of one forward product (both descriptors a_mode 0, b_mode 0, prec 2): `nn` = the leading row tiles of
 * y = x w^T that fill whole rounds of the CUs (plain), `tn` = the remaining row tiles (A / C advanced to their first row) split
 * over K with atomics onto zeroed rows — their short workgroups fill what the plain part's last round leaves idle. */
int ud_gemm_p3_pair(const ud_gemm_p3_desc* nn, const ud_gemm_p3_desc* tn, ud_stream_t stream);
/* A half-stored matrix X[R][C] (row stride ld elements, C % 8 == 0, 16-byte aligned) as ONE fp16 plane in the P32 layout, values
 * unchanged (*inv_scale = 1): the operand of ud_gemm_p3 prec 1 for the activations of the mixed-precision mode (BASELINE
 * configs[4]); pad columns of the last panel zero. */
int ud_planes_from_half(const void* x, long R, int C, long ld, uint16_t* planes, long panel_stride, float* inv_scale,
                        ud_stream_t stream);
/* x fp32 [R][C] (row stride ld; C, ld multiples of 4) -> three bf16 planes in the P32 layout above; columns C .. 32*ceil(C/32)-1
 * are written as zeros, slack rows are left untouched. */
int ud_split_planes(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride, long plane_stride,
                    ud_stream_t stream);
/* the prec-2 form with one scale per row: two fp16 planes of s_r * x[r][:] and inv_scale[r] = 1 / s_r (C <= 4096) */
int ud_split_planes_h2(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride, long plane_stride,
                       float* inv_scale, ud_stream_t stream);
/* the prec-2 form with one scale for the tensor: absmax[256] = partial maxima of |x| over the matrix, as the bit patterns of
 * non-negative floats (ud_absmax writes all 256; a producer kernel may fill them instead), then the split folds them and writes
 * the planes and *inv_scale = 1 / s */
int ud_absmax(const float* x, long R, int C, long ld, uint32_t* absmax, ud_stream_t stream);
int ud_split_planes_h2t(const float* x, long R, int C, long ld, uint16_t* planes, long panel_stride, long plane_stride,
                        const uint32_t* absmax, float* inv_scale, ud_stream_t stream);
/* Every weight matrix of a step in TWO launches (an absmax pass and a split pass over all of them) instead of two per matrix:
 * `items` is a DEVICE array of n descriptors sorted by their block prefixes — item i owns absmax blocks [amax_block0,
 * amax_block0 + amax_blocks) (amax_blocks <= 256: block b writes slot b of the item's 256 slots; the caller zeroes the
 * n * 256 slots ONCE, the unused ones stay zero) and split blocks [split_block0, split_block0 + split_bx * ceil(C / 32)),
 * split_bx = ceil(R / 64).  Same planes and scales as ud_absmax + ud_split_planes_h2t per matrix (tested bitwise). */
typedef struct {
    const float* x; uint16_t* out; float* inv_scale;
    long R, ld, panel, plane;
    int C, amax_block0, amax_blocks, split_block0, split_bx, pad_;
} ud_split_item;
int ud_split_planes_h2t_multi(const ud_split_item* items_dev, int n, uint32_t* slots, int amax_blocks_total,
                              int split_blocks_total, ud_stream_t stream);

/* A k x k conv (F.conv2d, any stride; model/resnet/exp.py:95-111, model/modules.py:111, the decoders) as a 1x1 conv on ud_gemm_p3:
 * ud_im2col_planes writes its im2col matrix [N Hout Wout] x [KH KW Cin] (g->transposed == 0, Cin % 32 == 0) DIRECTLY as prec-2
 * planes (P32 layout, one scale from `absmax`: 256 slots holding |x|max of the conv's input) — forward = planes . W^T, weight
 * gradient = dY^T . planes, data gradient = ud_col2im(dY . W): dx[n][ih][iw][ci] = sum over the taps of the [N Hout Wout] x
 * [KH KW Cin] fp32 matrix dcol (the gather adjoint to the im2col; dx written). */
int ud_im2col_planes(const float* x, const ud_conv_geom* g, uint16_t* planes, long panel_stride, long plane_stride,
                     const uint32_t* absmax, float* inv_scale, ud_stream_t stream);
int ud_col2im(const float* dcol, const ud_conv_geom* g, float* dx, ud_stream_t stream);

/* All k x k conv weights of a step into the [rows][tap][reduced channel] matrices the implicit-GEMM convs read, in ONE launch
 * (F.conv2d / ConvTranspose2d weights of model/unidefense.py:59-102, model/modules.py:111, the stem, model/resnet/exp.py:95-111):
 * src W[A][B][KH][KW] -> mode 0: dst[a][kh][kw][b]; mode 1: dst[b][KH-1-kh][KW-1-kw][a]; mode 2: dst[b][kh][kw][a].
 * items_dev: device table; item i owns blocks [block0, block0 + ceil(A B KH KW / 256)), blocks_total their sum. */
typedef struct {
    const float* src;
    float* dst;
    int A, B, KH, KW, mode, block0;
} ud_layout_item;
int ud_weight_layouts_multi(const ud_layout_item* items_dev, int n, int blocks_total, ud_stream_t stream);

/* ---- column reductions / normalisation on [G][R][C]  (C % 4 == 0) -----------------------------
 * Every reduction is a partial pass (fp64 per-workgroup totals stored into the scratch `ws`) plus a small
 * finalize launch; deterministic, no atomics.  ws: ud_reduce_ws_doubles(G, R, C) doubles of scratch, no
 * initialisation needed, contents undefined afterwards (one buffer can serve all calls issued in order on a stream).
 * Serves nn.BatchNorm2d/1d in training mode (model/efficientnet/model.py:67,77,91,186,222;
 * model/unidefense.py:104; model/modules.py:83,112), nn.InstanceNorm2d (model/unidefense.py:54),
 * MemoryEfficientSwish (utils.py:66-82), adaptive_avg_pool2d(x,1) / mean([-2,-1]). */
int ud_reduce_ws_doubles(int G, int R, int C);
/* mean[G][C], invstd[G][C] = 1/sqrt(biased var + eps); var_out optional; when running_mean != NULL and
 * G == 1 the running statistics are updated in place (momentum, unbiased variance) like nn.BatchNorm. */
int ud_norm_stats(const float* x, int G, int R, int C, float eps, double* ws, float* mean, float* invstd,
                  float* var_out, float momentum, float* running_mean, float* running_var, ud_stream_t stream);
/* torch.nn.SyncBatchNorm forward exchange (engine/forgery_engine.py:142), after an all_gather of every rank's
 * (mean, biased var) over `rows_per_rank` rows each: gathered[world][2][C] -> global mean / invstd, running
 * statistics updated with the unbiased variance over world*rows_per_rank rows (running_* may be NULL). */
int ud_syncbn_combine(const float* gathered, int world, int C, long rows_per_rank, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, ud_stream_t stream);
/* y = act(gamma * (x - mean[g]) * invstd[g] + beta) */
int ud_norm_apply_fwd(const float* x, int G, int R, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, float* y, ud_stream_t stream);
/* dz = dy * act'(z); s1 = sum dz, s2 = sum dz*xhat per (g,c); dgamma = sum_g s2, dbeta = sum_g s1;
 * dx = gamma*invstd*(dz - s1/R - xhat*s2/R)   (dx may be NULL: reductions only) */
int ud_norm_bwd(const float* x, const float* dy, int G, int R, int C, const float* mean, const float* invstd,
                const float* gamma, const float* beta, int act, double* ws, float* s1, float* s2, float* dgamma,
                float* dbeta, float* dx, ud_stream_t stream);
/* One-launch forms of the two above (csrc/norm.hip, round 6): statistics + apply / sums + apply in ONE kernel — the P workgroups
 * sharing a (group, column group) publish their fp64 partials with returning agent-scope atomic exchanges, meet at a counter and
 * fold the partials in a fixed order (deterministic; no fence, no L2 write-back), then apply out of L2.  For the launch-bound
 * InstanceNorms of the decoder (model/unidefense.py:59-102) and the ResNet variants' BatchNorms (model/resnet/exp.py:79-232).
 * slots: ud_norm_fused_ws_doubles doubles (contents arbitrary); counters: ud_norm_fused_counters ZERO 32-bit words.
 * ud_norm_bwd_fused: G > 1 needs s1 / s2 [G][C] (one extra launch sums them into dgamma / dbeta); dx may be NULL. */
long ud_norm_fused_ws_doubles(int G, int R, int C);
int ud_norm_fused_counters(int G, int R, int C);
int ud_norm_fwd_fused(const float* x, int G, int R, int C, const float* gamma, const float* beta, int act, float eps,
                      double* slots, uint32_t* counters, float* mean, float* invstd, float momentum, float* running_mean,
                      float* running_var, float* y, ud_stream_t stream);
int ud_norm_bwd_fused(const float* x, const float* dy, int G, int R, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, double* slots, uint32_t* counters, float* s1,
                      float* s2, float* dgamma, float* dbeta, float* dx, ud_stream_t stream);
/* the elementwise half of ud_norm_bwd with caller-provided sums (already all-reduced over the ranks) and
 * inv_count = 1 / (rows of all ranks): SyncBatchNorm backward (engine/forgery_engine.py:142) */
int ud_norm_bwd_apply(const float* x, const float* dy, int G, int R, int C, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* s1,
                      const float* s2, float inv_count, int act, float* dx, ud_stream_t stream);
/* out[g][c] = scale * sum_r x[g][r][c] ;  out[g][c] = scale * sum_r a*b */
int ud_group_colsum(const float* x, int G, int R, int C, float scale, double* ws, float* out, ud_stream_t stream);
int ud_group_coldot(const float* a, const float* b, int G, int R, int C, float scale, double* ws, float* out,
                    ud_stream_t stream);
/* out[n][p][c] = g[n][c] * scale   (gradient of the mean over the HW rows) */
int ud_bcast_rows(const float* g, float scale, float* out, int N, int HW, int C, ud_stream_t stream);

/* ---- depthwise k x k conv, k in {3,5}, stride in {1,2}; weights tap-major wt[k*k][C] ------------
 * Serves the depthwise Conv2dStaticSamePadding / the spatial branch of SFConv2dStaticSamePadding
 * (model/efficientnet/utils.py:277-280, exp.py:49-51); pad_t/pad_l = top/left of the static ZeroPad2d. */
int ud_dwconv_fwd(const void* x, const float* wt, void* y, int N, int H, int W, int C, int Ho, int Wo, int K,
    int stride, int pad_t, int pad_l, int f16, ud_stream_t stream);
/* add (may be NULL): another contribution to the same gradient, [N][H][W][C]; dx = conv-transpose(dy) + add
 * (the input of SFConv's spatial branch also feeds its frequency branch, exp.py:49-55) */
int ud_dwconv_bwd_data(const void* dy, const float* wt, const void* add, void* dx, int N, int H, int W, int C,
    int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int f16, ud_stream_t stream);
int ud_dwconv_bwd_weight_parts(int C, int chunks);   /* rows of K*K*C floats that `part` must hold */
/* every depthwise weight of the network to the tap-major layout in one launch.  table: layers x 4 int64 in DEVICE
 * memory (source pointer [C][K*K], C, K*K, destination offset in floats); dst[off + tap*C + c] = src[c*K*K + tap];
 * max_elems = the largest C*K*K */
int ud_dw_weights_tapmajor(const void* table, int layers, long max_elems, float* dst, ud_stream_t stream);
/* dwt: the gradient in the PARAMETER's layout [C][K*K] (nn.Conv2d weight [C,1,K,K]), not tap-major */
int ud_dwconv_bwd_weight(const float* x, const float* dy, float* dwt, float* part, int chunks, int N,
                         int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l,
                         ud_stream_t stream);

/* ---- batched real 2-D FFT, S x S planes, S in {8,16,32,64}, channel = lane --------------------
 * ud_rfft2 : x[N][S][S][C] -> Y[N][S][S/2+1][2C]  (Re in channels [0,C), Im in [C,2C)),
 *            Y = f(kx) * DFT2(x), f = scale for kx in {0,S/2}, scale*w_interior otherwise
 * ud_irfft2: Y -> x = scale * C2R(f(kx) * Y), f = 1 / w_interior likewise
 * torch.fft.rfft2/irfft2 (model/efficientnet/exp.py:55,60; model/unidefense.py:130-145):
 *   forward rfft2 = (scale,1), its adjoint = ud_irfft2(scale, 0.5);
 *   forward irfft2 = (scale,1), its adjoint = ud_rfft2(scale, 2). */
int ud_rfft2(const void* x, void* Y, int N, int S, int C, float scale, float w_interior, int f16, ud_stream_t
    stream);
int ud_irfft2(const void* Y, void* x, int N, int S, int C, float scale, float w_interior, int f16, ud_stream_t
    stream);
/* S = 32: the transform may be shared by a LANE PAIR (lanes l and l ^ 32 hold the even / odd decimated samples, the last
 * radix-2 stage crosses the halves with wavefront shuffles; a wave owns 32 channels = whole 128-byte lines).  0 auto (where the
 * 32-channel groups are full and fit one round of workgroups), 1 never, 2 always (env UD_FFT32_WAVE = 0 / 1: never / always).
 * Returns the previous mode. */
int ud_fft32_set_wave(int mode);

/* ---- small FC: y[n][o] = sum_i act_in(x[n][i]) W[o][i] + b[o]  (SE 1x1 convs, classifier) ------
 * model/efficientnet/model.py:119-121; model/modules.py:27.  Any of dx / dW / db may be NULL. */
int ud_fc_fwd(const float* x, const float* W, const float* b, float* y, int N, int I, int O, int act_in,
              ud_stream_t stream);
int ud_fc_bwd(const float* dy, const float* W, const float* x, float* dx, float* dW, float* db, int N,
              int I, int O, int act_in, ud_stream_t stream);

/* ---- SE gating (model/efficientnet/model.py:122): y = x * sigmoid(s[n][c]);
 *      dx = dy * sigmoid(s) + dpool[n][c] / HW;   v *= sigmoid'(s) */
int ud_se_scale_fwd(const float* x, const float* s, float* y, int N, int HW, int C, ud_stream_t stream);
int ud_se_scale_bwd(const float* dy, const float* s, const float* dpool, float* dx, int N, int HW, int C,
                    ud_stream_t stream);
int ud_sigmoid_grad_mul(const float* s, float* v, long n, ud_stream_t stream);

/* ---- SFConv mixing (model/efficientnet/exp.py:61-65): y = (1-a) spat + a P(freq), a = sigmoid(*alpha),
 *      P = identity or the 2x2 average pool (pool = 1: freq is [N][2Ho][2Wo][C]).
 *      bwd: part holds ud_sfmix_blocks() DOUBLES; dalpha[0] = sigmoid'(alpha) * sum dy (P(freq) - spat) */
int ud_sfmix_blocks(int N, int Ho, int Wo, int C);
int ud_sfmix_fwd(const void* spat, const void* freq, const float* alpha, void* y, int N, int Ho, int Wo, int C,
    int pool, int f16, ud_stream_t stream);
int ud_sfmix_bwd(const void* spat, const void* freq, const float* alpha, const void* dy, void* dspat, void*
    dfreq, double* part, float* dalpha, int N, int Ho, int Wo, int C, int pool, int f16, ud_stream_t stream);
/* same for two equal-shape tensors (fuse_coef, model/unidefense.py:153-154) */
int ud_gate_mix_blocks(long total);
int ud_gate_mix_fwd(const float* p, const float* q, const float* alpha, float* y, long total,
                    ud_stream_t stream);
int ud_gate_mix_bwd(const float* p, const float* q, const float* alpha, const float* dy, float* dp, float* dq,
                    double* part, float* dalpha, long total, ud_stream_t stream);

/* ---- elementwise --------------------------------------------------------------------------------
 * ud_residual_fwd: out = x * (keep[n] * inv_keep) + skip   (drop_connect + skip, model.py:130-134;
 *                  skip / keep may be NULL)
 * ud_axpby: out = alpha*a + beta*b (b may be NULL); ud_mask_scale: out = x*mask*scale (dropout by mask);
 * ud_absdiff: out = |a - b| (b may be NULL) */
int ud_residual_fwd(const float* x, const float* skip, const float* keep, float inv_keep, float* out,
                    long total, long per_sample, ud_stream_t stream);
int ud_axpby(const void* a, float alpha, const void* b, float beta, void* out, long total, int f16, ud_stream_t
    stream);
int ud_mask_scale(const float* x, const float* mask, float scale, float* out, long total,
                  ud_stream_t stream);
int ud_absdiff(const float* a, const float* b, float* out, long total, ud_stream_t stream);

/* ---- layout + image-domain tail --------------------------------------------------------------------
 * ud_pix_to_planes: [N][HW][C] -> [N][C][HW], mode 1 applies tanh (nn.Tanh, model/unidefense.py:101)
 * ud_planes_to_pix: reverse; mode 2 multiplies by (1 - aux^2), aux = saved tanh output (planes)
 * ud_bilinear_*   : F.interpolate(mode='bilinear', align_corners=True) on P planes (unidefense.py:16)
 * ud_l1_*         : out[n] = scale * sum |a - b| per sample and its gradient g[n]*scale*sign(a-b)
 *                   (torch.abs(...).mean(dim=[-3,-2,-1]), unidefense.py:245,251-253); b may be NULL */
int ud_pix_to_planes(const float* in, float* out, int N, int C, int HW, int mode, ud_stream_t stream);
int ud_planes_to_pix(const float* in, const float* aux, float* out, int N, int C, int HW, int mode,
                     ud_stream_t stream);
int ud_bilinear_fwd(const float* x, float* y, int P, int Hi, int Wi, int Ho, int Wo, ud_stream_t stream);
int ud_bilinear_bwd(const float* dy, float* dx, int P, int Hi, int Wi, int Ho, int Wo, ud_stream_t stream);
int ud_l1_chunks(long per_sample);
int ud_l1_fwd(const float* a, const float* b, float* part, float* out, int N, long per_sample, float scale,
              ud_stream_t stream);
int ud_l1_bwd(const float* a, const float* b, const float* g, float scale, int accumulate, float* da, int N,
              long per_sample, ud_stream_t stream);

/* ---- dynamic-filter mask (model/modules.py:94-104, 123-133), rows m over pixels:
 *      pre[m] = [mean_c proj, max_c proj, diff[m][0..D)], mask = sigmoid(w2 . pre), out = mask * x
 *      bwd: dx = mask*dout; dlogit[m]; dproj[m][c] = dlogit*(w2[0]/C + w2[1]*[c == argmax]) */
int ud_dynfilter_fwd(const float* proj, const float* diff, const float* w2, const float* x, float* pre,
                     int* argmax, float* mask, float* out, int M, int C, int D, int Cx, ud_stream_t stream);
int ud_dynfilter_bwd(const float* dout, const float* dmask_ext, const float* x, const float* mask,
                     const int* argmax, const float* w2, float* dx, float* dlogit, float* dproj, int M,
                     int C, int Cx, ud_stream_t stream);

/* ---- ResNet-variant helpers (model/resnet/module_exp.py, model/resnet/exp.py) ------------------------
 * ud_avgpool_*    : k x k average pool to [N][Ho][Wo][C] (F.adaptive_avg_pool2d to size/k, module_exp.py:30-31)
 * ud_maxpool3s2_* : nn.MaxPool2d(3, 2, 1) with the winning tap per output in arg (module_exp.py:73-75)
 * ud_add_act_fwd  : y = act(a + b) (act 0 / 2 = ReLU) — `x += shortcut; act(x)` (resnet/exp.py:146-147)
 * ud_relu_bwd     : g = dy * [y > 0]
 * ud_copy_cols    : dir 0: wide[m][off..off+Cn) = narrow[m][:] (torch.cat(dim=1), module_exp.py:32);
 *                   dir 1: the reverse (slice = gradient of the concat) */
int ud_avgpool_fwd(const float* x, float* y, int N, int Ho, int Wo, int C, int k, ud_stream_t stream);
int ud_avgpool_bwd(const float* dy, float* dx, int N, int Ho, int Wo, int C, int k, ud_stream_t stream);
/* F.adaptive_avg_pool2d to an output size that does not divide the input (model/efficientnet/exp.py:61-62 on the 95 x 95 map of
 * the 380 x 380 trunk: 95 -> 48), pixel-major [N][H][W][C] -> [N][Ho][Wo][C], Ho <= H, Wo <= W; windows [floor(o H / Ho),
 * ceil((o + 1) H / Ho)) as in ATen; _bwd is its adjoint (dx written, not accumulated). */
int ud_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int Ho, int Wo, int C, ud_stream_t stream);
int ud_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int Ho, int Wo, int C, ud_stream_t stream);
int ud_maxpool3s2_fwd(const float* x, float* y, unsigned char* arg, int N, int H, int W, int C,
                      ud_stream_t stream);
int ud_maxpool3s2_bwd(const float* dy, const unsigned char* arg, float* dx, int N, int H, int W, int C,
                      ud_stream_t stream);
int ud_add_act_fwd(const float* a, const float* b, int act, float* y, long total, ud_stream_t stream);
int ud_relu_bwd(const float* dy, const float* y, float* g, long total, ud_stream_t stream);
int ud_copy_cols(float* narrow, float* wide, long M, int Cn, int Cw, int off, int dir, ud_stream_t stream);

/* ---- asymmetrical weighted triplet loss (loss/triplet_loss.py:16-82) on feat[N][D]; anchors = the first n_real
 * rows (the batch is ordered [real...; fake...], triplet_loss.py:46-53).  Writes the scalar loss and
 * dfeat[N][D] = d loss / d feat in two launches (as torch ops: ~90 tiny kernels per feature).
 * ws: n_real * (N + 1) floats of scratch. */
int ud_aw_triplet(const float* feat, int N, int D, int n_real, float* loss, float* dfeat, float* ws,
                  ud_stream_t stream);

/* ---- the scalar tail of one pass's loss in two launches (engine/abstract_engine.py:241-281 of the reference: `softmax`
 * criterion on cls_out :250-254, the mask means :256-263, the triplet terms :232-240, the real / fake means of the per-sample
 * reconstruction and frequency terms :241-249, their weighted sum :264-270 — ~45 torch launches of 4-5 us per pass):
 *   vals[0] = w_cls CE(cls, tgt) + w_fm mean(fm) + w_sm mean(sm) + w_trip sum_f awtriplet(feat_f) + w_rec mean(spatial[:R])
 *             + w_freq mean(freq[:R]);  vals[1..8] = CE, sum of triplet terms, real / fake mean of spatial, real / fake mean
 *   of freq, mean(fm), mean(sm);  d* = d vals[0] / d (that input).  The first R rows are the real samples, the next F the fake
 * ones (triplet_loss.py:46-53).  NULL fm / sm / spatial / freq: the term is absent.  ws: ud_loss_tail_ws_floats floats. */
typedef struct {
    const float* feat[3]; float* dfeat[3]; int D[3]; int nfeat;          /* triplet features [N][D_f] */
    const float* cls; const long long* tgt; float* dcls; int N, C, R, F;   /* logits [N][C], int64 labels */
    const float* fm; float* dfm; int nfm; const float* sm; float* dsm; int nsm;
    const float* spatial; float* dspatial; const float* freq; float* dfreq; /* [N] */
    float w_cls, w_fm, w_sm, w_trip, w_rec, w_freq;
    float* vals;                                                            /* 9 floats */
    float* ws;
} ud_loss_tail;
int ud_loss_tail_ws_floats(int N, int n_real, int nfeat);
int ud_loss_tail_run(const ud_loss_tail* t, ud_stream_t stream);

/* ---- direct 3x3 conv for small channel counts (csrc/conv_small.hip): the image-resolution end of the decoder
 * (model/unidefense.py:59-102: 40 -> 20 -> 3 channels at 64x64 / 128x128), its data gradients / transposed conv, and
 * the stem conv (model/efficientnet/model.py:185).  Same operation and geometry as ud_gemm with a_mode 2:
 * y[N][Hout][Wout][Cout] = gather-conv(x[N][Hin][Win][Cin], wmat[Cout][9*Cin]); one thread per output pixel.
 * ud_conv_small_supported: 1 if a kernel exists for (Cin, Cout, KH, KW), else 0 (use ud_gemm). */
int ud_conv_small_supported(int Cin, int Cout, int KH, int KW);
int ud_conv_small(const ud_conv_geom* g, const float* x, const float* wmat, float* y, int Cout, ud_stream_t stream);
/* weight gradient of the same convs (ud_gemm's b_mode 2): out[Ma][9*Cin] = sum over the rows m = (n,oh,ow) of g's
 * output grid of a[m][Ma] (x) patch(x)[m][9*Cin].  part: ud_conv_small_wgrad_ws_floats(Cin, Ma) floats of scratch
 * (per-workgroup partial results, summed by a second launch; deterministic, no atomics). */
int ud_conv_small_wgrad_supported(int Cin, int Ma, int KH, int KW);
long ud_conv_small_wgrad_ws_floats(int Cin, int Ma);
int ud_conv_small_wgrad(const ud_conv_geom* g, const float* a, const float* x, float* part, float* out, int Ma,
                        ud_stream_t stream);

/* ---- pass-2 input perturbations (model/unidefense.py:177-198), NCHW planes x[planes][H][W], no gradients --------
 * ud_gather2d      : out[p][y][x] = in[p][iy[y]][ix[x]] — downscale (model/modules.py:19-21): the two nearest
 *                    F.interpolate calls composed into one gather (index vectors from ATen's float32 rule)
 * ud_blur5_reflect : torchvision gaussian_blur(k=5), model/modules.py:15-16: reflect pad 2, taps (k0,k1,k2,k1,k0)^2
 * ud_amp_mix       : FrequencyStyleTransfer's spectrum step (model/modules.py:44-52) on spectra laid out
 *                    [planes][2S][Whp] (rows [0,S) Re, [S,2S) Im; columns [0,S/2] valid):
 *                    out = w(kx) * (l|A| + (1-l)|B|) * exp(i angle(A)), l = lmda[plane / planes_per_sample],
 *                    w = 2 on the interior columns so the forward transform's adjoint yields irfft2
 * ud_efdm          : SpatialStyleTransfer (model/modules.py:58-76) on rows of L values:
 *                    out = (c + (1-l) * style_sorted[rank(c)]) - (1-l) * c ; ws from ud_efdm_ws_bytes (bytes, <0: bad
 *                    sizes); equal content values take the equal-rank style values in index order
 * ud_coral_moments : CORAL statistics (utils/operation.py:7-13,24-34) of x[N][3][HW]: per (sample, chunk) partial
 *                    sums part[N][chunks][9] = (sum x_c (3), sum x_c x_d for 00 01 02 11 12 22), fp64
 * ud_affine3       : out[n][c][i] = sum_k M[n][c][k] x[n][k][i] + M[n][c][3]   (M[N][3][4]; CORAL's transfer,
 *                    utils/operation.py:36-45, folded into one affine colour map per sample) */
int ud_gather2d(const float* in, float* out, const int* iy, const int* ix, long planes, int H, int W,
                ud_stream_t stream);
int ud_blur5_reflect(const float* in, float* out, long planes, int H, int W, float k0, float k1, float k2,
                     ud_stream_t stream);
int ud_amp_mix(const float* A, const float* B, const float* lmda, float* out, long planes, int S, int Whp,
               int planes_per_sample, ud_stream_t stream);
long ud_efdm_ws_bytes(int rows, int L);
int ud_efdm(const float* content, const float* style, const float* lmda, float* out, int rows, int L,
            int rows_per_sample, void* ws, long ws_bytes, ud_stream_t stream);
int ud_coral_moments(const float* x, double* part, int N, int HW, int chunks, ud_stream_t stream);
int ud_affine3(const float* x, const float* M, float* out, int N, int HW, ud_stream_t stream);

/* ---- deferred normalisation: the fused MBConv path (csrc/fused.hip) ------------------------------------------------
 * MBConvBlock.forward (model/efficientnet/model.py:94-135) alternates 1x1 convs / depthwise or SF convs with
 * BatchNorm + swish, squeeze-excite and the residual.  Here a BatchNorm is never applied as a pass of its own: its
 * batch statistics live as fp64 sums (sum x, sum x^2 per channel) in a zero-initialised accumulator that the
 * PRODUCING side fills with atomic adds (ud_colstats, ud_irfft2_mix), and every CONSUMER applies
 * act(gamma (x - mean) invstd + beta) while it loads x (ud_bn_ref).  The same holds for the backward sums
 * (sum dz, sum dz xhat).  No finalize launches, no normalised copy of the activation in HBM.
 * The accumulators are caller-owned fp64 buffers that must be zero before the producing kernel runs.
 * Data shape arguments (G, R, C): G samples x R pixels x C channels (C % 4 == 0).
 * gate_alpha / gate_mode: an optional scalar factor read on the device: 0 none, 1 sigmoid(alpha[0]),
 * 2 1 - sigmoid(alpha[0])  (the sf_coef mix of exp.py:61-65 carried into the backward kernels). */
typedef struct {
    const double* sum;       /* [G][C] sum of x over the rows (and ranks) the statistics cover */
    const double* sumsq;     /* [G][C] sum of x^2 */
    const float* gamma;      /* [C] */
    const float* beta;       /* [C] */
    double inv_count;        /* 1 / (number of values behind each sum) */
    double unbias;           /* count / (count - 1): running_var takes the unbiased variance (nn.BatchNorm) */
    float eps;
    float momentum;
    int act;                 /* activation applied after the affine map: 0 none, 1 swish */
    int G;                   /* groups of the STATISTICS: 1 = batch norm (one set for all samples) */
    float* running_mean;     /* optional [C]: updated once by the consuming kernel that is handed them */
    float* running_var;
} ud_bn_ref;

/* Storage type: entry points with an `int f16` parameter take their ACTIVATION tensors (declared const void* / void*) as
 * float (f16 = 0) or _Float16 (f16 = 1, "half storage", BASELINE configs[4]); per-channel / per-sample vectors,
 * weights, weight gradients and the fp64 accumulators keep their types.  Arithmetic is fp32 in registers either way;
 * half results are rounded to nearest even on store, and statistics are taken of the rounded values.
 *
 * Reducing entry points take `ws`: fp64 scratch of ud_fused_reduce_ws_doubles(G, R, C, per_group, min_rows) doubles
 * (per_group = 1 for [G][C] outputs, 0 for one [C] set; min_rows = 8) — they run as ONE launch with fp64 atomic adds
 * while at most 64 workgroups would add to the same addresses (ws unused, the helper returns 0) and as partials +
 * finalize otherwise.  ws may be NULL (always atomics). */
long ud_fused_reduce_ws_doubles(int G, int R, int C, int per_group, int min_rows);
/* sum[g][c] += sum_r x, sumsq[g][c] += sum_r x^2   (training-mode BatchNorm statistics, model.py:109,114,126) */
int ud_colstats(const void* x, int G, int R, int C, double* sum, double* sumsq, double* ws, int f16, ud_stream_t
    stream);
/* out[g][c] += sum_r act(bn(x))            (SE squeeze: adaptive_avg_pool2d of the activated tensor, model.py:118;
 *                                           head pooling unidefense.py:226)  — bn->running_* are updated here */
int ud_colsum_bn(const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double* ws, int f16,
    ud_stream_t stream);
/* out[g][c] += sum_r dy * act(bn(x))       (gradient of the SE gate) */
int ud_coldot_bn(const void* dy, const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double*
    ws, int f16, ud_stream_t stream);
/* y[n][o] = sum_i (xsum[n][i] * xscale) W[o][i] + b[o]      (SE reduce conv on the pooled sums) */
int ud_fc_fwd_d(const double* xsum, float xscale, const float* W, const float* b, float* y, int N, int I, int O,
                ud_stream_t stream);
/* y = act(bn(x)) * sigmoid(s[g][c])        (BN1 + swish + SE gate in one pass, model.py:114-122)
 * absmax (here and below; may be NULL): 256 caller-zeroed slots that receive |result|max as a side output — the scale of
 * ud_split_planes_h2t without a pass of its own, for results that feed a 1x1 conv on ud_gemm_p3 */
int ud_se_scale_bn(const void* x, const ud_bn_ref* bn, const float* s, void* y, int G, int R, int C, int f16,
    uint32_t* absmax, ud_stream_t stream);
/* ud_colsum_bn (fp32 storage) that also leaves max |act(bn(x))| in the 256 caller-zeroed slots `absmax`, and ud_se_scale_bn
 * writing its result y = act(bn(x)) * sigmoid(s) DIRECTLY as the fp16 x 2 planes of the project conv's operand (ud_gemm_p3 prec 2,
 * P32 layout over [G R] x C, C % 32 == 0) with the scale that maximum gives (|y| <= it: the gate is a sigmoid) — the SE squeeze
 * pass has just read the whole tensor, so the split pass (ud_split_planes_h2t) and the fp32 tensor between them are not needed. */
int ud_colsum_bn_amax(const void* x, const ud_bn_ref* bn, int G, int R, int C, double* out, double* ws, uint32_t* absmax,
                      ud_stream_t stream);
int ud_se_scale_bn_planes(const void* x, const ud_bn_ref* bn, const float* s, uint16_t* planes, long panel_stride,
                          long plane_stride, float* inv_scale, const uint32_t* amax_in, int G, int R, int C,
                          ud_stream_t stream);
/* half storage (the mixed-precision mode): ud_se_scale_bn whose half result is laid into the ONE fp16 plane of ud_gemm_p3 prec 1 */
int ud_se_scale_bn_plane_half(const void* x, const ud_bn_ref* bn, const float* s, uint16_t* plane, long panel_stride,
                              float* inv_scale, int G, int R, int C, ud_stream_t stream);
/* out = bn(x) * (keep[g] * inv_keep) + skip   (BN2 + drop_connect + residual, model.py:126-134; keep / skip may be
 * NULL; bn->running_* are updated here) */
int ud_residual_bn(const void* x, const ud_bn_ref* bn, const float* keep, float inv_keep, const void* skip,
    void* out, int G, int R, int C, int f16, uint32_t* absmax, ud_stream_t stream);
/* ud_residual_bn (fp32 storage) that ALSO writes its result as the fp16 x 2 planes of the next block's expand conv (ud_gemm_p3 prec 2,
 * P32 layout over [G R] x C) — no split pass over the block output.  Scale from an a-priori bound: |bn(x)_c| <= |gamma_c| sqrt(count)
 * + |beta_c| (times inv_keep under drop-connect) + max |skip| (skip_absmax: the 256 absmax slots its producer left; required with skip). */
int ud_residual_bn_planes(const float* x, const ud_bn_ref* bn, const float* keep, float inv_keep, const float* skip,
                          const uint32_t* skip_absmax, float* out, uint16_t* planes, long panel_stride, long plane_stride,
                          float* inv_scale, int G, int R, int C, uint32_t* absmax, ud_stream_t stream);
/* BatchNorm backward, reductions: dz = dy * (keep[g] * inv_keep) * act'(z)  (dy_is_dz: dz = dy);
 * s1[c] += sum dz, s2[c] += sum dz * xhat;  s3 (optional) [c] += sum dz^2, rounded up (ud_normbwd_apply_planes' energy) */
int ud_normbwd_sums(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn, int
    dy_is_dz, int G, int R, int C, double* s1, double* s2, double* s3, double* ws, int f16, ud_stream_t stream);
/* dx = gamma invstd (dz - s1 inv_count - xhat s2 inv_count); s1/s2: sums over ALL ranks, s1_local/s2_local: this
 * rank's sums -> dbeta / dgamma (NULL: not written) */
int ud_normbwd_apply(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn, int
    dy_is_dz, const double* s1, const double* s2, const double* s1_local, const double* s2_local, int G, int R,
    int C, void* dx, float* dgamma, float* dbeta, int f16, uint32_t* absmax, ud_stream_t stream);
/* ud_normbwd_apply (fp32) writing dx DIRECTLY as the fp16 x 2 planes of the GEMMs that read it (the 1x1 conv's weight and data
 * gradient on ud_gemm_p3 prec 2; P32 layout over [G R] x C, pad columns of the last panel zero) — no fp32 dx, no split pass.  One
 * power-of-two scale for the tensor from an a-priori bound: dx_c = gamma_c invstd_c P(dz_c) with P a contraction, so
 * |dx| <= max_c |gamma_c invstd_c| sqrt(energy[c]), energy[c] >= sum_rows dz_c^2 over the whole batch (ud_normbwd_sums' s3 or
 * ud_irfft2_dwbwd's s3, reduced over the ranks like s1 / s2); *inv_scale receives 1 / scale.  A bound above the true maximum
 * costs log2(bound / max) of the 18 binades in which an element keeps its 22 bits, nothing of the large elements' precision. */
int ud_normbwd_apply_planes(const float* x, const float* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                            int dy_is_dz, const double* s1, const double* s2, const double* s1_local,
                            const double* s2_local, const double* energy, int G, int R, int C, uint16_t* planes,
                            long panel_stride, long plane_stride, float* inv_scale, float* dgamma, float* dbeta,
                            ud_stream_t stream);
/* Backward of a THIN expand 1x1 conv (model/efficientnet/model.py:101-109: _expand_conv + _bn0 of the 128 x 128 / 64 x 64 blocks) in
 * ONE pass over its two big tensors: ud_normbwd_apply (dy_is_dz = 1) + the conv's weight gradient + its data gradient —
 *     de = gamma invstd (dz - s1 / count - xhat s2 / count)   formed in registers from (dz, e), never written
 *     dw[CE][CIN] = de^T x        dx[M][CIN] = de w (+ add; add may alias dx)
 * with gemm_x3's arithmetic (exact three-way bf16 split, six piece products, fp32 accumulation).  (CE, CIN) must be a pair
 * ud_pw_bwd_fused_ok accepts; part: ud_pw_bwd_fused_grid(M) * CE * CIN floats of scratch (one weight-gradient partial per workgroup,
 * folded in workgroup order: bit-reproducible).  dgamma / dbeta (optional) from this rank's sums, as ud_normbwd_apply. */
int ud_pw_bwd_fused_ok(int CE, int CIN);
/* launch form of ud_pw_bwd_fused (kernel bench, tools/bench_pwbwd.py): 0 the shipped form per shape; 1..4 = (tiles of loads in flight
 * per thread, workgroups per CU) in (1, 2), (2, 2), (3, 1), (4, 1) */
int ud_pw_bwd_set_form(int form);
long ud_pw_bwd_fused_grid(long M);
int ud_pw_bwd_fused(const float* e, const float* dz, const ud_bn_ref* bn, const double* s1, const double* s2,
                    const double* s1_local, const double* s2_local, const float* x, const float* w, const float* add, long M,
                    int CE, int CIN, float* dx, float* dw, float* part, float* dgamma, float* dbeta, ud_stream_t stream);
/* Backward of a THIN project 1x1 conv and the squeeze-excite gate in front of it (model/efficientnet/model.py:113-126 of the 64 x 64
 * blocks: c = swish(bn1(d)) sigmoid(s), p = c Wp^T) WITHOUT the conv's data gradient dc = dp Wp ever written: a 32-row tile of dc is
 * re-made from the thin dp [N HW][CO] inside each of the two passes over d that need it —
 *   ud_pj_bwd_fused_a:  dw[CO][CE] = dp^T c  (c re-made from d on load, as ud_se_scale_bn makes it)   and
 *                       dgate[n][ch] += sum_hw dc act(bn1(d))   (ud_coldot_bn's result; dgate zeroed by the caller);
 *                       part: ud_pj_bwd_fused_grid(N, HW, CE, CO) * CO * CE floats of scratch (partials folded in a fixed order);
 *   ud_pj_bwd_fused_b:  dz = (dc sigmoid(s) + dpool inv_hw) act'(bn1(d)),  s1[ch] += sum dz, s2[ch] += sum dz xhat
 *                       (ud_se_scale_bwd_bn's results).
 * gemm_x3's arithmetic for the products.  (CE, CO, HW) must be a triple ud_pj_bwd_fused_ok accepts (HW % 32 == 0: a tile of rows
 * belongs to one sample). */
int ud_pj_bwd_fused_ok(int CE, int CO, int HW);
int ud_pj_fwd_fused_ok(int CE, int CO, int HW);
/* ... and the forward of the same pair of layers in ONE pass over d: p[N HW][CO] = (act(bn1(d)) sigmoid(s)) w^T with the gated tensor
 * made on load (ud_se_scale_bn's values) and never written; sum / sumsq (optional, together): p's BatchNorm-2 statistics
 * (sum[co] += sum_rows p, sumsq[co] += sum_rows p^2).  (CE, CO, HW): a triple ud_pj_fwd_fused_ok accepts. */
int ud_pj_fwd_fused(const float* d, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE, int CO, float* p,
                    double* sum, double* sumsq, ud_stream_t stream);
long ud_pj_bwd_fused_grid(int N, int HW, int CE, int CO);
int ud_pj_bwd_fused_a(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* w, int N, int HW, int CE,
                      int CO, float* dw, double* dgate, float* part, ud_stream_t stream);
int ud_pj_bwd_fused_b(const float* d, const float* dp, const ud_bn_ref* bn, const float* s, const float* dpool, float inv_hw,
                      const float* w, int N, int HW, int CE, int CO, float* dz, double* s1, double* s2, ud_stream_t stream);
/* Half storage (the mixed-precision mode): ud_normbwd_apply whose half result is laid straight into the ONE fp16 plane (P32 layout,
 * scale 1) that ud_gemm_p3 prec 1 reads — the row-major tensor and the ud_planes_from_half pass over it are not needed. */
int ud_normbwd_apply_plane_half(const void* x, const void* dy, const float* keep, float inv_keep, const ud_bn_ref* bn,
                                int dy_is_dz, const double* s1, const double* s2, const double* s1_local,
                                const double* s2_local, int G, int R, int C, uint16_t* plane, long panel_stride,
                                float* inv_scale, float* dgamma, float* dbeta, ud_stream_t stream);
/* SE backward, the two small FC layers (model.py:119-121) in two launches:
 *   a: dpre = dgate[n][c] * sigmoid'(s2);  ds1[n][i] = swish'(s1) sum_c dpre W_e[c][i];
 *      dW_e[c][i] = sum_n dpre swish(s1[n][i]);  db_e[c] = sum_n dpre
 *   b: dpool[n][c] = sum_i ds1 W_r[i][c];  dW_r[i][c] = sum_n ds1[n][i] pool[n][c] * pool_scale;  db_r[i] = sum_n ds1 */
int ud_se_bwd_a(const double* dgate, const float* s2, const float* s1, const float* We, double* ds1_acc, float* dWe,
                float* dbe, int N, int C, int Cs, ud_stream_t stream);
int ud_se_bwd_b(const double* ds1_acc, const float* s1, const float* Wr, const double* pool, float pool_scale,
                float* dpool, float* dWr, float* dbr, int N, int C, int Cs, ud_stream_t stream);
/* db = dc * sigmoid(s[g][c]) + dpool[g][c] * inv_hw;  dz = db * act'(bn(x));  s1 += sum dz, s2 += sum dz xhat
 * (gradient through the SE gate and the swish of BN1, with BN1's backward sums) */
int ud_se_scale_bwd_bn(const void* dc, const void* x, const ud_bn_ref* bn, const float* s, const float* dpool,
    float inv_hw, void* dz, double* s1, double* s2, double* ws, int G, int R, int C, int f16, ud_stream_t
    stream);
/* ud_normbwd_apply (dy_is_dz) fused with the gradient of the SF mix y = (1-a) spat + a freq (exp.py:61-65):
 * writes dd = dL/dy and accumulates sum dd * diff, diff = freq - spat as stored by ud_irfft2_mix, into the 64 slots
 * dalpha_acc[0..64) (zeroed by the caller); energy (optional, C doubles zeroed by the caller): += sum_rows dd_c^2, rounded up —
 * the per-channel energy bound ud_rfft2_ex_planes takes for the transform of dd that follows */
int ud_normbwd_apply_mix(const void* x, const void* dz, const ud_bn_ref* bn, const double* s1, const double* s2,
    const double* s1_local, const double* s2_local, const void* diff, int G, int R, int C, void* dd,
    double* dalpha_acc, float* dgamma, float* dbeta, double* energy, int f16, ud_stream_t stream);
/* out[0] = sigmoid'(alpha[0]) * sum(acc[0..64))      (sf_coef gradient from the accumulator above) */
int ud_gate_grad_from_acc(const double* acc, const float* alpha, float* out, ud_stream_t stream);
/* y = act(bn(x)): the materialised form for consumers that re-read their input per tap (a plain depthwise conv and its
 * weight gradient: re-evaluating the swish per window load costs more than this pass); bn->running_* are updated here */
int ud_bn_apply(const void* x, const ud_bn_ref* bn, void* y, int G, int R, int C, int f16, ud_stream_t stream);
/* data gradient of the depthwise conv, scaled by the gate, plus `add`, pushed through the swish of the deferred
 * BatchNorm of its INPUT x:  da = gate * dwconv_bwd_data(dy) + add;  dz = da * act'(bn(x));  s1/s2 as above */
int ud_dwconv_bwd_data_bn(const void* dy, const float* gate_alpha, int gate_mode, const float* wt, const void*
    add, const void* x, const ud_bn_ref* bn, void* dz, double* s1, double* s2, double* ws, int N, int H, int W,
    int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int f16, ud_stream_t stream);
long ud_dwconv_bwd_data_bn_ws_doubles(int N, int H, int W, int C, int stride);
/* ud_dwconv_bwd_data with the gate factor on dy */
int ud_dwconv_bwd_data_ex(const void* dy, const float* gate_alpha, int gate_mode, const float* wt, const void*
    add, void* dx, int N, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int f16,
    ud_stream_t stream);
/* ud_dwconv_bwd_weight with the gate factor on dy */
int ud_dwconv_bwd_weight_ex(const void* x, const void* dy, const float* gate_alpha, int gate_mode, float* dwt,
    float* part, int chunks, int N, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int
    pad_l, int f16, ud_stream_t stream);
/* ud_rfft2 of act(bn(x)) (bn may be NULL), optionally also writing the activated input (act_out) and scaling the
 * result by the gate:  SFConv's spectral branch reading the expand conv's raw output (exp.py:55) and, as the
 * adjoint of irfft2, its backward (gate = sigmoid(sf_coef)).  gate_grad (optional): also finishes the gate's gradient,
 * gate_grad[0] = sigmoid'(alpha) * sum(gate_acc[0..64)), from the slots ud_normbwd_apply_mix filled just before. */
int ud_rfft2_ex(const void* x, void* Y, int N, int S, int C, float scale, float w_interior, const ud_bn_ref* bn,
    void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc, float* gate_grad, int f16,
    uint32_t* absmax, ud_stream_t stream);
/* ud_rfft2_ex whose result goes DIRECTLY into the fp16 x 2 planes ud_gemm_p3 (prec 2) reads — P32 panel layout over the matrix
 * [N S (S/2+1)] x [Re 0..C | Im 0..C], one power-of-two scale for the tensor — instead of fp32 + ud_split_planes_h2t (the
 * spectral 1x1 conv's operand, model/efficientnet/exp.py:55-57, and its output gradient in the backward).  The scale is taken
 * from an a-priori UPPER BOUND of |Y|: bound_pre * sqrt(max_c e_c) with e_c = gamma_c^2 + beta_c^2 (bn given: the input is
 * act(bn(x)), |act(z)| <= |z|; bound_pre = f_max S sqrt(count)) or e_c = energy[c] >= sum_{n,h,w} x_c^2 (bound_pre = f_max S);
 * a bound 2^b above the true maximum moves the 2^-18 window of full 22-bit precision up by b binades and changes nothing for
 * the large elements.  S in {8, 16, 32, 12, 24, 48}, (2 C) % 32 == 0, fp32 storage; *inv_scale receives 1 / scale.
 * dw_k = 3 / 5 (S in {8, 16, 32}; 0: off): the kernel ALSO writes dw_out[N][S][S][C] = the stride-1 depthwise dw_k x dw_k conv
 * (pads (dw_k - 1) / 2, taps dw_wt tap-major [dw_k^2][C]) of the same activated plane act(bn(x)) — SFConv's spatial branch
 * (exp.py:49-51) without a kernel of its own. */
int ud_rfft2_ex_planes(const void* x, uint16_t* planes, long panel_stride, long plane_stride, float* inv_scale, float bound_pre,
                       const double* energy, int N, int S, int C, float scale, float w_interior, const ud_bn_ref* bn,
                       void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc, float* gate_grad,
                       const float* dw_wt, void* dw_out, int dw_k, ud_stream_t stream);
/* irfft2 + SF mix + BN1 statistics (exp.py:60-65, stride 1):  freq = irfft2(Y) * scale;
 * y = (1 - a) spat + a freq, a = sigmoid(alpha[0]);  diff_out = freq - spat (what the backward needs of the two
 * branches: neither has to be kept);  sum[c] += sum y, sumsq[c] += sum y^2 */
int ud_irfft2_mix(const void* Y, void* y, int N, int S, int C, float scale, float w_interior, const void* spat,
    const float* alpha, void* diff_out, double* sum, double* sumsq, int f16, ud_stream_t stream);

/* Two-pass forms of the 32 x 32 / 64 x 64 transforms (torch.fft.rfft2 / irfft2 of exp.py:55,60 and their adjoints): a
 * row kernel and a column kernel with the half-spectrum between them in ws (ud_fft2_two_pass_ws_floats floats, fp32,
 * [N][S][S/2+1][Re 0..C | Im 0..C]); a wave owns 64 consecutive channels of one image row / one kx column, so every access
 * is a run of whole lines where the one-kernel forms above read 16 ... 64-byte pieces.  Same results to the last bit or
 * two.  bn / act_out / gate_*: as ud_rfft2_ex (all NULL / 0: plain
 * ud_rfft2); spat / alpha / freq_out / sum / sumsq: as ud_irfft2_mix (spat NULL: plain ud_irfft2). */
long ud_fft2_two_pass_ws_floats(int N, int S, int C);
int ud_rfft2_two_pass(const void* x, void* Y, float* ws, int N, int S, int C, float scale, float w_interior,
    const ud_bn_ref* bn, void* act_out, const float* gate_alpha, int gate_mode, const double* gate_acc,
    float* gate_grad, int f16, uint32_t* absmax, ud_stream_t stream);
int ud_irfft2_two_pass(const void* Y, void* y, float* ws, int N, int S, int C, float scale, float w_interior,
    const void* spat, const float* alpha, void* freq_out, double* sum, double* sumsq, int f16, ud_stream_t stream);

/* ---- multi-tensor AdamW (csrc/optim.hip) ------------------------------------------------------------------------
 * torch.optim.AdamW(amsgrad) over timm's weight-decay groups (engine/forgery_engine.py:149-156) with GradScaler's
 * unscale and found_inf skip (engine/abstract_engine.py:281-283) folded in: ONE launch for all parameter tensors.
 * table: device array of entries of 8 x int64 {p, g, m, v, vmax (0: no amsgrad) pointers, numel, group, 0}; chunk_map: device array of int32 pairs (tensor, chunk of ud_adamw_chunk_elems() elements),
 * one workgroup each; lr / wd: host arrays per group (<= 8); grad_scale, found_inf: device scalars or NULL;
 * step_in / step_out: device int32 counters (out = in + 1, or in when found_inf != 0 and nothing is updated). */
int ud_adamw_chunk_elems(void);
int ud_adamw_multi(const void* table, const void* chunk_map, int n_chunks, const float* lr, const float* wd, int n_groups,
                   double beta1, double beta2, double eps, int amsgrad, int maximize, const float* grad_scale,
                   const float* found_inf, const int* step_in, int* step_out, ud_stream_t stream);

/* ---- gradient accumulation of a second backward (csrc/optim.hip) --------------------------------------------------
 * The train step calls backward() twice per zero_grad() (engine/abstract_engine.py:281 and :374 under the one
 * optimizer.zero_grad() of engine/forgery_engine.py:241): torch's AccumulateGrad adds the second backward's gradients with one
 * launch per parameter tensor.  ud_multi_add: dst[i][0..numel[i]) += src[i][0..numel[i]) for n fp32 tensors in ceil(n / 120)
 * launches; dst / src / numel are HOST arrays (device pointers inside), read at call time — nothing staged on the device, so the
 * call can sit inside a captured step.  numel[i] < 2^31. */
int ud_multi_add(void* const* dst, const void* const* src, const long* numel, int n, ud_stream_t stream);

/* ---- LDS-tiled depthwise conv, stride 1 (csrc/dwtile.hip) -----------------------------------------------------------
 * The depthwise k x k conv of MBConvBlock.forward (model/efficientnet/model.py:112-115; SFConv's spatial branch
 * exp.py:49-51) with a halo tile of 32 channels staged in LDS once per workgroup, act(bn_in(src)) applied while staging
 * (the activated tensor never exists in HBM).  out(oh, ow) = sum_{i,j} src'(oh + i - P_t, ow + j - P_l) * w[tap(i, j)],
 * src' = act(bn_in(src)) (bn_in NULL: src), zero outside the image; flip = 0: tap = i*K + j (forward, P = the conv's
 * pads); flip = 1: tap = K*K-1 - (i*K + j) with P = K-1 - pad: the data gradient over dy.  wt: tap-major [K*K][C].
 * epi 0: store.  epi 1: also s1 += sum out, s2 += sum out^2 per channel (BN1 statistics of a plain depthwise block).
 * epi 2: out = gate(gate_alpha, gate_mode) * conv [+ add]; with bn_out: out *= act'(bn_out(xbn)) and s1 += sum out,
 * s2 += sum out * xhat (the BatchNorm backward sums).  ws: ud_dwtile_ws_doubles doubles when sums are taken.
 * stride 2 (the four down-sampling blocks): flip = 0 — the forward reads src(2 oh + i - P_t, ..) (epi 0 / 1); flip = 1 with
 * epi 2 — the data gradient: src = dy of the strided conv is read through a zero-stuffed grid (position v holds dy(v / 2)
 * for even v), P = K-1 - pad as for stride 1. */
long ud_dwtile_ws_doubles(int N, int Ho, int Wo, int C);
int ud_dwtile(const void* src, const ud_bn_ref* bn_in, const float* wt, void* out, int N, int Hs, int Ws, int C, int Ho,
              int Wo, int K, int P_t, int P_l, int flip, const float* gate_alpha, int gate_mode, const void* add,
              const void* xbn, const ud_bn_ref* bn_out, int epi, double* s1, double* s2, double* ws, int stride,
              int f16, ud_stream_t stream);
/* weight gradient dwt[C][K*K] = gate * sum_pixels act(bn_in(src))(oh + i - P_t, ow + j - P_l) * dy(oh, ow);
 * part: ud_dwtile_wgrad_part_rows(N, Ho, Wo) rows of K*K*C floats */
long ud_dwtile_wgrad_part_rows(int N, int Ho, int Wo);
int ud_dwtile_wgrad(const void* src, const ud_bn_ref* bn_in, const void* dy, const float* gate_alpha, int gate_mode,
                    float* dwt, float* part, long part_rows, int N, int Hs, int Ws, int C, int Ho, int Wo, int K, int P_t,
                    int P_l, int stride, int f16, ud_stream_t stream);
/* Backward of the stride-1 depthwise conv (autograd of model/efficientnet/model.py:112-115 / exp.py:49-51) in ONE pass over its
 * operands: both halo tiles (dy, act(bn(x))) staged once per image and workgroup;
 *   dz = (gate * conv_flipped(dy) [+ add]) * act'(bn(x))   (bn NULL: no activation factor, x is the conv's input itself),
 *   s1 += sum dz, s2 += sum dz * xhat (bn given; ws: ud_dwtile_ws_doubles(N, H, W, C) doubles),
 *   dwt[C][K*K] = gate * sum_pixels act(bn(x))(oh + i - P_t, ow + j - P_l) * dy(oh, ow)
 * (ud_dwtile epi 2 + ud_dwtile_wgrad, which read each operand twice).  wpart: ud_dwtile_wgrad_part_rows(N, H, W) rows.
 * dwt == NULL: the partial rows are left for ud_dwtile_wgrad_finalize_multi and their number is returned. */
int ud_dwtile_bwd(const void* dy, const void* x, const ud_bn_ref* bn, const float* wt, const float* gate_alpha, int gate_mode,
                  const void* add, void* dz, float* dwt, float* wpart, long part_rows, double* s1, double* s2, double* ws,
                  int N, int H, int W, int C, int K, int P_t, int P_l, int f16, ud_stream_t stream);
/* dwt[C][K*K] = gate * sum_p part[p][K*K][C] (fp64 accumulation): the fold of per-workgroup weight-gradient partials, for kernels
 * outside csrc/dwtile.hip that produce them (ud_irfft2_dwbwd) */
int ud_dwtile_wgrad_finalize(const float* part, int nparts, int K, int C, const float* gate_alpha, int gate_mode, float* dwt,
                             ud_stream_t stream);
/* The folds of ALL depthwise convs of a backward pass in one launch per 48 items (the items travel by value in the kernel
 * arguments: capturable).  ud_dwtile_wgrad / ud_dwtile_bwd called with dwt == NULL leave their partial rows in `part` / `wpart` and
 * RETURN their number (> 0) instead of folding; ud_irfft2_dwbwd's wpart holds N rows.  The partial buffers must stay untouched
 * until this call. */
typedef struct {
    const float* part;          /* [nparts][K*K][C] */
    float* dwt;                 /* [C][K*K] */
    const float* gate_alpha;    /* gate_mode != 0 */
    int nparts, K, C, gate_mode;
} ud_wgrad_fold;
int ud_dwtile_wgrad_finalize_multi(const ud_wgrad_fold* items, int n, ud_stream_t stream);
/* Half storage (the mixed-precision mode): ud_rfft2_ex of a half-stored x whose half result is laid straight into the ONE fp16
 * plane (P32 layout over [N S (S/2+1)] x 2C, scale 1) that ud_gemm_p3 prec 1 reads — no row-major spectrum, no layout pass.  The
 * one-kernel transform sizes (8, 16, 32, 12, 24, 48).  dw_wt / dw_out / dw_k as in ud_rfft2_ex_planes (the stride-1 depthwise conv
 * of the activated plane, half-stored result; S in {8, 16, 32}). */
int ud_rfft2_ex_plane_half(const void* x, uint16_t* plane, long panel_stride, float* inv_scale, int N, int S, int C, float scale,
                           float w_interior, const ud_bn_ref* bn, void* act_out, const float* gate_alpha, int gate_mode,
                           const double* gate_acc, float* gate_grad, const float* dw_wt, void* dw_out, int dw_k,
                           ud_stream_t stream);
/* Backward of an SF block's spatial branch inside the adjoint transform (csrc/fft.hip: irfft2_dwbwd_kernel; S in {8, 16}, K in
 * {3, 5}; f16: Y, dd, x, dz half-stored): da_f = scale * C2R(f(kx) Y) as ud_irfft2 (the adjoint of rfft2: w_interior = 1/2), then with dd = dL/d(conv output)
 * [N][S][S][C], x the conv's raw input and bn the BatchNorm in front of it:
 *   dz = (gate * conv_flipped(dd) + da_f) * act'(bn(x));  s1 += sum dz, s2 += sum dz * xhat;  s3 (optional) += sum dz^2, rounded
 *   up (the energy bound ud_normbwd_apply_planes takes);
 *   wpart[n][K*K][C] = sum_pixels act(bn(x))(window) * dd   of image n  (ud_dwtile_wgrad_finalize sums the N rows) — or, wacc
 *   given (C * K*K floats, zeroed): wacc[c][tap] += gate * that sum by fp32 atomics, no fold launch (N adds per address).
 * Replaces ud_irfft2 + the depthwise weight-gradient kernel + its finalize + the depthwise data-gradient kernel. */
int ud_irfft2_dwbwd(const void* Y, int N, int S, int C, float scale, float w_interior, const void* dd, const void* x,
                    const ud_bn_ref* bn, const float* wt, int K, const float* gate_alpha, int gate_mode, void* dz, double* s1,
                    double* s2, double* s3, float* wpart, float* wacc, int f16, ud_stream_t stream);

/* ---- large real 2-D FFT of image planes (csrc/fft_large.hip), S in {128, 256, 320} ------------------------------------
 * torch.fft.rfft2 on [N,3,S,S] images: the frequency reconstruction loss (model/unidefense.py:246-253; ResNet variants
 * :421-431, :615-625) and FrequencyStyleTransfer (model/modules.py:35-55), with their autograd adjoint.
 * Y[P][2S][Whp]: rows [0,S) = Re(ky), rows [S,2S) = Im(ky); columns [0, S/2] valid, the rest (Whp = ceil4(S/2+1)) zero.
 * ws: ud_rfft2_planes_ws_floats(P, S) floats of scratch.  The adjoint maps dY back to dx[P][S][S] (transpose of the
 * real-linear map x -> (Re Y, Im Y), same scale): with dY weighted 2 on the interior columns it is irfft2. */
long ud_rfft2_planes_ws_floats(long P, int S);
int ud_rfft2_planes(const float* x, float* Y, float* ws, long P, int S, float scale, ud_stream_t stream);
int ud_rfft2_planes_adjoint(const float* dY, float* dx, float* ws, long P, int S, float scale, ud_stream_t stream);

/* ---- one-shot SyncBatchNorm exchange (csrc/xchg.hip) ---------------------------------------------------------------
 * Replaces the per-BatchNorm library collectives of nn.SyncBatchNorm (engine/forgery_engine.py:142) on one node: every
 * rank owns a mailbox in fine-grained device memory mapped into its peers through HIP IPC; ud_xchg_allreduce is ONE
 * small kernel (a thread per double) that writes the rank's doubles, tagged with the exchange's sequence number, into
 * every mailbox (peer writes over xGMI), polls the own mailbox until all ranks' words carry the tag and sums the rows
 * in rank order (bit-identical on all ranks) — no fences, no separate flags.  seq_counter: two zero-initialised device
 * words advanced by the kernel (hipGraph replays keep counting); a peer that does not arrive within timeout_ms of
 * wall-clock time sets *err (1 + its rank) and turns the affected sums into NaN (never a partial sum) instead of hanging.  world <= 16.  Setup: ud_xchg_create on every rank, handles exchanged by the
 * host (64 bytes each), ud_xchg_open on every peer's handle, the `world` pointers (own base at [rank]) copied to a
 * device array.  max_doubles bounds n; slots >= 2 (a rank is never more than one exchange ahead of the slowest).
 * local_out (may be NULL): receives this rank's OWN n doubles as they were before the sum (the local dgamma / dbeta of a
 * BatchNorm backward) — the copy a separate launch made before round 6. */
long ud_xchg_bytes(int world, int max_doubles, int slots);
int ud_xchg_create(int world, int max_doubles, int slots, void** base, char* handle);
int ud_xchg_open(const char* handle, void** ptr);
int ud_xchg_close(void* ptr);
int ud_xchg_destroy(void* base);
int ud_xchg_allreduce(double* acc, int n, void* const* peers, int rank, int world, int max_doubles, int slots,
                      unsigned long long* seq_counter, int* err, long timeout_ms, double* local_out, ud_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
