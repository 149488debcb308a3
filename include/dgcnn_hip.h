/*
 * dgcnn_hip.h -- C ABI of libdgcnn_hip.so, the MI355X (gfx950) EdgeConv path.
 *
 * The reference (BenCurran98/AHN3-dgcnn.pytorch) has no FFI/plugin layer: its
 * boundary for this path is the Python surface of model.py.
 * Each entry point below states which reference lines it replaces; the Python
 * host side (ahn3-dgcnn.pytorch_amd/model.py) mirrors the reference's names and
 * argument meaning and binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a borrowed DEVICE pointer unless stated otherwise;
 *     outputs and workspaces are caller-allocated; nothing is freed or kept.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every
 *     entry point only enqueues work on it (graph-capturable, no host sync).
 *   - return value: 0 = ok, non-zero = error; dgcnn_last_error() then returns a
 *     thread-local message.  No C++ exceptions or torch types cross the ABI.
 *   - "point-major" = [B][N][C] fp32 contiguous (one point's features are one
 *     contiguous row); this is the layout every kernel consumes and produces.
 *   - neighbour indices are int32, local to the batch element (0..N-1).
 */
#ifndef DGCNN_HIP_H
#define DGCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------- */
const char *dgcnn_last_error(void);
/* ABI version, bumped on any signature change. */
int dgcnn_abi_version(void);

/* ---- layout helper ------------------------------------------------------------
 * Replaces x.transpose(2,1).contiguous() (model.py:73) and absorbs the strided
 * inputs the callers hand to forward (train.py:176, test.py:119,304-306):
 * src is a [B][C][N] view with element strides (sb, sc, sn); dst is point-major
 * [B][N][C].  Also used in the other direction (point-major -> [B][C][N]) by
 * swapping the roles of the strides. */
int dgcnn_to_point_major(const float *src, int B, int C, int N,
                         int64_t sb, int64_t sc, int64_t sn,
                         float *dst, void *stream);
int dgcnn_from_point_major(const float *src, int B, int C, int N,
                           float *dst, int64_t sb, int64_t sc, int64_t sn,
                           void *stream);

/* ---- kNN graph build ------------------------------------------------------------
 * Replaces knn(x, k) (model.py:25-40): pairwise -||xi-xl||^2 and topk, fused; the
 * B x N x N distance tensor is never materialised.
 *   x   : point-major [B][N][C]
 *   idx : [B][N][k] int32, sorted by (distance ascending, index ascending); the
 *         arithmetic is the one oracle/knn_exact.c specifies (bit-exact gate).
 * Requires 1 <= k <= min(N, 96), 1 <= C <= 128. */
size_t dgcnn_knn_workspace_bytes(int B, int C, int N, int k);
int dgcnn_knn(const float *x, int B, int N, int C, int k, int32_t *idx,
              void *workspace, size_t workspace_bytes, void *stream);

/* ---- materialised edge features (public get_graph_feature API only) ------------------
 * Replaces the gather / repeat / sub / cat of get_graph_feature (model.py:64-81):
 *   out[b][n][j][0..C)  = x[b][idx[b][n][j]][:] - x[b][n][:]
 *   out[b][n][j][C..2C) = x[b][n][:]
 * out is [B][N][k][2C] contiguous -- exactly the memory the reference returns through a
 * permuted [B][2C][N][k] view.  The EdgeConv kernels never materialise this tensor.
 * The backward OVERWRITES gx: centre term by a local sum over k, neighbour term (autograd's
 * IndexBackward0 of model.py:74, a scatter-add with atomics there) as a gather over the reverse
 * neighbour lists, which the call builds in `workspace` sorted by edge id: the additions happen in
 * a fixed order and the result is bit-stable from run to run.  k <= 128, N <= 40000 (the list kernels keep N ints
 * per wave in LDS), B * N * k < 2^31; the workspace query returns 0 for a shape the backward would refuse
 * (dgcnn_last_error names the limit). */
int dgcnn_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k,
                            float *out, void *stream);
size_t dgcnn_graph_feature_bwd_workspace_bytes(int B, int N, int C, int k);
int dgcnn_graph_feature_bwd(const float *gout, const int32_t *idx, int B, int N, int C, int k,
                            float *gx, void *workspace, size_t workspace_bytes, void *stream);

/* ---- EdgeConv block ---------------------------------------------------------------------
 * One call = one EdgeConv block of DGCNN.forward, i.e. (model.py:147-150 / 154-157 / 161-163)
 *     e = get_graph_feature(x);  e = convA(e);  [e = convB(e);]  y = e.max(dim=-1)[0]
 * with convX = Sequential(Conv2d(c_in, 64, 1, bias=False), BatchNorm2d(64), LeakyReLU(slope))
 * (model.py:116-130).  The kNN indices are an input (dgcnn_knn), so that the graph can be
 * shared, cached or injected by tests.
 *
 * x   : point-major [B][N][C]            idx : [B][N][k] int32
 * y   : point-major [B][N][64]
 * saved : opaque buffer of dgcnn_edgeconv_saved_bytes() that the forward fills and the
 *         backward reads (per-point projections, selected pre-activation extremes, BatchNorm
 *         batch statistics; after a TRAINING forward with N <= 8192 and k <= 64 also the first two
 *         passes of the backward's reverse neighbour lists and the backward's counters, which the
 *         backward leaves ready for another backward from the same buffer).  The backward must be
 *         given the descriptor the forward was given (the layout follows it), and a buffer belongs
 *         to ONE forward at a time.
 * BatchNorm semantics are PyTorch's: training = biased batch variance for normalisation,
 * running_var updated with the unbiased one, running = (1 - momentum) * running + momentum *
 * batch; eval = running statistics.  num_batches_tracked: incremented by a training-mode forward when the
 * descriptor's dgcnn_bn_params carries its address, otherwise the caller's business.
 * Limits (checked, reported through dgcnn_last_error): C <= 64, k <= 96 (two-stage blocks: k <= 64), B * N < 2^24
 * points, B * N * k < 2^31 edges, slope >= 0 (the block takes max over k before the activation, which therefore has
 * to be non-decreasing).  The backward -- training AND eval mode -- additionally needs N <= 40000 (reverse neighbour
 * lists: N ints per wave in LDS); dgcnn_edgeconv_bwd_workspace_bytes returns 0 for a shape the backward would
 * refuse, with the limit in dgcnn_last_error.
 * One process drives one GPU: per-kernel attributes are set for the process's current device. */
typedef struct dgcnn_bn_params {
    const float *weight;  /* gamma [64] */
    const float *bias;    /* beta  [64] */
    float *running_mean;  /* [64]; read when training == 0, updated when training == 1 */
    float *running_var;   /* [64] */
    int64_t *num_batches_tracked; /* optional (may be NULL): nn.BatchNorm's counter, incremented by one in the launch that
                                     finishes the batch statistics of a training-mode forward (dgcnn_edgeconv_fwd only) */
} dgcnn_bn_params;

typedef struct dgcnn_edgeconv_desc {
    int32_t B, N, C, k;
    int32_t n_stage;   /* 1: conv(2C->64)+BN+LReLU;  2: followed by conv(64->64)+BN+LReLU */
    int32_t training;  /* 1: batch statistics (+ running update);  0: running statistics */
    float eps, momentum, slope;
    const float *w1;   /* [64][2C]  = Conv2d(2C, 64, 1).weight */
    dgcnn_bn_params bn1;
    const float *w2;   /* [64][64]  = Conv2d(64, 64, 1).weight, or NULL when n_stage == 1 */
    dgcnn_bn_params bn2;
} dgcnn_edgeconv_desc;

typedef struct dgcnn_edgeconv_grads {
    float *gx;      /* [B][N][C] point-major, or NULL (block 1: the network input needs no grad) */
    float *gw1;     /* [64][2C] */
    float *gbn1_w;  /* [64] */
    float *gbn1_b;  /* [64] */
    float *gw2;     /* [64][64] or NULL */
    float *gbn2_w;  /* [64] or NULL */
    float *gbn2_b;  /* [64] or NULL */
} dgcnn_edgeconv_grads;

size_t dgcnn_edgeconv_saved_bytes(const dgcnn_edgeconv_desc *d);
size_t dgcnn_edgeconv_fwd_workspace_bytes(const dgcnn_edgeconv_desc *d);
size_t dgcnn_edgeconv_bwd_workspace_bytes(const dgcnn_edgeconv_desc *d);

int dgcnn_edgeconv_fwd(const dgcnn_edgeconv_desc *d, const float *x, const int32_t *idx, float *y,
                       void *saved, void *workspace, size_t workspace_bytes, void *stream);

/* gy: dL/dy point-major [B][N][64].  Every gradient is OVERWRITTEN (not accumulated).  The backward is
 * deterministic: the scatter-add of autograd's IndexBackward0 (model.py:74) runs as a gather over reverse
 * neighbour lists sorted by edge id, every reduction has a fixed association, no floating-point atomics:
 * two calls on the same inputs return the same bits (provided no row of idx repeats a neighbour -- a kNN
 * graph never does).  One qualification: in the one-stage block the few max-over-k winners of a destination's
 * in-edges meet in LDS through float adds of ONE wave instruction on one address; the library relies on the LDS unit
 * of gfx950 resolving those in a fixed lane order (a property of this hardware, not an ISA guarantee; held by
 * tests/test_edgeconv_gpu.py::test_backward_is_bit_stable on every run).  N <= 40000. */
int dgcnn_edgeconv_bwd(const dgcnn_edgeconv_desc *d, const float *x, const int32_t *idx,
                       const float *y, const float *gy, const void *saved,
                       const dgcnn_edgeconv_grads *g, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---- segmentation head: BatchNorm1d + LeakyReLU on point-major rows ("next") ---------
 * Replaces the BatchNorm1d + LeakyReLU(slope) of conv6 / conv7 / conv8 (model.py:132-139, applied at :172,:184,
 * :187) for x = the 1x1 convolution's output as rows [P][C] (C a multiple of 4).  The convolutions themselves stay
 * plain library GEMMs.  stats [4][C] (mean, rstd, gamma*rstd, beta - mean*gamma*rstd) is written by the forward and
 * read by the backward.  BatchNorm semantics as in dgcnn_edgeconv_*. */
size_t dgcnn_bn_lrelu_workspace_bytes(int P, int C);
int dgcnn_bn_lrelu_fwd(const float *x, int P, int C, const dgcnn_bn_params *bn, int training, float eps,
                       float momentum, float slope, float *y, float *stats, void *workspace,
                       size_t workspace_bytes, void *stream);
int dgcnn_bn_lrelu_bwd(const float *x, const float *y, const float *gy, const float *stats, int P, int C,
                       int training, float slope, float *gx, float *gweight, float *gbias, void *workspace,
                       size_t workspace_bytes, void *stream);

/* The same BatchNorm1d + LeakyReLU followed by the maximum over the N points of each batch element: conv6 -> bn6 ->
 * LeakyReLU -> max (model.py:172-175) without writing the activated [B*N][C] tensor.  x rows [B*N][C] (point-major,
 * batch element b = rows b*N .. b*N+N-1), g [B][C], arg [B][C] = the winning point of every (b, column) (first
 * point attaining the extreme), stats as above.  The backward takes gg = dL/dg [B][C] and returns dL/dx dense. */
size_t dgcnn_bn_lrelu_max_workspace_bytes(int B, int N, int C);
int dgcnn_bn_lrelu_max_fwd(const float *x, int B, int N, int C, const dgcnn_bn_params *bn, int training, float eps,
                           float momentum, float slope, float *g, int32_t *arg, float *stats, void *workspace,
                           size_t workspace_bytes, void *stream);
int dgcnn_bn_lrelu_max_bwd(const float *x, const float *gg, const int32_t *arg, const float *stats, int B, int N, int C,
                           int training, float slope, float *gx, float *gweight, float *gbias, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Weight gradient of a 1x1 convolution on point-major rows: dw[M][N] = sum over the P rows of a[p][m] * b[p][n]
 * (a = dL/d(output) rows with leading dimension lda >= M, b = the convolution's input rows, ldb >= N).  The
 * contraction over the points is split across workgroups and folded in a fixed order.  With 16-byte aligned rows and
 * M, N, lda, ldb multiples of 4 it runs on the fp16 matrix cores with split operands, like dgcnn_rows_gemm (amax_a /
 * amax_b: the operands' dgcnn_rows_absmax words, or NULL to have them found here); otherwise on fp32 MFMA. */
size_t dgcnn_rows_wgrad_workspace_bytes(int M, int N, int P);
int dgcnn_rows_wgrad(const float *a, int lda, int M, const float *b, int ldb, int N, int P, const uint32_t *amax_a,
                     const uint32_t *amax_b, float *dw, void *workspace, size_t workspace_bytes, void *stream);

/* A 1x1 convolution on point-major rows, y[P][N] = x[P][K] . w[N][K]^T (+ add[row / rows_per_add][N] when add != NULL:
 * conv7's share of the pooled feature, one row per batch element, model.py:179-184): the forward and input-gradient
 * products of conv6 .. conv8 (model.py:172, :184, :187; for dL/dx pass w = W^T).  fp32 in / fp32 out on the fp16 matrix
 * cores with both operands split in two halves at an exact power-of-two scale (three MFMAs per step, fp32 accumulate;
 * one scale per TENSOR: an element is represented to 2^-23 relative while its low half stays a normal fp16 number --
 * i.e. down to 2^-14 of the tensor's largest entry -- and to 2^-39 of that largest entry, absolute, below: rows far
 * smaller than the largest lose relative accuracy; dgcnn_head_tail_* scales gradient operands per 128-row tile
 * instead) -- see head_gemm.hip.  amax_x / amax_w: the operands' dgcnn_rows_absmax words.  x rows
 * have leading dimension ldx, w rows ldw, y rows ldy; K a multiple of 32, ldx and ldw multiples of 4, x and w 16-byte
 * aligned. */
int dgcnn_rows_gemm(const float *x, int ldx, const float *w, int ldw, int P, int K, int N, const uint32_t *amax_x,
                    const uint32_t *amax_w, const float *add, int rows_per_add, float *y, int ldy, void *stream);
/* amax[0] <- the bits of max |x[r][c]| over a [rows][cols] array with leading dimension ld: the operand magnitude the
 * two products above take their power-of-two scale from.  One word per operand; a tensor used by several products
 * (dL/dy in the input and weight gradients, a layer's input in forward and weight gradient) is measured once. */
int dgcnn_rows_absmax(const float *x, int ld, int rows, int cols, uint32_t *amax, void *stream);

/* conv9, the classifier (model.py:141 applied at :191): y[P][NC] = x[P][K] . w[NC][K]^T for NC <= 8 classes as a
 * streaming pass (too few columns for a matrix tile), and its backward in ONE pass over x and gy: gx[P][K] = gy . w
 * (may be NULL), gw[NC][K] = gy^T x (may be NULL; per-workgroup records added in a fixed order).  K a multiple of 4,
 * 16-byte aligned operands. */
int dgcnn_cls_fwd(const float *x, const float *w, int P, int K, int NC, float *y, void *stream);
size_t dgcnn_cls_bwd_workspace_bytes(int P, int K, int NC);
int dgcnn_cls_bwd(const float *x, const float *w, const float *gy, int P, int K, int NC, float *gx, float *gw,
                  void *workspace, size_t workspace_bytes, void *stream);

/* conv6 -> bn6 -> LeakyReLU -> max over the points (model.py:172-175) in one pass over cat [B*N][F] (leading dimension
 * ldc): the [B*N][C] product is never written -- the matrix kernel's epilogue keeps the column sums (BatchNorm1d batch
 * statistics; running statistics updated when training) and, per (batch element, column), the extreme that survives the
 * max (BatchNorm + LeakyReLU is monotone per column) with the first point attaining it.  Outputs: g [B][C], arg [B][C],
 * uwin [B][C] = the pre-activation at the winner (what dgcnn_embed_max_bwd_prepare needs), stats [4][C].  amax_cat /
 * amax_w: the operands' dgcnn_rows_absmax words.  F a multiple of 32, ldc / ldw multiples of 4, 16-byte aligned rows. */
size_t dgcnn_embed_max_fwd_workspace_bytes(int B, int N, int C);
int dgcnn_embed_max_fwd(const float *cat, int ldc, const float *w, int ldw, int B, int N, int F, int C,
                        const uint32_t *amax_cat, const uint32_t *amax_w, const dgcnn_bn_params *bn, int training, float eps,
                        float momentum, float slope, float *g, int32_t *arg, float *uwin, float *stats, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The same pass and its backward as ONE call each (round 3; head_fused.hip): the forward splits conv6's weight once into
 * fp16 planes scaled per output channel (no operand-magnitude pass over w); the backward runs the whole of ops._EmbedMax's
 * algebra -- with u = cat . W^T the BatchNorm mean terms of dL/du are affine in u, dL/du[p][c] = A_c + B_c u[p][c] + (s[b][c]
 * at the winning point), hence  dcat = 1 (W^T A)^T + cat (W^T diag(B) W) + [winners],  dW = A (sum_p cat)^T + diag(B) W
 * (cat^T cat) + [winners]: two F x F products over the points instead of two C-wide ones -- in eight launches without a
 * library call.  catsum [F] = the column sums of cat (dgcnn_rows_stats; training mode only), gg = dL/dg [B][C]; dcat [B N][F]
 * and dw [C][F] are OVERWRITTEN, contiguous.  N >= 128, F a multiple of 32 (<= 256), C a multiple of 4. */
size_t dgcnn_embed_fwd_workspace_bytes(int B, int N, int F, int C);
int dgcnn_embed_fwd(const float *cat, int ldc, const float *w, int ldw, int B, int N, int F, int C, const uint32_t *amax_cat,
                    const dgcnn_bn_params *bn, int training, float eps, float momentum, float slope, float *g, int32_t *arg,
                    float *uwin, float *stats, void *workspace, size_t workspace_bytes, void *stream);
size_t dgcnn_embed_bwd_workspace_bytes(int B, int N, int F, int C);
int dgcnn_embed_bwd(const float *cat, int ldc, const uint32_t *amax_cat, const float *catsum, const float *w, int ldw,
                    const int32_t *arg, const float *uwin, const float *stats, const float *gg, int B, int N, int F, int C,
                    int training, float slope, float *dcat, int lddcat, float *dw, float *gbn_w, float *gbn_b, void *workspace,
                    size_t workspace_bytes, void *stream);
/* amax[0] <- the bits of max |x| (as dgcnn_rows_absmax) and colsum[c] <- sum over the rows of x[r][c], in one pass over a
 * [rows][cols] array (cols a multiple of 4, <= 256; fp64 records folded in a fixed order). */
size_t dgcnn_rows_stats_workspace_bytes(int rows, int cols);
int dgcnn_rows_stats(const float *x, int ld, int rows, int cols, uint32_t *amax, float *colsum, void *workspace,
                     size_t workspace_bytes, void *stream);

/* Backward of conv6 -> bn6 -> LeakyReLU -> max without the dense [B*N][C] gradient (model.py:172-175; host side in
 * ops.py, _EmbedMax).  With u = cat . W^T the BatchNorm mean terms of dL/du are affine in u, dL/du[p][c] = A_c +
 * B_c u[p][c] + (s[b][c] at the winning point), so the backward needs 192-wide library products only.
 * prepare: uwin [B][C] = u at the winners, gg = dL/dg [B][C], stats of the forward -> gweight, gbias [C] (BatchNorm
 * gradients), coef_a, coef_b [C] (zero in eval mode), s [B][C].
 * scatter: dcat[b*N + arg[b][c]][f] += s[b][c] * w[c][f] for every (b, c), w [C][F]; race-free, fixed order. */
int dgcnn_embed_max_bwd_prepare(const float *uwin, const float *gg, const float *stats, int B, int N, int C, int training,
                                float slope, float *gweight, float *gbias, float *coef_a, float *coef_b, float *s,
                                void *stream);
int dgcnn_embed_max_bwd_scatter(const int32_t *arg, const float *s, const float *w, int B, int N, int C, int F,
                                float *dcat, void *stream);

/* ---- the tail of the head in one call (conv7 .. conv9) --------------------------------------------
 * Replaces model.py:179-191 on point-major rows, for cat = [x1 | x2 | x3] rows [P = B N][F] (leading dimension ldcat) and
 * add7 [B][C7] = W7[:, :emb] . g_b, conv7's share of the pooled feature (the repeat / cat of :179-180 never exists):
 *     u7 = cat . W7c^T + add7[b];  z7 = LReLU(BN7(u7));  u8 = z7 . W8^T;  z8 = LReLU(BN8(u8));
 *     logits [P][NC] = dropout(z8, p_drop) . W9^T                                    (dropout only when training)
 * Only u7 and u8 are written: BatchNorm statistics are the products' epilogues, BatchNorm + LeakyReLU (+ dropout) are
 * applied where the next product loads its operand, and the backward forms every BatchNorm input gradient the same way
 * (head_fused.hip).  Products on the fp16 matrix cores with split operands, fp32 accumulate: weights scaled per output
 * channel, gradient operands per 128-row tile, activations per tensor (exact maximum from the column extrema) -- an
 * operand element is represented to 2^-22 relative (2^-40 of its tile's / row's largest entry where the low half
 * leaves fp16's normal range).
 * w7c: conv7.weight[:, emb:] as a [C7][F] view with row stride ldw7; w8 [C8][C7]; w9 [NC][C8].  BatchNorm semantics as
 * dgcnn_edgeconv_* (num_batches_tracked is the caller's business here).  amax_cat: the dgcnn_rows_absmax word of cat.
 * Dropout: element (p, k) is kept when dgcnn_dropout_mask says so for (seed, p_drop) -- a counter-based generator, the
 * backward regenerates the mask from the seed; survivors are scaled by the reciprocal of the realised keep probability
 * (a multiple of 2^-16).
 * Limits: F, C7, C8 multiples of 32, C7 and C8 <= 1024, NC <= 8, N >= 128, B <= 64, B N < 2^23; 16-byte aligned rows.
 * saved: opaque, forward -> backward (u7, u8, the statistics, this step's weight planes).  Every gradient is
 * OVERWRITTEN; all reductions have a fixed association (bit-stable). */
typedef struct dgcnn_head_tail_desc {
    int32_t B, N;           /* P = B N rows; add7 row b belongs to rows b N .. b N + N - 1 */
    int32_t F, C7, C8, NC;  /* 192, 512, 256, classes in the reference */
    int32_t ldcat, ldw7;
    int32_t training;       /* 1: batch statistics (+ running update) and dropout; 0: running statistics, no dropout */
    float eps7, eps8, momentum7, momentum8, slope, p_drop;
    uint64_t seed;          /* of the dropout mask of this step */
    const float *w7c;
    dgcnn_bn_params bn7;    /* [C7] */
    const float *w8;
    dgcnn_bn_params bn8;    /* [C8] */
    const float *w9;
} dgcnn_head_tail_desc;

typedef struct dgcnn_head_tail_grads {
    float *gcat;   /* [P][F] rows with leading dimension ldgcat */
    int64_t ldgcat;
    float *gadd7;  /* [B][C7] */
    float *gw7c;   /* [C7][F] dense */
    float *gbn7_w, *gbn7_b;
    float *gw8;    /* [C8][C7] */
    float *gbn8_w, *gbn8_b;
    float *gw9;    /* [NC][C8] */
} dgcnn_head_tail_grads;

size_t dgcnn_head_tail_saved_bytes(const dgcnn_head_tail_desc *d);
size_t dgcnn_head_tail_fwd_workspace_bytes(const dgcnn_head_tail_desc *d);
size_t dgcnn_head_tail_bwd_workspace_bytes(const dgcnn_head_tail_desc *d);
int dgcnn_head_tail_fwd(const dgcnn_head_tail_desc *d, const float *cat, const uint32_t *amax_cat, const float *add7,
                        float *logits, void *saved, void *workspace, size_t workspace_bytes, void *stream);
/* glogits: dL/dlogits [P][NC] */
int dgcnn_head_tail_bwd(const dgcnn_head_tail_desc *d, const float *cat, const uint32_t *amax_cat, const float *glogits,
                        const void *saved, const dgcnn_head_tail_grads *g, void *workspace, size_t workspace_bytes,
                        void *stream);
/* mask [P][K] (K a multiple of 4) <- 1 where dropout keeps the element, 0 where it drops it: the generator of
 * dgcnn_head_tail_* (tests; a caller that wants the mask of a step). */
int dgcnn_dropout_mask(uint64_t seed, int P, int K, float p_drop, uint8_t *mask, void *stream);

/* ---- the whole network as one forward and one backward call (DGCNN.forward at depth 10, model.py:144-193) -------------
 * What a caller of the reference's training / inference loops needs per step: three kNN builds, three EdgeConv blocks,
 * conv6 -> bn6 -> max over the points, the pooled feature's share of conv7, and the head's tail, enqueued by ONE call each
 * way on the caller's stream.  Compared with calling the per-layer entry points from the host language this removes every
 * framework op between them: the blocks write x1 | x2 | x3 into ONE [P][192] buffer and read their inputs from its column
 * slices (no torch.cat, model.py:167), the gradients that meet in x1 / x2 / x3 (block b + 1's edge features, the cat of :167
 * and the cat of :180) are accumulated where they are produced, the pooled-feature product W7[:, :emb] . g_b (model.py:179)
 * and its three backward products are kernels of this library, and the eight `num_batches_tracked` counters are bumped by
 * kernels that run anyway.  Results are those of the per-layer entry points (the same kernels run).
 *
 * w[i] = conv(i+1).weight: [64][2F], [64][64], [64][128], [64][64], [64][128], [emb][192], [c7][emb + 192], [c8][c7],
 * [num_classes][c8], all dense.  bn[i] = bn(i+1) (64 x5, emb, c7, c8 channels); eps / momentum per BatchNorm (the two of a
 * block share eps[first], momentum[first]).  idx[b]: int32 [B][N][k] of block b + 1 -- written by the forward (idx_given = 0)
 * or supplied by the caller (idx_given = 1: the test hook that separates kNN parity from edge-MLP parity); the backward
 * reads the same three arrays.  x: [B][F][N] by element strides; logits / glogits: [B][num_classes][N] by element strides.
 * Limits: F <= 64, k <= 64, N >= 128, emb a multiple of 32, c7, c8 multiples of 32 (<= 1024), 1 <= num_classes <= 8,
 * B <= 64, B N < 2^23 (dgcnn_model_*_bytes return 0 and dgcnn_last_error says which).  The network input gets no gradient. */
typedef struct dgcnn_model_desc {
    int32_t B, N, F, k;
    int32_t num_classes, emb, c7, c8;
    int32_t training;   /* 1: batch statistics (+ running update, counters) and dropout; 0: running statistics */
    int32_t idx_given;
    float slope, p_drop;
    uint64_t seed;      /* of the dropout mask of this step (dgcnn_dropout_mask) */
    const float *w[9];
    dgcnn_bn_params bn[8];
    float eps[8], momentum[8];
    int32_t *idx[3];
} dgcnn_model_desc;

typedef struct dgcnn_model_grads {
    float *gw[9];       /* shapes of w[] */
    float *gbn_w[8];
    float *gbn_b[8];
} dgcnn_model_grads;

size_t dgcnn_model_saved_bytes(const dgcnn_model_desc *d);
size_t dgcnn_model_fwd_workspace_bytes(const dgcnn_model_desc *d);
size_t dgcnn_model_bwd_workspace_bytes(const dgcnn_model_desc *d);
int dgcnn_model_fwd(const dgcnn_model_desc *d, const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn, float *logits,
                    int64_t l_sb, int64_t l_sc, int64_t l_sn, void *saved, void *workspace, size_t workspace_bytes,
                    void *stream);
/* x: the forward's input again (read in place when its strides are those of point-major memory -- (N F, 1, F), the permuted
 * view train.py:176 hands over -- otherwise the forward's own copy in `saved` is used).  Every gradient is OVERWRITTEN;
 * deterministic as dgcnn_edgeconv_bwd is. */
int dgcnn_model_bwd(const dgcnn_model_desc *d, const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn, const float *glogits,
                    int64_t g_sb, int64_t g_sc, int64_t g_sn, const void *saved, const dgcnn_model_grads *g, void *workspace,
                    size_t workspace_bytes, int phases, void *stream);
/* phases = 3: the whole backward.  phases = 1 then phases = 2 on the SAME workspace (untouched in between): first the head --
 * after it the gradients of conv6 .. conv9 and bn6 .. bn8 are final (97 % of the model's gradient bytes), so a data-parallel
 * caller can start their all-reduce on another stream --, then the three EdgeConv blocks. */

/* ---- the steps either side of DGCNN.forward in the reference's loops ("next") -------
 * Logits are addressed by element strides, (b, c, n) at b*stride_b + c*stride_c + n*stride_n, so both the
 * [B, classes, N] tensor DGCNN.forward returns and a point-major [B, N, classes] buffer are read in place.
 *
 * dgcnn_seg_loss: one training / validation loss evaluation of the reference -- softmax over the classes
 * (train.py:180, :267), cal_loss(pred = the PROBABILITIES, gold, smoothing) = label-smoothed cross entropy through a
 * second log_softmax (util.py:17-34; smoothing = 0.2 there, 0 gives plain cross entropy) averaged over all B*N
 * points, and the arg-max prediction (train.py:201, :271).  mask (optional, uint8 [B*N]) selects the points that
 * take part, as train.py:182-196 does it: rows get one extra column, (p, 0) with the point's label where mask != 0
 * and (0, ..., 0, 1) with label `classes` elsewhere.  loss: one float; grad_logits (same strides as logits) =
 * d loss / d logits, computed in the same pass; pred int64 [B*N].  classes <= 32.  A label outside [0, classes)
 * (where the reference's scatter raises) makes the loss NaN. */
size_t dgcnn_seg_loss_workspace_bytes(int B, int N);
int dgcnn_seg_loss(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                   int64_t stride_n, const int64_t *labels, const uint8_t *mask, float smoothing, float *loss,
                   float *grad_logits, int64_t *pred, void *workspace, size_t workspace_bytes, void *stream);
/* the same with the labels as the reference's loader hands them to the loop: float64 [B*N] (data.py:169-182, train.py:177;
 * labels_f64 = 1) -- no conversion pass in front of the loss -- or int64 (labels_f64 = 0) */
int dgcnn_seg_loss_labels(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                          int64_t stride_n, const void *labels, int labels_f64, const uint8_t *mask, float smoothing,
                          float *loss, float *grad_logits, int64_t *pred, void *workspace, size_t workspace_bytes,
                          void *stream);

/* Inference post-step (test.py:123-127, :308-311): softmax over the classes, maximum and arg-max; points whose
 * maximum probability is below min_confidence get the label `unclassified` (31 in the reference).  pred int64
 * [B*N]; prob (optional) fp32 [B*N] = the maximum probability. */
int dgcnn_seg_predict(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                      int64_t stride_n, float min_confidence, int unclassified, int64_t *pred, float *prob,
                      void *stream);

/* One torch.optim.Adam step (train.py:118: lr, betas, eps, L2 weight_decay added to the gradient, no amsgrad) on flat
 * fp32 buffers of n elements; step counts from 1. */
int dgcnn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, void *stream);

/* float64 -> float32 of a point block (the `.float()` of train.py:176 / test.py:119 on the device). */
int dgcnn_cast_f64_f32(const double *x, size_t n, float *y, void *stream);

/* ---- exchange step: RCCL behind the C ABI -----------------------------------------------------------
 * One process per GPU; batch elements are sharded across the ranks (independent units: no collective in the data path)
 * and a training step ends with ONE all-reduce of the flat fp32 gradient buffer (980 864 floats = 3.9 MB for the
 * reference's model, train.py:118,198-199); the reference itself has no distributed code (train.py:110-111).
 * Bootstrap: rank 0 makes an id (dgcnn_comm_unique_id, DGCNN_COMM_ID_BYTES host bytes) and hands it to the other ranks over
 * any side channel (the launcher's store, a file, MPI ...); then EVERY rank calls dgcnn_comm_init with the same id on its
 * own current HIP device (a collective call).  buf is a device pointer; the collectives are enqueued on `stream` and
 * operate in place; average != 0 divides the sum by the number of ranks inside the collective (ncclAvg).
 * RCCL is bound at run time (dlopen of librccl.so.1: the copy the process already holds, e.g. PyTorch's, else ROCm's):
 * the library links without it and these calls fail with a message where it is absent. */
#define DGCNN_COMM_ID_BYTES 128
typedef struct dgcnn_comm dgcnn_comm;
int dgcnn_comm_rccl_version(void);  /* NCCL_VERSION_CODE of the bound library, -1 if it cannot be loaded */
int dgcnn_comm_unique_id(uint8_t *id);
int dgcnn_comm_init(dgcnn_comm **comm, const uint8_t *id, int world, int rank);
int dgcnn_comm_destroy(dgcnn_comm *comm);
int dgcnn_comm_world(const dgcnn_comm *comm);
int dgcnn_comm_rank(const dgcnn_comm *comm);
int dgcnn_comm_allreduce_f32(dgcnn_comm *comm, float *buf, size_t count, int average, void *stream);
int dgcnn_comm_broadcast_f32(dgcnn_comm *comm, float *buf, size_t count, int root, void *stream);

/* ---- measurement hook (bench.py) ----------------------------------------------------------------------
 * HIP events around the large kernels of the path, recorded on the stream each kernel is launched on.  Regions
 * (dgcnn_profile_region_name): the fp16 threshold pass, the fp16 filter and the exact refine of dgcnn_knn, the
 * forward gather/reduce pass, the backward reverse gather and the second convolution's two kernels of dgcnn_edgeconv_*,
 * and the seven products of the head (dgcnn_embed_fwd, dgcnn_head_tail_fwd / _bwd).  enable(1) resets the sample
 * rings; mean_ms(region) returns the mean time of the samples recorded since (call it after the stream has been
 * synchronised; at most 256 samples per region are kept).  enable(0) stops recording and keeps the samples,
 * resume() continues without resetting them (bench.py samples every few steps: an event record costs a few
 * microseconds of GPU time).  No effect on results. */
int dgcnn_profile_enable(int on);
int dgcnn_profile_resume(void);
int dgcnn_profile_regions(void);
const char *dgcnn_profile_region_name(int region);
float dgcnn_profile_mean_ms(int region);
int dgcnn_profile_samples(int region);

#ifdef __cplusplus
}
#endif
#endif /* DGCNN_HIP_H */
