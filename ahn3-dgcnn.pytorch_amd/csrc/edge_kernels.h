// edge_kernels.h -- host-side launchers of the EdgeConv kernels (one translation unit each).
#pragma once
#include "edge_common.h"
#include "fold.h"

namespace dgcnn {

// ---- pointgemm.hip : per-point fp32 MFMA contractions --------------------------------------
// forward per-point product of a block's first convolution w1 [64][2C]:
//   W1 . [x_j - x_i ; x_i] = Wa . x_j + (Wb - Wa) . x_i,  Wcat = [Wa ; Wb - Wa]:  TP = x Wa^T, TQ = x (Wb - Wa)^T.
// Every workgroup builds Wcat^T in LDS from w1; workgroup 0 also leaves wcat [128][C] for the backward's dgrad,
// zeroes zero_words[0 .. nzero) (the fold.h counters of the statistics passes behind it) and, in eval mode (nbn = 1 or 2),
// writes the BatchNorm affine pairs of the block's layers from their running statistics (ST_* layout of edge_common.h).
// the counting pass of the reverse neighbour lists as extra workgroups of the forward's per-point product (see rev_count_rider)
// (training forward of the 64-channel blocks: the product is bound by its matrix instructions, the counting pass by its LDS
// atomics -- 23.7 + 17.0 us as two launches, 27.0 us as one at B = 8, N = 7000, k = 30); the rider's first workgroup also zeroes
// zero_words[0 .. nzero): the counters of the BACKWARD call that will use the counts
struct RevCountRider {
    const int32_t *idx;
    int N, k, chunks, rows_per, nblocks, main_blocks;
    int *cnt;
    int *zero_words;
    int nzero;
};
// does launch_rowgemm_fwd carry the rider for this shape?  (the table of a chunk lives in the product's 32 KB of LDS)
bool rowgemm_fwd_carries_count(int C, int N, int k);
struct FwdSide {
    RevCountRider rc;
    float *wcat;
    int *zero_words;
    int nzero;
    int nbn;
    float eps;
    dgcnn_bn_params bn[2];
    float *stats[2];
};
// row strides (in floats) of the operands that may be column slices of a wider row (the model's [P][192] concatenation of
// the block outputs): A0's rows, Y0's rows, and whether Y0 is added to instead of overwritten (the second gradient that flows
// into a block's output)
struct RowLd {
    int a0, y0, y0_acc;
};
// ldx: row stride of x (0 = dense, C)
void launch_rowgemm_fwd(const float *x, int C, const float *w1, float *TP, float *TQ, const FwdSide &sd, int P, hipStream_t st,
                        int ldx = 0);
// Y(p, n) = sum_k A(p, k) W[k][n];  A = [A0 | A1] (K0 + K1 columns), Y = [Y0 | Y1] (NO0 + rest)
// ldy0: row stride of Y0 (0 = dense, NO0); y0_accumulate: Y0 += instead of =
void launch_rowgemm(const float *A0, int K0, const float *A1, int K1, const float *W, int NO, float *Y0, int NO0,
                    float *Y1, int P, hipStream_t st, int ldy0 = 0, int y0_accumulate = 0);
// partial[chunk][m][n] = sum_{p in chunk} A(p, m) X[p][n];  A = [A0 | A1] (M0 + M1 <= 128 columns)
int wgrad_chunks(int P, int C);
void launch_wgrad(const float *A0, int M0, const float *A1, int M1, const float *X, int C, int P, float *partial,
                  hipStream_t st, int ldx = 0);
// G [128][C] = sum of the partials, then gw1[o][c]:  c < C: G[o][c] - G[64 + o][c];  c >= C: G[64 + o][c - C]
void launch_finish_dw1(const float *partial, int nchunk, int C, float *gw1, hipStream_t st, int *zero_words = nullptr, int nzero = 0);
// that fold as extra workgroups of the input-gradient product's launch
struct Dw1Rider {
    const float *partial;
    int nchunk, C;
    float *gw1;
    int main_blocks;
    int *zero_words;  // counters the call leaves at zero behind itself (its last launch), or NULL
    int nzero;
    // INSTEAD of the fold: the weight-gradient product itself rides (wgrad_split64_body; x, its row stride, the partials)
    const float *wg_x;
    int wg_ldx;
    float *wg_partial;
};
bool dgrad_wgrad64_fused(int C);
void launch_dgrad_wgrad64(const float *dR, const float *dD, const float *wcat, float *gx, int ldgx, int gx_acc, int P, const float *x,
                          int ldx, float *partial, float *gw1, hipStream_t st, int *zero_words = nullptr, int nzero = 0);
void launch_dgrad_and_dw1(const float *dR, const float *dD, const float *wcat, int C, float *gx, int ldgx, int gx_acc, int P,
                          const float *partial, int nchunk, float *gw1, hipStream_t st, int *zero_words = nullptr, int nzero = 0);
// out[e] = sum_chunk partial[chunk][e]   (fixed order: deterministic)
void launch_sum_partials(const float *partial, int nchunk, int n, float *out, hipStream_t st);

// ---- edgeconv.hip : the block with row strides (0 = dense) -- what dgcnn_edgeconv_fwd / _bwd run with all strides dense.
// x, y, gy, gx: row strides in floats of the block's input, output, output gradient and input gradient when those are
// column slices of wider rows; gx_acc: the input gradient is added to its slice instead of overwriting it.  The second
// convolution's backward takes gy == y == 64 or 192 only.
// gy2: dL/dy = gy + gy2 with gy2 dense [P][64] (added by the first kernel of the backward)
// no_fwd_lists: the forward does not run the counting pass of the reverse lists and the backward does not expect its result
// (the model builds the lists of its three blocks in one go where that is the faster way)
struct EdgeLd {
    int x, y, gy, gx, gx_acc;
    const float *gy2;
    int no_fwd_lists = 0;
};
int edgeconv_fwd_ld(const dgcnn_edgeconv_desc *d, const EdgeLd &ld, const float *x, const int32_t *idx, float *y, void *saved,
                    void *workspace, size_t workspace_bytes, void *stream);
// counters of a backward call that must be zero when its kernels start (zeroed by the first kernel of the call -- the counting
// pass of the reverse lists -- or, with RevPrebuilt, by launch_rev_build_merged)
enum {
    Z_FOLD_A = 0,
    Z_FOLD_B = FOLD_CTR_WORDS,
    Z_FOLD_C = 2 * FOLD_CTR_WORDS,   // the fp32 fallback of edge_mlp_bwd16_kernel
    Z_GMAX = 3 * FOLD_CTR_WORDS,     // bits of max |dL/dy . LReLU'| (bwd_point -> edge_mlp_bwd16)
    Z_GUARD = Z_GMAX + 1,            // raised by edge_mlp_bwd16_kernel when an operand leaves the fp16 range
    Z_ALLOC = 3 * FOLD_CTR_WORDS + 8,  // [0] reverse-list space, [1] extra (point, segment) pairs.  LAST: where the forward has
                                     // allotted the lists, the backward's last launch zeroes the words in front of this pair only
    Z_WORDS = 3 * FOLD_CTR_WORDS + 16
};
// reverse lists built outside the call (the model builds those of its three blocks in one go): the call's Z_WORDS counters
// (zeroed, the list allocation at Z_ALLOC filled in) and the arrays of THIS graph
struct RevPrebuilt {
    int *zero, *indeg, *rowptr, *segbase, *pcnt, *rev, *extra;
};
// does the training forward of this block prepare the backward's reverse lists (edgeconv.hip, lists_in_forward)?
bool edgeconv_lists_in_forward(const dgcnn_edgeconv_desc *d);
int edgeconv_bwd_ld(const dgcnn_edgeconv_desc *d, const EdgeLd &ld, const float *x, const int32_t *idx, const float *y,
                    const float *gy, const void *saved, const dgcnn_edgeconv_grads *g, void *workspace, size_t workspace_bytes,
                    void *stream, const RevPrebuilt *pre = nullptr);

// ---- knn.hip : dgcnn_knn over rows with a stride (ldx = 0: dense) -------------------------------------------------
int knn_ld(const float *x, int ldx, int B, int N, int C, int k, int32_t *idx, void *workspace, size_t workspace_bytes,
           void *stream);

// ---- edge_stream.hip : gather / reduce passes over the edges (no MFMA) -------------------------
int edge_grid(int P);
// stats of u = TP[n(i,j)] + TQ[i] over all edges and, if usel != NULL, the per (point, channel) extreme of u that
// survives max-over-k (max if gamma >= 0, min otherwise); usum[i] = sum_j u[i][j]; hitj[i] = slot j of the first edge
// attaining the extreme (with usel only).  fin != NULL (training): the pass finishes the BatchNorm batch statistics
// itself (fold.h; fs.ctr must be zero at launch) -> fin->stats, running statistics updated.
struct RevSlotsArgs;
void launch_edge_reduce(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *gamma,
                        float *usel, float *usum, uint8_t *hitj, const FoldSet &fs, const BnBatchStats *fin, hipStream_t st,
                        const struct RevSlotsArgs *slots = nullptr);
// y = LeakyReLU(sc * usel + sh)
void launch_edge_out(const float *usel, const float *stats, float slope, int P, float *y, hipStream_t st, int ldy = 0);
// last-stage BN backward sums from per-point quantities, finished in the same launch: gbn_w, gbn_b and cc = [c1 | c2]
// gmax_bits (zero at launch): receives the bits of max |gy . LReLU'| (an atomic max: order-independent)
struct RevSlotsArgs;
struct RevPlaceArgs;
void launch_bwd_point(const float *gy, const float *usel, const float *stats, float slope, int P, double count,
                      int training, const FoldSet &fs, float *gbn_w, float *gbn_b, float *cc, unsigned *gmax_bits,
                      const struct RevSlotsArgs *slots, hipStream_t st, int ldgy, const float *gy2, float *gdv,
                      const struct RevPlaceArgs *place = nullptr);
// Reverse adjacency of the neighbour graph (depends on idx only), every list sorted by edge id.  Scratch: cnt
// [B * rev_chunks][N]; results: rowptr [P] (list start), indeg [P], segbase [P], pcnt [P] (zeroed), rev [E], alloc [2]
// (alloc[1] = number of extra (point, segment) pairs), extra [2 (E / REV_SEG + 1)] ints.  The first kernel also zeroes
// zero_words[0 .. nzero) (which must contain alloc).
int rev_chunks(int B, int N, int k);
constexpr int REV_RIDER_MAX_N = 8192;  // the list passes that ride in other kernels' launches keep their table in 32 KB of static LDS
constexpr int REV_MAX_N = 40000;  // points per batch element: the list kernels keep an N-int table per wave in LDS
                                  // (160 KB per CU); every backward builds the lists, so this bounds N there
// with_slots / with_place = false: the caller runs those passes itself -- the list-space pass inside launch_bwd_point's
// launch (rev_slots_args), the placement with launch_rev_place.
struct RevSlotsArgs {
    int N, chunks, P;
    int *cnt, *rowptr, *indeg, *segbase, *pcnt, *alloc;
    int2 *extra;
    // a merged build of nblk graphs of Pblk points each (launch_rev_build_merged): alloc / extra of graph g at
    // alloc + g alloc_stride ints, extra + g extra_stride pairs; one graph: nblk = 1, Pblk = P
    int nblk, Pblk, alloc_stride, extra_stride;
    int *cnt_out = nullptr;  // where the first slots go (NULL: over the counts, in place).  The counts of a training forward stay as
                             // they are when the list-space pass runs in the backward: a second backward reads them again
};
RevSlotsArgs rev_slots_args(int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase, int *pcnt, int *alloc,
                            int *extra);
int launch_rev_build(const int32_t *idx, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase, int *pcnt,
                     int32_t *rev, int *alloc, int *extra, int *zero_words, int nzero, bool with_slots, bool with_place,
                     hipStream_t st);
// the second and third pass alone, from counts that another call left (read only): the first slots go to `slots` [B rev_chunks][N]
int launch_rev_slots_place(const int32_t *idx, int B, int N, int k, const int *counts, int *slots, int *indeg, int *rowptr, int *segbase,
                           int *pcnt, int32_t *rev, int *alloc, int *extra, hipStream_t st);
int launch_rev_place(const int32_t *idx, int B, int N, int k, const int *cnt, int32_t *rev, hipStream_t st);
struct RevPlaceArgs {
    const int32_t *idx;
    int N, k, chunks, rows_per, nblocks;  // nblocks = B * chunks
    const int *cnt;
    int32_t *rev;
    int Bblk, Eblk;  // merged build: batch elements per graph and list entries per graph (one graph: Bblk = B, Eblk = 0)
};
int launch_rev_build_merged(const int32_t *idx, int nblk, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase,
                            int *pcnt, int32_t *rev, int *extra, int extra_stride, int *zero_words, int zero_stride, int alloc_off,
                            hipStream_t st);

// out[e] = sum over nchunk partials [nchunk][n] (chunk order), as extra workgroups of the per-point pass (dense form)
struct SumRider {
    const float *partial;
    int nchunk, n;
    float *out;
    int main_blocks;
};
// stage-1 backward as a gather over the reverse lists (see edge_stream.hip): dR and dD are overwritten.
// dv == NULL: sparse dv from (gdv = dL/dy . LReLU'(y) as launch_bwd_point leaves it, hitj) -- hits [E] x 32 bytes is scratch, G = gdv again (read where a hit record overflows; a per-point copy sc . gdv until round 4); else dense dv [E][64] with
// dvsum [P][64] = its sums over k.  part0 [P][128], partx [E / REV_SEG + 1][128] floats: segment partials of long lists.
// cc = 0 (eval mode) drops the BatchNorm mean terms.
int launch_edge_bwd(const float *TP, const float *TQ, int B, int N, int k, const float *stats, const float *cc, float slope,
                    const float *gdv, const uint8_t *hitj, const float *usum, const float *dv,
                    const float *dvsum, const int *rowptr, const int *indeg, const int *segbase, int *pcnt,
                    const int32_t *rev, const int *alloc, const int *extra, void *hits, float *G, float *part0,
                    float *partx, float *dR, float *dD, hipStream_t st, const struct SumRider *rider = nullptr);
void launch_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k, float *out,
                              hipStream_t st);
void launch_graph_feature_bwd(const float *gout, const int *rowptr, const int *indeg, const int32_t *rev, int B, int N,
                              int C, int k, float *gx, hipStream_t st);

// ---- edge_mlp.hip : per-edge 64 -> 64 contraction on MFMA (second stage of blocks 1, 2) ------
int mlp_grid(int P);      // backward: number of gW2 / BN1 partial records
int mlp_grid_fwd(int P);  // forward: number of BN2 partial records
// fin != NULL (training): BatchNorm-2 batch statistics finished in the same launch (fold.h).  guard: one int, zero at
// launch -- raised when |a1| or |W2| leaves the fp16 range, in which case the fp32 instance of the kernel (fold counters
// fs_fallback) redoes the pass in the same call
void launch_edge_mlp_fwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                         const float *stats1, const float *w2, const float *gamma2, float slope, float *usel2,
                         uint8_t *hit2, const FoldSet &fs, const FoldSet &fs_fallback, const BnBatchStats *fin, int *guard,
                         hipStream_t st);
// dv1 [E][64]; dvsum [P][64] = sum over each point's k edges of dv1; partial_dw2 [grid][64][64]; the BatchNorm-1
// backward sums (sum dv1, sum dv1 * uhat1) are finished in the same launch (fold.h) -> fin1
// gmax_bits: bits of max |dL/dy . LReLU'| (launch_bwd_point); guard: one int, zero at launch -- raised when an operand
// leaves the fp16 range, in which case the fp32 form of the pass (fold counters fs_fallback) redoes it in the same call
int launch_edge_mlp_bwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                         const float *stats1, const float *stats2, const float *cc2, const float *w2, float slope,
                         int training, const float *gdv, const uint8_t *hit2, const unsigned *gmax_bits,
                         float *dv1, float *dvsum, float *partial_dw2, const FoldSet &fs, const FoldSet &fs_fallback,
                         const BnBackwardSums &fin1, int *guard, hipStream_t st);

}  // namespace dgcnn
