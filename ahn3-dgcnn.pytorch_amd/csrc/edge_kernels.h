// edge_kernels.h -- host-side launchers of the EdgeConv kernels (one translation unit each).
#pragma once
#include "edge_common.h"

namespace dgcnn {

// ---- pointgemm.hip : per-point fp32 MFMA contractions --------------------------------------
// wcatT [Ke][128] (k-major) and wcat [128][C] from the conv weight w1 [64][2C]:
//   W1 . [x_j - x_i ; x_i] = Wa . x_j + (Wb - Wa) . x_i,  Wcat = [Wa ; Wb - Wa]
void launch_prep_wcat(const float *w1, int C, int Ke, float *wcatT, float *wcat, hipStream_t st);
// Y(p, n) = sum_k A(p, k) W[k][n];  A = [A0 | A1] (K0 + K1 columns), Y = [Y0 | Y1] (NO0 + rest)
void launch_rowgemm(const float *A0, int K0, const float *A1, int K1, const float *W, int NO, float *Y0, int NO0,
                    float *Y1, int P, hipStream_t st);
// partial[chunk][m][n] = sum_{p in chunk} A(p, m) X[p][n];  A = [A0 | A1] (M0 + M1 <= 128 columns)
int wgrad_chunks(int P);
void launch_wgrad(const float *A0, int M0, const float *A1, int M1, const float *X, int C, int P, float *partial,
                  hipStream_t st);
// G [128][C] = sum of the partials, then gw1[o][c]:  c < C: G[o][c] - G[64 + o][c];  c >= C: G[64 + o][c - C]
void launch_finish_dw1(const float *partial, int nchunk, int C, float *gw1, hipStream_t st);
// out[e] = sum_chunk partial[chunk][e]   (fixed order: deterministic)
void launch_sum_partials(const float *partial, int nchunk, int n, float *out, hipStream_t st);

// ---- edge_stream.hip : gather / reduce / scatter passes over the edges (no MFMA) ------------
int edge_grid(int P);
// stats of u = TP[n(i,j)] + TQ[i] over all edges (fp64 partials) and, if usel != NULL, the per
// (point, channel) extreme of u that survives max-over-k (max if gamma >= 0, min otherwise)
// usum[i] = sum_j u[i][j]; hitj[i] = slot j of the first edge attaining the extreme (with usel only)
void launch_edge_reduce(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                        const float *gamma, float *usel, float *usum, uint8_t *hitj, double *partial, hipStream_t st);
// training-mode stage-1 backward as a gather over the reverse adjacency of the kNN graph (see edge_stream.hip).
// launch_rev_build: the lists (depends on idx only).  Scratch: cnt [B * rev_chunks][N], indeg [P]; results: rowptr
// [P + 1], rev [E], extra [2 + 2 (E / 128 + 1)] ints.
// launch_edge_bwd_nogather: sparse dv only: hits [E] x 32 bytes, G [P][64].  dR is overwritten (no zero fill).
// dv == NULL: sparse dv from (gy, y, hitj); else dense dv [E][64] with dvsum [P][64] = its sums over k.
int rev_chunks(int B, int N, int k);
int launch_rev_build(const int32_t *idx, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int32_t *rev, int *extra,
                     hipStream_t st);
int launch_edge_bwd_nogather(const float *TP, const float *TQ, int B, int N, int k, const float *stats, const float *cc,
                             float slope, const float *gy, const float *y, const uint8_t *hitj, const float *usum,
                             const float *dv, const float *dvsum, const int *rowptr, const int32_t *rev, const int *extra,
                             void *hits,
                             float *G, float *dR, float *dD, hipStream_t st);
// BatchNorm statistics from the partial sums -> stats[ST_*]; running-statistics update
void launch_bn_finalize(double *partial, int nparts, double count, int training, const dgcnn_bn_params &bn,
                        float eps, float momentum, float *stats, hipStream_t st);
// y = LeakyReLU(sc * usel + sh)
void launch_edge_out(const float *usel, const float *stats, float slope, int P, float *y, hipStream_t st);
// last-stage BN backward sums from per-point quantities: gbn_w, gbn_b and cc = [c1 | c2]
void launch_bwd_point(const float *gy, const float *usel, const float *stats, float slope, int P, double count,
                      int training, double *partial, float *gbn_w, float *gbn_b, float *cc, hipStream_t st);
// stage-1 backward over the edges: dR (scatter-add into neighbours, zero-filled by the caller)
// and dD (sum over k at the centre).  dv == NULL: sparse dv from (gy, y);  else dense dv [E][64].
void launch_edge_bwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats,
                     const float *cc, float slope, int training, const float *gy, const float *y, const float *dv,
                     float *dR, float *dD, hipStream_t st);
void launch_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k, float *out,
                              hipStream_t st);
void launch_graph_feature_bwd(const float *gout, const int32_t *idx, int B, int N, int C, int k, float *gx,
                              hipStream_t st);
// generic: finalize (sum_b, sum_g) fp64 partials into gbn_w, gbn_b, cc  (used by both BN stages)
void launch_bn_bwd_finalize(double *partial, int nparts, double count, int training, float *gbn_w,
                            float *gbn_b, float *cc, hipStream_t st);

// ---- edge_mlp.hip : per-edge 64 -> 64 contraction on MFMA (second stage of blocks 1, 2) ------
int mlp_grid(int P);      // backward: number of gW2 / BN1 partial records
int mlp_grid_fwd(int P);  // forward: number of BN2 partial records
void launch_edge_mlp_fwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                         const float *stats1, const float *w2, const float *gamma2, float slope, float *usel2,
                         uint8_t *hit2, double *partial, hipStream_t st);
// dv1 [E][64]; dvsum [P][64] = sum over each point's k edges of dv1; partial_dw2 [grid][64][64]; partial_bn1
// [grid][2][64] doubles (sum dv1, sum dv1 * uhat1)
void launch_edge_mlp_bwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                         const float *stats1, const float *stats2, const float *cc2, const float *w2, float slope,
                         int training, const float *gy, const float *y, const uint8_t *hit2, float *dv1, float *dvsum,
                         float *partial_dw2, double *partial_bn1, hipStream_t st);

}  // namespace dgcnn
