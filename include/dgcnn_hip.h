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

#ifdef __cplusplus
}
#endif
#endif /* DGCNN_HIP_H */
