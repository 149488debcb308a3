// abi.hip -- library-level entry points and the layout helpers of libdgcnn_hip.so.
#include <stdarg.h>

#include "common.h"

namespace dgcnn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// One thread per (b, n, c) element; dst-contiguous order so stores coalesce.
// to_pm  : dst[b][n][c] = src[b*sb + c*sc + n*sn]
// from_pm: dst[b*sb + c*sc + n*sn] = src[b][n][c]  (thread order follows dst when sn == 1)
__global__ void to_point_major_kernel(const float *__restrict__ src, int C, int N, int64_t sb, int64_t sc,
                                      int64_t sn, float *__restrict__ dst, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t bn = i / C;
    const int n = (int)(bn % N);
    const size_t b = bn / N;
    dst[i] = src[(int64_t)b * sb + (int64_t)c * sc + (int64_t)n * sn];
}

__global__ void from_point_major_kernel(const float *__restrict__ src, int C, int N, float *__restrict__ dst,
                                        int64_t sb, int64_t sc, int64_t sn, size_t total)
{
    // thread index enumerates (b, c, n) with n fastest: coalesced stores for the usual
    // [B][C][N]-contiguous destination; the point-major reads hit L2.
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % N);
    const size_t bc = i / N;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    dst[(int64_t)b * sb + (int64_t)c * sc + (int64_t)n * sn] = src[(b * N + n) * C + c];
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" const char *dgcnn_last_error(void) { return g_err; }
extern "C" int dgcnn_abi_version(void) { return 6; }  // 6: dgcnn_model_* (5: dgcnn_head_tail_*, dgcnn_dropout_mask; 4: dgcnn_bn_params.num_batches_tracked)

extern "C" int dgcnn_to_point_major(const float *src, int B, int C, int N, int64_t sb, int64_t sc, int64_t sn,
                                    float *dst, void *stream)
{
    DG_REQUIRE(src && dst, "dgcnn_to_point_major: null pointer");
    DG_REQUIRE(B > 0 && C > 0 && N > 0, "dgcnn_to_point_major: bad shape");
    const size_t total = (size_t)B * C * N;
    hipLaunchKernelGGL(to_point_major_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, C, N, sb, sc, sn, dst, total);
    DG_CHECK_LAUNCH("to_point_major_kernel");
    return 0;
}

extern "C" int dgcnn_from_point_major(const float *src, int B, int C, int N, float *dst, int64_t sb, int64_t sc,
                                      int64_t sn, void *stream)
{
    DG_REQUIRE(src && dst, "dgcnn_from_point_major: null pointer");
    DG_REQUIRE(B > 0 && C > 0 && N > 0, "dgcnn_from_point_major: bad shape");
    const size_t total = (size_t)B * C * N;
    hipLaunchKernelGGL(from_point_major_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, C, N, dst, sb, sc, sn, total);
    DG_CHECK_LAUNCH("from_point_major_kernel");
    return 0;
}
