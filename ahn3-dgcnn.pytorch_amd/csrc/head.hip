// head.hip -- BatchNorm1d + LeakyReLU on point-major rows [P][C] for the segmentation head
// (reference model.py:132-139 applied at :172,:184,:187: Sequential(Conv1d 1x1, BatchNorm1d, LeakyReLU(0.2))).
// SURVEY.md section 8f row N1 ("next" row after the EdgeConv path): the 1x1 convolutions of the head stay plain
// library GEMMs on point-major rows; this file replaces the BatchNorm/activation around them, whose stock
// kernels are slow in this layout (945 + 689 us per layer at P = 56 000, C = 1024).
//
// Same statistics discipline as the EdgeConv kernels: fp64 partial sums, fixed reduction order.
#include <stdlib.h>

#include "edge_kernels.h"

namespace dgcnn {

constexpr int HEAD_ROWS_PER_BLOCK = 256;  // rows reduced by one workgroup (4 waves x 64 rows)

// partial[chunk][2][C]: sum and sum of squares (fp64) of x over the chunk's rows.  A lane owns four adjacent columns
// (one 16-byte load per row: 1 KB per wave and row), the four waves of a workgroup take every fourth row, two rows in
// flight per wave; C is a multiple of 4.
constexpr int HEAD_COLS_PER_BLOCK = 256;
__global__ __launch_bounds__(256) void colstats_kernel(const float *__restrict__ x, int P, int C,
                                                       double *__restrict__ partial)
{
    __shared__ double red[4][2][HEAD_COLS_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * HEAD_COLS_PER_BLOCK + 4 * lane;
    const int r0 = blockIdx.y * HEAD_ROWS_PER_BLOCK;
    const int r1 = r0 + HEAD_ROWS_PER_BLOCK < P ? r0 + HEAD_ROWS_PER_BLOCK : P;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int r = r0 + wave;
        for (; r + 4 < r1; r += 8) {
            const float4 v0 = *reinterpret_cast<const float4 *>(x + (size_t)r * C + c);
            const float4 v1 = *reinterpret_cast<const float4 *>(x + (size_t)(r + 4) * C + c);
            const float a[4] = {v0.x, v0.y, v0.z, v0.w}, b[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += (double)a[e];
                ss[e] = __builtin_fma((double)a[e], (double)a[e], ss[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += (double)b[e];
                ss[e] = __builtin_fma((double)b[e], (double)b[e], ss[e]);
            }
        }
        for (; r < r1; r += 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(x + (size_t)r * C + c);
            const float a[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += (double)a[e];
                ss[e] = __builtin_fma((double)a[e], (double)a[e], ss[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[wave][0][4 * lane + e] = s[e];
        red[wave][1][4 * lane + e] = ss[e];
    }
    __syncthreads();
    const int cc = blockIdx.x * HEAD_COLS_PER_BLOCK + threadIdx.x;
    if (cc < C) {
        const int t = threadIdx.x;
        partial[((size_t)blockIdx.y * 2 + 0) * C + cc] = (red[0][0][t] + red[1][0][t]) + (red[2][0][t] + red[3][0][t]);
        partial[((size_t)blockIdx.y * 2 + 1) * C + cc] = (red[0][1][t] + red[1][1][t]) + (red[2][1][t] + red[3][1][t]);
    }
}

// sum over the chunks of partial[chunk][2][C] for column c: the FOLD_W waves of the workgroup take every FOLD_W-th chunk,
// eight chunks in flight per wave (at B = 32, N = 7000 there are 875 chunks: with four waves and one load in flight the
// finalize kernels took 77-83 us each), added in chunk order; then the wave sums are added in wave order (fixed
// association: deterministic)
constexpr int FOLD_W = 16;
__device__ __forceinline__ void fold_chunks(const double *__restrict__ partial, int nchunk, int C, int c, bool live,
                                            double (*red)[2][64], double &s0, double &s1)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (live) {
        int g = wave;
        for (; g + 7 * FOLD_W < nchunk; g += 8 * FOLD_W) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = partial[((size_t)(g + u * FOLD_W) * 2 + 0) * C + c];
                vb[u] = partial[((size_t)(g + u * FOLD_W) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += va[u];
                b += vb[u];
            }
        }
        for (; g < nchunk; g += FOLD_W) {
            a += partial[((size_t)g * 2 + 0) * C + c];
            b += partial[((size_t)g * 2 + 1) * C + c];
        }
    }
    red[wave][0][lane] = a;
    red[wave][1][lane] = b;
    __syncthreads();
    s0 = 0.0;
    s1 = 0.0;
#pragma unroll
    for (int w = 0; w < FOLD_W; ++w) {
        s0 += red[w][0][lane];
        s1 += red[w][1][lane];
    }
}

// stats[4][C] = mean, rstd, sc = gamma rstd, sh = beta - mean sc; running statistics updated when training.
// One workgroup of four waves per 64 columns (lane = column); fixed summation order (deterministic).
__global__ __launch_bounds__(64 * FOLD_W) void colstats_finalize_kernel(const double *__restrict__ partial, int nchunk, int P, int C, int training,
                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                         float *__restrict__ rmean, float *__restrict__ rvar, float eps,
                                         float momentum, float *__restrict__ stats)
{
    __shared__ double red[FOLD_W][2][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s = 0.0, ss = 0.0;
    if (training) fold_chunks(partial, nchunk, C, c, c < C, red, s, ss);
    if (c >= C || threadIdx.x >= 64) return;
    float mean, var;
    if (training) {
        const double m = s / P;
        double v = ss / P - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        if (rmean) {
            const double unbiased = P > 1 ? v * P / (P - 1.0) : v;
            rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    const float rstd = (float)(1.0 / sqrt((double)var + (double)eps));
    const float sc = gamma[c] * rstd;
    stats[0 * (size_t)C + c] = mean;
    stats[1 * (size_t)C + c] = rstd;
    stats[2 * (size_t)C + c] = sc;
    stats[3 * (size_t)C + c] = beta[c] - mean * sc;
}

// y = LeakyReLU(sc x + sh); C is a multiple of 4
__global__ void bn_lrelu_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats, int C, float slope,
                                      size_t n4, float *__restrict__ y)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const int c = (int)((t * 4) % C);
    const float4 v = reinterpret_cast<const float4 *>(x)[t];
    const float4 sc = *reinterpret_cast<const float4 *>(stats + 2 * (size_t)C + c);
    const float4 sh = *reinterpret_cast<const float4 *>(stats + 3 * (size_t)C + c);
    float4 o;
    o.x = lrelu(__builtin_fmaf(v.x, sc.x, sh.x), slope);
    o.y = lrelu(__builtin_fmaf(v.y, sc.y, sh.y), slope);
    o.z = lrelu(__builtin_fmaf(v.z, sc.z, sh.z), slope);
    o.w = lrelu(__builtin_fmaf(v.w, sc.w, sh.w), slope);
    reinterpret_cast<float4 *>(y)[t] = o;
}

// backward sums: partial[chunk][2][C] = sum dv, sum dv * xhat  with dv = gy * LeakyReLU'(v), sign(v) = sign(y);
// same thread layout as colstats_kernel
__global__ __launch_bounds__(256) void bn_lrelu_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                  const float *__restrict__ gy,
                                                                  const float *__restrict__ stats, int P, int C,
                                                                  float slope, double *__restrict__ partial)
{
    __shared__ double red[4][2][HEAD_COLS_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * HEAD_COLS_PER_BLOCK + 4 * lane;
    const int r0 = blockIdx.y * HEAD_ROWS_PER_BLOCK;
    const int r1 = r0 + HEAD_ROWS_PER_BLOCK < P ? r0 + HEAD_ROWS_PER_BLOCK : P;
    double sb[4] = {0.0, 0.0, 0.0, 0.0}, sg[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        const float4 m4 = *reinterpret_cast<const float4 *>(stats + c), r4 = *reinterpret_cast<const float4 *>(stats + (size_t)C + c);
        const float mean[4] = {m4.x, m4.y, m4.z, m4.w}, rstd[4] = {r4.x, r4.y, r4.z, r4.w};
        for (int r = r0 + wave; r < r1; r += 4) {
            const size_t o = (size_t)r * C + c;
            const float4 g4 = *reinterpret_cast<const float4 *>(gy + o), y4 = *reinterpret_cast<const float4 *>(y + o);
            const float4 x4 = *reinterpret_cast<const float4 *>(x + o);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = g[e] * (yv[e] > 0.0f ? 1.0f : slope);
                const float xh = (xv[e] - mean[e]) * rstd[e];
                sb[e] += (double)dv;
                sg[e] = __builtin_fma((double)dv, (double)xh, sg[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[wave][0][4 * lane + e] = sb[e];
        red[wave][1][4 * lane + e] = sg[e];
    }
    __syncthreads();
    const int cc = blockIdx.x * HEAD_COLS_PER_BLOCK + threadIdx.x;
    if (cc < C) {
        const int t = threadIdx.x;
        partial[((size_t)blockIdx.y * 2 + 0) * C + cc] = (red[0][0][t] + red[1][0][t]) + (red[2][0][t] + red[3][0][t]);
        partial[((size_t)blockIdx.y * 2 + 1) * C + cc] = (red[0][1][t] + red[1][1][t]) + (red[2][1][t] + red[3][1][t]);
    }
}

// gbeta, ggamma and the two mean terms cc[2][C] (zero in eval mode)
__global__ __launch_bounds__(64 * FOLD_W) void bn_bwd_finalize_cols_kernel(const double *__restrict__ partial, int nchunk, int P, int C, int training,
                                            float *__restrict__ ggamma, float *__restrict__ gbeta, float *__restrict__ cc)
{
    __shared__ double red[FOLD_W][2][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double sb, sg;
    fold_chunks(partial, nchunk, C, c, c < C, red, sb, sg);
    if (c >= C || threadIdx.x >= 64) return;
    gbeta[c] = (float)sb;
    ggamma[c] = (float)sg;
    cc[c] = training ? (float)(sb / P) : 0.0f;
    cc[(size_t)C + c] = training ? (float)(sg / P) : 0.0f;
}

// gx = sc (dv - c1 - c2 xhat)
__global__ void bn_lrelu_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                          const float *__restrict__ gy, const float *__restrict__ stats,
                                          const float *__restrict__ cc, int C, float slope, size_t n4,
                                          float *__restrict__ gx)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const int c = (int)((t * 4) % C);
    const float4 xv = reinterpret_cast<const float4 *>(x)[t], yv = reinterpret_cast<const float4 *>(y)[t];
    const float4 g = reinterpret_cast<const float4 *>(gy)[t];
    const float *mean = stats + c, *rstd = stats + (size_t)C + c, *sc = stats + 2 * (size_t)C + c;
    const float *c1 = cc + c, *c2 = cc + (size_t)C + c;
    float4 o;
    o.x = sc[0] * (g.x * (yv.x > 0.0f ? 1.0f : slope) - c1[0] - c2[0] * ((xv.x - mean[0]) * rstd[0]));
    o.y = sc[1] * (g.y * (yv.y > 0.0f ? 1.0f : slope) - c1[1] - c2[1] * ((xv.y - mean[1]) * rstd[1]));
    o.z = sc[2] * (g.z * (yv.z > 0.0f ? 1.0f : slope) - c1[2] - c2[2] * ((xv.z - mean[2]) * rstd[2]));
    o.w = sc[3] * (g.w * (yv.w > 0.0f ? 1.0f : slope) - c1[3] - c2[3] * ((xv.w - mean[3]) * rstd[3]));
    reinterpret_cast<float4 *>(gx)[t] = o;
}

// ---- BatchNorm1d + LeakyReLU + max over the N points of a batch element (conv6 -> bn6 -> max, model.py:172-175) ----
// BN-affine + LeakyReLU is monotone per column (and so is every fp32 rounding in it), so the maximum of the
// activation is the activation of the column's extreme pre-activation (maximum if sc >= 0, minimum otherwise): the
// activated [P][C] tensor (229 MB at P = 56 000, C = 1024) is never written.  keys[b][c] = ordered(extreme) << 32 |
// ~n (the first point attaining it wins), combined over the row slices with 64-bit atomic max; zero-filled first.
typedef unsigned long long hkey;

__device__ __forceinline__ unsigned head_ordered(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float head_unordered(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ x, const float *__restrict__ stats, int N,
                                                     int C, hkey *__restrict__ keys)
{
    __shared__ hkey red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.z;
    const int r0 = blockIdx.y * HEAD_ROWS_PER_BLOCK;
    const int r1 = r0 + HEAD_ROWS_PER_BLOCK < N ? r0 + HEAD_ROWS_PER_BLOCK : N;
    hkey best = 0;
    if (c < C) {
        const unsigned flip = stats[2 * (size_t)C + c] < 0.0f ? 0x80000000u : 0u;  // sc < 0: track the minimum
        const float *xb = x + (size_t)b * N * C + c;
        for (int r = r0 + wave; r < r1; r += 4) {
            const unsigned o = head_ordered(__uint_as_float(__float_as_uint(xb[(size_t)r * C]) ^ flip));
            const hkey k = ((hkey)o << 32) | (hkey)(0xffffffffu - (unsigned)r);
            best = k > best ? k : best;
        }
    }
    red[wave][lane] = best;
    __syncthreads();
    if (wave == 0 && c < C) {
        hkey k0 = red[0][lane], k1 = red[1][lane], k2 = red[2][lane], k3 = red[3][lane];
        k0 = k0 > k1 ? k0 : k1;
        k2 = k2 > k3 ? k2 : k3;
        k0 = k0 > k2 ? k0 : k2;
        atomicMax(keys + (size_t)b * C + c, k0);
    }
}

// g[b][c] = LeakyReLU(sc ext + sh), arg[b][c] = the point that supplied the extreme
__global__ void colmax_finish_kernel(const hkey *__restrict__ keys, const float *__restrict__ stats, int B, int C,
                                     float slope, float *__restrict__ g, int32_t *__restrict__ arg,
                                     float *__restrict__ uwin = nullptr)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int c = t % C;
    const float sc = stats[2 * (size_t)C + c], sh = stats[3 * (size_t)C + c];
    const unsigned flip = sc < 0.0f ? 0x80000000u : 0u;
    const hkey k = keys[t];
    const float ext = __uint_as_float(__float_as_uint(head_unordered((unsigned)(k >> 32))) ^ flip);
    g[t] = lrelu(__builtin_fmaf(ext, sc, sh), slope);
    arg[t] = (int32_t)(0xffffffffu - (unsigned)k);
    if (uwin) uwin[t] = ext;  // the pre-activation at the winner (the backward of the fused conv6 needs nothing else of u)
}

// backward sums over the B winners of a column: gbias = sum dv, gweight = sum dv xhat, cc = (c1, c2) = the means over
// all P rows (training) or zero (eval); dvb[b][c] = dv of the winner = gg * LeakyReLU'(v)
__global__ void colmax_bwd_sums_kernel(const float *__restrict__ x, const float *__restrict__ gg,
                                       const int32_t *__restrict__ arg, const float *__restrict__ stats, int B, int N,
                                       int C, int training, float slope, float *__restrict__ gweight,
                                       float *__restrict__ gbias, float *__restrict__ cc, float *__restrict__ dvb)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c], rstd = stats[(size_t)C + c], sc = stats[2 * (size_t)C + c], sh = stats[3 * (size_t)C + c];
    double sb = 0.0, sg = 0.0;
    for (int b = 0; b < B; ++b) {
        const float xv = x[((size_t)b * N + arg[(size_t)b * C + c]) * C + c];
        const float v = __builtin_fmaf(xv, sc, sh);
        const float dv = gg[(size_t)b * C + c] * (v > 0.0f ? 1.0f : slope);
        dvb[(size_t)b * C + c] = dv;
        sb += (double)dv;
        sg += (double)dv * (double)((xv - mean) * rstd);
    }
    gbias[c] = (float)sb;
    gweight[c] = (float)sg;
    const double P = (double)B * N;
    cc[c] = training ? (float)(sb / P) : 0.0f;
    cc[C + c] = training ? (float)(sg / P) : 0.0f;
}

// gx[p][c] = sc (dv[p][c] - c1 - c2 xhat), dv non-zero only in the winner's row; C is a multiple of 4
__global__ void colmax_bwd_apply_kernel(const float *__restrict__ x, const int32_t *__restrict__ arg,
                                        const float *__restrict__ stats, const float *__restrict__ cc,
                                        const float *__restrict__ dvb, int N, int C, size_t n4, float *__restrict__ gx)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const size_t e = t * 4;
    const int c = (int)(e % C);
    const size_t p = e / C;
    const int b = (int)(p / N), n = (int)(p - (size_t)b * N);
    const float4 xv = reinterpret_cast<const float4 *>(x)[t];
    const float4 mean = *reinterpret_cast<const float4 *>(stats + c), rstd = *reinterpret_cast<const float4 *>(stats + (size_t)C + c);
    const float4 sc = *reinterpret_cast<const float4 *>(stats + 2 * (size_t)C + c);
    const float4 c1 = *reinterpret_cast<const float4 *>(cc + c), c2 = *reinterpret_cast<const float4 *>(cc + C + c);
    const int4 a = *reinterpret_cast<const int4 *>(arg + (size_t)b * C + c);
    const float4 dv = *reinterpret_cast<const float4 *>(dvb + (size_t)b * C + c);
    float4 o;
    o.x = sc.x * ((a.x == n ? dv.x : 0.0f) - c1.x - c2.x * ((xv.x - mean.x) * rstd.x));
    o.y = sc.y * ((a.y == n ? dv.y : 0.0f) - c1.y - c2.y * ((xv.y - mean.y) * rstd.y));
    o.z = sc.z * ((a.z == n ? dv.z : 0.0f) - c1.z - c2.z * ((xv.z - mean.z) * rstd.z));
    o.w = sc.w * ((a.w == n ? dv.w : 0.0f) - c1.w - c2.w * ((xv.w - mean.w) * rstd.w));
    reinterpret_cast<float4 *>(gx)[t] = o;
}


// ---- conv6 -> bn6 -> LeakyReLU -> max, backward WITHOUT the dense [P][C] gradient -------------------------------
// dL/du[p][c] = A_c + B_c u[p][c] + (s[b][c] at the winning point of (b, c)): the BatchNorm mean terms are affine in
// u = cat . W^T, so the two 1024-wide products of the backward collapse to 192-wide ones (ops.py: _EmbedMax):
//   dcat = 1 (A W)^T + cat (W^T diag(B) W) + [winners]      dW = A (sum cat)^T + diag(B) W (cat^T cat) + [winners]
// This kernel turns the winners' pre-activations uwin [B][C] and dL/dg into gbias, gweight, A, B and s = sc dv.
// Workgroups behind the channel workgroups (sorted != NULL): workgroup ncb + b sorts the (winning point, channel) keys of batch
// element b ONCE for embed_bwd_sorted_scatter_kernel (the order depends on arg alone; sorting inside every scatter workgroup
// -- 32 per batch element -- was 45 us of the backward at any batch size).
__global__ __launch_bounds__(256) void embed_bwd_coef_kernel(const float *__restrict__ uwin, const float *__restrict__ gg,
                                      const float *__restrict__ stats, int B, int N, int C, int training, float slope,
                                      float *__restrict__ gweight, float *__restrict__ gbias, float *__restrict__ coefA,
                                      float *__restrict__ coefB, float *__restrict__ s, const int32_t *__restrict__ arg,
                                      int C2, unsigned long long *__restrict__ sorted)
{
    extern __shared__ unsigned long long skeys[];  // [C2] (sort workgroups only)
    const int ncb = (C + 255) / 256;
    if ((int)blockIdx.x >= ncb) {
        // C2 <= 2048: C2 / 256 workgroups per batch element, each ranks 256 of its keys (rank sort); else one per element (bitonic)
        const int parts = C2 <= 2048 ? (C2 + 255) / 256 : 1;
        const int b = ((int)blockIdx.x - ncb) / parts, part = ((int)blockIdx.x - ncb) % parts;
        for (int i = threadIdx.x; i < C2; i += 256)
            skeys[i] = i < C ? ((unsigned long long)(unsigned)arg[(size_t)b * C + i] << 32) | (unsigned)i : ~0ull;
        __syncthreads();
        if (C2 <= 2048) {
            // rank sort: the keys are distinct (the channel is their low word), so the number of smaller keys IS the position;
            // a thread ranks ONE key against all of them with broadcast LDS reads -- no barrier-separated steps (the bitonic
            // network below takes 55 of them for 1024 keys: 30 us in one workgroup; this: 1024 compares per thread)
            const int i = part * 256 + (int)threadIdx.x;
            const unsigned long long mine = i < C2 ? skeys[i] : ~0ull;
            int rank = 0;
            for (int j = 0; j < C2; j += 8) {
                unsigned long long kj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kj[u] = skeys[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += kj[u] < mine ? 1 : 0;
            }
            // (the padding keys ~0 are equal among themselves: they keep their positions behind the C real keys)
            if (i < C) sorted[(size_t)b * C2 + rank] = mine;
            else if (i < C2) sorted[(size_t)b * C2 + i] = ~0ull;
            return;
        }
        for (int k = 2; k <= C2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < C2; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = skeys[i], c = skeys[l];
                        const bool up = (i & k) == 0;
                        if ((a > c) == up) {
                            skeys[i] = c;
                            skeys[l] = a;
                        }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < C2; i += 256) sorted[(size_t)b * C2 + i] = skeys[i];
        return;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c], rstd = stats[(size_t)C + c], sc = stats[2 * (size_t)C + c], sh = stats[3 * (size_t)C + c];
    double sb = 0.0, sg = 0.0;
    for (int b = 0; b < B; ++b) {
        const float u = uwin[(size_t)b * C + c];
        const float dv = gg[(size_t)b * C + c] * (__builtin_fmaf(u, sc, sh) > 0.0f ? 1.0f : slope);
        s[(size_t)b * C + c] = sc * dv;
        sb += (double)dv;
        sg += (double)dv * (double)((u - mean) * rstd);
    }
    gbias[c] = (float)sb;
    gweight[c] = (float)sg;
    const double P = (double)B * N;
    const double Bc = training ? -(double)sc * (sg / P) * (double)rstd : 0.0;
    coefB[c] = (float)Bc;
    coefA[c] = training ? (float)(-(double)sc * (sb / P) - Bc * (double)mean) : 0.0f;
}

// The scatter over keys sorted beforehand (embed_bwd_coef_kernel): workgroup (share, b) takes a contiguous share of the sorted
// list of batch element b, moved to the next point boundary at both ends, one thread per column f; the additions of one
// (point, column) happen in list order inside ONE thread: race-free, fixed order.
constexpr int EMB_SORTED_SHARES = 128;
__global__ __launch_bounds__(256) void embed_bwd_sorted_scatter_kernel(const unsigned long long *__restrict__ sorted, int C2,
                                                                       const float *__restrict__ s, const float *__restrict__ W,
                                                                       int N, int C, int F, float *__restrict__ dcat, int ldd)
{
    const int b = blockIdx.y;
    const unsigned long long *keys = sorted + (size_t)b * C2;
    const int per = (C + EMB_SORTED_SHARES - 1) / EMB_SORTED_SHARES;
    int i0 = blockIdx.x * per, i1 = i0 + per < C ? i0 + per : C;
    if (i0 >= C) return;
    // a point's run belongs to the share it starts in
    while (i0 > 0 && i0 < C && (keys[i0] >> 32) == (keys[i0 - 1] >> 32)) ++i0;
    while (i1 > 0 && i1 < C && (keys[i1] >> 32) == (keys[i1 - 1] >> 32)) ++i1;
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc = 0.0f;
        for (int i = i0; i < i1; i += 4) {
            float sv[4], wv[4];
            int nn[4], nx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long key = keys[i + u < i1 ? i + u : i1 - 1];
                const int c = (int)(unsigned)key;
                nn[u] = (int)(key >> 32);
                nx[u] = i + u + 1 < i1 ? (int)(keys[i + u + 1] >> 32) : -1;
                sv[u] = s[(size_t)b * C + c];
                wv[u] = W[(size_t)c * F + f];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u < i1) {
                    acc = __builtin_fmaf(sv[u], wv[u], acc);
                    if (nx[u] != nn[u]) {
                        dcat[((size_t)b * N + nn[u]) * ldd + f] += acc;
                        acc = 0.0f;
                    }
                }
            }
        }
    }
}

// dcat[b N + arg[b][c]][f] += s[b][c] W[c][f] over the C winners of batch element b, race-free and in a fixed order:
// every workgroup sorts the (point, channel) keys of its batch element in LDS, takes a contiguous share of the sorted
// list moved to the next point boundary at both ends, and walks it with one thread per column f.
constexpr int EMB_SHARES = 32;  // (8 shares: 91 us -- each workgroup walks its entries one dependent load at a time)
__global__ __launch_bounds__(256) void embed_bwd_scatter_kernel(const int32_t *__restrict__ arg, const float *__restrict__ s,
                                                                const float *__restrict__ W, int N, int C, int F, int C2,
                                                                float *__restrict__ dcat, int ldd)
{
    extern __shared__ unsigned long long keys[];  // C2 = C rounded up to a power of two
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C2; i += 256)
        keys[i] = i < C ? ((unsigned long long)(unsigned)arg[(size_t)b * C + i] << 32) | (unsigned)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= C2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < C2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) {
                        keys[i] = c;
                        keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    const int per = (C + EMB_SHARES - 1) / EMB_SHARES;
    int i0 = blockIdx.x * per, i1 = i0 + per < C ? i0 + per : C;
    // a point's run belongs to the share it starts in
    while (i0 > 0 && i0 < C && (keys[i0] >> 32) == (keys[i0 - 1] >> 32)) ++i0;
    while (i1 > 0 && i1 < C && (keys[i1] >> 32) == (keys[i1 - 1] >> 32)) ++i1;
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc = 0.0f;
        for (int i = i0; i < i1; i += 4) {
            // four entries' operands in flight, added in list order
            float sv[4], wv[4];
            int nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long key = keys[i + u < i1 ? i + u : i1 - 1];
                const int c = (int)(unsigned)key;
                nn[u] = (int)(key >> 32);
                sv[u] = s[(size_t)b * C + c];
                wv[u] = W[(size_t)c * F + f];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u < i1) {
                    acc = __builtin_fmaf(sv[u], wv[u], acc);
                    const int nxt = i + u + 1 < i1 ? (int)(keys[i + u + 1] >> 32) : -1;
                    if (nxt != nn[u]) {
                        dcat[((size_t)b * N + nn[u]) * ldd + f] += acc;
                        acc = 0.0f;
                    }
                }
            }
        }
    }
}

// ---- weight gradient of a 1x1 convolution on point-major rows: dW[M][Nn] = sum_p A[p][m] B[p][n] --------------------
// (A = dL/d(conv output) [P][M], B = the conv input [P][Nn]).  The contraction runs over the P points and the output
// is small, which library GEMMs handle poorly here (measured: 29 TFLOP/s for [256 x P] x [P x 512]); so split the
// points: workgroup (tile, chunk) accumulates a 128 x 64 tile over its chunk of points on v_mfma_f32_32x32x2_f32 --
// four waves take a quarter of the chunk each, operands straight from global memory (128-byte row segments, the next
// step's loads in flight under the current MFMAs) -- folds its waves through LDS in a fixed order and writes one
// partial tile; the partials are added in chunk order.  The tiles of one chunk read the same rows: workgroups are
// dealt round-robin over the 8 XCDs, so the grid index is remapped to give every XCD a contiguous range of
// (chunk, tile) pairs -- the tiles of a chunk then run side by side on ONE XCD and share its L2 (otherwise every XCD
// pulls every row through the fabric: measured 233 -> see DESIGN.md).
// VEC: lane i of the A operand takes columns 4 i .. 4 i + 3 of the tile with one 16-byte load (they become the four
// M-tiles: a tile is any 32 columns, here every fourth), lane j of B columns 2 j, 2 j + 1 -- a quarter of the load
// instructions, 512 contiguous bytes per half wave.  Needs 16-byte aligned rows and M % 4 == N % 2 == 0.
constexpr int RW_MT = 4, RW_NT = 2, RW_UN = 8;
template <bool VEC>
__global__ __launch_bounds__(256) void rows_wgrad_kernel(const float *__restrict__ A, int lda, int M,
                                                         const float *__restrict__ Bm, int ldb, int Nn, int P, int chunk,
                                                         int ntiles_n, float *__restrict__ partial)
{
    __shared__ float fold[2][RW_MT * 32][RW_NT * 32 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int tiles = ntiles_n * ((M + RW_MT * 32 - 1) / (RW_MT * 32));
    const int v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // gridDim.x is a multiple of 8
    const int cidx = v / tiles, tile = v - cidx * tiles;
    if (cidx * chunk >= P) return;  // padding workgroups
    const int tm = tile / ntiles_n, tn = tile - tm * ntiles_n;
    const int m0 = tm * RW_MT * 32, n0 = tn * RW_NT * 32;
    const int cbeg = cidx * chunk, cend = cbeg + chunk < P ? cbeg + chunk : P;
    const int quarter = ((cend - cbeg + 3) / 4 + 1) & ~1;  // even: the two lane halves take alternate points
    const int pbeg = cbeg + wave * quarter;
    const int pend = pbeg + quarter < cend ? pbeg + quarter : cend;
    int acol[RW_MT], bcol[RW_NT];
    bool aval[RW_MT], bval[RW_NT];
#pragma unroll
    for (int mt = 0; mt < RW_MT; ++mt) {
        const int col = VEC ? m0 + 4 * i32 + mt : m0 + mt * 32 + i32;
        aval[mt] = col < M;
        acol[mt] = aval[mt] ? col : 0;
    }
#pragma unroll
    for (int nt = 0; nt < RW_NT; ++nt) {
        const int col = VEC ? n0 + 2 * i32 + nt : n0 + nt * 32 + i32;
        bval[nt] = col < Nn;
        bcol[nt] = bval[nt] ? col : 0;
    }
    f32x16 acc[RW_MT][RW_NT];
#pragma unroll
    for (int mt = 0; mt < RW_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < RW_NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    float a[RW_UN][RW_MT], b[RW_UN][RW_NT], an[RW_UN][RW_MT], bn[RW_UN][RW_NT];
    auto fetch = [&](int p, float(&fa)[RW_UN][RW_MT], float(&fb)[RW_UN][RW_NT]) {
#pragma unroll
        for (int u = 0; u < RW_UN; ++u) {
            const int pp = p + 2 * u + h;
            const bool pv = pp < pend;
            const size_t row = (size_t)(pv ? pp : cbeg);  // clamped row: unconditional loads, masked values
            if (VEC) {
                const float4 va = *reinterpret_cast<const float4 *>(A + row * lda + acol[0]);
                const float2 vb = *reinterpret_cast<const float2 *>(Bm + row * ldb + bcol[0]);
                const bool av = pv && aval[0], bv = pv && bval[0];
                fa[u][0] = av ? va.x : 0.0f;
                fa[u][1] = av ? va.y : 0.0f;
                fa[u][2] = av ? va.z : 0.0f;
                fa[u][3] = av ? va.w : 0.0f;
                fb[u][0] = bv ? vb.x : 0.0f;
                fb[u][1] = bv ? vb.y : 0.0f;
            } else {
#pragma unroll
                for (int mt = 0; mt < RW_MT; ++mt) {
                    const float v = A[row * lda + acol[mt]];
                    fa[u][mt] = pv && aval[mt] ? v : 0.0f;
                }
#pragma unroll
                for (int nt = 0; nt < RW_NT; ++nt) {
                    const float v = Bm[row * ldb + bcol[nt]];
                    fb[u][nt] = pv && bval[nt] ? v : 0.0f;
                }
            }
        }
    };
    if (pbeg < pend) {
        fetch(pbeg, a, b);
        for (int p = pbeg; p < pend; p += 2 * RW_UN) {
            fetch(p + 2 * RW_UN, an, bn);  // past the end: all rows masked
#pragma unroll
            for (int u = 0; u < RW_UN; ++u)
#pragma unroll
                for (int mt = 0; mt < RW_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < RW_NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mt], b[u][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < RW_UN; ++u) {
#pragma unroll
                for (int mt = 0; mt < RW_MT; ++mt) a[u][mt] = an[u][mt];
#pragma unroll
                for (int nt = 0; nt < RW_NT; ++nt) b[u][nt] = bn[u][nt];
            }
        }
    }
    // fold the four waves: (2, 3) -> LDS, (0, 1) add; 1 -> LDS, 0 adds and stores
    auto put = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < RW_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) fold[slot][mt * 32 + mfma32_row(r, h)][nt * 32 + i32] = acc[mt][nt][r];
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < RW_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += fold[slot][mt * 32 + mfma32_row(r, h)][nt * 32 + i32];
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) {
        add(0);
        float *out = partial + (size_t)cidx * M * Nn;
#pragma unroll
        for (int mt = 0; mt < RW_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = mfma32_row(r, h);
                    const int m = VEC ? m0 + 4 * i + mt : m0 + mt * 32 + i, n = VEC ? n0 + 2 * i32 + nt : n0 + nt * 32 + i32;
                    if (m < M && n < Nn) out[(size_t)m * Nn + n] = acc[mt][nt][r];
                }
    }
}

int dgcnn_rows_wgrad16_launch(const float *a, int lda, int M, const float *b, int ldb, int Nn, int P, int chunk, float *partial,
                              const unsigned *amax_a, const unsigned *amax_b, unsigned *scratch, hipStream_t st);  // head_gemm.hip

// the fp16-matrix-core product needs 16-byte row pieces of both operands and whole 16-point MFMA steps per stage
static bool rows_wgrad_fast(const float *a, int lda, int M, const float *b, int ldb, int Nn)
{
    static const int off = getenv("DGCNN_WGRAD_F32") ? atoi(getenv("DGCNN_WGRAD_F32")) : 0;  // experiments / tests
    return !off && (M & 3) == 0 && (Nn & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && ((uintptr_t)a & 15) == 0 &&
           ((uintptr_t)b & 15) == 0 && M >= 32 && Nn >= 32;
}
static int rows_wgrad16_chunks(int M, int Nn, int P)
{
    const int tiles = ceil_div(M, 128) * ceil_div(Nn, 128);
    int chunks = 768 / tiles;  // about three workgroups per CU
    if (chunks < 8) chunks = 8;
    const int most = ceil_div(P, 128);
    if (chunks > most) chunks = most;
    return chunks < 1 ? 1 : chunks;
}

static int rows_wgrad_chunks(int M, int Nn, int P)
{
    const int tiles = ceil_div(M, RW_MT * 32) * ceil_div(Nn, RW_NT * 32);
    int chunks = 1024 / tiles;  // about four workgroups per CU
    if (chunks < 8) chunks = 8;
    const int most = ceil_div(P, 64);
    if (chunks > most) chunks = most;
    return chunks < 1 ? 1 : chunks;
}

static int head_chunks(int P) { return ceil_div(P, HEAD_ROWS_PER_BLOCK); }

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_bn_lrelu_workspace_bytes(int P, int C)
{
    if (P <= 0 || C <= 0) return 0;
    return align_up((size_t)head_chunks(P) * 2 * C * sizeof(double), 256) + align_up((size_t)2 * C * sizeof(float), 256);
}

extern "C" int dgcnn_bn_lrelu_fwd(const float *x, int P, int C, const dgcnn_bn_params *bn, int training, float eps,
                                  float momentum, float slope, float *y, float *stats, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && bn && y && stats && workspace, "dgcnn_bn_lrelu_fwd: null pointer");
    DG_REQUIRE(P > 0 && C > 0 && (C & 3) == 0, "dgcnn_bn_lrelu_fwd: bad shape P=%d C=%d (C must be a multiple of 4)", P, C);
    DG_REQUIRE(bn->weight && bn->bias, "dgcnn_bn_lrelu_fwd: BatchNorm parameters missing");
    DG_REQUIRE(training || (bn->running_mean && bn->running_var), "dgcnn_bn_lrelu_fwd: eval mode needs running statistics");
    DG_REQUIRE(workspace_bytes >= dgcnn_bn_lrelu_workspace_bytes(P, C), "dgcnn_bn_lrelu_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    const int nchunk = head_chunks(P);
    if (training)
        hipLaunchKernelGGL(colstats_kernel, dim3(ceil_div(C, HEAD_COLS_PER_BLOCK), nchunk), dim3(256), 0, st, x, P, C, partial);
    hipLaunchKernelGGL(colstats_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * FOLD_W), 0, st, partial, nchunk, P, C, training,
                       bn->weight, bn->bias, bn->running_mean, bn->running_var, eps, momentum, stats);
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(bn_lrelu_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, stats, C, slope, n4, y);
    DG_CHECK_LAUNCH("dgcnn_bn_lrelu_fwd");
    return 0;
}

extern "C" int dgcnn_bn_lrelu_bwd(const float *x, const float *y, const float *gy, const float *stats, int P, int C,
                                  int training, float slope, float *gx, float *gweight, float *gbias, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && y && gy && stats && gx && gweight && gbias && workspace, "dgcnn_bn_lrelu_bwd: null pointer");
    DG_REQUIRE(P > 0 && C > 0 && (C & 3) == 0, "dgcnn_bn_lrelu_bwd: bad shape P=%d C=%d", P, C);
    DG_REQUIRE(workspace_bytes >= dgcnn_bn_lrelu_workspace_bytes(P, C), "dgcnn_bn_lrelu_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = head_chunks(P);
    double *partial = (double *)workspace;
    float *cc = (float *)((char *)workspace + align_up((size_t)nchunk * 2 * C * sizeof(double), 256));
    hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel, dim3(ceil_div(C, HEAD_COLS_PER_BLOCK), nchunk), dim3(256), 0, st, x, y, gy, stats, P, C,
                       slope, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_cols_kernel, dim3(ceil_div(C, 64)), dim3(64 * FOLD_W), 0, st, partial, nchunk, P, C,
                       training, gweight, gbias, cc);
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(bn_lrelu_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, y, gy, stats,
                       cc, C, slope, n4, gx);
    DG_CHECK_LAUNCH("dgcnn_bn_lrelu_bwd");
    return 0;
}

extern "C" size_t dgcnn_bn_lrelu_max_workspace_bytes(int B, int N, int C)
{
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    return dgcnn_bn_lrelu_workspace_bytes(B * N, C) + align_up((size_t)B * C * sizeof(hkey), 256) +
           align_up((size_t)B * C * sizeof(float), 256);
}

extern "C" int dgcnn_bn_lrelu_max_fwd(const float *x, int B, int N, int C, const dgcnn_bn_params *bn, int training,
                                      float eps, float momentum, float slope, float *g, int32_t *arg, float *stats,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && bn && g && arg && stats && workspace, "dgcnn_bn_lrelu_max_fwd: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0 && (C & 3) == 0, "dgcnn_bn_lrelu_max_fwd: bad shape B=%d N=%d C=%d", B, N, C);
    DG_REQUIRE((size_t)B * N < ((size_t)1 << 31), "dgcnn_bn_lrelu_max_fwd: more than 2^31 rows");
    DG_REQUIRE(bn->weight && bn->bias, "dgcnn_bn_lrelu_max_fwd: BatchNorm parameters missing");
    DG_REQUIRE(training || (bn->running_mean && bn->running_var), "dgcnn_bn_lrelu_max_fwd: eval mode needs running statistics");
    DG_REQUIRE(workspace_bytes >= dgcnn_bn_lrelu_max_workspace_bytes(B, N, C), "dgcnn_bn_lrelu_max_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int P = B * N;
    double *partial = (double *)workspace;
    hkey *keys = (hkey *)((char *)workspace + dgcnn_bn_lrelu_workspace_bytes(P, C));
    const int nchunk = head_chunks(P);
    if (training)
        hipLaunchKernelGGL(colstats_kernel, dim3(ceil_div(C, HEAD_COLS_PER_BLOCK), nchunk), dim3(256), 0, st, x, P, C, partial);
    hipLaunchKernelGGL(colstats_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * FOLD_W), 0, st, partial, nchunk, P, C, training,
                       bn->weight, bn->bias, bn->running_mean, bn->running_var, eps, momentum, stats);
    if (hipMemsetAsync(keys, 0, (size_t)B * C * sizeof(hkey), st) != hipSuccess) DG_FAIL("dgcnn_bn_lrelu_max_fwd: memset failed");
    hipLaunchKernelGGL(colmax_kernel, dim3(ceil_div(C, 64), head_chunks(N), B), dim3(256), 0, st, x, stats, N, C, keys);
    hipLaunchKernelGGL(colmax_finish_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, keys, stats, B, C, slope, g, arg);
    DG_CHECK_LAUNCH("dgcnn_bn_lrelu_max_fwd");
    return 0;
}

namespace dgcnn {
void launch_rows_gemm_colred(const float *x, int ldx, const float *w, int ldw, int P, int K, int N, const unsigned *amax_x,
                             const unsigned *amax_w, int rows_per_batch, const float *gamma, int want_sums, double *partial,
                             unsigned long long *keys, hipStream_t st);  // head_gemm.hip
int rows_gemm_row_blocks(int P);
}  // namespace dgcnn

namespace dgcnn {
// launchers of this file's kernels for head_fused.hip (dgcnn_embed_fwd / _bwd)
void launch_colstats_finalize(const double *partial, int nchunk, int P, int C, int training, const dgcnn_bn_params *bn, float eps,
                              float momentum, float *stats, hipStream_t st)
{
    hipLaunchKernelGGL(colstats_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * FOLD_W), 0, st, partial, nchunk, P, C, training,
                       bn->weight, bn->bias, bn->running_mean, bn->running_var, eps, momentum, stats);
}
void launch_colmax_finish(const unsigned long long *keys, const float *stats, int B, int C, float slope, float *g, int32_t *arg,
                          float *uwin, hipStream_t st)
{
    hipLaunchKernelGGL(colmax_finish_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, keys, stats, B, C, slope, g, arg, uwin);
}
void launch_embed_bwd_coef(const float *uwin, const float *gg, const float *stats, int B, int N, int C, int training, float slope,
                           float *gweight, float *gbias, float *coef_a, float *coef_b, float *s, hipStream_t st, const int32_t *arg,
                           unsigned long long *sorted)
{
    int C2 = 1;
    while (C2 < C) C2 <<= 1;
    const int sort_wgs = sorted ? B * (C2 <= 2048 ? ceil_div(C2, 256) : 1) : 0;
    hipLaunchKernelGGL(embed_bwd_coef_kernel, dim3(ceil_div(C, 256) + sort_wgs), dim3(256),
                       sorted ? (size_t)C2 * sizeof(unsigned long long) : 0, st, uwin, gg, stats, B, N, C, training, slope, gweight, gbias,
                       coef_a, coef_b, s, arg, C2, sorted);
}
int embed_sorted_len(int C)
{
    int C2 = 1;
    while (C2 < C) C2 <<= 1;
    return C2;
}
void launch_embed_bwd_sorted_scatter(const unsigned long long *sorted, const float *s, const float *w, int B, int N, int C, int F,
                                     float *dcat, hipStream_t st, int ldd)
{
    hipLaunchKernelGGL(embed_bwd_sorted_scatter_kernel, dim3(EMB_SORTED_SHARES, B), dim3(256), 0, st, sorted, embed_sorted_len(C), s, w,
                       N, C, F, dcat, ldd > 0 ? ldd : F);
}
void launch_embed_bwd_scatter(const int32_t *arg, const float *s, const float *w, int B, int N, int C, int F, float *dcat,
                              hipStream_t st, int ldd)
{
    if (ldd <= 0) ldd = F;
    int C2 = 1;
    while (C2 < C) C2 <<= 1;
    hipLaunchKernelGGL(embed_bwd_scatter_kernel, dim3(EMB_SHARES, B), dim3(256), (size_t)C2 * sizeof(unsigned long long), st, arg,
                       s, w, N, C, F, C2, dcat, ldd);
}
}  // namespace dgcnn

extern "C" size_t dgcnn_embed_max_fwd_workspace_bytes(int B, int N, int C)
{
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    return align_up((size_t)rows_gemm_row_blocks(B * N) * 2 * C * sizeof(double), 256) + align_up((size_t)B * C * sizeof(hkey), 256);
}

extern "C" int dgcnn_embed_max_fwd(const float *cat, int ldc, const float *w, int ldw, int B, int N, int F, int C,
                                   const uint32_t *amax_cat, const uint32_t *amax_w, const dgcnn_bn_params *bn, int training,
                                   float eps, float momentum, float slope, float *g, int32_t *arg, float *uwin, float *stats,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(cat && w && amax_cat && amax_w && bn && g && arg && uwin && stats && workspace, "dgcnn_embed_max_fwd: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && F > 0 && C > 0 && ldc >= F && ldw >= F, "dgcnn_embed_max_fwd: bad shape B=%d N=%d F=%d C=%d", B, N, F, C);
    DG_REQUIRE((F & 31) == 0 && (ldc & 3) == 0 && (ldw & 3) == 0 && ((uintptr_t)cat & 15) == 0 && ((uintptr_t)w & 15) == 0,
               "dgcnn_embed_max_fwd: F must be a multiple of 32, rows 16-byte aligned");
    DG_REQUIRE((size_t)B * N < ((size_t)1 << 31), "dgcnn_embed_max_fwd: more than 2^31 rows");
    DG_REQUIRE(bn->weight && bn->bias, "dgcnn_embed_max_fwd: BatchNorm parameters missing");
    DG_REQUIRE(training || (bn->running_mean && bn->running_var), "dgcnn_embed_max_fwd: eval mode needs running statistics");
    DG_REQUIRE(workspace_bytes >= dgcnn_embed_max_fwd_workspace_bytes(B, N, C), "dgcnn_embed_max_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int P = B * N, nrb = rows_gemm_row_blocks(P);
    double *partial = (double *)workspace;
    hkey *keys = (hkey *)((char *)workspace + align_up((size_t)nrb * 2 * C * sizeof(double), 256));
    if (hipMemsetAsync(keys, 0, (size_t)B * C * sizeof(hkey), st) != hipSuccess) DG_FAIL("dgcnn_embed_max_fwd: memset failed");
    launch_rows_gemm_colred(cat, ldc, w, ldw, P, F, C, amax_cat, amax_w, N, bn->weight, training, partial, keys, st);
    hipLaunchKernelGGL(colstats_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * FOLD_W), 0, st, partial, nrb, P, C, training,
                       bn->weight, bn->bias, bn->running_mean, bn->running_var, eps, momentum, stats);
    hipLaunchKernelGGL(colmax_finish_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, keys, stats, B, C, slope, g, arg, uwin);
    DG_CHECK_LAUNCH("dgcnn_embed_max_fwd");
    return 0;
}

extern "C" int dgcnn_bn_lrelu_max_bwd(const float *x, const float *gg, const int32_t *arg, const float *stats, int B, int N,
                                      int C, int training, float slope, float *gx, float *gweight, float *gbias,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && gg && arg && stats && gx && gweight && gbias && workspace, "dgcnn_bn_lrelu_max_bwd: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0 && (C & 3) == 0, "dgcnn_bn_lrelu_max_bwd: bad shape B=%d N=%d C=%d", B, N, C);
    DG_REQUIRE(workspace_bytes >= dgcnn_bn_lrelu_max_workspace_bytes(B, N, C), "dgcnn_bn_lrelu_max_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int P = B * N;
    char *base = (char *)workspace;
    float *cc = (float *)base;  // the statistics partials are not needed here: c1, c2 take their place
    float *dvb = (float *)(base + dgcnn_bn_lrelu_workspace_bytes(P, C) + align_up((size_t)B * C * sizeof(hkey), 256));
    hipLaunchKernelGGL(colmax_bwd_sums_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, x, gg, arg, stats, B, N, C, training,
                       slope, gweight, gbias, cc, dvb);
    const size_t n4 = (size_t)P * C / 4;
    hipLaunchKernelGGL(colmax_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, arg, stats, cc, dvb,
                       N, C, n4, gx);
    DG_CHECK_LAUNCH("dgcnn_bn_lrelu_max_bwd");
    return 0;
}

extern "C" int dgcnn_embed_max_bwd_prepare(const float *uwin, const float *gg, const float *stats, int B, int N, int C,
                                           int training, float slope, float *gweight, float *gbias, float *coef_a,
                                           float *coef_b, float *s, void *stream)
{
    DG_REQUIRE(uwin && gg && stats && gweight && gbias && coef_a && coef_b && s, "dgcnn_embed_max_bwd_prepare: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0, "dgcnn_embed_max_bwd_prepare: bad shape B=%d N=%d C=%d", B, N, C);
    hipLaunchKernelGGL(embed_bwd_coef_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, uwin, gg, stats, B, N,
                       C, training, slope, gweight, gbias, coef_a, coef_b, s, (const int32_t *)nullptr, 0, (unsigned long long *)nullptr);
    DG_CHECK_LAUNCH("dgcnn_embed_max_bwd_prepare");
    return 0;
}

extern "C" int dgcnn_embed_max_bwd_scatter(const int32_t *arg, const float *s, const float *w, int B, int N, int C, int F,
                                           float *dcat, void *stream)
{
    DG_REQUIRE(arg && s && w && dcat, "dgcnn_embed_max_bwd_scatter: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0 && F > 0, "dgcnn_embed_max_bwd_scatter: bad shape B=%d N=%d C=%d F=%d", B, N, C, F);
    DG_REQUIRE(C <= 8192, "dgcnn_embed_max_bwd_scatter: C=%d > 8192 unsupported", C);
    int C2 = 1;
    while (C2 < C) C2 <<= 1;
    hipLaunchKernelGGL(embed_bwd_scatter_kernel, dim3(EMB_SHARES, B), dim3(256), (size_t)C2 * sizeof(unsigned long long),
                       (hipStream_t)stream, arg, s, w, N, C, F, C2, dcat, F);
    DG_CHECK_LAUNCH("dgcnn_embed_max_bwd_scatter");
    return 0;
}

extern "C" size_t dgcnn_rows_wgrad_workspace_bytes(int M, int Nn, int P)
{
    if (M <= 0 || Nn <= 0 || P <= 0) return 0;
    const int c0 = rows_wgrad_chunks(M, Nn, P), c1 = rows_wgrad16_chunks(M, Nn, P);
    return 256 + align_up((size_t)(c0 > c1 ? c0 : c1) * M * Nn * sizeof(float), 256);
}

extern "C" int dgcnn_rows_wgrad(const float *a, int lda, int M, const float *b, int ldb, int Nn, int P,
                                const uint32_t *amax_a, const uint32_t *amax_b, float *dw, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(a && b && dw && workspace, "dgcnn_rows_wgrad: null pointer");
    DG_REQUIRE(M > 0 && Nn > 0 && P > 0 && lda >= M && ldb >= Nn, "dgcnn_rows_wgrad: bad shape M=%d N=%d P=%d lda=%d ldb=%d", M,
               Nn, P, lda, ldb);
    DG_REQUIRE((size_t)M * Nn < ((size_t)1 << 31), "dgcnn_rows_wgrad: output too large");
    DG_REQUIRE(workspace_bytes >= dgcnn_rows_wgrad_workspace_bytes(M, Nn, P), "dgcnn_rows_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (rows_wgrad_fast(a, lda, M, b, ldb, Nn)) {
        const int chunks16 = rows_wgrad16_chunks(M, Nn, P);
        const int chunk16 = (ceil_div(P, chunks16) + 31) & ~31;  // whole stages of 32 points
        if (dgcnn_rows_wgrad16_launch(a, lda, M, b, ldb, Nn, P, chunk16, (float *)((char *)workspace + 256), amax_a, amax_b,
                                      (unsigned *)workspace, st))
            DG_FAIL("dgcnn_rows_wgrad: hipMemsetAsync failed");
        launch_sum_partials((const float *)((char *)workspace + 256), ceil_div(P, chunk16), M * Nn, dw, st);
        DG_CHECK_LAUNCH("dgcnn_rows_wgrad");
        return 0;
    }
    const int chunks = rows_wgrad_chunks(M, Nn, P);
    const int chunk = ceil_div(P, chunks);
    const int tn = ceil_div(Nn, RW_NT * 32), tm = ceil_div(M, RW_MT * 32);
    const bool vec = (M & 3) == 0 && (Nn & 1) == 0 && (lda & 3) == 0 && (ldb & 1) == 0 && ((uintptr_t)a & 15) == 0 &&
                     ((uintptr_t)b & 7) == 0;
    const dim3 grid((unsigned)align_up((size_t)tm * tn * ceil_div(P, chunk), 8));
    if (vec)
        hipLaunchKernelGGL(rows_wgrad_kernel<true>, grid, dim3(256), 0, st, a, lda, M, b, ldb, Nn, P, chunk, tn,
                           (float *)((char *)workspace + 256));
    else
        hipLaunchKernelGGL(rows_wgrad_kernel<false>, grid, dim3(256), 0, st, a, lda, M, b, ldb, Nn, P, chunk, tn,
                           (float *)((char *)workspace + 256));
    launch_sum_partials((const float *)((char *)workspace + 256), ceil_div(P, chunk), M * Nn, dw, st);
    DG_CHECK_LAUNCH("dgcnn_rows_wgrad");
    return 0;
}
