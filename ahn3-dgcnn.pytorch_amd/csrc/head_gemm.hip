// head_gemm.hip -- the 1x1 convolutions of the segmentation head on point-major rows (conv6 .. conv8 of the reference,
// model.py:132-139 applied at :172, :184, :187, and their input gradients): y[P][N] = x[P][K] . w[N][K]^T in fp32 in /
// fp32 out, evaluated on the fp16 matrix cores.
//
// Both operands are scaled by an EXACT power of two into the fp16 range (scale from the operand's largest magnitude,
// found by a reduction pass ahead of the product) and split in two halves of the same scale, x' = hi + lo (hi = x' with
// the low 13 mantissa bits cleared: exactly a fp16 number; lo = x' - hi, rounded to fp16): hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation -- three matrix instructions of 32 cycles per 32x32x16 step where the
// fp32-input v_mfma_f32_32x32x2_f32 (which shares the vector ALU's FMA lanes on gfx950) needs eight of 64.  Error per
// operand element: 2^-23 relative (2^-25 of the scaled value where lo runs into fp16's subnormals: 2^-39 of the
// operand's largest entry), i.e. at the level of fp32's own accumulation error over K >= 192 terms; the parity tests hold
// the results to 1e-4 of PyTorch-CPU like every other kernel.
//
// Tiling: a workgroup of 4 waves computes 128 rows x 128 columns; wave w owns rows 32 w .. 32 w + 31 and all four
// 32-column tiles (12 MFMAs per 16-wide k step against 2 + 8 ds_read_b128).  K is walked in chunks of 32: the fp32
// operands of the next chunk are in flight (registers) while the current one is multiplied, then split and written to
// LDS as fp16 planes [row][k] (row stride 40 halves = 20 banks: conflict-free 16-byte reads).  Column blocks of one row
// block are adjacent in the grid, so the x rows of a row block are re-read from L2, not from HBM.
#include "common.h"
#include "dgcnn_hip.h"

namespace dgcnn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int GM = 128, GN = 128, GK = 32;  // workgroup tile, k chunk
constexpr int GSTR = GK + 8;                // halves per LDS row

// ---- operand magnitude: bits of max |x| over a strided [rows][cols] array (non-negative floats order like their
// bits; the word is zeroed by a memset node ahead of the kernel)
__global__ __launch_bounds__(256) void rows_absmax_kernel(const float *__restrict__ x, int ld, int rows, int cols,
                                                          unsigned *__restrict__ out)
{
    __shared__ float red[4];
    const size_t total = (size_t)rows * cols;
    float m = 0.0f;
    if (ld == cols && (cols & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        const size_t n4 = total / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
            m = fmaxf(m, fabsf(x[(i / cols) * ld + i % cols]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const unsigned bits = __float_as_uint(m);  // (NaN compares false everywhere above: it never enters m)
        if (bits > *out) atomicMax(out, bits);
    }
}

// the power of two s with 2^(top - 1) <= m s < 2^top (m > 0 and finite; else 1)
__device__ __forceinline__ float scale_to(float m, int top)
{
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    if (!(m > 0.0f) || e > 126) return 1.0f;
    int s = top - 1 - e;
    s = s > 120 ? 120 : (s < -120 ? -120 : s);
    return __uint_as_float((unsigned)(s + 127) << 23);
}

// four consecutive fp32 values (scaled) -> their hi halves and lo halves, 8 bytes each
__device__ __forceinline__ void split4(const float4 v, float s, u32x2 &hi, u32x2 &lo)
{
    const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __uint_as_float(__float_as_uint(x[i]) & 0xffffe000u);
    hi[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{h[0], h[1]}, f16x2));
    hi[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{h[2], h[3]}, f16x2));
    lo[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x[0] - h[0], x[1] - h[1]}, f16x2));
    lo[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x[2] - h[2], x[3] - h[3]}, f16x2));
}

// one 128 x 32 fp32 operand tile: thread t holds float4 number t + 256 i (i < 4): row (t + 256 i) / 8, columns 4 ((t + 256 i) % 8)
struct StageRegs {
    float4 v[4];
};
__device__ __forceinline__ void stage_load(const float *__restrict__ src, int ld, int row0, int nrows, int k0, StageRegs &r)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, row = f >> 3, c = (f & 7) * 4;
        const int rr = row0 + row < nrows ? row0 + row : nrows - 1;  // clamped: rows past the end are never stored
        r.v[i] = *reinterpret_cast<const float4 *>(src + (size_t)rr * ld + k0 + c);
    }
}
__device__ __forceinline__ void stage_store(const StageRegs &r, float s, _Float16 (*hi)[GSTR], _Float16 (*lo)[GSTR])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, row = f >> 3, c = (f & 7) * 4;
        u32x2 h, l;
        split4(r.v[i], s, h, l);
        *reinterpret_cast<u32x2 *>(&hi[row][c]) = h;
        *reinterpret_cast<u32x2 *>(&lo[row][c]) = l;
    }
}

// y = x . w^T (+ add[row / rows_per_add][col]);  K a multiple of 32, ldx / ldw multiples of 4, 16-byte aligned bases
template <bool HAS_ADD>
__global__ __launch_bounds__(256) void rows_gemm16_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                                                          int ldw, int P, int K, int N, const unsigned *__restrict__ amax,
                                                          const float *__restrict__ add, int rows_per_add,
                                                          float *__restrict__ Y, int ldy, int ncb)
{
    __shared__ __attribute__((aligned(16))) _Float16 xh[GM][GSTR], xl[GM][GSTR], wh[GN][GSTR], wl[GN][GSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int cb = (int)blockIdx.x % ncb, rb = (int)blockIdx.x / ncb;  // column blocks of a row block are neighbours
    const int row0 = rb * GM, col0 = cb * GN;
    const float sx = scale_to(__uint_as_float(amax[0]), 14), sw = scale_to(__uint_as_float(amax[1]), 14);

    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    StageRegs rx, rw;
    stage_load(X, ldx, row0, P, 0, rx);
    stage_load(W, ldw, col0, N, 0, rw);
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();  // the previous chunk's LDS reads are done
        stage_store(rx, sx, xh, xl);
        stage_store(rw, sw, wh, wl);
        __syncthreads();
        if (k0 + GK < K) {  // next chunk in flight during the MFMAs
            stage_load(X, ldx, row0, P, k0 + GK, rx);
            stage_load(W, ldw, col0, N, k0 + GK, rw);
        }
#pragma unroll
        for (int s = 0; s < GK / 16; ++s) {
            const half8 ahi = *reinterpret_cast<const half8 *>(&xh[wave * 32 + i32][16 * s + 8 * h]);
            const half8 alo = *reinterpret_cast<const half8 *>(&xl[wave * 32 + i32][16 * s + 8 * h]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const half8 bhi = *reinterpret_cast<const half8 *>(&wh[nt * 32 + i32][16 * s + 8 * h]);
                const half8 blo = *reinterpret_cast<const half8 *>(&wl[nt * 32 + i32][16 * s + 8 * h]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[nt], 0, 0, 0);
            }
        }
    }
    const float un = 1.0f / (sx * sw);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = col0 + nt * 32 + i32;
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wave * 32 + mfma32_row(r, h);
                if (row < P) {
                    float v = acc[nt][r] * un;
                    if (HAS_ADD) v += add[(size_t)(row / rows_per_add) * N + col];
                    Y[(size_t)row * ldy + col] = v;
                }
            }
        }
    }
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_rows_gemm_workspace_bytes(int P, int K, int N)
{
    (void)P, (void)K, (void)N;
    return 256;  // the two operand maxima
}

extern "C" int dgcnn_rows_gemm(const float *x, int ldx, const float *w, int ldw, int P, int K, int N, const float *add,
                               int rows_per_add, float *y, int ldy, void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && w && y && workspace, "dgcnn_rows_gemm: null pointer");
    DG_REQUIRE(P > 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N, "dgcnn_rows_gemm: bad shape P=%d K=%d N=%d", P, K, N);
    DG_REQUIRE((K & 31) == 0 && (ldx & 3) == 0 && (ldw & 3) == 0, "dgcnn_rows_gemm: K=%d must be a multiple of 32, ldx / ldw of 4", K);
    DG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "dgcnn_rows_gemm: operands must be 16-byte aligned");
    DG_REQUIRE(!add || rows_per_add > 0, "dgcnn_rows_gemm: rows_per_add must be positive");
    DG_REQUIRE(workspace_bytes >= 256, "dgcnn_rows_gemm: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned *amax = (unsigned *)workspace;
    if (hipMemsetAsync(amax, 0, 16, st) != hipSuccess) DG_FAIL("dgcnn_rows_gemm: hipMemsetAsync failed");
    int gx = ceil_div((int)(((size_t)P * K / 4 + 255) / 256), 8);
    gx = gx > 2048 ? 2048 : (gx < 1 ? 1 : gx);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(gx), dim3(256), 0, st, x, ldx, P, K, amax);
    int gw = ceil_div((int)(((size_t)N * K / 4 + 255) / 256), 8);
    gw = gw > 256 ? 256 : (gw < 1 ? 1 : gw);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(gw), dim3(256), 0, st, w, ldw, N, K, amax + 1);
    const int ncb = ceil_div(N, GN), nrb = ceil_div(P, GM);
    if (add)
        hipLaunchKernelGGL(rows_gemm16_kernel<true>, dim3(ncb * nrb), dim3(256), 0, st, x, ldx, w, ldw, P, K, N, amax, add,
                           rows_per_add, y, ldy, ncb);
    else
        hipLaunchKernelGGL(rows_gemm16_kernel<false>, dim3(ncb * nrb), dim3(256), 0, st, x, ldx, w, ldw, P, K, N, amax, add,
                           rows_per_add, y, ldy, ncb);
    DG_CHECK_LAUNCH("dgcnn_rows_gemm");
    return 0;
}
