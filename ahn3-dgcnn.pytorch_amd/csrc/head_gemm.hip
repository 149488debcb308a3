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
#include <stdlib.h>

#include "common.h"
#include "dgcnn_hip.h"

namespace dgcnn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int GM = 128, GN = 128;  // workgroup tile; the k chunk GK (32 or 64) is a template parameter: LDS rows hold
                                   // GK + 8 halves (20 or 36 banks: conflict-free 16-byte reads either way)

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

// one 128 x GK fp32 operand tile: thread t holds float4 number t + 256 i (i < GK / 8): row f / (GK / 4), columns 4 (f % (GK / 4))
template <int GK>
struct StageRegs {
    float4 v[GK / 8];
};
template <int GK>
__device__ __forceinline__ void stage_load(const float *__restrict__ src, int ld, int row0, int nrows, int k0, StageRegs<GK> &r)
{
#pragma unroll
    for (int i = 0; i < GK / 8; ++i) {
        const int f = threadIdx.x + 256 * i, row = f / (GK / 4), c = (f % (GK / 4)) * 4;
        const int rr = row0 + row < nrows ? row0 + row : nrows - 1;  // clamped: rows past the end are never stored
        r.v[i] = *reinterpret_cast<const float4 *>(src + (size_t)rr * ld + k0 + c);
    }
}
template <int GK>
__device__ __forceinline__ void stage_store(const StageRegs<GK> &r, float s, _Float16 (*hi)[GK + 8], _Float16 (*lo)[GK + 8])
{
#pragma unroll
    for (int i = 0; i < GK / 8; ++i) {
        const int f = threadIdx.x + 256 * i, row = f / (GK / 4), c = (f % (GK / 4)) * 4;
        u32x2 h, l;
        split4(r.v[i], s, h, l);
        *reinterpret_cast<u32x2 *>(&hi[row][c]) = h;
        *reinterpret_cast<u32x2 *>(&lo[row][c]) = l;
    }
}

// y = x . w^T (+ add[row / rows_per_add][col]);  K a multiple of 32, ldx / ldw multiples of 4, 16-byte aligned bases
template <bool HAS_ADD, int GK>
__global__ __launch_bounds__(256) void rows_gemm16_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                                                          int ldw, int P, int K, int N, const unsigned *__restrict__ amax_x,
                                                          const unsigned *__restrict__ amax_w,
                                                          const float *__restrict__ add, int rows_per_add,
                                                          float *__restrict__ Y, int ldy, int ncb)
{
    constexpr int GSTR = GK + 8;
    __shared__ __attribute__((aligned(16))) _Float16 xh[GM][GSTR], xl[GM][GSTR], wh[GN][GSTR], wl[GN][GSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int cb = (int)blockIdx.x % ncb, rb = (int)blockIdx.x / ncb;  // column blocks of a row block are neighbours
    const int row0 = rb * GM, col0 = cb * GN;
    const float sx = scale_to(__uint_as_float(*amax_x), 14), sw = scale_to(__uint_as_float(*amax_w), 14);

    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    StageRegs<GK> rx, rw;
    stage_load<GK>(X, ldx, row0, P, 0, rx);
    stage_load<GK>(W, ldw, col0, N, 0, rw);
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();  // the previous chunk's LDS reads are done
        stage_store<GK>(rx, sx, xh, xl);
        stage_store<GK>(rw, sw, wh, wl);
        __syncthreads();
        if (k0 + GK < K) {  // next chunk in flight during the MFMAs
            stage_load<GK>(X, ldx, row0, P, k0 + GK, rx);
            stage_load<GK>(W, ldw, col0, N, k0 + GK, rw);
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

// ---- conv6 -> bn6 -> LeakyReLU -> max over the points (model.py:172-175) without ever writing u = cat . W6^T ---------
// The same product as rows_gemm16_kernel, but the [P][C] result stays in the accumulators: the epilogue leaves
//   * per column the sum and the sum of squares over the tile's rows (fp64 records partial[row block][2][C]: the
//     BatchNorm1d batch statistics, finished by colstats_finalize_kernel in record order), and
//   * per (batch element, column) the extreme that survives the max over the points -- BatchNorm + LeakyReLU is monotone
//     per column (increasing for gamma >= 0, else decreasing), so the extreme of u is enough -- as a 64-bit key
//     {ordered(u ^ flip), ~point}: larger value first, then the LOWER point index (PyTorch's first maximum); keys of a tile
//     meet in LDS, then one atomicMax per (batch element, column) and tile on keys[B][C] (a maximum: order-independent).
// At B = 8, N = 7000 that is 229 MB never written and never re-read twice (column statistics, column maxima).
typedef unsigned long long hkey64;
__device__ __forceinline__ unsigned ordered_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void rows_gemm16_colred_kernel(const float *__restrict__ X, int ldx,
                                                                 const float *__restrict__ W, int ldw, int P, int K, int N,
                                                                 const unsigned *__restrict__ amax_x,
                                                                 const unsigned *__restrict__ amax_w, int rows_per_batch,
                                                                 const float *__restrict__ gamma, int want_sums,
                                                                 double *__restrict__ partial, hkey64 *__restrict__ keys,
                                                                 int ncb)
{
    constexpr int GK = 32, GSTR = GK + 8;
    __shared__ __attribute__((aligned(16))) _Float16 xh[GM][GSTR], xl[GM][GSTR], wh[GN][GSTR], wl[GN][GSTR];
    __shared__ hkey64 kbest[2][GN];       // the tile's first two batch elements
    __shared__ double csum[4][2][GN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int cb = (int)blockIdx.x % ncb, rb = (int)blockIdx.x / ncb;
    const int row0 = rb * GM, col0 = cb * GN;
    const float sx = scale_to(__uint_as_float(*amax_x), 14), sw = scale_to(__uint_as_float(*amax_w), 14);
    for (int t = threadIdx.x; t < 2 * GN; t += 256) kbest[t / GN][t % GN] = 0;

    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    StageRegs<GK> rx, rw;
    stage_load<GK>(X, ldx, row0, P, 0, rx);
    stage_load<GK>(W, ldw, col0, N, 0, rw);
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();
        stage_store<GK>(rx, sx, xh, xl);
        stage_store<GK>(rw, sw, wh, wl);
        __syncthreads();
        if (k0 + GK < K) {
            stage_load<GK>(X, ldx, row0, P, k0 + GK, rx);
            stage_load<GK>(W, ldw, col0, N, k0 + GK, rw);
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
    // ---- epilogue: lane (n, h) holds column nt 32 + n, rows mfma32_row(r, h) (ascending in r) of this wave's 32 rows
    const float un = 1.0f / (sx * sw);
    const int bfirst = row0 / rows_per_batch;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int cl = nt * 32 + i32, col = col0 + cl;
        const bool cv = col < N;
        const unsigned flip = cv && gamma[col] < 0.0f ? 0x80000000u : 0u;  // gamma < 0: the activation's maximum is u's minimum
        double s1 = 0.0, s2 = 0.0;  // (fp64 from the first addition on: |mean| >> sigma costs an fp32 sum of squares its digits)
        hkey64 best = 0;
        int bcur = -1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 32 + mfma32_row(r, h);
            if (row < P && cv) {
                const float u = acc[nt][r] * un;
                s1 += (double)u;
                s2 = __builtin_fma((double)u, (double)u, s2);
                const int b = row / rows_per_batch, rl = row - b * rows_per_batch;
                if (b != bcur) {  // the rows of a lane ascend: a batch element's rows are consecutive
                    if (bcur >= 0) {
                        if (bcur - bfirst < 2) atomicMax(&kbest[bcur - bfirst][cl], best);
                        else atomicMax(keys + (size_t)bcur * N + col, best);
                    }
                    bcur = b;
                    best = 0;
                }
                const hkey64 key = ((hkey64)ordered_bits(__uint_as_float(__float_as_uint(u) ^ flip)) << 32) |
                                   (hkey64)(0xffffffffu - (unsigned)rl);
                best = key > best ? key : best;
            }
        }
        if (bcur >= 0) {
            if (bcur - bfirst < 2) atomicMax(&kbest[bcur - bfirst][cl], best);
            else atomicMax(keys + (size_t)bcur * N + col, best);
        }
        if (want_sums) {
            const double d1 = s1 + __shfl_xor(s1, 32), d2 = s2 + __shfl_xor(s2, 32);
            if (h == 0) {
                csum[wave][0][cl] = d1;
                csum[wave][1][cl] = d2;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * GN; t += 256) {
        const int bb = t / GN, cl = t % GN;
        const hkey64 kv = kbest[bb][cl];
        if (kv != 0 && col0 + cl < N) atomicMax(keys + (size_t)(bfirst + bb) * N + col0 + cl, kv);
    }
    if (want_sums && threadIdx.x < GN && col0 + (int)threadIdx.x < N) {
        const int cl = threadIdx.x;
        partial[((size_t)rb * 2 + 0) * N + col0 + cl] = (csum[0][0][cl] + csum[1][0][cl]) + (csum[2][0][cl] + csum[3][0][cl]);
        partial[((size_t)rb * 2 + 1) * N + col0 + cl] = (csum[0][1][cl] + csum[1][1][cl]) + (csum[2][1][cl] + csum[3][1][cl]);
    }
}

// ---- weight gradient of a 1x1 convolution on point-major rows: dW[M][Nn] = sum_p A[p][m] B[p][n] --------------------
// (A = dL/d(conv output) [P][M], B = the conv input [P][Nn]; model.py:172-187 backward).  The contraction runs over the
// points, so both MFMA operands need runs of consecutive POINTS for a fixed column: a stage of 32 points is loaded in
// 16-byte row pieces (thread = 4 columns x 4 points), split, and written to LDS TRANSPOSED as fp16 planes [column][point]
// (one 8-byte write per column and plane), from where the operands are 16-byte reads.  Workgroup = one 128 x 128 tile of
// dW over one chunk of the points (wave = a 64 x 64 quadrant: 12 MFMAs per 16 points against 8 ds_read_b128); partial
// tiles are added in chunk order by sum_partials_kernel (fixed association).  Same operand scaling and splitting as
// rows_gemm16_kernel.  Workgroups are dealt round-robin over the 8 XCDs: the grid index is remapped so that the tiles of
// one chunk, which read the same rows, run on ONE XCD and share its L2.
constexpr int WK = 32;          // points per stage
constexpr int WSTR = WK + 8;    // halves per LDS row

struct WStage {
    float4 v[4];  // v[i] = row p0 + 4 pq + i, columns c0 + 4 cq .. + 3
};
__device__ __forceinline__ void wstage_load(const float *__restrict__ src, int ld, int ncols, int c0, int p0, int pend,
                                            WStage &r)
{
    const int cq = threadIdx.x & 31, pq = threadIdx.x >> 5;
    const int col = c0 + 4 * cq < ncols ? c0 + 4 * cq : 0;  // columns past the matrix: any valid address (never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + 4 * pq + i;
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)(p < pend ? p : pend - 1) * ld + col);
        r.v[i] = p < pend ? v : float4{0.0f, 0.0f, 0.0f, 0.0f};  // points past the chunk add nothing
    }
}
__device__ __forceinline__ void wstage_store(const WStage &r, float s, _Float16 (*hi)[WSTR], _Float16 (*lo)[WSTR])
{
    const int cq = threadIdx.x & 31, pq = threadIdx.x >> 5;
    const float4 t[4] = {float4{r.v[0].x, r.v[1].x, r.v[2].x, r.v[3].x}, float4{r.v[0].y, r.v[1].y, r.v[2].y, r.v[3].y},
                         float4{r.v[0].z, r.v[1].z, r.v[2].z, r.v[3].z}, float4{r.v[0].w, r.v[1].w, r.v[2].w, r.v[3].w}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32x2 h, l;
        split4(t[c], s, h, l);
        *reinterpret_cast<u32x2 *>(&hi[4 * cq + c][4 * pq]) = h;
        *reinterpret_cast<u32x2 *>(&lo[4 * cq + c][4 * pq]) = l;
    }
}

__global__ __launch_bounds__(256) void rows_wgrad16_kernel(const float *__restrict__ A, int lda, int M,
                                                           const float *__restrict__ Bm, int ldb, int Nn, int P, int chunk,
                                                           int ntiles_n, const unsigned *__restrict__ amax_a,
                                                           const unsigned *__restrict__ amax_b, float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) _Float16 ah[128][WSTR], al[128][WSTR], bh[128][WSTR], bl[128][WSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int tiles = ntiles_n * ((M + 127) / 128);
    const int v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // gridDim.x is a multiple of 8
    const int cidx = v / tiles, tile = v - cidx * tiles;
    if (cidx * chunk >= P) return;  // padding workgroups
    const int tm = tile / ntiles_n, tn = tile - tm * ntiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int pbeg = cidx * chunk, pend = pbeg + chunk < P ? pbeg + chunk : P;
    const float sa = scale_to(__uint_as_float(*amax_a), 14), sb = scale_to(__uint_as_float(*amax_b), 14);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a_][b_][r] = 0.0f;

    WStage ra, rb;
    wstage_load(A, lda, M, m0, pbeg, pend, ra);
    wstage_load(Bm, ldb, Nn, n0, pbeg, pend, rb);
    for (int p0 = pbeg; p0 < pend; p0 += WK) {
        __syncthreads();
        wstage_store(ra, sa, ah, al);
        wstage_store(rb, sb, bh, bl);
        __syncthreads();
        if (p0 + WK < pend) {
            wstage_load(A, lda, M, m0, p0 + WK, pend, ra);
            wstage_load(Bm, ldb, Nn, n0, p0 + WK, pend, rb);
        }
#pragma unroll
        for (int s = 0; s < WK / 16; ++s) {
            half8 xhi[2], xlo[2], yhi[2], ylo[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xhi[t] = *reinterpret_cast<const half8 *>(&ah[wm + t * 32 + i32][16 * s + 8 * h]);
                xlo[t] = *reinterpret_cast<const half8 *>(&al[wm + t * 32 + i32][16 * s + 8 * h]);
                yhi[t] = *reinterpret_cast<const half8 *>(&bh[wn + t * 32 + i32][16 * s + 8 * h]);
                ylo[t] = *reinterpret_cast<const half8 *>(&bl[wn + t * 32 + i32][16 * s + 8 * h]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xhi[mt], yhi[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xhi[mt], ylo[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlo[mt], yhi[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    }
    const float un = 1.0f / (sa * sb);
    float *out = partial + (size_t)cidx * M * Nn;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wn + nt * 32 + i32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + mt * 32 + mfma32_row(r, h);
                if (m < M && n < Nn) out[(size_t)m * Nn + n] = acc[mt][nt][r] * un;
            }
        }
}

}  // namespace dgcnn

using namespace dgcnn;

static void launch_absmax(const float *x, int ld, int rows, int cols, unsigned *out, hipStream_t st)
{
    int g = ceil_div((int)(((size_t)rows * cols / 4 + 255) / 256), 8);
    g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(g), dim3(256), 0, st, x, ld, rows, cols, out);
}

extern "C" int dgcnn_rows_absmax(const float *x, int ld, int rows, int cols, uint32_t *amax, void *stream)
{
    DG_REQUIRE(x && amax, "dgcnn_rows_absmax: null pointer");
    DG_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "dgcnn_rows_absmax: bad shape rows=%d cols=%d ld=%d", rows, cols, ld);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(amax, 0, 4, st) != hipSuccess) DG_FAIL("dgcnn_rows_absmax: hipMemsetAsync failed");
    launch_absmax(x, ld, rows, cols, amax, st);
    DG_CHECK_LAUNCH("dgcnn_rows_absmax");
    return 0;
}

extern "C" int dgcnn_rows_gemm(const float *x, int ldx, const float *w, int ldw, int P, int K, int N, const uint32_t *amax_x,
                               const uint32_t *amax_w, const float *add, int rows_per_add, float *y, int ldy, void *stream)
{
    DG_REQUIRE(x && w && y && amax_x && amax_w, "dgcnn_rows_gemm: null pointer");
    DG_REQUIRE(P > 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N, "dgcnn_rows_gemm: bad shape P=%d K=%d N=%d", P, K, N);
    DG_REQUIRE((K & 31) == 0 && (ldx & 3) == 0 && (ldw & 3) == 0, "dgcnn_rows_gemm: K=%d must be a multiple of 32, ldx / ldw of 4", K);
    DG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "dgcnn_rows_gemm: operands must be 16-byte aligned");
    DG_REQUIRE(!add || rows_per_add > 0, "dgcnn_rows_gemm: rows_per_add must be positive");
    hipStream_t st = (hipStream_t)stream;
    const int ncb = ceil_div(N, GN), nrb = ceil_div(P, GM);
    static const int gk_env = getenv("DGCNN_GEMM_GK") ? atoi(getenv("DGCNN_GEMM_GK")) : 0;  // experiments
    const bool k64 = gk_env == 64 && (K & 63) == 0;  // (measured: 64-wide chunks halve the barriers but cost a workgroup per CU: 5-15 % slower)
#define DG_GEMM(ADD_, GK_)                                                                                              \
    hipLaunchKernelGGL((rows_gemm16_kernel<ADD_, GK_>), dim3(ncb * nrb), dim3(256), 0, st, x, ldx, w, ldw, P, K, N, amax_x, \
                       amax_w, add, rows_per_add, y, ldy, ncb)
    if (add) {
        if (k64) DG_GEMM(true, 64);
        else DG_GEMM(true, 32);
    } else {
        if (k64) DG_GEMM(false, 64);
        else DG_GEMM(false, 32);
    }
#undef DG_GEMM
    DG_CHECK_LAUNCH("dgcnn_rows_gemm");
    return 0;
}

// dgcnn_rows_wgrad's product on the fp16 matrix cores (called from head.hip when the operands allow 16-byte row pieces):
// partial [chunks][M][Nn] in `partial`; amax_a / amax_b: the operand maxima, or NULL -> found here into scratch[0 / 1]
namespace dgcnn {
int dgcnn_rows_wgrad16_launch(const float *a, int lda, int M, const float *b, int ldb, int Nn, int P, int chunk, float *partial,
                              const unsigned *amax_a, const unsigned *amax_b, unsigned *scratch, hipStream_t st)
{
    if (!amax_a || !amax_b) {
        if (hipMemsetAsync(scratch, 0, 16, st) != hipSuccess) return 1;
        if (!amax_a) launch_absmax(a, lda, P, M, scratch, st), amax_a = scratch;
        if (!amax_b) launch_absmax(b, ldb, P, Nn, scratch + 1, st), amax_b = scratch + 1;
    }
    const int tn = ceil_div(Nn, 128), tm = ceil_div(M, 128);
    const dim3 grid((unsigned)align_up((size_t)tm * tn * ceil_div(P, chunk), 8));
    hipLaunchKernelGGL(rows_wgrad16_kernel, grid, dim3(256), 0, st, a, lda, M, b, ldb, Nn, P, chunk, tn, amax_a, amax_b, partial);
    return 0;
}
}  // namespace dgcnn

namespace dgcnn {
// the product + column reductions of dgcnn_embed_max_fwd (head.hip): records partial [ceil(P / 128)][2][C], keys [B][C] (zeroed)
void launch_rows_gemm_colred(const float *x, int ldx, const float *w, int ldw, int P, int K, int N, const unsigned *amax_x,
                             const unsigned *amax_w, int rows_per_batch, const float *gamma, int want_sums, double *partial,
                             unsigned long long *keys, hipStream_t st)
{
    const int ncb = ceil_div(N, GN), nrb = ceil_div(P, GM);
    hipLaunchKernelGGL(rows_gemm16_colred_kernel, dim3(ncb * nrb), dim3(256), 0, st, x, ldx, w, ldw, P, K, N, amax_x, amax_w,
                       rows_per_batch, gamma, want_sums, partial, keys, ncb);
}
int rows_gemm_row_blocks(int P) { return ceil_div(P, GM); }
}  // namespace dgcnn

// ---- conv9: the classifier, y[P][NC] = x[P][K] . w[NC][K]^T with NC <= 8 classes (model.py:141,191) -----------------
// Far too few output columns for a matrix tile (the library's product takes 279 us for 57 GFLOP-less work at B = 32): a
// streaming pass instead.  A lane owns four consecutive k (one 16-byte load of the row), a wave a row at a time, eight
// rows in flight; the NC partial dot products are folded across the wave with DPP/shuffle steps.  The backward makes ONE
// pass over x and dL/dy: dL/dx rows are written as they go (16-byte stores), dL/dw accumulates in registers (NC x 4 per
// lane) and leaves one record per workgroup, added in record order by sum_partials_kernel.  K a multiple of 4, K <= 1024.
namespace dgcnn {
constexpr int CLS_MAX = 8, CLS_ROWS = 8;

template <int NC>
__global__ __launch_bounds__(256) void cls_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, int P, int K,
                                                      float *__restrict__ y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nk = K / 4;  // float4 pieces per row
    for (int p0 = (blockIdx.x * 4 + wave) * CLS_ROWS; p0 < P; p0 += gridDim.x * 4 * CLS_ROWS) {
        float acc[CLS_ROWS][NC];
#pragma unroll
        for (int r = 0; r < CLS_ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[r][c] = 0.0f;
        for (int q = lane; q < nk; q += 64) {
            float4 wv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) wv[c] = reinterpret_cast<const float4 *>(w + (size_t)c * K)[q];
            float4 xv[CLS_ROWS];
#pragma unroll
            for (int r = 0; r < CLS_ROWS; ++r) {
                const int p = p0 + r < P ? p0 + r : P - 1;
                xv[r] = reinterpret_cast<const float4 *>(x + (size_t)p * K)[q];
            }
#pragma unroll
            for (int r = 0; r < CLS_ROWS; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    acc[r][c] = __builtin_fmaf(xv[r].x, wv[c].x, acc[r][c]);
                    acc[r][c] = __builtin_fmaf(xv[r].y, wv[c].y, acc[r][c]);
                    acc[r][c] = __builtin_fmaf(xv[r].z, wv[c].z, acc[r][c]);
                    acc[r][c] = __builtin_fmaf(xv[r].w, wv[c].w, acc[r][c]);
                }
        }
#pragma unroll
        for (int r = 0; r < CLS_ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float v = acc[r][c];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                acc[r][c] = v;
            }
        if (lane < CLS_ROWS * NC) {
            const int r = lane / NC, c = lane - r * NC;
            float v = 0.0f;
#pragma unroll
            for (int rr = 0; rr < CLS_ROWS; ++rr)
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) v = (rr == r && cc == c) ? acc[rr][cc] : v;
            if (p0 + r < P) y[(size_t)(p0 + r) * NC + c] = v;
        }
    }
}

// gx[p][k] = sum_c gy[p][c] w[c][k];  partial[block][c][k] = sum over the block's rows of gy[p][c] x[p][k]
template <int NC>
__global__ __launch_bounds__(256) void cls_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ gy, int P, int K, int rows_per_block,
                                                      float *__restrict__ gx, float *__restrict__ partial)
{
    __shared__ float red[4][CLS_MAX * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nk = K / 4;
    const int pb = blockIdx.x * rows_per_block, pe = pb + rows_per_block < P ? pb + rows_per_block : P;
    for (int q0 = 0; q0 < nk; q0 += 64) {
        const int q = q0 + lane;
        const bool qv = q < nk;
        float4 wv[NC];
        float4 dw[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            wv[c] = qv ? reinterpret_cast<const float4 *>(w + (size_t)c * K)[q] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            dw[c] = float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        for (int p0 = pb + wave * 4; p0 < pe; p0 += 16) {  // a wave takes 4 consecutive rows of every 16, in row order
            float g[4][NC];
            float4 xv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + r < pe ? p0 + r : pe - 1;
#pragma unroll
                for (int c = 0; c < NC; ++c) g[r][c] = p0 + r < pe ? gy[(size_t)p * NC + c] : 0.0f;
                xv[r] = qv ? reinterpret_cast<const float4 *>(x + (size_t)p * K)[q] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 o = float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    o.x = __builtin_fmaf(g[r][c], wv[c].x, o.x);
                    o.y = __builtin_fmaf(g[r][c], wv[c].y, o.y);
                    o.z = __builtin_fmaf(g[r][c], wv[c].z, o.z);
                    o.w = __builtin_fmaf(g[r][c], wv[c].w, o.w);
                    dw[c].x = __builtin_fmaf(g[r][c], xv[r].x, dw[c].x);
                    dw[c].y = __builtin_fmaf(g[r][c], xv[r].y, dw[c].y);
                    dw[c].z = __builtin_fmaf(g[r][c], xv[r].z, dw[c].z);
                    dw[c].w = __builtin_fmaf(g[r][c], xv[r].w, dw[c].w);
                }
                if (gx && qv && p0 + r < pe) reinterpret_cast<float4 *>(gx + (size_t)(p0 + r) * K)[q] = o;
            }
        }
        // fold the four waves in wave order, one record per workgroup
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            red[wave][c * 4 + 0][lane] = dw[c].x;
            red[wave][c * 4 + 1][lane] = dw[c].y;
            red[wave][c * 4 + 2][lane] = dw[c].z;
            red[wave][c * 4 + 3][lane] = dw[c].w;
        }
        __syncthreads();
        if (wave == 0 && qv && partial) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float4 t;
                t.x = (red[0][c * 4 + 0][lane] + red[1][c * 4 + 0][lane]) + (red[2][c * 4 + 0][lane] + red[3][c * 4 + 0][lane]);
                t.y = (red[0][c * 4 + 1][lane] + red[1][c * 4 + 1][lane]) + (red[2][c * 4 + 1][lane] + red[3][c * 4 + 1][lane]);
                t.z = (red[0][c * 4 + 2][lane] + red[1][c * 4 + 2][lane]) + (red[2][c * 4 + 2][lane] + red[3][c * 4 + 2][lane]);
                t.w = (red[0][c * 4 + 3][lane] + red[1][c * 4 + 3][lane]) + (red[2][c * 4 + 3][lane] + red[3][c * 4 + 3][lane]);
                reinterpret_cast<float4 *>(partial + ((size_t)blockIdx.x * NC + c) * K)[q] = t;
            }
        }
        __syncthreads();
    }
}

void launch_sum_partials(const float *partial, int nchunk, int n, float *out, hipStream_t st);  // pointgemm.hip

static int cls_blocks(int P) { int b = ceil_div(P, 128); return b > 1024 ? 1024 : b; }
}  // namespace dgcnn

extern "C" size_t dgcnn_cls_bwd_workspace_bytes(int P, int K, int NC)
{
    if (P <= 0 || K <= 0 || NC <= 0) return 0;
    return align_up((size_t)cls_blocks(P) * NC * K * sizeof(float), 256);
}

#define DG_CLS_DISPATCH(NC_, CALL)      \
    switch (NC_) {                      \
        case 1: { constexpr int NCV = 1; CALL; } break; \
        case 2: { constexpr int NCV = 2; CALL; } break; \
        case 3: { constexpr int NCV = 3; CALL; } break; \
        case 4: { constexpr int NCV = 4; CALL; } break; \
        case 5: { constexpr int NCV = 5; CALL; } break; \
        case 6: { constexpr int NCV = 6; CALL; } break; \
        case 7: { constexpr int NCV = 7; CALL; } break; \
        default: { constexpr int NCV = 8; CALL; } break; \
    }

extern "C" int dgcnn_cls_fwd(const float *x, const float *w, int P, int K, int NC, float *y, void *stream)
{
    DG_REQUIRE(x && w && y, "dgcnn_cls_fwd: null pointer");
    DG_REQUIRE(P > 0 && K > 0 && (K & 3) == 0 && NC >= 1 && NC <= CLS_MAX, "dgcnn_cls_fwd: bad shape P=%d K=%d NC=%d (K %% 4 == 0, NC <= 8)", P, K, NC);
    DG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "dgcnn_cls_fwd: operands must be 16-byte aligned");
    int grid = ceil_div(P, 4 * CLS_ROWS);
    grid = grid > 4096 ? 4096 : grid;
    DG_CLS_DISPATCH(NC, hipLaunchKernelGGL((cls_fwd_kernel<NCV>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, P, K, y));
    DG_CHECK_LAUNCH("dgcnn_cls_fwd");
    return 0;
}

extern "C" int dgcnn_cls_bwd(const float *x, const float *w, const float *gy, int P, int K, int NC, float *gx, float *gw,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && w && gy && workspace && (gx || gw), "dgcnn_cls_bwd: null pointer");
    DG_REQUIRE(P > 0 && K > 0 && (K & 3) == 0 && NC >= 1 && NC <= CLS_MAX, "dgcnn_cls_bwd: bad shape P=%d K=%d NC=%d", P, K, NC);
    DG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && (!gx || ((uintptr_t)gx & 15) == 0),
               "dgcnn_cls_bwd: operands must be 16-byte aligned");
    DG_REQUIRE(workspace_bytes >= dgcnn_cls_bwd_workspace_bytes(P, K, NC), "dgcnn_cls_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = cls_blocks(P), rows = ceil_div(P, blocks);
    float *partial = gw ? (float *)workspace : nullptr;
    DG_CLS_DISPATCH(NC, hipLaunchKernelGGL((cls_bwd_kernel<NCV>), dim3(ceil_div(P, rows)), dim3(256), 0, st, x, w, gy, P, K, rows, gx, partial));
    if (gw) launch_sum_partials(partial, ceil_div(P, rows), NC * K, gw, st);
    DG_CHECK_LAUNCH("dgcnn_cls_bwd");
    return 0;
}
