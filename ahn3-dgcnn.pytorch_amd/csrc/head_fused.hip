// head_fused.hip -- the tail of the segmentation head as ONE forward call and ONE backward call:
//
//     u7 = cat . W7c^T + add7[b]        z7 = LReLU(BN7(u7))       (conv7 on cat(repeat(g), x1, x2, x3), model.py:179-184;
//     u8 = z7 . W8^T                    z8 = LReLU(BN8(u8))        add7 = W7g . g_b is the pooled feature's share)
//     logits = dropout(z8) . W9^T                                  (conv8, dp1, conv9: model.py:187-191)
//
// Only the PRE-activations u7 [P][512] and u8 [P][256] ever reach memory.  BatchNorm1d statistics are column sums in the
// producing product's epilogue; BatchNorm + LeakyReLU (+ dropout) are applied where the NEXT product loads its operand;
// the backward forms every BatchNorm input gradient du = sc (dv - c1 - c2 uhat) where a product loads it and every
// dv = dL/dz . LReLU' where the product that makes dL/dz stores it.  Per training step and layer this replaces the
// passes colstats, apply, (dropout,) bwd_reduce, bwd_apply over the [P][C] tensors (1.6 ms of a 13.4 ms step at B = 32)
// and the operand-magnitude passes in front of every product.
//
// Products: fp32 in / fp32 out on the fp16 matrix cores with both operands split in two halves at an exact power-of-two
// scale (hi.hi + hi.lo + lo.hi, fp32 accumulate), as head_gemm.hip -- but
//   * weights are split ONCE per step (wprep_kernel: one wave per output row, which also finds the row's own scale: the
//     scale of w is per output channel), in both orientations, as fp16 planes the products stage without vector work;
//   * the scale of a GRADIENT operand is per 128-row tile (the rows of dL/dy span orders of magnitude between points), from
//     maxima its producer leaves behind; activations take one scale per tensor (BatchNorm has normalised their columns),
//     exact because BatchNorm + LeakyReLU is monotone per column: max |z| follows from the column extrema of u.
#include <stdlib.h>

#include "dgcnn_hip.h"
#include "hsplit.h"
#include "prof.h"

namespace dgcnn {
void launch_sum_partials(const float *partial, int nchunk, int n, float *out, hipStream_t st);  // pointgemm.hip
int embed_bwd_ld(const float *cat, int ldc, const uint32_t *amax_cat, const float *catsum, const float *w, int ldw,
                 const int32_t *arg, const float *uwin, const float *stats, const float *gg, int B, int N, int F, int C,
                 int training, float slope, float *dcat, int lddcat, int dcat_acc, float *dw, float *gbn_w, float *gbn_b,
                 void *workspace, size_t workspace_bytes, void *stream);  // (model_parts.h)

namespace hf {

constexpr int GM = 128, GN = 128, GK = 32, GSTR = GK + 8;  // workgroup tile; LDS rows of 40 halves: conflict-free 16-byte reads

// ---- weights: fp16 planes + per-row scales ------------------------------------------------------------------------------
struct WPrepJob {
    const float *src;
    int ld, rows, cols, transposed;  // OUTPUT shape rows x cols; out[r][c] = transposed ? src[c ld + r] : src[r ld + c]
    _Float16 *hi, *lo;               // planes [rows][cols]
    float *scale;                    // [rows]
};
constexpr int WPREP_MAX = 4;
struct WPrepArgs {
    WPrepJob job[WPREP_MAX];
    int first_row[WPREP_MAX + 1];
    int njobs;
    unsigned long long *zero;  // side job (or NULL): nzero 64-bit words the NEXT kernel of the call takes maxima into
    size_t nzero;              // (a hipMemsetAsync of their own was a 4.8 us launch)
};

__global__ __launch_bounds__(256) void wprep_kernel(WPrepArgs a)
{
    if (a.zero)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.nzero; i += (size_t)gridDim.x * 256) a.zero[i] = 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gr = blockIdx.x * 4 + wave;
    if (gr >= a.first_row[a.njobs]) return;
    int j = 0;
    while (gr >= a.first_row[j + 1]) ++j;
    const WPrepJob J = a.job[j];
    const int r = gr - a.first_row[j];
    const size_t rs = J.transposed ? 1 : (size_t)J.ld, cs = J.transposed ? (size_t)J.ld : 1;
    const float *row = J.src + (size_t)r * rs;
    float m = 0.0f;
    for (int c = lane; c < J.cols; c += 64) m = fmaxf(m, fabsf(row[(size_t)c * cs]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float s = pow2_scale(m, H_TOP);
    unsigned *hi = reinterpret_cast<unsigned *>(J.hi + (size_t)r * J.cols), *lo = reinterpret_cast<unsigned *>(J.lo + (size_t)r * J.cols);
    for (int c = 2 * lane; c < J.cols; c += 128) {  // (cols is even)
        unsigned H, L;
        split_pair(row[(size_t)c * cs] * s, row[(size_t)(c + 1) * cs] * s, H, L);
        hi[c >> 1] = H;
        lo[c >> 1] = L;
    }
    if (lane == 0) J.scale[r] = s;
}

// ---- y = xeff . w^T with a prologue on x and an epilogue on y -----------------------------------------------------------
enum { X_RAW = 0, X_BN = 1, X_DU = 2 };
enum { E_PLAIN = 0, E_STATS = 1, E_DV = 2, E_COLRED = 3 };
struct HGemmArgs {
    const float *X;  // X_RAW: x;  X_BN: u of the previous layer;  X_DU: dv of this layer
    int ldx;
    const float *X2;  // X_DU: u of this layer
    int ldx2;
    int P, K, N;
    const _Float16 *Wh, *Wl;  // planes [N][K]
    const float *wscale;      // [N]
    // scale of x:  X_RAW / X_BN: the bits of (an upper bound of) max |xeff| at xmax[0]
    //              X_DU: bound of the row tile = sb[0] (ta[rb] + sb[1] + sb[2] tb[rb]),  ta / tb = bits of max |dv|, max |uhat|
    const unsigned *xmax, *ta, *tb;
    const unsigned *sb;  // bits of max |sc|, max |c1|, max |c2|
    // constants per k:  X_BN: pc[0] = sc, pc[1] = sh;   X_DU: pc[0..4] = mean, rstd, sc, c1, c2
    const float *pc[5];
    float slope;
    const float *add;  // E_PLAIN / E_STATS: + add[row / rows_per_batch][col] (or NULL)
    int rows_per_batch;
    float *Y;
    int ldy;
    int yacc;        // E_PLAIN: Y += (the product is the second gradient that reaches Y) instead of Y =
    double *psum;    // E_STATS: [nrb][2][N] sum, sum of squares (or NULL: extrema only)
    float *pext;     // E_STATS: [nrb][2][N] min, max
    unsigned *zero;  // E_STATS: one word zeroed by workgroup 0 (the finalize kernel behind this launch takes a maximum into it)
    // E_COLRED (conv6 -> bn6 -> LeakyReLU -> max over the points, model.py:172-175): nothing is stored; psum as E_STATS and
    // keys[b][col] <- max over the rows of batch element b of {ordered(u ^ flip), ~row}: BatchNorm + LeakyReLU is monotone per
    // column (decreasing where gamma < 0: flip), so the extreme of u is the winner of the max; the first row among equals
    unsigned long long *keys;  // [nbatch][N], zero at launch
    const float *gamma;
    // E_DV: the product is dL/dz of the PREVIOUS layer; Y = acc . LReLU'(ec_sc U + ec_sh) = its dv (U = that layer's
    // pre-activation, ec = its statistics), with the BatchNorm backward sums of that layer as column sums
    const float *U;
    int ldu;
    const float *ec[4];  // per column n: mean, rstd, sc, sh
    double *pdv;         // [nrb][2][3][N]: per batch-element slot of the tile: sum dv, sum dv uhat, sum uhat
    unsigned *oa, *ob;   // [nrb + 1]: bits of max |dv|, max |uhat| of the row tile (zero at launch); [nrb] = over all tiles
    int ncb, nrb;
    int csplit;          // hrow_kernel: workgroups per row block, each taking a contiguous share of the column blocks (0 / 1: one)
};

struct XRegs {
    float4 v[4];  // v[i]: row (t + 256 i) / 8 of the tile, columns 4 (t % 8) .. + 3 of the chunk
};
__device__ __forceinline__ void load_x(const float *__restrict__ src, int ld, int row0, int nrows, int k0, XRegs &r)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, row = f >> 3, c = (f & 7) * 4;
        const int rr = row0 + row < nrows ? row0 + row : nrows - 1;  // clamped: rows past the end are never stored
        r.v[i] = *reinterpret_cast<const float4 *>(src + (size_t)rr * ld + k0 + c);
    }
}
struct WRegs {
    u32x4 h[2], l[2];  // piece q = t + 256 j: row q / 4 of the tile, halves 8 (q % 4) .. + 7 of the chunk
};
__device__ __forceinline__ void load_w(const _Float16 *__restrict__ Wh, const _Float16 *__restrict__ Wl, int K, int col0, int N,
                                       int k0, WRegs &r)
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = threadIdx.x + 256 * j, row = q >> 2, pc = (q & 3) * 8;
        const int rr = col0 + row < N ? col0 + row : N - 1;
        r.h[j] = *reinterpret_cast<const u32x4 *>(Wh + (size_t)rr * K + k0 + pc);
        r.l[j] = *reinterpret_cast<const u32x4 *>(Wl + (size_t)rr * K + k0 + pc);
    }
}
__device__ __forceinline__ void store_w(const WRegs &r, _Float16 (*wh)[GSTR], _Float16 (*wl)[GSTR])
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = threadIdx.x + 256 * j, row = q >> 2, pc = (q & 3) * 8;
        *reinterpret_cast<u32x4 *>(&wh[row][pc]) = r.h[j];
        *reinterpret_cast<u32x4 *>(&wl[row][pc]) = r.l[j];
    }
}
__device__ __forceinline__ float f4(const float4 &v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// Prologue constants per k, folded with the operand's power-of-two scale sx (exact) and kept in LDS for the whole kernel:
//   X_BN:  xeff sx = LReLU(x (sc sx) + sh sx)                                      kt[0] = sc sx, kt[1] = sh sx
//   X_DU:  du sx = sc sx (dv - c1 - c2 rstd (u - mean)) = A dv + (Bc u + Cc)      kt[0] = A = sc sx, kt[1] = Bc = -A c2 rstd,
//                                                                                 kt[2] = Cc = A (c2 rstd mean - c1)
//          (two fused multiply-adds per element; Bc u + Cc cancels where |mean| >> sigma, at an error of 6e-8 |mean| / sigma
//          relative to the c2 term -- the correction of the gradient by its projection on uhat -- which itself is O(1) of dv)
constexpr int KT_MAX = 1024;
template <int XMODE>
__device__ __forceinline__ void fill_ktable(const HGemmArgs &a, float sx, float (*kt)[KT_MAX])
{
    for (int k = threadIdx.x; k < a.K; k += 256) {
        if (XMODE == X_BN) {
            kt[0][k] = a.pc[0][k] * sx;
            kt[1][k] = a.pc[1][k] * sx;
        } else if (XMODE == X_DU) {
            const float mean = a.pc[0][k], rstd = a.pc[1][k], A = a.pc[2][k] * sx, c1 = a.pc[3][k], c2 = a.pc[4][k];
            kt[0][k] = A;
            kt[1][k] = -A * c2 * rstd;
            kt[2][k] = A * __builtin_fmaf(c2 * rstd, mean, -c1);
        }
    }
}

template <int XMODE>
__device__ __forceinline__ void store_x(const XRegs &r, const XRegs &r2, const float (*kt)[KT_MAX], int k0, float sx, float slope,
                                        _Float16 (*xh)[GSTR], _Float16 (*xl)[GSTR])
{
    const int c = (threadIdx.x & 7) * 4;  // the thread's four k of the chunk: the same in every one of its rows
    float4 q0, q1, q2;
    if (XMODE != X_RAW) {
        q0 = *reinterpret_cast<const float4 *>(&kt[0][k0 + c]);
        q1 = *reinterpret_cast<const float4 *>(&kt[1][k0 + c]);
        if (XMODE == X_DU) q2 = *reinterpret_cast<const float4 *>(&kt[2][k0 + c]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, row = f >> 3;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = f4(r.v[i], e);
            if (XMODE == X_RAW) v[e] = x * sx;
            else if (XMODE == X_BN) v[e] = lrelu_f(__builtin_fmaf(x, f4(q0, e), f4(q1, e)), slope);
            else v[e] = __builtin_fmaf(f4(q0, e), x, __builtin_fmaf(f4(q1, e), f4(r2.v[i], e), f4(q2, e)));
        }
        u32x2 h, l;
        split4(v, h, l);
        *reinterpret_cast<u32x2 *>(&xh[row][c]) = h;
        *reinterpret_cast<u32x2 *>(&xl[row][c]) = l;
    }
}

constexpr int EST = 68;  // floats per row of the epilogue's LDS image of 64 accumulator columns (272 bytes: 16-byte aligned rows)

template <int XMODE, int EMODE, int OCC>
__global__ __launch_bounds__(256, OCC) void hgemm_kernel(const HGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[4][GM][GSTR];  // one block: the epilogue reuses it
    __shared__ __attribute__((aligned(16))) float kt[XMODE == X_RAW ? 1 : (XMODE == X_BN ? 2 : 3)][XMODE == X_RAW ? 4 : KT_MAX];
    static_assert(GM == GN && sizeof(lds) >= 4 * 32 * EST * sizeof(float) && sizeof(lds) >= 4 * 2 * 3 * GN * sizeof(double),
                  "epilogue images fit the staging area");
    _Float16(*xh)[GSTR] = lds[0], (*xl)[GSTR] = lds[1], (*wh)[GSTR] = lds[2], (*wl)[GSTR] = lds[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, each with its own L2, so the column blocks of
    // one row block -- which read the same x rows -- must be neighbours on ONE XCD, not in blockIdx: the grid (a multiple
    // of 8) is cut into 8 contiguous runs of logical tiles, run x on XCD x.  (With the column blocks adjacent in blockIdx
    // every one of them fetched its x tile from HBM: 2.8 GB instead of 1.4 for dL/dz7 at B = 32, PMC-measured, and the
    // kernel ran at the HBM roofline on re-reads.)
    const int lt = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (lt >= a.ncb * a.nrb) return;  // padding workgroups
    const int cb = lt % a.ncb, rb = lt / a.ncb;
    const int row0 = rb * GM, col0 = cb * GN;
    const int P = a.P, K = a.K, N = a.N;
    float xbound;
    if (XMODE == X_DU)
        xbound = __uint_as_float(a.sb[0]) * (__uint_as_float(a.ta[rb]) + __uint_as_float(a.sb[1]) + __uint_as_float(a.sb[2]) * __uint_as_float(a.tb[rb]));
    else
        xbound = __uint_as_float(*a.xmax);
    const float sx = pow2_scale(xbound, H_TOP);
    if (EMODE == E_STATS && lt == 0 && threadIdx.x == 0 && a.zero) *a.zero = 0u;
    if (XMODE != X_RAW) fill_ktable<XMODE>(a, sx, reinterpret_cast<float(*)[KT_MAX]>(kt));  // (visible behind the loop's first barrier)

    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    auto mfma_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < GK / 16; ++s) {
            const half8 ahi = *reinterpret_cast<const half8 *>(&xh[wave * 32 + i32][16 * s + 8 * h]);
            const half8 alo = *reinterpret_cast<const half8 *>(&xl[wave * 32 + i32][16 * s + 8 * h]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const half8 bhi = *reinterpret_cast<const half8 *>(&wh[nt * 32 + i32][16 * s + 8 * h]);
                const half8 blo = *reinterpret_cast<const half8 *>(&wl[nt * 32 + i32][16 * s + 8 * h]);
                acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
            }
        }
    };
    {
        XRegs rx, rx2;
        WRegs rw;
        load_x(a.X, a.ldx, row0, P, 0, rx);
        if (XMODE == X_DU) load_x(a.X2, a.ldx2, row0, P, 0, rx2);
        load_w(a.Wh, a.Wl, K, col0, N, 0, rw);
        for (int k0 = 0; k0 < K; k0 += GK) {
            __syncthreads();  // the previous chunk's LDS reads are done
            store_x<XMODE>(rx, rx2, reinterpret_cast<const float(*)[KT_MAX]>(kt), k0, sx, a.slope, xh, xl);
            store_w(rw, wh, wl);
            __syncthreads();
            if (k0 + GK < K) {  // next chunk in flight during the MFMAs
                load_x(a.X, a.ldx, row0, P, k0 + GK, rx);
                if (XMODE == X_DU) load_x(a.X2, a.ldx2, row0, P, k0 + GK, rx2);
                load_w(a.Wh, a.Wl, K, col0, N, k0 + GK, rw);
            }
            mfma_chunk();
        }
    }

    // ---- epilogue through LDS: the wave's 32 x 128 accumulator tile goes, 64 columns at a time, from the MFMA layout (lane =
    // column) to rows -- lane (lr, lc) holds columns lc .. lc + 3 of rows lr, lr + 4, .., lr + 28 -- so that Y, U and add move
    // in 16-byte pieces of whole 256-byte row segments
    __syncthreads();  // every wave is done with the staging area
    float *wbuf = reinterpret_cast<float *>(&lds[0][0][0]) + wave * (32 * EST);
    const int lr = lane >> 4, lc = (lane & 15) * 4;
    const float isx = 1.0f / sx;
    const int wrow0 = row0 + wave * 32;
    const int nvalid = P - wrow0 < 0 ? 0 : (P - wrow0 > 32 ? 32 : P - wrow0);  // rows of this wave inside the matrix
    // the tile spans at most two batch elements (rows_per_batch >= GM): rows >= rsplit belong to the second
    const int b0 = (EMODE == E_DV || EMODE == E_COLRED || a.add) ? row0 / a.rows_per_batch : 0;
    const int rsplit = (EMODE == E_DV || EMODE == E_COLRED || a.add) ? (b0 + 1) * a.rows_per_batch : 0x7fffffff;
    // per-lane results of the two 64-column halves, kept until the staging area is free for the cross-wave step
    double st_sum[2][4], st_sq[2][4];  // E_STATS
    float st_mn[2][4], st_mx[2][4];
    unsigned long long kb[2][4][2];    // E_COLRED: [half][column][batch-element slot] best key
    float dvs[2][4][6];                // E_DV: [half][column][slot 0: dv, dv uhat, uhat; slot 1: the same]
    float amax_dv = 0.0f, amax_uh = 0.0f;
    // (a tile inside the matrix and inside one batch element -- all but one in 55 at N = 7000 -- runs without row masks and
    // without the second batch-element slot: the compiler drops the selects and the slot's sums)
    auto halves = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int r = 0; r < 16; ++r) wbuf[mfma32_row(r, h) * EST + n2 * 32 + i32] = acc[2 * half + n2][r];
        __builtin_amdgcn_wave_barrier();
        const int cbase = col0 + half * 64 + lc;
        const bool cv = cbase < N;  // (N is a multiple of 4: the lane's four columns are inside or outside together)
        const int cc = cv ? cbase : 0;
        const float4 ws4 = *reinterpret_cast<const float4 *>(a.wscale + cc);
        const float un[4] = {isx / ws4.x, isx / ws4.y, isx / ws4.z, isx / ws4.w};
        float4 ad0 = float4{0.0f, 0.0f, 0.0f, 0.0f}, ad1 = ad0;
        if (EMODE != E_DV && a.add) {
            ad0 = *reinterpret_cast<const float4 *>(a.add + (size_t)b0 * N + cc);
            // (the second batch element's row only where the tile really reaches into it: past the last batch element
            // there is no such row to read)
            ad1 = *reinterpret_cast<const float4 *>(a.add + (size_t)(wrow0 + 31 >= rsplit && rsplit < P ? b0 + 1 : b0) * N + cc);
        }
        float4 e_mean, e_rstd, e_sc, e_sh;
        if (EMODE == E_DV) {
            e_mean = *reinterpret_cast<const float4 *>(a.ec[0] + cc);
            e_rstd = *reinterpret_cast<const float4 *>(a.ec[1] + cc);
            e_sc = *reinterpret_cast<const float4 *>(a.ec[2] + cc);
            e_sh = *reinterpret_cast<const float4 *>(a.ec[3] + cc);
        }
        float4 uv[4];  // (the rows of U, four at a time: eight in flight cost the kernel a workgroup per CU in registers)
        float S1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, S2[4] = {0.0f, 0.0f, 0.0f, 0.0f}, cshift[4];
        unsigned long long kbest[4][2] = {{0ull, 0ull}, {0ull, 0ull}, {0ull, 0ull}, {0ull, 0ull}};
        unsigned kflip[4] = {0u, 0u, 0u, 0u};
        if (EMODE == E_COLRED) {
            const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) kflip[e] = f4(gm, e) < 0.0f ? 0x80000000u : 0u;
        }
        float mn[4], mx[4];
        float ds[4][6];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mn[e] = __builtin_inff();
            mx[e] = -__builtin_inff();
#pragma unroll
            for (int q = 0; q < 6; ++q) ds[e][q] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (EMODE == E_DV && (i & 3) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rj = wrow0 + lr + 4 * (i + j);
                    uv[j] = *reinterpret_cast<const float4 *>(a.U + (size_t)(rj < P ? rj : P - 1) * a.ldu + cc);
                }
            }
            const int row = wrow0 + lr + 4 * i;
            const bool on = cv && (FULL || row < P);
            const float4 t4 = *reinterpret_cast<const float4 *>(&wbuf[(lr + 4 * i) * EST + lc]);
            float o[4];
            const bool second = !FULL && row >= rsplit;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = f4(t4, e) * un[e];
                if (EMODE == E_DV) {
                    const float u = f4(uv[i & 3], e);
                    const float dv = __builtin_fmaf(u, f4(e_sc, e), f4(e_sh, e)) > 0.0f ? g : g * a.slope;
                    const float uh = (u - f4(e_mean, e)) * f4(e_rstd, e);
                    o[e] = dv;
                    const float dvm = on ? dv : 0.0f, uhm = on ? uh : 0.0f;
                    ds[e][0] += second ? 0.0f : dvm;
                    ds[e][1] = __builtin_fmaf(second ? 0.0f : dvm, uhm, ds[e][1]);
                    ds[e][2] += second ? 0.0f : uhm;
                    ds[e][3] += second ? dvm : 0.0f;
                    ds[e][4] = __builtin_fmaf(second ? dvm : 0.0f, uhm, ds[e][4]);
                    ds[e][5] += second ? uhm : 0.0f;
                    amax_dv = fmaxf(amax_dv, fabsf(dvm));
                    amax_uh = fmaxf(amax_uh, fabsf(uhm));
                } else {
                    const float v = g + (second ? f4(ad1, e) : f4(ad0, e));
                    o[e] = v;
                    if (EMODE == E_COLRED && on) {
                        const unsigned ub = __float_as_uint(v) ^ kflip[e];
                        const unsigned ord = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);
                        const unsigned rl = (unsigned)(row - (second ? rsplit : rsplit - a.rows_per_batch));
                        const unsigned long long key = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - rl);
                        kbest[e][second ? 1 : 0] = key > kbest[e][second ? 1 : 0] ? key : kbest[e][second ? 1 : 0];
                    }
                    if (EMODE == E_STATS || EMODE == E_COLRED) {
                        // sums shifted by the value of the wave's first row (fp32 is enough for sum (v - c), sum (v - c)^2
                        // over 32 rows whatever the column's mean is; the shift is undone in fp64 below)
                        if (i == 0) cshift[e] = __shfl(v, lane & 15);
                        const float d = on ? v - cshift[e] : 0.0f;
                        S1[e] += d;
                        S2[e] = __builtin_fmaf(d, d, S2[e]);
                        mn[e] = on ? fminf(mn[e], v) : mn[e];
                        mx[e] = on ? fmaxf(mx[e], v) : mx[e];
                    }
                }
            }
            if (EMODE != E_COLRED && on) {
                float4 *yp = reinterpret_cast<float4 *>(a.Y + (size_t)row * a.ldy + cbase);
                if (EMODE == E_PLAIN && a.yacc) {
                    const float4 y0 = *yp;
                    *yp = float4{y0.x + o[0], y0.y + o[1], y0.z + o[2], y0.w + o[3]};
                } else {
                    *yp = float4{o[0], o[1], o[2], o[3]};
                }
            }
        }
        // fold the four row groups of the wave (lanes lr = 0 .. 3 of a column group): lanes with lr == 0 end up with the sums
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (EMODE == E_COLRED) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    unsigned long long kv = kbest[e][sl];
#pragma unroll
                    for (int o_ = 16; o_ <= 32; o_ <<= 1) {
                        const unsigned long long ko = __shfl_xor(kv, o_);
                        kv = ko > kv ? ko : kv;
                    }
                    kb[half][e][sl] = kv;
                }
            }
            if (EMODE == E_STATS || EMODE == E_COLRED) {
#pragma unroll
                for (int o_ = 16; o_ <= 32; o_ <<= 1) {
                    S1[e] += __shfl_xor(S1[e], o_);
                    S2[e] += __shfl_xor(S2[e], o_);
                    mn[e] = fminf(mn[e], __shfl_xor(mn[e], o_));
                    mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], o_));
                }
                const double c = (double)cshift[e], n = (double)nvalid;
                st_sum[half][e] = nvalid ? (double)S1[e] + n * c : 0.0;
                st_sq[half][e] = nvalid ? (double)S2[e] + 2.0 * c * (double)S1[e] + n * c * c : 0.0;
                st_mn[half][e] = mn[e];
                st_mx[half][e] = mx[e];
            } else if (EMODE == E_DV) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    float v = ds[e][q];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    dvs[half][e][q] = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the image is read: the next half may overwrite it
    }
    };
    if (row0 + GM <= P && row0 + GM <= rsplit) halves(std::true_type{});
    else halves(std::false_type{});
    if (EMODE == E_PLAIN) return;
    __syncthreads();  // all images are consumed: the staging area becomes the scratch of the cross-wave step
    if (EMODE == E_COLRED) {
        // keys of the tile's two batch-element slots meet in LDS, then one atomicMax per (batch element, column) and tile
        unsigned long long(*kl)[GN] = reinterpret_cast<unsigned long long(*)[GN]>(reinterpret_cast<char *>(&lds[0][0][0]) + 4 * 2 * GN * 8 + 4 * 2 * GN * 4);  // [2][GN] behind csum / cext
        for (int t = threadIdx.x; t < 2 * GN; t += 256) kl[t / GN][t % GN] = 0ull;
        __syncthreads();
        if (lr == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                        if (kb[half][e][sl]) atomicMax(&kl[sl][half * 64 + lc + e], kb[half][e][sl]);
        }
    }
    if (EMODE == E_STATS || EMODE == E_COLRED) {
        double(*csum)[2][GN] = reinterpret_cast<double(*)[2][GN]>(&lds[0][0][0]);                                            // [4][2][GN]
        float(*cext)[2][GN] = reinterpret_cast<float(*)[2][GN]>(reinterpret_cast<char *>(&lds[0][0][0]) + 4 * 2 * GN * 8);  // [4][2][GN]
        if (lr == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cl = half * 64 + lc + e;
                    csum[wave][0][cl] = st_sum[half][e];
                    csum[wave][1][cl] = st_sq[half][e];
                    cext[wave][0][cl] = st_mn[half][e];
                    cext[wave][1][cl] = st_mx[half][e];
                }
        }
        __syncthreads();
        if (threadIdx.x < GN && col0 + (int)threadIdx.x < N) {
            const int cl = threadIdx.x;
            if (a.psum) {
                a.psum[((size_t)rb * 2 + 0) * N + col0 + cl] = (csum[0][0][cl] + csum[1][0][cl]) + (csum[2][0][cl] + csum[3][0][cl]);
                a.psum[((size_t)rb * 2 + 1) * N + col0 + cl] = (csum[0][1][cl] + csum[1][1][cl]) + (csum[2][1][cl] + csum[3][1][cl]);
            }
            if (EMODE == E_STATS) {
                a.pext[((size_t)rb * 2 + 0) * N + col0 + cl] = fminf(fminf(cext[0][0][cl], cext[1][0][cl]), fminf(cext[2][0][cl], cext[3][0][cl]));
                a.pext[((size_t)rb * 2 + 1) * N + col0 + cl] = fmaxf(fmaxf(cext[0][1][cl], cext[1][1][cl]), fmaxf(cext[2][1][cl], cext[3][1][cl]));
            } else {
                unsigned long long(*kl)[GN] = reinterpret_cast<unsigned long long(*)[GN]>(reinterpret_cast<char *>(&lds[0][0][0]) + 4 * 2 * GN * 8 + 4 * 2 * GN * 4);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (kl[sl][cl]) atomicMax(a.keys + (size_t)(b0 + sl) * N + col0 + cl, kl[sl][cl]);  // a maximum: order-independent
            }
        }
    } else {  // E_DV
        double(*cdv)[2][3][GN] = reinterpret_cast<double(*)[2][3][GN]>(&lds[0][0][0]);  // [4][2][3][GN]
        __shared__ float smax[4][2];
        if (lr == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cl = half * 64 + lc + e;
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                        for (int q = 0; q < 3; ++q) cdv[wave][sl][q][cl] = (double)dvs[half][e][sl * 3 + q];
                }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            amax_dv = fmaxf(amax_dv, __shfl_xor(amax_dv, o));
            amax_uh = fmaxf(amax_uh, __shfl_xor(amax_uh, o));
        }
        if (lane == 0) {
            smax[wave][0] = amax_dv;
            smax[wave][1] = amax_uh;
        }
        __syncthreads();
        if (threadIdx.x < GN && col0 + (int)threadIdx.x < N) {
            const int cl = threadIdx.x;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    a.pdv[(((size_t)rb * 2 + sl) * 3 + q) * N + col0 + cl] =
                        (cdv[0][sl][q][cl] + cdv[1][sl][q][cl]) + (cdv[2][sl][q][cl] + cdv[3][sl][q][cl]);
        }
        if (threadIdx.x == 0) {
            const unsigned ba = __float_as_uint(fmaxf(fmaxf(smax[0][0], smax[1][0]), fmaxf(smax[2][0], smax[3][0])));
            const unsigned bb = __float_as_uint(fmaxf(fmaxf(smax[0][1], smax[1][1]), fmaxf(smax[2][1], smax[3][1])));
            atomicMax(a.oa + rb, ba);  // (non-negative floats order like their bits; a maximum: order-independent)
            atomicMax(a.ob + rb, bb);
            atomicMax(a.oa + a.nrb, ba);
            atomicMax(a.ob + a.nrb, bb);
        }
    }
}

// ---- the same product for a SHORT contraction (K = 192: conv6 and conv7 on cat, model.py:172,181), x held in registers ----
// hgemm_kernel gives every (row block, column block) pair a workgroup of its own: with six chunks of K the workgroup's life is
// its start-up (first loads from HBM with nothing to overlap), six load -> split -> LDS -> product steps with one chunk of
// look-ahead, and the epilogue; every column block splits the same x rows again (conv6: eight times).  Measured at B = 32:
// conv6 479 us, conv7 252 us, the matrix pipe a quarter busy.  Here a workgroup owns a ROW block and walks all column blocks:
// the x tile is read and split ONCE (all six chunks in flight together), each wave keeps the A fragments of its 32 rows in 96
// registers, and only W -- fp16 planes, resident in L2 -- streams through a two-stage LDS ring, the next chunk in registers,
// across the column blocks without a break.  The epilogue works in the accumulator layout (a lane owns a column of a 32-column
// block: the column sums, extrema and keys are per-lane sequences, one exchange between the two lane halves), Y leaves as
// 128-byte row segments.
constexpr int RK = 192, RCH = RK / GK;
__device__ __forceinline__ void load_w_flat(const HGemmArgs &a, int f, WRegs &r)
{
    const int t = f / RCH, kc = f - t * RCH;
    load_w(a.Wh, a.Wl, RK, t * GN, a.N, kc * GK, r);
}
template <int EMODE>
__global__ __launch_bounds__(256, 2) void hrow_kernel(const HGemmArgs a)
{
    static_assert(EMODE == E_STATS || EMODE == E_COLRED, "epilogues of the two forward products on cat");
    __shared__ __attribute__((aligned(16))) _Float16 ring[2][2][GM][GSTR];  // [stage][plane]
    __shared__ __attribute__((aligned(16))) double csum[4][2][GN];           // per wave: sum, sum of squares
    __shared__ __attribute__((aligned(16))) unsigned long long cx[4][2][GN];  // E_COLRED: keys per batch-element slot; E_STATS: min, max (floats)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    // (csplit workgroups per row block: each prepares the row block's operand itself and walks its share of the column blocks --
    // between the row-stationary form, one workgroup per row block, which leaves CUs empty below ~320 row blocks, and the tiled
    // product, which prepares the operand once per column block)
    const int S = a.csplit > 1 ? a.csplit : 1;
    const int rb = (int)blockIdx.x / S, cs = (int)blockIdx.x - rb * S, row0 = rb * GM;
    const int t0 = cs * a.ncb / S, t1 = (cs + 1) * a.ncb / S;
    const int P = a.P, N = a.N;
    const float sx = pow2_scale(__uint_as_float(*a.xmax), H_TOP);
    if (EMODE == E_STATS && blockIdx.x == 0 && threadIdx.x == 0 && a.zero) *a.zero = 0u;
    const int total = t1 * RCH;  // end of this workgroup's walk over the W chunks (flat index f = t RCH + kc)

    half8 Ah[RCH][2], Al[RCH][2];
    WRegs rw;
    {
        XRegs rx[RCH];
#pragma unroll
        for (int c = 0; c < RCH; ++c) load_x(a.X, a.ldx, row0, P, c * GK, rx[c]);
        load_w_flat(a, t0 * RCH, rw);
#pragma unroll
        for (int c = 0; c < RCH; ++c) {
            // (stage c & 1 was last read two chunks ago, before the barrier of the chunk between)
            store_x<X_RAW>(rx[c], rx[c], nullptr, c * GK, sx, a.slope, ring[c & 1][0], ring[c & 1][1]);
            __syncthreads();
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                Ah[c][s_] = *reinterpret_cast<const half8 *>(&ring[c & 1][0][wave * 32 + i32][16 * s_ + 8 * h]);
                Al[c][s_] = *reinterpret_cast<const half8 *>(&ring[c & 1][1][wave * 32 + i32][16 * s_ + 8 * h]);
            }
        }
    }
    store_w(rw, ring[0][0], ring[0][1]);  // (stage 0 held x chunk 4: read before the barrier of chunk 5)
    load_w_flat(a, t0 * RCH + 1, rw);
    __syncthreads();

    const float isx = 1.0f / sx;
    const int wrow0 = row0 + wave * 32;
    const int b0 = row0 / a.rows_per_batch, rsplit = (b0 + 1) * a.rows_per_batch;
    const bool two = rsplit < P && row0 + GM > rsplit;  // the tile reaches into a second batch element
    for (int t = t0; t < t1; ++t) {
        f32x16 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < RCH; ++kc) {
            const int f = t * RCH + kc;
            _Float16(*wh)[GSTR] = ring[kc & 1][0], (*wl)[GSTR] = ring[kc & 1][1];
            if (f + 1 < total) store_w(rw, ring[(kc + 1) & 1][0], ring[(kc + 1) & 1][1]);  // (chunk f + 1, in flight since step f - 1; that stage was read in step f - 1)
            if (f + 2 < total) load_w_flat(a, f + 2, rw);  // (an LDS write takes its data at issue: the registers are free)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const half8 bhi = *reinterpret_cast<const half8 *>(&wh[nt * 32 + i32][16 * s_ + 8 * h]);
                    const half8 blo = *reinterpret_cast<const half8 *>(&wl[nt * 32 + i32][16 * s_ + 8 * h]);
                    acc[nt] = mfma3(Ah[kc][s_], Al[kc][s_], bhi, blo, acc[nt]);
                }
            }
            __syncthreads();
        }
        // ---- epilogue of column block t in the accumulator layout: lane (i32, h) owns column nt 32 + i32, rows mfma32_row(r, h).
        // Tiles inside the matrix and inside one batch element (all but one in 55 at N = 7000) take the form without row masks
        // and with one key slot: eight vector instructions per element instead of twenty-two.
        const int col0 = t * GN;
        auto epilogue = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int cl = nt * 32 + i32, col = col0 + cl;
                const float un = isx / a.wscale[col];
                float ad0 = 0.0f, ad1 = 0.0f;
                if (EMODE == E_STATS && a.add) {
                    ad0 = a.add[(size_t)b0 * N + col];
                    ad1 = a.add[(size_t)(two ? b0 + 1 : b0) * N + col];
                }
                const unsigned kflip = (EMODE == E_COLRED && a.gamma[col] < 0.0f) ? 0x80000000u : 0u;
                float S1 = 0.0f, S2 = 0.0f, mn = __builtin_inff(), mx = -__builtin_inff(), shift = 0.0f;
                float bw0 = -__builtin_inff(), bw1 = -__builtin_inff();  // E_COLRED: best of u ^ flip per slot, and its r
                int br0 = -1, br1 = -1;
                int cnt = FULL ? 16 : 0;
                float *yrow = EMODE == E_STATS ? a.Y + (size_t)(wrow0 + 4 * h) * a.ldy + col : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wrow0 + mfma32_row(r, h);
                    const bool on = FULL || row < P, second = !FULL && row >= rsplit;
                    const float v = __builtin_fmaf(acc[nt][r], un, second ? ad1 : ad0);
                    if (EMODE == E_STATS && on) yrow[(size_t)mfma32_row(r, 0) * a.ldy] = v;
                    // sums shifted by the lane's first value (fp32 is enough for sum (v - c), sum (v - c)^2 over 16 rows whatever
                    // the column's mean is; the shift is undone in fp64 below)
                    if (r == 0) shift = v;
                    const float d = on ? v - shift : 0.0f;
                    S1 += d;
                    S2 = __builtin_fmaf(d, d, S2);
                    if (!FULL) cnt += on ? 1 : 0;
                    if (EMODE == E_STATS) {
                        mn = on ? fminf(mn, v) : mn;
                        mx = on ? fmaxf(mx, v) : mx;
                    } else {
                        // (rows ascend with r: a strict comparison keeps the first row among equals)
                        const float w_ = __uint_as_float(__float_as_uint(v) ^ kflip);
                        const bool b0_ = on && !second && w_ > bw0, b1_ = on && second && w_ > bw1;
                        bw0 = b0_ ? w_ : bw0;
                        br0 = b0_ ? r : br0;
                        if (!FULL) {
                            bw1 = b1_ ? w_ : bw1;
                            br1 = b1_ ? r : br1;
                        }
                    }
                }
                const double c = (double)shift, n = (double)cnt;
                double s1 = (double)S1 + n * c, s2 = (double)S2 + 2.0 * c * (double)S1 + n * c * c;
                s1 += __shfl_xor(s1, 32);  // the two lane halves: rows 4 h + ..
                s2 += __shfl_xor(s2, 32);
                unsigned long long k0 = 0ull, k1 = 0ull;
                if (EMODE == E_STATS) {
                    mn = fminf(mn, __shfl_xor(mn, 32));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                } else {
                    auto key_of = [&](float bw, int br, int base) -> unsigned long long {
                        if (br < 0) return 0ull;
                        const unsigned ub = __float_as_uint(bw);
                        const unsigned ord = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);
                        const unsigned rl = (unsigned)(wrow0 + (br & 3) + 8 * (br >> 2) + 4 * h - base);
                        return ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - rl);
                    };
                    k0 = key_of(bw0, br0, rsplit - a.rows_per_batch);
                    if (!FULL) k1 = key_of(bw1, br1, rsplit);
                    const unsigned long long o0 = __shfl_xor(k0, 32), o1 = FULL ? 0ull : __shfl_xor(k1, 32);
                    k0 = o0 > k0 ? o0 : k0;
                    k1 = o1 > k1 ? o1 : k1;
                }
                if (h == 0) {
                    csum[wave][0][cl] = s1;
                    csum[wave][1][cl] = s2;
                    if (EMODE == E_STATS) {
                        reinterpret_cast<float *>(&cx[wave][0][cl])[0] = mn;
                        reinterpret_cast<float *>(&cx[wave][0][cl])[1] = mx;
                    } else {
                        cx[wave][0][cl] = k0;
                        cx[wave][1][cl] = k1;
                    }
                }
            }
        };
        if (row0 + GM <= P && !two) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        __syncthreads();  // (the exchange area is rewritten one column block -- six barriers -- later)
        if (threadIdx.x < GN) {
            const int cl = threadIdx.x, col = col0 + cl;
            if (a.psum) {
                a.psum[((size_t)rb * 2 + 0) * N + col] = (csum[0][0][cl] + csum[1][0][cl]) + (csum[2][0][cl] + csum[3][0][cl]);
                a.psum[((size_t)rb * 2 + 1) * N + col] = (csum[0][1][cl] + csum[1][1][cl]) + (csum[2][1][cl] + csum[3][1][cl]);
            }
            if (EMODE == E_STATS) {
                float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll
                for (int w_ = 0; w_ < 4; ++w_) {
                    lo = fminf(lo, reinterpret_cast<const float *>(&cx[w_][0][cl])[0]);
                    hi = fmaxf(hi, reinterpret_cast<const float *>(&cx[w_][0][cl])[1]);
                }
                a.pext[((size_t)rb * 2 + 0) * N + col] = lo;
                a.pext[((size_t)rb * 2 + 1) * N + col] = hi;
            } else {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    unsigned long long kv = cx[0][sl][cl];
#pragma unroll
                    for (int w_ = 1; w_ < 4; ++w_) kv = cx[w_][sl][cl] > kv ? cx[w_][sl][cl] : kv;
                    if (kv) atomicMax(a.keys + (size_t)(b0 + sl) * N + col, kv);  // a maximum: order-independent
                }
            }
        }
    }
}
// 0: the tiled product; otherwise the row-stationary form with that many workgroups per row block (HGemmArgs::csplit)
static int row_form(const HGemmArgs &a)
{
    // DGCNN_HF_ROW: 0 = never, 2 = whenever the shape allows (tests of the small, ragged shapes), otherwise by size
    static const int mode = getenv("DGCNN_HF_ROW") ? atoi(getenv("DGCNN_HF_ROW")) : 1;
    static const int split_lo = getenv("DGCNN_HF_ROW_SPLIT") ? atoi(getenv("DGCNN_HF_ROW_SPLIT")) : 160;  // experiments: 0 = no split form
    const bool off = mode == 0;
    if (mode == 2) return a.K == RK && a.N % GN == 0 ? 1 : 0;
    // (two workgroups per row block between 160 and 320 row blocks -- B = 3 .. 5 at N = 7000 --, when there are at least two
    // column blocks to share)
    if (!off && split_lo > 0 && a.K == RK && a.N % GN == 0 && a.nrb >= split_lo && a.nrb < 320 && a.ncb >= 2) return 2;
    // (four between 80 and 160 -- B = 2 --: measured below)
    if (!off && split_lo > 0 && a.K == RK && a.N % GN == 0 && a.nrb >= split_lo / 2 && a.nrb < split_lo && a.ncb >= 4) return 4;
    // (below about one and a half row blocks per CU the walk leaves CUs with one workgroup or none, and the product with a
    // workgroup per (row block, column block) is the faster one again: training step at B = 4, 219 row blocks: 2.134 ms
    // against 2.146; at B = 8, 438: 3.282 against 3.254)
    return !off && a.K == RK && a.N % GN == 0 && a.nrb >= 320 ? 1 : 0;
}

// ---- weight gradient dW[M][Nn] = sum_p aeff[p][m] beff[p][n] with prologues on both operands ---------------------------
// (the structure of rows_wgrad16_kernel, head_gemm.hip: stages of 32 points, loaded in 16-byte row pieces, transformed,
// split and written to LDS TRANSPOSED as fp16 planes [column][point]; one 128 x 128 tile of dW over one chunk of the
// points per workgroup, wave = a 64 x 64 quadrant; partial tiles are added in chunk order by sum_partials_kernel)
enum { A_RAW = 0, A_DU = 2, B_RAW = 0, B_BN = 1 };
constexpr int WK = 32, WSTR = WK + 8;
struct HWgradArgs {
    const float *A;  // A_DU: dv
    int lda;
    const float *A2;  // A_DU: u
    int lda2, M;
    const float *Bm;  // B_BN: u of the previous layer;  B_RAW: the layer's input
    int ldb, Nn, P, chunk, ntiles_n;
    const float *ac[5];  // A_DU, per column m: mean, rstd, sc, c1, c2
    const float *bc[2];  // B_BN, per column n: sc, sh
    float slope;
    const unsigned *ta, *tb, *sb;  // A_DU: scale bound = sb[0] (ta[gi] + sb[1] + sb[2] tb[gi])
    int gi;
    const unsigned *amax_b;  // bits of max |beff|
    float *partial;
};
// (Measured and dropped: thread (cq = t / 8, pq = t % 8) instead of (t % 32, t / 32), which also takes the transposed 8-byte
// LDS writes of a wave to a 2-way bank conflict, but makes a wave's loads 8 points x 128 bytes instead of 2 points x 512:
// 462 us against 429 / 409.  The row permutation wrow() below gets the same without touching the loads.)
struct WStage {
    float4 v[4];  // v[i] = row p0 + 4 pq + i, columns c0 + 4 cq .. + 3
};
__device__ __forceinline__ void wstage_load(const float *__restrict__ src, int ld, int ncols, int c0, int p0, int pend, WStage &r)
{
    const int cq = threadIdx.x & 31, pq = threadIdx.x >> 5;
    const int col = c0 + 4 * cq < ncols ? c0 + 4 * cq : 0;  // columns past the matrix: any valid address (never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + 4 * pq + i;
        r.v[i] = *reinterpret_cast<const float4 *>(src + (size_t)(p < pend ? p : pend - 1) * ld + col);
    }
}
// LDS row of column j of the 128-column tile: 32 (j & 3) + (j >> 2).  A thread holds four consecutive columns (one 16-byte
// load), so with row = column the 32 lanes of a store -- columns 4 cq + c -- are 320 bytes apart, i.e. on TWO 8-byte bank
// positions: a 16-way conflict on every transposed write (SQ_LDS_BANK_CONFLICT was 78 % of these kernels' LDS cycles).
// Here consecutive lanes write consecutive rows (80 bytes apart: eight positions, 2-way), the loads stay whole 512-byte row
// pieces, and the matrix instructions read rows as before: only the column a tile row STANDS FOR changes (wcol).
__device__ __forceinline__ constexpr int wrow(int j) { return 32 * (j & 3) + (j >> 2); }
__device__ __forceinline__ constexpr int wcol(int r) { return 4 * (r & 31) + (r >> 5); }
// the four columns of this thread, transformed, for the stage's four points -> transposed, split, stored
__device__ __forceinline__ void wstage_store(const float (&t)[4][4] /* [column][point] */, _Float16 (*hi)[WSTR], _Float16 (*lo)[WSTR])
{
    const int cq = threadIdx.x & 31, pq = threadIdx.x >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32x2 h, l;
        split4(t[c], h, l);
        *reinterpret_cast<u32x2 *>(&hi[wrow(4 * cq + c)][4 * pq]) = h;
        *reinterpret_cast<u32x2 *>(&lo[wrow(4 * cq + c)][4 * pq]) = l;
    }
}

template <int AMODE, int BMODE, int OCC>
__global__ __launch_bounds__(256, OCC) void hwgrad_kernel(const HWgradArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 ah[128][WSTR], al[128][WSTR], bh[128][WSTR], bl[128][WSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int M = a.M, Nn = a.Nn, P = a.P;
    const int tiles = a.ntiles_n * ((M + 127) / 128);
    const int v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // gridDim.x is a multiple of 8: a chunk's tiles on ONE XCD
    const int cidx = v / tiles, tile = v - cidx * tiles;
    if (cidx * a.chunk >= P) return;  // padding workgroups
    const int tm = tile / a.ntiles_n, tn = tile - tm * a.ntiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int pbeg = cidx * a.chunk, pend = pbeg + a.chunk < P ? pbeg + a.chunk : P;
    float abound;
    if (AMODE == A_DU)
        abound = __uint_as_float(a.sb[0]) * (__uint_as_float(a.ta[a.gi]) + __uint_as_float(a.sb[1]) + __uint_as_float(a.sb[2]) * __uint_as_float(a.tb[a.gi]));
    else
        abound = __uint_as_float(*a.ta);  // (A_RAW: ta points at the bits of max |a|)
    const float sa = pow2_scale(abound, H_TOP), sb = pow2_scale(__uint_as_float(*a.amax_b), H_TOP);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int cq = threadIdx.x & 31, pq = threadIdx.x >> 5;

    // per-thread column constants (the thread's four columns are the same in every stage), folded with the operand scales
    // as in fill_ktable:  du sa = A dv + (Bc u + Cc);  z sb = LReLU(u (sc sb) + sh sb)
    float qa[3][4], qb[2][4];
    {
        const int cm = m0 + 4 * cq < M ? m0 + 4 * cq : 0, cn = n0 + 4 * cq < Nn ? n0 + 4 * cq : 0;
        if (AMODE == A_DU) {
            const float4 mean = *reinterpret_cast<const float4 *>(a.ac[0] + cm), rstd = *reinterpret_cast<const float4 *>(a.ac[1] + cm);
            const float4 sc = *reinterpret_cast<const float4 *>(a.ac[2] + cm), c1 = *reinterpret_cast<const float4 *>(a.ac[3] + cm);
            const float4 c2 = *reinterpret_cast<const float4 *>(a.ac[4] + cm);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float A = f4(sc, c) * sa;
                qa[0][c] = A;
                qa[1][c] = -A * f4(c2, c) * f4(rstd, c);
                qa[2][c] = A * __builtin_fmaf(f4(c2, c) * f4(rstd, c), f4(mean, c), -f4(c1, c));
            }
        }
        if (BMODE == B_BN) {
            const float4 sc = *reinterpret_cast<const float4 *>(a.bc[0] + cn), sh = *reinterpret_cast<const float4 *>(a.bc[1] + cn);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                qb[0][c] = f4(sc, c) * sb;
                qb[1][c] = f4(sh, c) * sb;
            }
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a_][b_][r] = 0.0f;

    WStage ra, ra2, rb;
    wstage_load(a.A, a.lda, M, m0, pbeg, pend, ra);
    if (AMODE == A_DU) wstage_load(a.A2, a.lda2, M, m0, pbeg, pend, ra2);
    wstage_load(a.Bm, a.ldb, Nn, n0, pbeg, pend, rb);
    for (int p0 = pbeg; p0 < pend; p0 += WK) {
        __syncthreads();
        auto stage = [&](auto full_c) {  // (only a chunk's last stage can reach past its end: the others run without masks)
            constexpr bool FULL = decltype(full_c)::value;
            float ta_[4][4], tb_[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool on = FULL || p0 + 4 * pq + i < pend;  // points past the chunk add nothing
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = f4(ra.v[i], c);
                    x = AMODE == A_DU ? __builtin_fmaf(qa[0][c], x, __builtin_fmaf(qa[1][c], f4(ra2.v[i], c), qa[2][c])) : x * sa;
                    ta_[c][i] = on ? x : 0.0f;
                    float y = f4(rb.v[i], c);
                    y = BMODE == B_BN ? lrelu_f(__builtin_fmaf(y, qb[0][c], qb[1][c]), a.slope) : y * sb;
                    tb_[c][i] = on ? y : 0.0f;
                }
            }
            wstage_store(ta_, ah, al);
            wstage_store(tb_, bh, bl);
        };
        if (p0 + WK <= pend) stage(std::true_type{});
        else stage(std::false_type{});
        __syncthreads();
        if (p0 + WK < pend) {
            wstage_load(a.A, a.lda, M, m0, p0 + WK, pend, ra);
            if (AMODE == A_DU) wstage_load(a.A2, a.lda2, M, m0, p0 + WK, pend, ra2);
            wstage_load(a.Bm, a.ldb, Nn, n0, p0 + WK, pend, rb);
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
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma3(xhi[mt], xlo[mt], yhi[nt], ylo[nt], acc[mt][nt]);
        }
    }
    const float un = 1.0f / (sa * sb);
    float *out = a.partial + (size_t)cidx * M * Nn;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wcol(wn + nt * 32 + i32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wcol(wm + mt * 32 + mfma32_row(r, h));
                if (m < M && n < Nn) out[(size_t)m * Nn + n] = acc[mt][nt][r] * un;
            }
        }
}


// ---- the classifier on u8 (conv9 after bn8 + LeakyReLU + dropout, model.py:187-191) ------------------------------------
// Five output columns are no matrix tile: streaming passes, as dgcnn_cls_fwd / dgcnn_cls_bwd (head_gemm.hip) -- but the
// input is the PRE-activation u8: BatchNorm, LeakyReLU and the dropout mask are applied to the loaded row in registers.
struct ClsArgs {
    const float *u;
    int P, K, nrb, nblk;
    const float *mean, *rstd, *sc, *sh;  // stats of bn8
    const float *w;                      // [NC][K]
    float slope, inv_keep;
    unsigned keep16;
    unsigned long long seed;
    float *y;         // forward: logits [P][NC]
    const float *gy;  // backward: dL/dlogits [P][NC]
    float *dv;        // backward: [P][K] = dL/dz8 . LReLU'
    float *pdw;       // backward: [nblk][NC][K]
    double *pbn;      // backward: [nblk][2][K]: sum dv, sum dv uhat
    unsigned *oa, *ob;  // backward: [nrb + 1] bits of max |dv|, max |uhat| per 128-row tile (zero at launch), [nrb] = over all
};
constexpr int CLS_ROWS = 8, CLS_MAX = 8;

// sum over the 16 lanes of a DPP row (every lane ends up with the sum): xor 1, xor 2 inside the quads, then the mirrored
// half-row and the mirrored row
__device__ __forceinline__ float row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// Sixteen lanes per row (lane part p of a DPP row takes the 16-byte pieces p, p + 16, .. of its row: 256 contiguous bytes
// per row and instruction), four rows per wave instruction, sixteen rows per iteration; the NC dot products of a row are
// folded over its sixteen lanes with four DPP additions each.  (The first form -- 64 lanes on one row, eight rows in flight
// -- spent its time in 240 wave-wide shuffle reductions per eight rows: 98 us for 229 MB at B = 32.)
constexpr int CLS_GROUPS = 4;  // groups of four rows per iteration
template <int NC, bool DROP>
__global__ __launch_bounds__(256) void cls_fwd_fused_kernel(const ClsArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane & 15, rsub = lane >> 4;
    const int nk = a.K / 4, P = a.P;
    const int rows_it = 4 * CLS_GROUPS;
    for (int p0 = (blockIdx.x * 4 + wave) * rows_it; p0 < P; p0 += gridDim.x * 4 * rows_it) {
        float acc[CLS_GROUPS][NC];
#pragma unroll
        for (int g = 0; g < CLS_GROUPS; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[g][c] = 0.0f;
        for (int q = part; q < nk; q += 16) {
            float4 uv[CLS_GROUPS];
#pragma unroll
            for (int g = 0; g < CLS_GROUPS; ++g) {
                const int p = p0 + 4 * g + rsub < P ? p0 + 4 * g + rsub : P - 1;
                uv[g] = reinterpret_cast<const float4 *>(a.u + (size_t)p * a.K)[q];
            }
            float4 wv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) wv[c] = reinterpret_cast<const float4 *>(a.w + (size_t)c * a.K)[q];
            const float4 sc = reinterpret_cast<const float4 *>(a.sc)[q], sh = reinterpret_cast<const float4 *>(a.sh)[q];
#pragma unroll
            for (int g = 0; g < CLS_GROUPS; ++g) {
                const int p = p0 + 4 * g + rsub < P ? p0 + 4 * g + rsub : P - 1;
                const unsigned long long mix = DROP ? drop_mix(a.seed, (unsigned long long)p * nk + q) : 0ull;
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = lrelu_f(__builtin_fmaf(f4(uv[g], e), f4(sc, e), f4(sh, e)), a.slope);
                    d[e] = DROP ? (drop_keep(mix, e, a.keep16) ? z * a.inv_keep : 0.0f) : z;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    acc[g][c] = __builtin_fmaf(d[0], wv[c].x, acc[g][c]);
                    acc[g][c] = __builtin_fmaf(d[1], wv[c].y, acc[g][c]);
                    acc[g][c] = __builtin_fmaf(d[2], wv[c].z, acc[g][c]);
                    acc[g][c] = __builtin_fmaf(d[3], wv[c].w, acc[g][c]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < CLS_GROUPS; ++g) {
            float mine = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float t = row16_sum(acc[g][c]);
                mine = part == c ? t : mine;
            }
            const int p = p0 + 4 * g + rsub;
            if (part < NC && p < P) a.y[(size_t)p * NC + part] = mine;
        }
    }
}

// backward of the classifier in ONE pass over u8 and dL/dlogits:
//   dL/dd = gy . W9,  dL/dz8 = dL/dd . mask / (1 - p),  dv = dL/dz8 . LReLU'(v8)          -> dv [P][K]
//   dW9 += gy^T d  (d = dropout(z8), recomputed)                                           -> one record per workgroup
//   BatchNorm-8 backward sums: sum dv, sum dv uhat8 (fp32 per 128-row tile, fp64 across)   -> one record per workgroup
//   max |dv|, max |uhat8| per 128-row tile (the products behind take dv's scale bound from them)
// A workgroup walks 128-row tiles blk, blk + nblk, ...; a wave takes 4 consecutive rows of every 16, in row order.
template <int NC, bool DROP>
__global__ __launch_bounds__(256) void cls_bwd_fused_kernel(const ClsArgs a)
{
    __shared__ float red[4][CLS_MAX * 4][64];
    __shared__ double redd[4][8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nk = a.K / 4, P = a.P;
    for (int q0 = 0; q0 < nk; q0 += 64) {
        const int q = q0 + lane;
        const bool qv = q < nk;
        const int qq = qv ? q : 0;
        float4 wv[NC], dw[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            wv[c] = qv ? reinterpret_cast<const float4 *>(a.w + (size_t)c * a.K)[qq] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            dw[c] = float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const float4 mean = reinterpret_cast<const float4 *>(a.mean)[qq], rstd = reinterpret_cast<const float4 *>(a.rstd)[qq];
        const float4 sc = reinterpret_cast<const float4 *>(a.sc)[qq], sh = reinterpret_cast<const float4 *>(a.sh)[qq];
        double sdv[4] = {0.0, 0.0, 0.0, 0.0}, sdu[4] = {0.0, 0.0, 0.0, 0.0};
        for (int tile = blockIdx.x; tile < a.nrb; tile += a.nblk) {
            const int pb = tile * 128, pe = pb + 128 < P ? pb + 128 : P;
            float tdv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, tdu[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float amax_dv = 0.0f, amax_uh = 0.0f;
            for (int p0 = pb + wave * 4; p0 < pe; p0 += 16) {
                float g[4][NC];
                float4 uv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = p0 + r < pe ? p0 + r : pe - 1;
#pragma unroll
                    for (int c = 0; c < NC; ++c) g[r][c] = p0 + r < pe ? a.gy[(size_t)p * NC + c] : 0.0f;
                    uv[r] = reinterpret_cast<const float4 *>(a.u + (size_t)p * a.K)[qq];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = qv && p0 + r < pe;
                    const int p = p0 + r < pe ? p0 + r : pe - 1;
                    const unsigned long long mix = DROP ? drop_mix(a.seed, (unsigned long long)p * nk + qq) : 0ull;
                    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        o[0] = __builtin_fmaf(g[r][c], wv[c].x, o[0]);
                        o[1] = __builtin_fmaf(g[r][c], wv[c].y, o[1]);
                        o[2] = __builtin_fmaf(g[r][c], wv[c].z, o[2]);
                        o[3] = __builtin_fmaf(g[r][c], wv[c].w, o[3]);
                    }
                    float d[4], dvv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = f4(uv[r], e);
                        const float v = __builtin_fmaf(u, f4(sc, e), f4(sh, e));
                        const float z = lrelu_f(v, a.slope);
                        const bool keep = DROP ? drop_keep(mix, e, a.keep16) : true;
                        d[e] = keep ? (DROP ? z * a.inv_keep : z) : 0.0f;
                        const float gz = keep ? (DROP ? o[e] * a.inv_keep : o[e]) : 0.0f;
                        dvv[e] = on ? (v > 0.0f ? gz : gz * a.slope) : 0.0f;
                        const float uh = (u - f4(mean, e)) * f4(rstd, e);
                        tdv[e] += dvv[e];
                        tdu[e] = __builtin_fmaf(dvv[e], uh, tdu[e]);
                        amax_dv = fmaxf(amax_dv, fabsf(dvv[e]));
                        amax_uh = on ? fmaxf(amax_uh, fabsf(uh)) : amax_uh;
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {  // (g = 0 in rows past the tile)
                        dw[c].x = __builtin_fmaf(g[r][c], d[0], dw[c].x);
                        dw[c].y = __builtin_fmaf(g[r][c], d[1], dw[c].y);
                        dw[c].z = __builtin_fmaf(g[r][c], d[2], dw[c].z);
                        dw[c].w = __builtin_fmaf(g[r][c], d[3], dw[c].w);
                    }
                    if (on) reinterpret_cast<float4 *>(a.dv + (size_t)(p0 + r) * a.K)[q] = float4{dvv[0], dvv[1], dvv[2], dvv[3]};
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sdv[e] += (double)tdv[e];
                sdu[e] += (double)tdu[e];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                amax_dv = fmaxf(amax_dv, __shfl_xor(amax_dv, o));
                amax_uh = fmaxf(amax_uh, __shfl_xor(amax_uh, o));
            }
            if (lane == 0) {
                atomicMax(a.oa + tile, __float_as_uint(amax_dv));
                atomicMax(a.ob + tile, __float_as_uint(amax_uh));
                atomicMax(a.oa + a.nrb, __float_as_uint(amax_dv));
                atomicMax(a.ob + a.nrb, __float_as_uint(amax_uh));
            }
        }
        // fold the four waves in wave order: one record per workgroup
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            red[wave][c * 4 + 0][lane] = dw[c].x;
            red[wave][c * 4 + 1][lane] = dw[c].y;
            red[wave][c * 4 + 2][lane] = dw[c].z;
            red[wave][c * 4 + 3][lane] = dw[c].w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            redd[wave][e][lane] = sdv[e];
            redd[wave][4 + e][lane] = sdu[e];
        }
        __syncthreads();
        if (wave == 0 && qv) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float4 t;
                t.x = (red[0][c * 4 + 0][lane] + red[1][c * 4 + 0][lane]) + (red[2][c * 4 + 0][lane] + red[3][c * 4 + 0][lane]);
                t.y = (red[0][c * 4 + 1][lane] + red[1][c * 4 + 1][lane]) + (red[2][c * 4 + 1][lane] + red[3][c * 4 + 1][lane]);
                t.z = (red[0][c * 4 + 2][lane] + red[1][c * 4 + 2][lane]) + (red[2][c * 4 + 2][lane] + red[3][c * 4 + 2][lane]);
                t.w = (red[0][c * 4 + 3][lane] + red[1][c * 4 + 3][lane]) + (red[2][c * 4 + 3][lane] + red[3][c * 4 + 3][lane]);
                reinterpret_cast<float4 *>(a.pdw + ((size_t)blockIdx.x * NC + c) * a.K)[q] = t;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a.pbn[((size_t)blockIdx.x * 2 + 0) * a.K + 4 * q + e] = (redd[0][e][lane] + redd[1][e][lane]) + (redd[2][e][lane] + redd[3][e][lane]);
                a.pbn[((size_t)blockIdx.x * 2 + 1) * a.K + 4 * q + e] =
                    (redd[0][4 + e][lane] + redd[1][4 + e][lane]) + (redd[2][4 + e][lane] + redd[3][4 + e][lane]);
            }
        }
        __syncthreads();
    }
}

// ---- finalizers: fold per-workgroup records in record order (16 waves take every 16th record, 8 in flight; the wave sums
// are added in wave order: the association depends on the record count only) ---------------------------------------------
constexpr int FW = 16;

// forward: BatchNorm1d batch statistics (or the running ones) -> stats [4][C] = mean, rstd, sc, sh; running statistics
// updated when training; zmax <- bits of max |LReLU(BN(u))| over the whole tensor, exact: per column the activation is
// monotone in u, so its extremes are the images of the column's extremes.
// gridDim.y > 1 (fold_psum / fold_pext given): the launch only FOLDS -- slice y takes a contiguous run of the records and
// leaves one folded record in the same formats -- and a second launch (gridDim.y = 1) over the gridDim.y folded records
// finalizes: one workgroup per 64 columns reads 2.7 MB of records at B = 32 at the bandwidth of ONE CU (55 us); eight
// slices read an eighth each.  The association stays a function of the record count.  pext may be NULL (no extrema).
__global__ __launch_bounds__(64 * FW) void stats_finalize_kernel(const double *__restrict__ psum, const float *__restrict__ pext,
                                                                 int nrec_all, int P, int C, int training,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 float *__restrict__ rmean, float *__restrict__ rvar, float eps,
                                                                 float momentum, float slope, float *__restrict__ stats,
                                                                 unsigned *__restrict__ zmax, double *__restrict__ fold_psum,
                                                                 float *__restrict__ fold_pext)
{
    __shared__ double red[FW][2][64];
    __shared__ float rex[FW][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < C;
    const bool fold_only = gridDim.y > 1;
    const int per = (nrec_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int rbeg = (int)blockIdx.y * per, nrec = rbeg + per < nrec_all ? rbeg + per : nrec_all;
    double a0 = 0.0, a1 = 0.0;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (live) {
        int g = rbeg + wave;
        for (; g + 7 * FW < nrec; g += 8 * FW) {
            double va[8], vb[8];
            float ea[8], eb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = (size_t)(g + u * FW) * 2 * C + c;
                va[u] = training ? psum[o] : 0.0;
                vb[u] = training ? psum[o + C] : 0.0;
                ea[u] = pext ? pext[o] : 0.0f;
                eb[u] = pext ? pext[o + C] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 += va[u];
                a1 += vb[u];
                mn = fminf(mn, ea[u]);
                mx = fmaxf(mx, eb[u]);
            }
        }
        if (g < nrec) {  // the last, partial batch: the same eight loads in flight, past-the-end records masked (one record per
                         // round trip here was six dependent trips at 219 records: 9 -> 5 us per launch, three per step)
            double va[8], vb[8];
            float ea[8], eb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gu = g + u * FW < nrec ? g + u * FW : g;
                const size_t o = (size_t)gu * 2 * C + c;
                va[u] = training ? psum[o] : 0.0;
                vb[u] = training ? psum[o + C] : 0.0;
                ea[u] = pext ? pext[o] : 0.0f;
                eb[u] = pext ? pext[o + C] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g + u * FW < nrec) {
                    a0 += va[u];
                    a1 += vb[u];
                    mn = fminf(mn, ea[u]);
                    mx = fmaxf(mx, eb[u]);
                }
            }
        }
    }
    red[wave][0][lane] = a0;
    red[wave][1][lane] = a1;
    rex[wave][0][lane] = mn;
    rex[wave][1][lane] = mx;
    __syncthreads();
    if (wave != 0) return;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int w = 0; w < FW; ++w) {
        s += red[w][0][lane];
        ss += red[w][1][lane];
        mn = fminf(mn, rex[w][0][lane]);
        mx = fmaxf(mx, rex[w][1][lane]);
    }
    if (fold_only) {
        if (live) {
            const size_t o = (size_t)blockIdx.y * 2 * C + c;
            if (training) {
                fold_psum[o] = s;
                fold_psum[o + C] = ss;
            }
            if (pext) {
                fold_pext[o] = mn;
                fold_pext[o + C] = mx;
            }
        }
        return;
    }
    float zm = 0.0f;
    if (live) {
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
        const float sc = gamma[c] * rstd, sh = beta[c] - mean * sc;
        stats[0 * (size_t)C + c] = mean;
        stats[1 * (size_t)C + c] = rstd;
        stats[2 * (size_t)C + c] = sc;
        stats[3 * (size_t)C + c] = sh;
        zm = fmaxf(fabsf(lrelu_f(__builtin_fmaf(mn, sc, sh), slope)), fabsf(lrelu_f(__builtin_fmaf(mx, sc, sh), slope)));
        if (!(zm <= 3.0e38f)) zm = 0.0f;  // (a non-finite activation: any scale is as good)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) zm = fmaxf(zm, __shfl_xor(zm, o));
    if (lane == 0 && zmax) atomicMax(zmax, __float_as_uint(zm));
}

// BatchNorm backward constants of one column from the two sums; sb <- bits of max |sc|, max |c1|, max |c2| (zero at launch)
__device__ __forceinline__ void bn_bwd_consts(double sdv, double sdu, int P, int training, int c, bool live, float sc,
                                              float *__restrict__ gw, float *__restrict__ gb, float *__restrict__ cc, int C,
                                              unsigned *__restrict__ sb, float &c1, float &c2)
{
    c1 = live && training ? (float)(sdv / P) : 0.0f;
    c2 = live && training ? (float)(sdu / P) : 0.0f;
    if (live) {
        gb[c] = (float)sdv;
        gw[c] = (float)sdu;
        cc[c] = c1;
        cc[C + c] = c2;
    }
    float m0 = live ? fabsf(sc) : 0.0f, m1 = fabsf(c1), m2 = fabsf(c2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
        m2 = fmaxf(m2, __shfl_xor(m2, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(sb + 0, __float_as_uint(m0));
        atomicMax(sb + 1, __float_as_uint(m1));
        atomicMax(sb + 2, __float_as_uint(m2));
    }
}

// backward, from the classifier pass: records pbn [nrec][2][C] -> gw = sum dv uhat, gb = sum dv, cc = [c1 | c2]
__global__ __launch_bounds__(64 * FW) void bwd_finalize_kernel(const double *__restrict__ pbn, int nrec, int P, int C,
                                                               int training, const float *__restrict__ stats,
                                                               float *__restrict__ gw, float *__restrict__ gb,
                                                               float *__restrict__ cc, unsigned *__restrict__ sb)
{
    __shared__ double red[FW][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < C;
    double a0 = 0.0, a1 = 0.0;
    if (live) {
        int g = wave;
        for (; g + 7 * FW < nrec; g += 8 * FW) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = pbn[(size_t)(g + u * FW) * 2 * C + c];
                vb[u] = pbn[(size_t)(g + u * FW) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 += va[u];
                a1 += vb[u];
            }
        }
        if (g < nrec) {  // the last, partial batch with its loads in flight together (past-the-end records masked)
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gu = g + u * FW < nrec ? g + u * FW : g;
                va[u] = pbn[(size_t)gu * 2 * C + c];
                vb[u] = pbn[(size_t)gu * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g + u * FW < nrec) {
                    a0 += va[u];
                    a1 += vb[u];
                }
            }
        }
    }
    red[wave][0][lane] = a0;
    red[wave][1][lane] = a1;
    __syncthreads();
    if (wave != 0) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int w = 0; w < FW; ++w) {
        s0 += red[w][0][lane];
        s1 += red[w][1][lane];
    }
    float c1, c2;
    bn_bwd_consts(s0, s1, P, training, c, live, live ? stats[2 * (size_t)C + c] : 0.0f, gw, gb, cc, C, sb, c1, c2);
}

// backward, from a product's E_DV epilogue: records pdv [nrb][2 slots][3][C] (sum dv, sum dv uhat, sum uhat per
// batch-element slot of a 128-row tile).  Wave w adds up batch elements w, w + 16, ... (tiles in order), the totals are the
// wave sums in wave order; then gw, gb, cc as above and the gradient of the per-batch-element row that was added to u:
//   gadd[b][c] = sum over the rows of b of du = sc (sum_b dv - N c1 - c2 sum_b uhat).
constexpr int SLOT_B = 4;  // batch elements per wave: B <= 64
// Two launches when there are many tiles (bsum given): gridDim.y = B -- workgroup (column group, b) adds up batch element b's
// tiles into bsum [B][3][C] (the 128 waves of the one-launch form read 40 MB at B = 32: 64 us) -- then gridDim.y = 1 takes the
// per-batch-element sums from bsum.
__global__ __launch_bounds__(64 * FW) void bwd_finalize_slots_kernel(const double *__restrict__ pdv, int nrb, int B, int N, int C,
                                                                     int training, const float *__restrict__ stats,
                                                                     float *__restrict__ gw, float *__restrict__ gb,
                                                                     float *__restrict__ cc, unsigned *__restrict__ sb,
                                                                     float *__restrict__ gadd, double *__restrict__ bsum)
{
    __shared__ double red[FW][2][64];
    __shared__ float scc[2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < C;
    const bool first_level = gridDim.y > 1;  // (launched with 64 threads: wave 0 only)
    double bdv[SLOT_B], bdu[SLOT_B], buh[SLOT_B];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < SLOT_B; ++i) {
        const int b = first_level ? (i == 0 ? (int)blockIdx.y : B) : wave + i * FW;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (live && b < B && bsum && !first_level) {
            s0 = bsum[((size_t)b * 3 + 0) * C + c];
            s1 = bsum[((size_t)b * 3 + 1) * C + c];
            s2 = bsum[((size_t)b * 3 + 2) * C + c];
        } else if (live && b < B) {
            const int t0 = (int)(((long long)b * N) / 128);
            int t1 = (int)((((long long)(b + 1)) * N - 1) / 128);
            t1 = t1 < nrb - 1 ? t1 : nrb - 1;
            // eight tiles' records in flight, added in tile order (one record at a time made this launch 77 us at B = 32:
            // 56 dependent round trips per batch element)
            for (int tb = t0; tb <= t1; tb += 8) {
                double v0[8], v1[8], v2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = tb + u <= t1 ? tb + u : t1;
                    const int sl = b - (int)(((long long)t * 128) / N);  // 0: b is the tile's first batch element, 1: its second
                    const double *r = pdv + (((size_t)t * 2 + sl) * 3) * C + c;
                    v0[u] = r[0];
                    v1[u] = r[C];
                    v2[u] = r[2 * (size_t)C];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool on = tb + u <= t1;
                    s0 += on ? v0[u] : 0.0;
                    s1 += on ? v1[u] : 0.0;
                    s2 += on ? v2[u] : 0.0;
                }
            }
        }
        if (first_level && live && b < B) {
            bsum[((size_t)b * 3 + 0) * C + c] = s0;
            bsum[((size_t)b * 3 + 1) * C + c] = s1;
            bsum[((size_t)b * 3 + 2) * C + c] = s2;
        }
        bdv[i] = s0;
        bdu[i] = s1;
        buh[i] = s2;
        a0 += s0;
        a1 += s1;
    }
    if (first_level) return;
    red[wave][0][lane] = a0;
    red[wave][1][lane] = a1;
    __syncthreads();
    if (wave == 0) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int w = 0; w < FW; ++w) {
            s0 += red[w][0][lane];
            s1 += red[w][1][lane];
        }
        float c1, c2;
        bn_bwd_consts(s0, s1, B * N, training, c, live, live ? stats[2 * (size_t)C + c] : 0.0f, gw, gb, cc, C, sb, c1, c2);
        scc[0][lane] = c1;
        scc[1][lane] = c2;
    }
    __syncthreads();
    if (!live || !gadd) return;
    const float c1 = scc[0][lane], c2 = scc[1][lane], sc = stats[2 * (size_t)C + c];
#pragma unroll
    for (int i = 0; i < SLOT_B; ++i) {
        const int b = wave + i * FW;
        if (b < B) gadd[(size_t)b * C + c] = sc * (float)(bdv[i] - (double)N * (double)c1 - (double)c2 * buh[i]);
    }
}

__global__ void dropout_mask_kernel(unsigned long long seed, int P, int K, unsigned keep16, unsigned char *mask)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nq = (size_t)P * (K / 4);
    if (q >= nq) return;
    const unsigned long long mix = drop_mix(seed, q);
#pragma unroll
    for (int e = 0; e < 4; ++e) mask[4 * q + e] = drop_keep(mix, e, keep16) ? 1 : 0;
}

// ---- conv6 -> bn6 -> LeakyReLU -> max over the points (model.py:172-175): the small pieces of its backward -------------------
// With u = cat W^T the BatchNorm mean terms of dL/du are affine in u:  dL/du[p][c] = A_c + B_c u[p][c] + (s[b][c] at the winner),
//   dcat = 1 (W^T A)^T + cat (W^T diag(B) W) + [winners]        dW = A (sum_p cat)^T + diag(B) W (cat^T cat) + [winners]
// (ops._EmbedMax): two F x F products over the points instead of two C-wide ones.  Here: M = W^T diag(B) W and v = W^T A ...
// partial[cs][f][g], g < F: the share of channel chunk cs (MV_CH channels) of M[f][g] = sum_c W[c][f] B_c W[c][g];
// partial[cs][F * F + f]: its share of v[f] = sum_c W[c][f] A_c.  One workgroup = a 32 x 32 tile of (f, g) -- the tile column
// behind the last g tile carries v -- over one chunk; launch_sum_partials adds the chunks in order.
constexpr int MV_CH = 128;
__global__ __launch_bounds__(256) void embed_mv_kernel(const float *__restrict__ W, int ldw, int C, int F,
                                                       const float *__restrict__ coef_a, const float *__restrict__ coef_b,
                                                       float *__restrict__ partial)
{
    __shared__ float sA[MV_CH][33], sB[MV_CH][33];
    const int f0 = blockIdx.x * 32, g0 = blockIdx.y * 32, c0 = blockIdx.z * MV_CH;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // tx: g within the tile; ty: four rows f
    for (int t = threadIdx.x; t < MV_CH * 32; t += 256) {
        const int cc = t >> 5, j = t & 31, c = c0 + cc;
        const bool cv = c < C;
        sA[cc][j] = cv && f0 + j < F ? W[(size_t)c * ldw + f0 + j] : 0.0f;
        const int g = g0 + j;
        sB[cc][j] = !cv ? 0.0f : (g < F ? coef_b[c] * W[(size_t)c * ldw + g] : (g == F ? coef_a[c] : 0.0f));
    }
    __syncthreads();
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int cc = 0; cc < MV_CH; ++cc) {
        const float b = sB[cc][tx];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(sA[cc][ty * 4 + r], b, acc[r]);
    }
    float *out = partial + (size_t)blockIdx.z * ((size_t)F * F + F);
    const int g = g0 + tx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty * 4 + r;
        if (f < F) {
            if (g < F) out[(size_t)f * F + g] = acc[r];
            else if (g == F) out[(size_t)F * F + f] = acc[r];
        }
    }
}

// ... and dW[c][f] = sum_b s[b][c] cat[b N + arg[b][c]][f]  +  B_c sum_g W[c][g] G[g][f]  +  A_c catsum[f]   (G = cat^T cat),
// four channels per workgroup, thread = column f; every sum in a fixed order
constexpr int DW_CH = 4;
__global__ __launch_bounds__(256) void embed_dw_kernel(const float *__restrict__ cat, int ldc, const float *__restrict__ W, int ldw,
                                                       const int32_t *__restrict__ arg, const float *__restrict__ s,
                                                       const float *__restrict__ coef_a, const float *__restrict__ coef_b,
                                                       const float *__restrict__ G, const float *__restrict__ catsum, int B, int N,
                                                       int C, int F, int training, float *__restrict__ dW)
{
    __shared__ float wrow[DW_CH][256];  // (F <= 256)
    const int c0 = blockIdx.x * DW_CH;
    if (training)
        for (int t = threadIdx.x; t < DW_CH * F; t += 256) {
            const int cc = t / F, g = t - cc * F;
            wrow[cc][g] = c0 + cc < C ? W[(size_t)(c0 + cc) * ldw + g] : 0.0f;
        }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc[DW_CH];
#pragma unroll
        for (int cc = 0; cc < DW_CH; ++cc) acc[cc] = 0.0f;
        if (training) {
            for (int g = 0; g < F; ++g) {
                const float gv = G[(size_t)g * F + f];
#pragma unroll
                for (int cc = 0; cc < DW_CH; ++cc) acc[cc] = __builtin_fmaf(wrow[cc][g], gv, acc[cc]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < DW_CH; ++cc) {
            const int c = c0 + cc;
            if (c >= C) continue;
            float d = training ? __builtin_fmaf(coef_b[c], acc[cc], coef_a[c] * catsum[f]) : 0.0f;
            // the winners' rows, eight batch elements in flight (arg -> row is a dependent load), added in batch order
            for (int b0 = 0; b0 < B; b0 += 8) {
                float sv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + u < B ? b0 + u : B - 1;
                    sv[u] = s[(size_t)b * C + c];
                    xv[u] = cat[((size_t)b * N + arg[(size_t)b * C + c]) * ldc + f];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b0 + u < B) d = __builtin_fmaf(sv[u], xv[u], d);
            }
            dW[(size_t)c * F + f] = d;
        }
    }
}

// max |x| (bits, into a zeroed word) and the column sums of a [rows][cols] array in one pass: a workgroup takes 256 rows
// (a wave every fourth row, a lane four adjacent columns) and leaves one fp64 record; colsum_fold_kernel adds the records
// in order.  cols a multiple of 4, cols <= 256.
__global__ __launch_bounds__(256) void rows_stats_kernel(const float *__restrict__ x, int ld, int rows, int cols,
                                                         float *__restrict__ pmax, double *__restrict__ partial)
{
    __shared__ double red[4][256];
    __shared__ float rmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = 4 * lane;
    const int r0 = blockIdx.x * 256, r1 = r0 + 256 < rows ? r0 + 256 : rows;
    double sm[4] = {0.0, 0.0, 0.0, 0.0};
    float m = 0.0f;
    if (c < cols)
        for (int rr = r0 + wave; rr < r1; rr += 32) {  // eight of the wave's rows in flight, added in row order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rr + 4 * u < r1 ? rr + 4 * u : r1 - 1;
                v[u] = *reinterpret_cast<const float4 *>(x + (size_t)r * ld + c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (rr + 4 * u < r1) {
                    sm[0] += (double)v[u].x;
                    sm[1] += (double)v[u].y;
                    sm[2] += (double)v[u].z;
                    sm[3] += (double)v[u].w;
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
                }
            }
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][4 * lane + e] = sm[e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) rmax[wave] = m;
    __syncthreads();
    if ((int)threadIdx.x < cols) partial[(size_t)blockIdx.x * cols + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    // (the workgroup's maximum goes to a record of its own: colsum_fold_kernel folds them -- no word to zero before the pass)
    if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
}
__global__ __launch_bounds__(64 * FW) void colsum_fold_kernel(const double *__restrict__ partial, int nrec, int cols,
                                                              float *__restrict__ colsum, const float *__restrict__ pmax,
                                                              unsigned *__restrict__ amax)
{
    __shared__ double red[FW][64];
    __shared__ float mred[FW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && amax) {  // max |x| = the maximum of the workgroups' records (a maximum has no order)
        float m = 0.0f;
        for (int g = threadIdx.x; g < nrec; g += 64 * FW) m = fmaxf(m, pmax[g]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) mred[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = mred[0];
            for (int w = 1; w < FW; ++w) t = fmaxf(t, mred[w]);
            *amax = __float_as_uint(t);
        }
    }
    const int c = blockIdx.x * 64 + lane;
    double a = 0.0;
    if (c < cols)
        for (int g = wave; g < nrec; g += FW) a += partial[(size_t)g * cols + c];
    red[wave][lane] = a;
    __syncthreads();
    if (wave != 0 || c >= cols) return;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < FW; ++w) t += red[w][lane];
    colsum[c] = (float)t;
}

// out[e] = sum over nchunk partials [nchunk][n] (chunk order) for up to three jobs in ONE launch: the tail's three weight
// gradients (dW9's records, dW8's and dW7c's chunk partials) were three launches of 5 - 11 us.  1024 threads = 16 waves x
// 64 elements; a wave adds its contiguous share of the chunks (eight in flight), the 16 sums are added in wave order.
struct FoldJob {
    const float *partial;
    int nchunk, n;
    float *out;
    int first_block;
};
struct Fold3Args {
    FoldJob job[3];
    int njobs;
};
__global__ __launch_bounds__(1024) void fold3_kernel(const Fold3Args a)
{
    __shared__ float red[16][64];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.job[j + 1].first_block) ++j;
    const FoldJob J = a.job[j];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - J.first_block) * 64 + lane;
    const int q = (J.nchunk + 15) / 16;
    const int c0 = w * q, c1 = c0 + q < J.nchunk ? c0 + q : J.nchunk;
    float s = 0.0f;
    if (e < J.n) {
        int ch = c0;
        for (; ch + 8 <= c1; ch += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = J.partial[(size_t)(ch + u) * J.n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        if (ch < c1) {  // the last, partial batch (at 48 - 96 chunks a wave's whole share): loads in flight together, masked
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = J.partial[(size_t)(ch + u < c1 ? ch + u : ch) * J.n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += ch + u < c1 ? v[u] : 0.0f;
        }
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < J.n) {
        float t = red[0][lane];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i][lane];
        J.out[e] = t;
    }
}

// ---- host side: layouts and launch sequences ----------------------------------------------------------------------------
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base((char *)p), off(0) {}
    template <typename T>
    T *take(size_t n)
    {
        T *r = (T *)(base + off);
        off += align_up(n * sizeof(T), 256);
        return r;
    }
};
struct Planes {
    _Float16 *hi, *lo;
    float *scale;
};
template <typename C>
static Planes take_planes(C &c, int rows, int cols)
{
    Planes p;
    p.hi = c.template take<_Float16>((size_t)rows * cols);
    p.lo = c.template take<_Float16>((size_t)rows * cols);
    p.scale = c.template take<float>(rows);
    return p;
}
struct Saved {
    float *u7, *u8, *stats7, *stats8;
    unsigned *zmax;  // [0] max |z7|, [1] max |z8| (unused)
    Planes w7c, w8, w8t, w7ct;
    size_t bytes;
};
static Saved saved_layout(const dgcnn_head_tail_desc *d, void *p)
{
    Carver c(p);
    Saved s;
    const size_t P = (size_t)d->B * d->N;
    s.u7 = c.take<float>(P * d->C7);
    s.u8 = c.take<float>(P * d->C8);
    s.stats7 = c.take<float>(4 * (size_t)d->C7);
    s.stats8 = c.take<float>(4 * (size_t)d->C8);
    s.zmax = c.take<unsigned>(4);
    s.w7c = take_planes(c, d->C7, d->F);
    s.w8 = take_planes(c, d->C8, d->C7);
    s.w8t = take_planes(c, d->C7, d->C8);
    s.w7ct = take_planes(c, d->F, d->C7);
    s.bytes = c.off;
    return s;
}
static int row_blocks(const dgcnn_head_tail_desc *d) { return ceil_div(d->B * d->N, GM); }
struct FwdWs {
    double *psum7, *psum8, *fpsum;
    float *pext7, *pext8, *fpext;
    size_t bytes;
};
static FwdWs fwd_ws(const dgcnn_head_tail_desc *d, void *p)
{
    Carver c(p);
    FwdWs w;
    const size_t nrb = row_blocks(d);
    w.psum7 = c.take<double>(nrb * 2 * d->C7);
    w.pext7 = c.take<float>(nrb * 2 * d->C7);
    w.psum8 = c.take<double>(nrb * 2 * d->C8);
    w.pext8 = c.take<float>(nrb * 2 * d->C8);
    const size_t cmax = d->C7 > d->C8 ? d->C7 : d->C8;
    w.fpsum = c.take<double>((size_t)8 * 2 * cmax);  // FIN_SLICES folded records (launch_stats_finalize)
    w.fpext = c.take<float>((size_t)8 * 2 * cmax);
    w.bytes = c.off;
    return w;
}
static int wgrad_chunks(int M, int Nn, int P)
{
    const int tiles = ceil_div(M, 128) * ceil_div(Nn, 128);
    // (one and a half workgroups per CU.  768 -- three per CU -- until round 4: every chunk leaves a 128 x 128 partial tile, 88 MB
    // for dW8 and dW7c together whatever the batch; at B = 4 the two products 54 + 53 -> 48 + 45 us and the fold of the partials
    // 30 -> 20 us, at B = 32 the products 320 + 306 -> 337 + 304 and the fold 34 -> 20; 192: 62 + 60 us at B = 4)
    static const int target = getenv("DGCNN_WGRAD_TARGET") ? atoi(getenv("DGCNN_WGRAD_TARGET")) : 384;  // experiments
    int chunks = target / tiles;
    if (chunks < 8) chunks = 8;
    const int most = ceil_div(P, 128);
    if (chunks > most) chunks = most;
    return chunks < 1 ? 1 : chunks;
}
static int cls_blocks(int nrb) { return nrb < 512 ? nrb : 512; }
struct BwdWs {
    unsigned *zero;  // one block, zeroed by a memset node: oa8, ob8, oa7, ob7 [nrb + 1] each, sb8 [4], sb7 [4]
    size_t zero_words;
    unsigned *oa8, *ob8, *oa7, *ob7, *sb8, *sb7;
    float *dv8, *dv7, *pdw9, *cc8, *cc7, *wpart, *wpart7;
    double *pbn8, *pdv7, *bsum7;
    size_t bytes;
};
static BwdWs bwd_ws(const dgcnn_head_tail_desc *d, void *p)
{
    Carver c(p);
    BwdWs w;
    const size_t P = (size_t)d->B * d->N, nrb = row_blocks(d);
    w.zero_words = 4 * (nrb + 1) + 8;
    w.zero = c.take<unsigned>(w.zero_words);
    w.oa8 = w.zero;
    w.ob8 = w.oa8 + nrb + 1;
    w.oa7 = w.ob8 + nrb + 1;
    w.ob7 = w.oa7 + nrb + 1;
    w.sb8 = w.ob7 + nrb + 1;
    w.sb7 = w.sb8 + 4;
    w.dv8 = c.take<float>(P * d->C8);
    w.dv7 = c.take<float>(P * d->C7);
    w.pdw9 = c.take<float>((size_t)cls_blocks((int)nrb) * d->NC * d->C8);
    w.pbn8 = c.take<double>((size_t)cls_blocks((int)nrb) * 2 * d->C8);
    w.cc8 = c.take<float>(2 * (size_t)d->C8);
    w.pdv7 = c.take<double>(nrb * 2 * 3 * d->C7);
    w.cc7 = c.take<float>(2 * (size_t)d->C7);
    w.bsum7 = c.take<double>((size_t)d->B * 3 * d->C7);
    const size_t wp8 = (size_t)wgrad_chunks(d->C8, d->C7, (int)P) * d->C8 * d->C7, wp7 = (size_t)wgrad_chunks(d->C7, d->F, (int)P) * d->C7 * d->F;
    // (one buffer each: the three weight gradients of the tail are folded by ONE launch at the end of the backward)
    w.wpart = c.take<float>(wp8);
    w.wpart7 = c.take<float>(wp7);
    w.bytes = c.off;
    return w;
}

static int check_desc(const dgcnn_head_tail_desc *d, const char *who, bool params)
{
    DG_REQUIRE(d, "%s: null descriptor", who);
    DG_REQUIRE(d->B > 0 && d->N >= GM && d->B <= FW * SLOT_B, "%s: needs 1 <= B <= %d and N >= %d (B=%d N=%d)", who, FW * SLOT_B, GM,
               d->B, d->N);
    DG_REQUIRE((size_t)d->B * d->N < ((size_t)1 << 23), "%s: more than 2^23 points", who);
    DG_REQUIRE(d->F > 0 && d->F % 32 == 0 && d->C7 > 0 && d->C7 % 32 == 0 && d->C8 > 0 && d->C8 % 32 == 0 && d->C7 <= KT_MAX &&
                   d->C8 <= KT_MAX,
               "%s: F, C7, C8 must be multiples of 32 (C7, C8 <= %d: the per-k constants of a product live in LDS): F=%d C7=%d C8=%d",
               who, KT_MAX, d->F, d->C7, d->C8);
    DG_REQUIRE(d->NC >= 1 && d->NC <= CLS_MAX, "%s: 1 <= classes <= %d (NC=%d)", who, CLS_MAX, d->NC);
    DG_REQUIRE(d->ldcat >= d->F && d->ldcat % 4 == 0 && d->ldw7 >= d->F && d->ldw7 % 4 == 0, "%s: ldcat / ldw7 must be multiples of 4 and >= F", who);
    DG_REQUIRE(d->p_drop >= 0.0f && d->p_drop < 1.0f, "%s: dropout probability %g outside [0, 1)", who, (double)d->p_drop);
    if (!params) return 0;
    DG_REQUIRE(d->w7c && d->w8 && d->w9 && d->bn7.weight && d->bn7.bias && d->bn8.weight && d->bn8.bias, "%s: parameters missing", who);
    DG_REQUIRE(((uintptr_t)d->w7c & 15) == 0 && ((uintptr_t)d->w8 & 15) == 0 && ((uintptr_t)d->w9 & 15) == 0, "%s: weights must be 16-byte aligned", who);
    DG_REQUIRE(d->training || (d->bn7.running_mean && d->bn7.running_var && d->bn8.running_mean && d->bn8.running_var),
               "%s: eval mode needs running statistics", who);
    return 0;
}

#define HF_CLS_DISPATCH(NC_, CALL)                      \
    switch (NC_) {                                      \
        case 1: { constexpr int NCV = 1; CALL; } break; \
        case 2: { constexpr int NCV = 2; CALL; } break; \
        case 3: { constexpr int NCV = 3; CALL; } break; \
        case 4: { constexpr int NCV = 4; CALL; } break; \
        case 5: { constexpr int NCV = 5; CALL; } break; \
        case 6: { constexpr int NCV = 6; CALL; } break; \
        case 7: { constexpr int NCV = 7; CALL; } break; \
        default: { constexpr int NCV = 8; CALL; } break; \
    }

// experiments: DGCNN_HF_OCC=3 asks the gradient-operand kernels for three workgroups per CU (168 registers, a few spilled)
static bool occ3()
{
    static const int v = getenv("DGCNN_HF_OCC") ? atoi(getenv("DGCNN_HF_OCC")) : 2;
    return v == 3;
}
// BatchNorm statistics from the products' records: one launch for few records, a fold over FIN_SLICES slices + the finalizer
// for many (fold buffers: fpsum [FIN_SLICES][2][C] doubles, fpext the same in floats)
constexpr int FIN_SLICES = 8;
static void launch_stats_finalize(const double *psum, const float *pext, int nrec, int P, int C, int training,
                                  const dgcnn_bn_params &bn, float eps, float momentum, float slope, float *stats, unsigned *zmax,
                                  double *fpsum, float *fpext, hipStream_t st)
{
    const dim3 block(64 * FW);
    const int gx = ceil_div(C, 64);
    if (nrec > 256 && fpsum && (training || pext)) {
        hipLaunchKernelGGL(stats_finalize_kernel, dim3(gx, FIN_SLICES), block, 0, st, psum, pext, nrec, P, C, training, bn.weight,
                           bn.bias, bn.running_mean, bn.running_var, eps, momentum, slope, stats, zmax, fpsum, pext ? fpext : nullptr);
        hipLaunchKernelGGL(stats_finalize_kernel, dim3(gx, 1), block, 0, st, (const double *)fpsum, pext ? (const float *)fpext : nullptr,
                           FIN_SLICES, P, C, training, bn.weight, bn.bias, bn.running_mean, bn.running_var, eps, momentum, slope,
                           stats, zmax, (double *)nullptr, (float *)nullptr);
    } else {
        hipLaunchKernelGGL(stats_finalize_kernel, dim3(gx, 1), block, 0, st, psum, pext, nrec, P, C, training, bn.weight, bn.bias,
                           bn.running_mean, bn.running_var, eps, momentum, slope, stats, zmax, (double *)nullptr, (float *)nullptr);
    }
}
static unsigned keep16_of(float p) { return (unsigned)((1.0 - (double)p) * 65536.0 + 0.5); }
// the survivors' scale: the reciprocal of the keep probability the mask REALISES (a multiple of 2^-16), so that the
// expectation of dropout(z) is z exactly
static float inv_keep_of(float p) { return (float)(65536.0 / (double)keep16_of(p)); }

}  // namespace hf
}  // namespace dgcnn

using namespace dgcnn;
using namespace dgcnn::hf;

extern "C" size_t dgcnn_head_tail_saved_bytes(const dgcnn_head_tail_desc *d)
{
    return check_desc(d, "dgcnn_head_tail_saved_bytes", false) == 0 ? saved_layout(d, nullptr).bytes : 0;
}
extern "C" size_t dgcnn_head_tail_fwd_workspace_bytes(const dgcnn_head_tail_desc *d)
{
    return check_desc(d, "dgcnn_head_tail_fwd_workspace_bytes", false) == 0 ? fwd_ws(d, nullptr).bytes : 0;
}
extern "C" size_t dgcnn_head_tail_bwd_workspace_bytes(const dgcnn_head_tail_desc *d)
{
    return check_desc(d, "dgcnn_head_tail_bwd_workspace_bytes", false) == 0 ? bwd_ws(d, nullptr).bytes : 0;
}

extern "C" int dgcnn_head_tail_fwd(const dgcnn_head_tail_desc *d, const float *cat, const uint32_t *amax_cat, const float *add7,
                                   float *logits, void *saved, void *workspace, size_t workspace_bytes, void *stream)
{
    if (check_desc(d, "dgcnn_head_tail_fwd", true)) return 1;
    DG_REQUIRE(cat && amax_cat && logits && saved && workspace, "dgcnn_head_tail_fwd: null pointer");
    DG_REQUIRE(((uintptr_t)cat & 15) == 0, "dgcnn_head_tail_fwd: cat must be 16-byte aligned");
    const FwdWs w = fwd_ws(d, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_head_tail_fwd: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    const Saved s = saved_layout(d, saved);
    hipStream_t st = (hipStream_t)stream;
    const int P = d->B * d->N, nrb = row_blocks(d), F = d->F, C7 = d->C7, C8 = d->C8;

    {   // the weights of this step as fp16 planes, both orientations (the backward reads the transposed ones)
        WPrepArgs a{};
        a.job[0] = WPrepJob{d->w7c, d->ldw7, C7, F, 0, s.w7c.hi, s.w7c.lo, s.w7c.scale};
        a.job[1] = WPrepJob{d->w8, C7, C8, C7, 0, s.w8.hi, s.w8.lo, s.w8.scale};
        a.job[2] = WPrepJob{d->w8, C7, C7, C8, 1, s.w8t.hi, s.w8t.lo, s.w8t.scale};
        a.job[3] = WPrepJob{d->w7c, d->ldw7, F, C7, 1, s.w7ct.hi, s.w7ct.lo, s.w7ct.scale};
        a.njobs = 4;
        a.first_row[0] = 0;
        for (int j = 0; j < 4; ++j) a.first_row[j + 1] = a.first_row[j] + a.job[j].rows;
        hipLaunchKernelGGL(wprep_kernel, dim3(ceil_div(a.first_row[4], 4)), dim3(256), 0, st, a);
    }
    {   // u7 = cat . W7c^T + add7[b]; column sums and extrema in the epilogue
        HGemmArgs a{};
        a.X = cat, a.ldx = d->ldcat, a.P = P, a.K = F, a.N = C7;
        a.Wh = s.w7c.hi, a.Wl = s.w7c.lo, a.wscale = s.w7c.scale;
        a.xmax = amax_cat;
        a.slope = d->slope;
        a.add = add7, a.rows_per_batch = d->N;
        a.Y = s.u7, a.ldy = C7;
        a.psum = d->training ? w.psum7 : nullptr, a.pext = w.pext7, a.zero = s.zmax;
        a.ncb = ceil_div(C7, GN), a.nrb = nrb;
        prof_begin(PROF_HEAD_CONV7, st);
        if ((a.csplit = row_form(a)) > 0) hipLaunchKernelGGL((hrow_kernel<E_STATS>), dim3(nrb * a.csplit), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hgemm_kernel<X_RAW, E_STATS, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        prof_end(PROF_HEAD_CONV7, st);
        launch_stats_finalize(w.psum7, w.pext7, nrb, P, C7, d->training, d->bn7, d->eps7, d->momentum7, d->slope, s.stats7, s.zmax,
                              w.fpsum, w.fpext, st);
    }
    {   // u8 = LReLU(BN7(u7)) . W8^T: the activation is formed where the product loads its operand
        HGemmArgs a{};
        a.X = s.u7, a.ldx = C7, a.P = P, a.K = C7, a.N = C8;
        a.Wh = s.w8.hi, a.Wl = s.w8.lo, a.wscale = s.w8.scale;
        a.xmax = s.zmax;
        a.pc[0] = s.stats7 + 2 * (size_t)C7, a.pc[1] = s.stats7 + 3 * (size_t)C7;
        a.slope = d->slope;
        a.Y = s.u8, a.ldy = C8;
        a.psum = d->training ? w.psum8 : nullptr, a.pext = w.pext8, a.zero = s.zmax + 1;
        a.ncb = ceil_div(C8, GN), a.nrb = nrb;
        prof_begin(PROF_HEAD_CONV8, st);
        hipLaunchKernelGGL((hgemm_kernel<X_BN, E_STATS, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        prof_end(PROF_HEAD_CONV8, st);
        launch_stats_finalize(w.psum8, w.pext8, nrb, P, C8, d->training, d->bn8, d->eps8, d->momentum8, d->slope, s.stats8,
                              s.zmax + 1, w.fpsum, w.fpext, st);
    }
    {   // logits = dropout(LReLU(BN8(u8))) . W9^T
        ClsArgs a{};
        a.u = s.u8, a.P = P, a.K = C8, a.nrb = nrb;
        a.mean = s.stats8, a.rstd = s.stats8 + C8, a.sc = s.stats8 + 2 * (size_t)C8, a.sh = s.stats8 + 3 * (size_t)C8;
        a.w = d->w9, a.slope = d->slope;
        const bool drop = d->training && d->p_drop > 0.0f;
        a.keep16 = keep16_of(d->p_drop), a.inv_keep = inv_keep_of(d->p_drop), a.seed = d->seed;
        a.y = logits;
        int grid = ceil_div(P, 4 * 4 * CLS_GROUPS);
        grid = grid > 4096 ? 4096 : grid;
        if (drop) {
            HF_CLS_DISPATCH(d->NC, hipLaunchKernelGGL((cls_fwd_fused_kernel<NCV, true>), dim3(grid), dim3(256), 0, st, a));
        } else {
            HF_CLS_DISPATCH(d->NC, hipLaunchKernelGGL((cls_fwd_fused_kernel<NCV, false>), dim3(grid), dim3(256), 0, st, a));
        }
    }
    DG_CHECK_LAUNCH("dgcnn_head_tail_fwd");
    return 0;
}

extern "C" int dgcnn_head_tail_bwd(const dgcnn_head_tail_desc *d, const float *cat, const uint32_t *amax_cat,
                                   const float *glogits, const void *saved, const dgcnn_head_tail_grads *g, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (check_desc(d, "dgcnn_head_tail_bwd", true)) return 1;
    DG_REQUIRE(cat && amax_cat && glogits && saved && g && workspace, "dgcnn_head_tail_bwd: null pointer");
    DG_REQUIRE(g->gcat && g->gadd7 && g->gw7c && g->gbn7_w && g->gbn7_b && g->gw8 && g->gbn8_w && g->gbn8_b && g->gw9,
               "dgcnn_head_tail_bwd: gradient outputs missing");
    DG_REQUIRE(g->ldgcat >= d->F, "dgcnn_head_tail_bwd: ldgcat < F");
    const BwdWs w = bwd_ws(d, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_head_tail_bwd: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    const Saved s = saved_layout(d, const_cast<void *>(saved));
    hipStream_t st = (hipStream_t)stream;
    const int P = d->B * d->N, nrb = row_blocks(d), F = d->F, C7 = d->C7, C8 = d->C8, NC = d->NC;
    const float *mean7 = s.stats7, *rstd7 = s.stats7 + C7, *sc7 = s.stats7 + 2 * (size_t)C7, *sh7 = s.stats7 + 3 * (size_t)C7;
    const float *mean8 = s.stats8, *rstd8 = s.stats8 + C8, *sc8 = s.stats8 + 2 * (size_t)C8, *sh8 = s.stats8 + 3 * (size_t)C8;

    if (hipMemsetAsync(w.zero, 0, w.zero_words * sizeof(unsigned), st) != hipSuccess) DG_FAIL("dgcnn_head_tail_bwd: hipMemsetAsync failed");
    int nch8 = 0, nch7 = 0;
    {   // classifier backward: dv8, dW9 records, BatchNorm-8 sums, tile maxima
        ClsArgs a{};
        a.u = s.u8, a.P = P, a.K = C8, a.nrb = nrb, a.nblk = cls_blocks(nrb);
        a.mean = mean8, a.rstd = rstd8, a.sc = sc8, a.sh = sh8;
        a.w = d->w9, a.slope = d->slope;
        const bool drop = d->training && d->p_drop > 0.0f;
        a.keep16 = keep16_of(d->p_drop), a.inv_keep = inv_keep_of(d->p_drop), a.seed = d->seed;
        a.gy = glogits, a.dv = w.dv8, a.pdw = w.pdw9, a.pbn = w.pbn8, a.oa = w.oa8, a.ob = w.ob8;
        if (drop) {
            HF_CLS_DISPATCH(NC, hipLaunchKernelGGL((cls_bwd_fused_kernel<NCV, true>), dim3(a.nblk), dim3(256), 0, st, a));
        } else {
            HF_CLS_DISPATCH(NC, hipLaunchKernelGGL((cls_bwd_fused_kernel<NCV, false>), dim3(a.nblk), dim3(256), 0, st, a));
        }
        hipLaunchKernelGGL(bwd_finalize_kernel, dim3(ceil_div(C8, 64)), dim3(64 * FW), 0, st, w.pbn8, a.nblk, P, C8, d->training,
                           s.stats8, g->gbn8_w, g->gbn8_b, w.cc8, w.sb8);
    }
    {   // dL/dz7 = du8 . W8 with du8 formed from (dv8, u8) at the operand load; dv7 = dL/dz7 . LReLU'(v7) and the
        // BatchNorm-7 sums in the epilogue
        HGemmArgs a{};
        a.X = w.dv8, a.ldx = C8, a.X2 = s.u8, a.ldx2 = C8, a.P = P, a.K = C8, a.N = C7;
        a.Wh = s.w8t.hi, a.Wl = s.w8t.lo, a.wscale = s.w8t.scale;
        a.ta = w.oa8, a.tb = w.ob8, a.sb = w.sb8;
        a.pc[0] = mean8, a.pc[1] = rstd8, a.pc[2] = sc8, a.pc[3] = w.cc8, a.pc[4] = w.cc8 + C8;
        a.slope = d->slope;
        a.rows_per_batch = d->N;
        a.Y = w.dv7, a.ldy = C7;
        a.U = s.u7, a.ldu = C7;
        a.ec[0] = mean7, a.ec[1] = rstd7, a.ec[2] = sc7, a.ec[3] = sh7;
        a.pdv = w.pdv7, a.oa = w.oa7, a.ob = w.ob7;
        a.ncb = ceil_div(C7, GN), a.nrb = nrb;
        prof_begin(PROF_HEAD_DZ7, st);
        if (occ3()) hipLaunchKernelGGL((hgemm_kernel<X_DU, E_DV, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hgemm_kernel<X_DU, E_DV, 2>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        prof_end(PROF_HEAD_DZ7, st);
    }
    {   // dW8 = du8^T z7: both operands formed at the load
        HWgradArgs a{};
        a.A = w.dv8, a.lda = C8, a.A2 = s.u8, a.lda2 = C8, a.M = C8;
        a.Bm = s.u7, a.ldb = C7, a.Nn = C7, a.P = P;
        const int chunks = wgrad_chunks(C8, C7, P);
        a.chunk = (ceil_div(P, chunks) + 31) & ~31;
        a.ntiles_n = ceil_div(C7, 128);
        a.ac[0] = mean8, a.ac[1] = rstd8, a.ac[2] = sc8, a.ac[3] = w.cc8, a.ac[4] = w.cc8 + C8;
        a.bc[0] = sc7, a.bc[1] = sh7;
        a.slope = d->slope;
        a.ta = w.oa8, a.tb = w.ob8, a.sb = w.sb8, a.gi = nrb;
        a.amax_b = s.zmax;
        a.partial = w.wpart;
        const int nch = ceil_div(P, a.chunk);
        const dim3 grid((unsigned)align_up((size_t)ceil_div(C8, 128) * a.ntiles_n * nch, 8));
        prof_begin(PROF_HEAD_DW8, st);
        if (occ3()) hipLaunchKernelGGL((hwgrad_kernel<A_DU, B_BN, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hwgrad_kernel<A_DU, B_BN, 2>), grid, dim3(256), 0, st, a);
        prof_end(PROF_HEAD_DW8, st);
        nch8 = nch;
    }
    if (nrb > 256) {
        hipLaunchKernelGGL(bwd_finalize_slots_kernel, dim3(ceil_div(C7, 64), d->B), dim3(64), 0, st, w.pdv7, nrb, d->B, d->N, C7,
                           d->training, s.stats7, g->gbn7_w, g->gbn7_b, w.cc7, w.sb7, g->gadd7, w.bsum7);
        hipLaunchKernelGGL(bwd_finalize_slots_kernel, dim3(ceil_div(C7, 64), 1), dim3(64 * FW), 0, st, w.pdv7, nrb, d->B, d->N, C7,
                           d->training, s.stats7, g->gbn7_w, g->gbn7_b, w.cc7, w.sb7, g->gadd7, w.bsum7);
    } else {
        hipLaunchKernelGGL(bwd_finalize_slots_kernel, dim3(ceil_div(C7, 64), 1), dim3(64 * FW), 0, st, w.pdv7, nrb, d->B, d->N, C7,
                           d->training, s.stats7, g->gbn7_w, g->gbn7_b, w.cc7, w.sb7, g->gadd7, (double *)nullptr);
    }
    {   // dL/dcat = du7 . W7c
        HGemmArgs a{};
        a.X = w.dv7, a.ldx = C7, a.X2 = s.u7, a.ldx2 = C7, a.P = P, a.K = C7, a.N = F;
        a.Wh = s.w7ct.hi, a.Wl = s.w7ct.lo, a.wscale = s.w7ct.scale;
        a.ta = w.oa7, a.tb = w.ob7, a.sb = w.sb7;
        a.pc[0] = mean7, a.pc[1] = rstd7, a.pc[2] = sc7, a.pc[3] = w.cc7, a.pc[4] = w.cc7 + C7;
        a.slope = d->slope;
        a.Y = g->gcat, a.ldy = g->ldgcat;
        a.ncb = ceil_div(F, GN), a.nrb = nrb;
        prof_begin(PROF_HEAD_DCAT, st);
        if (occ3()) hipLaunchKernelGGL((hgemm_kernel<X_DU, E_PLAIN, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hgemm_kernel<X_DU, E_PLAIN, 2>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        prof_end(PROF_HEAD_DCAT, st);
    }
    {   // dW7c = du7^T cat
        HWgradArgs a{};
        a.A = w.dv7, a.lda = C7, a.A2 = s.u7, a.lda2 = C7, a.M = C7;
        a.Bm = cat, a.ldb = d->ldcat, a.Nn = F, a.P = P;
        const int chunks = wgrad_chunks(C7, F, P);
        a.chunk = (ceil_div(P, chunks) + 31) & ~31;
        a.ntiles_n = ceil_div(F, 128);
        a.ac[0] = mean7, a.ac[1] = rstd7, a.ac[2] = sc7, a.ac[3] = w.cc7, a.ac[4] = w.cc7 + C7;
        a.slope = d->slope;
        a.ta = w.oa7, a.tb = w.ob7, a.sb = w.sb7, a.gi = nrb;
        a.amax_b = amax_cat;
        a.partial = w.wpart7;
        const int nch = ceil_div(P, a.chunk);
        const dim3 grid((unsigned)align_up((size_t)ceil_div(C7, 128) * a.ntiles_n * nch, 8));
        prof_begin(PROF_HEAD_DW7, st);
        if (occ3()) hipLaunchKernelGGL((hwgrad_kernel<A_DU, B_RAW, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hwgrad_kernel<A_DU, B_RAW, 2>), grid, dim3(256), 0, st, a);
        prof_end(PROF_HEAD_DW7, st);
        nch7 = nch;
    }
    {   // the three weight gradients' partials, folded by one launch
        Fold3Args f{};
        f.job[0] = FoldJob{w.pdw9, cls_blocks(nrb), NC * C8, g->gw9, 0};
        f.job[1] = FoldJob{w.wpart, nch8, C8 * C7, g->gw8, ceil_div(NC * C8, 64)};
        f.job[2] = FoldJob{w.wpart7, nch7, C7 * F, g->gw7c, f.job[1].first_block + ceil_div(C8 * C7, 64)};
        f.njobs = 3;
        hipLaunchKernelGGL(fold3_kernel, dim3(f.job[2].first_block + ceil_div(C7 * F, 64)), dim3(1024), 0, st, f);
    }
    DG_CHECK_LAUNCH("dgcnn_head_tail_bwd");
    return 0;
}

extern "C" int dgcnn_dropout_mask(uint64_t seed, int P, int K, float p_drop, uint8_t *mask, void *stream)
{
    DG_REQUIRE(mask && P > 0 && K > 0 && (K & 3) == 0 && p_drop >= 0.0f && p_drop < 1.0f, "dgcnn_dropout_mask: bad arguments");
    const size_t nq = (size_t)P * (K / 4);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)seed, P, K, keep16_of(p_drop), mask);
    DG_CHECK_LAUNCH("dgcnn_dropout_mask");
    return 0;
}

// ---- conv6 -> bn6 -> LeakyReLU -> max over the points as one forward and one backward call ---------------------------------
namespace dgcnn {
void launch_colstats_finalize(const double *partial, int nchunk, int P, int C, int training, const dgcnn_bn_params *bn, float eps,
                              float momentum, float *stats, hipStream_t st);  // head.hip
void launch_colmax_finish(const unsigned long long *keys, const float *stats, int B, int C, float slope, float *g, int32_t *arg,
                          float *uwin, hipStream_t st);
void launch_embed_bwd_coef(const float *uwin, const float *gg, const float *stats, int B, int N, int C, int training, float slope,
                           float *gweight, float *gbias, float *coef_a, float *coef_b, float *s, hipStream_t st, const int32_t *arg,
                           unsigned long long *sorted);
int embed_sorted_len(int C);
void launch_embed_bwd_sorted_scatter(const unsigned long long *sorted, const float *s, const float *w, int B, int N, int C, int F,
                                     float *dcat, hipStream_t st, int ldd);
void launch_embed_bwd_scatter(const int32_t *arg, const float *s, const float *w, int B, int N, int C, int F, float *dcat,
                              hipStream_t st, int ldd);
namespace hf {
struct EmbedFwdWs {
    double *psum, *fpsum;
    unsigned long long *keys;
    Planes w;
    size_t bytes;
};
static EmbedFwdWs embed_fwd_ws(int B, int N, int F, int C, void *p)
{
    Carver c(p);
    EmbedFwdWs w;
    w.psum = c.take<double>((size_t)ceil_div(B * N, GM) * 2 * C);
    w.keys = c.take<unsigned long long>((size_t)B * C);
    w.fpsum = c.take<double>((size_t)8 * 2 * C);
    w.w = take_planes(c, C, F);
    w.bytes = c.off;
    return w;
}
struct EmbedBwdWs {
    float *coef_a, *coef_b, *s, *M, *v, *G, *wpart, *mvpart;
    unsigned long long *sorted;  // [B][C2]: (winning point, channel) keys of every batch element, ascending
    Planes m;
    size_t bytes;
};
static EmbedBwdWs embed_bwd_ws(int B, int N, int F, int C, void *p)
{
    Carver c(p);
    EmbedBwdWs w;
    w.coef_a = c.take<float>(C);
    w.coef_b = c.take<float>(C);
    w.s = c.take<float>((size_t)B * C);
    w.sorted = c.take<unsigned long long>((size_t)B * embed_sorted_len(C));
    w.M = c.take<float>((size_t)F * F + F);  // M [F][F], then v [F]: one fold of the channel chunks' shares
    w.v = w.M + (size_t)F * F;
    w.mvpart = c.take<float>((size_t)ceil_div(C, MV_CH) * ((size_t)F * F + F));
    w.G = c.take<float>((size_t)F * F);
    w.wpart = c.take<float>((size_t)wgrad_chunks(F, F, B * N) * F * F);
    w.m = take_planes(c, F, F);
    w.bytes = c.off;
    return w;
}
static int embed_shape_ok(int B, int N, int F, int C, int ldc, int ldw, const char *who)
{
    DG_REQUIRE(B > 0 && N >= GM && F > 0 && C > 0, "%s: needs B > 0, N >= %d (B=%d N=%d F=%d C=%d)", who, GM, B, N, F, C);
    DG_REQUIRE(F % 32 == 0 && F <= 256 && C % 4 == 0 && C <= 8192, "%s: F a multiple of 32 (<= 256), C a multiple of 4 (<= 8192)", who);
    DG_REQUIRE(ldc >= F && ldc % 4 == 0 && ldw >= F && ldw % 4 == 0, "%s: ldc / ldw must be multiples of 4 and >= F", who);
    DG_REQUIRE((size_t)B * N < ((size_t)1 << 23), "%s: more than 2^23 points", who);
    return 0;
}
}  // namespace hf
}  // namespace dgcnn

extern "C" size_t dgcnn_rows_stats_workspace_bytes(int rows, int cols)
{
    return rows > 0 && cols > 0 ? align_up((size_t)ceil_div(rows, 256) * cols * sizeof(double), 256) + align_up((size_t)ceil_div(rows, 256) * sizeof(float), 256) : 0;
}
extern "C" int dgcnn_rows_stats(const float *x, int ld, int rows, int cols, uint32_t *amax, float *colsum, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && amax && colsum && workspace, "dgcnn_rows_stats: null pointer");
    DG_REQUIRE(rows > 0 && cols > 0 && cols <= 256 && (cols & 3) == 0 && ld >= cols && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0,
               "dgcnn_rows_stats: cols a multiple of 4 (<= 256), 16-byte aligned rows (rows=%d cols=%d ld=%d)", rows, cols, ld);
    DG_REQUIRE(workspace_bytes >= dgcnn_rows_stats_workspace_bytes(rows, cols), "dgcnn_rows_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nrec = ceil_div(rows, 256);
    float *pmax = (float *)((char *)workspace + align_up((size_t)nrec * cols * sizeof(double), 256));
    hipLaunchKernelGGL(rows_stats_kernel, dim3(nrec), dim3(256), 0, st, x, ld, rows, cols, pmax, (double *)workspace);
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(ceil_div(cols, 64)), dim3(64 * FW), 0, st, (const double *)workspace, nrec, cols, colsum,
                       (const float *)pmax, amax);
    DG_CHECK_LAUNCH("dgcnn_rows_stats");
    return 0;
}

extern "C" size_t dgcnn_embed_fwd_workspace_bytes(int B, int N, int F, int C)
{
    return embed_shape_ok(B, N, F, C, F, F, "dgcnn_embed_fwd_workspace_bytes") == 0 ? embed_fwd_ws(B, N, F, C, nullptr).bytes : 0;
}
extern "C" int dgcnn_embed_fwd(const float *cat, int ldc, const float *w, int ldw, int B, int N, int F, int C,
                               const uint32_t *amax_cat, const dgcnn_bn_params *bn, int training, float eps, float momentum,
                               float slope, float *g, int32_t *arg, float *uwin, float *stats, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(cat && w && amax_cat && bn && g && arg && uwin && stats && workspace, "dgcnn_embed_fwd: null pointer");
    if (embed_shape_ok(B, N, F, C, ldc, ldw, "dgcnn_embed_fwd")) return 1;
    DG_REQUIRE(((uintptr_t)cat & 15) == 0 && ((uintptr_t)w & 15) == 0, "dgcnn_embed_fwd: operands must be 16-byte aligned");
    DG_REQUIRE(bn->weight && bn->bias, "dgcnn_embed_fwd: BatchNorm parameters missing");
    DG_REQUIRE(training || (bn->running_mean && bn->running_var), "dgcnn_embed_fwd: eval mode needs running statistics");
    const EmbedFwdWs ws = embed_fwd_ws(B, N, F, C, workspace);
    DG_REQUIRE(workspace_bytes >= ws.bytes, "dgcnn_embed_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int P = B * N, nrb = ceil_div(P, GM);
    {
        WPrepArgs a{};
        a.job[0] = WPrepJob{w, ldw, C, F, 0, ws.w.hi, ws.w.lo, ws.w.scale};
        a.njobs = 1;
        a.first_row[0] = 0;
        a.first_row[1] = C;
        a.zero = ws.keys, a.nzero = (size_t)B * C;  // the winners' keys start at zero (atomic maxima of the product's tiles)
        hipLaunchKernelGGL(wprep_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, a);
    }
    {
        HGemmArgs a{};
        a.X = cat, a.ldx = ldc, a.P = P, a.K = F, a.N = C;
        a.Wh = ws.w.hi, a.Wl = ws.w.lo, a.wscale = ws.w.scale;
        a.xmax = amax_cat;
        a.slope = slope;
        a.rows_per_batch = N;
        a.psum = training ? ws.psum : nullptr;
        a.keys = ws.keys, a.gamma = bn->weight;
        a.ncb = ceil_div(C, GN), a.nrb = nrb;
        prof_begin(PROF_HEAD_CONV6, st);
        if ((a.csplit = row_form(a)) > 0) hipLaunchKernelGGL((hrow_kernel<E_COLRED>), dim3(nrb * a.csplit), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((hgemm_kernel<X_RAW, E_COLRED, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        prof_end(PROF_HEAD_CONV6, st);
    }
    launch_stats_finalize(ws.psum, nullptr, nrb, P, C, training, *bn, eps, momentum, slope, stats, nullptr, ws.fpsum, nullptr, st);
    launch_colmax_finish(ws.keys, stats, B, C, slope, g, arg, uwin, st);
    DG_CHECK_LAUNCH("dgcnn_embed_fwd");
    return 0;
}

extern "C" size_t dgcnn_embed_bwd_workspace_bytes(int B, int N, int F, int C)
{
    return embed_shape_ok(B, N, F, C, F, F, "dgcnn_embed_bwd_workspace_bytes") == 0 ? embed_bwd_ws(B, N, F, C, nullptr).bytes : 0;
}
extern "C" int dgcnn_embed_bwd(const float *cat, int ldc, const uint32_t *amax_cat, const float *catsum, const float *w, int ldw,
                               const int32_t *arg, const float *uwin, const float *stats, const float *gg, int B, int N, int F, int C,
                               int training, float slope, float *dcat, int lddcat, float *dw, float *gbn_w, float *gbn_b,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(lddcat == F, "dgcnn_embed_bwd: dcat must be contiguous (lddcat = F)");
    return embed_bwd_ld(cat, ldc, amax_cat, catsum, w, ldw, arg, uwin, stats, gg, B, N, F, C, training, slope, dcat, lddcat, 0, dw,
                        gbn_w, gbn_b, workspace, workspace_bytes, stream);
}

// dgcnn_embed_bwd with a row stride of dcat and, with dcat_acc, dL/dcat ADDED to what dcat holds (model.hip: the head's tail
// has left its share of dL/dcat there -- model.py:167 and :180 both read the concatenation)
int dgcnn::embed_bwd_ld(const float *cat, int ldc, const uint32_t *amax_cat, const float *catsum, const float *w, int ldw,
                        const int32_t *arg, const float *uwin, const float *stats, const float *gg, int B, int N, int F, int C,
                        int training, float slope, float *dcat, int lddcat, int dcat_acc, float *dw, float *gbn_w, float *gbn_b,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(cat && amax_cat && w && arg && uwin && stats && gg && dcat && dw && gbn_w && gbn_b && workspace, "dgcnn_embed_bwd: null pointer");
    DG_REQUIRE(!training || catsum, "dgcnn_embed_bwd: training mode needs the column sums of cat");
    if (embed_shape_ok(B, N, F, C, ldc, ldw, "dgcnn_embed_bwd")) return 1;
    DG_REQUIRE(lddcat >= F && (lddcat & 3) == 0 && ldw == F, "dgcnn_embed_bwd: lddcat >= F (a multiple of 4), w contiguous (ldw = F)");
    DG_REQUIRE(((uintptr_t)cat & 15) == 0 && ((uintptr_t)dcat & 15) == 0, "dgcnn_embed_bwd: rows must be 16-byte aligned");
    const EmbedBwdWs ws = embed_bwd_ws(B, N, F, C, workspace);
    DG_REQUIRE(workspace_bytes >= ws.bytes, "dgcnn_embed_bwd: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int P = B * N, nrb = ceil_div(P, GM);
    launch_embed_bwd_coef(uwin, gg, stats, B, N, C, training, slope, gbn_w, gbn_b, ws.coef_a, ws.coef_b, ws.s, st, arg, ws.sorted);
    if (training) {
        {
            const int ncs = ceil_div(C, MV_CH);
            hipLaunchKernelGGL(embed_mv_kernel, dim3(ceil_div(F, 32), F / 32 + 1, ncs), dim3(256), 0, st, w, ldw, C, F, ws.coef_a,
                               ws.coef_b, ws.mvpart);
            launch_sum_partials(ws.mvpart, ncs, F * F + F, ws.M, st);
        }
        {   // M is symmetric: its rows are the columns the product contracts with
            WPrepArgs a{};
            a.job[0] = WPrepJob{ws.M, F, F, F, 0, ws.m.hi, ws.m.lo, ws.m.scale};
            a.njobs = 1;
            a.first_row[0] = 0;
            a.first_row[1] = F;
            hipLaunchKernelGGL(wprep_kernel, dim3(ceil_div(F, 4)), dim3(256), 0, st, a);
        }
        {   // dcat = cat M + 1 v^T
            HGemmArgs a{};
            a.X = cat, a.ldx = ldc, a.P = P, a.K = F, a.N = F;
            a.Wh = ws.m.hi, a.Wl = ws.m.lo, a.wscale = ws.m.scale;
            a.xmax = amax_cat;
            a.slope = slope;
            a.add = ws.v, a.rows_per_batch = P;
            a.Y = dcat, a.ldy = lddcat, a.yacc = dcat_acc;
            a.ncb = ceil_div(F, GN), a.nrb = nrb;
            hipLaunchKernelGGL((hgemm_kernel<X_RAW, E_PLAIN, 3>), dim3((unsigned)align_up((size_t)a.ncb * nrb, 8)), dim3(256), 0, st, a);
        }
        {   // G = cat^T cat
            HWgradArgs a{};
            a.A = cat, a.lda = ldc, a.M = F;
            a.Bm = cat, a.ldb = ldc, a.Nn = F, a.P = P;
            const int chunks = wgrad_chunks(F, F, P);
            a.chunk = (ceil_div(P, chunks) + 31) & ~31;
            a.ntiles_n = ceil_div(F, 128);
            a.slope = slope;
            a.ta = amax_cat, a.amax_b = amax_cat;
            a.partial = ws.wpart;
            const int nch = ceil_div(P, a.chunk);
            const dim3 grid((unsigned)align_up((size_t)ceil_div(F, 128) * a.ntiles_n * nch, 8));
            hipLaunchKernelGGL((hwgrad_kernel<A_RAW, B_RAW, 3>), grid, dim3(256), 0, st, a);
            launch_sum_partials(ws.wpart, nch, F * F, ws.G, st);
        }
    } else if (!dcat_acc) {   // eval mode: only the winners' rows are non-zero
        if (hipMemset2DAsync(dcat, (size_t)lddcat * sizeof(float), 0, (size_t)F * sizeof(float), (size_t)P, st) != hipSuccess)
            DG_FAIL("dgcnn_embed_bwd: memset failed");
    }
    hipLaunchKernelGGL(embed_dw_kernel, dim3(ceil_div(C, DW_CH)), dim3(256), 0, st, cat, ldc, w, ldw, arg, ws.s, ws.coef_a, ws.coef_b,
                       ws.G, catsum, B, N, C, F, training, dw);
    launch_embed_bwd_sorted_scatter(ws.sorted, ws.s, w, B, N, C, F, dcat, st, lddcat);
    DG_CHECK_LAUNCH("dgcnn_embed_bwd");
    return 0;
}
