// knn.hip -- fused kNN graph build for gfx950 (MI355X).
//
// Replaces knn(x, k) of the reference (/root/reference/model.py:25-40): the
// B x N x N matrix of -||x_i - x_l||^2 (model.py:35-37) is produced tile by tile in
// MFMA accumulators and consumed immediately by a streaming top-k (model.py:39), so it
// never exists in HBM.
//
// Arithmetic (must stay bit-identical to oracle/knn_exact.c):
//   dot  = fp32 FMA chain over channels c = 0..C-1 from +0  -> v_mfma_f32_32x32x2_f32,
//          which on gfx950 is bitwise a k-ordered fmaf chain (one rounding per product)
//   xx_l = the same chain with both operands x_l             (knn_norm_kernel)
//   t    = fma(2, dot, -xx_l);  pd = (t - xx_i) + 0
//   order: pd descending, candidate index ascending among equal pd
//
// Structure ("wave-autonomous"): one workgroup = ONE wave = 32 query rows (the MFMA M
// dimension) of one batch element, for the whole candidate stream.  Its A operand (the query
// features) stays in registers; the candidates arrive as B operands straight from global
// memory / L2 into registers, one 32-candidate tile ahead, from a copy of x that knn_tile_kernel
// lays out in MFMA operand order (so every load is a fully coalesced 1 KB wave access; all
// waves of a batch element walk the same 1.8 MB through one XCD's L2).  No LDS staging and no
// barrier exists: LDS holds only the top-k state, 16 KB per wave, so 8 waves share a CU and
// one wave's selection work overlaps the other waves' MFMAs.
//
// Streaming top-k per query row: a buffer of CAP 64-bit keys in LDS
//   key = orderable(pd) << 32 | ~candidate      (u64 max == pd desc, index asc)
// holds the current best k plus pending candidates; a candidate is appended only if it
// beats tau = the k-th best pd at the last compaction (ballot + popcount give the slot, the
// per-row counts live in registers: no atomics, no LDS read on the append path).  When a
// row's buffer would overflow, the wave sorts it (bitonic network across the 64 lanes; DPP
// row operations for exchange distances 1, 2, 4, 8), keeps the best k and tightens tau.
#include "common.h"

namespace dgcnn {

constexpr int KNN_ROWPAD = 128;  // query/candidate rows are padded to a multiple of this

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------
// prep 1: squared norms with the oracle's chain order; +inf for padding rows, which makes
// pd = -inf: a padded candidate can never enter a list.
__global__ void knn_norm_kernel(const float *__restrict__ x, int N, int C, int Npad, float *__restrict__ xx)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= Npad) return;
    float acc = __builtin_inff();
    if (n < N) {
        const float *src = x + ((size_t)b * N + n) * C;
        acc = 0.0f;
        for (int c = 0; c < C; ++c) acc = __builtin_fmaf(src[c], src[c], acc);
    }
    xx[(size_t)b * Npad + n] = acc;
}

// prep 2: point-major [B][N][C] -> MFMA operand order.  For the 32x32x2 MFMA, lane (j, h) of a
// 32-point tile supplies channel k = 2 s + h of point j at step s.  Per batch element:
//   xt[tile][g][lane][e] = x[tile * 32 + (lane & 31)][2 * (g * VEC + e) + (lane >> 5)]
// (VEC = min(4, CP / 2) floats per lane and load; zero beyond N or C).  One thread per vector.
template <int VEC>
__global__ void knn_tile_kernel(const float *__restrict__ x, int N, int C, int Npad, int CP,
                                float *__restrict__ xt)
{
    const int G = (CP / 2) / VEC;
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // vector index within the batch element
    const int b = blockIdx.y;
    const size_t nvec = (size_t)(Npad / 32) * G * 64;
    if (v >= nvec) return;
    const int lane = (int)(v & 63);
    const int g = (int)((v >> 6) % G);
    const int tile = (int)((v >> 6) / G);
    const int n = tile * 32 + (lane & 31), h = lane >> 5;
    float out[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = 2 * (g * VEC + e) + h;
        out[e] = (n < N && c < C) ? x[((size_t)b * N + n) * C + c] : 0.0f;
    }
    float *dst = xt + (size_t)b * Npad * CP + v * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = out[e];
}

// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pd_to_ordered(float pd)
{
    const unsigned u = __float_as_uint(pd + 0.0f);  // +0 turns -0 into +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_pd(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ u64 make_key(float pd, int cand)
{
    return ((u64)pd_to_ordered(pd) << 32) | (u64)(0xffffffffu - (unsigned)cand);
}

// value of lane (l ^ J): DPP row operations inside a row of 16 lanes, LDS-crossbar permute beyond
template <int J>
__device__ __forceinline__ unsigned xchg32(unsigned v)
{
    if constexpr (J == 1)
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    else if constexpr (J == 2)
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        // banks 0,2 (lanes 0-3, 8-11) take lane + 4 (row_shl:4); banks 1,3 take lane - 4 (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
    } else if constexpr (J == 8)
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    else
        return (unsigned)__shfl_xor((int)v, J);
}
template <int J>
__device__ __forceinline__ u64 xchg64(u64 v)
{
    const unsigned lo = xchg32<J>((unsigned)v), hi = xchg32<J>((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

template <int EPL, int K2, int J>
__device__ __forceinline__ void bitonic_step(u64 (&key)[EPL], int lane)
{
    if constexpr (J >= 64) {
        // EPL == 2, J == 64, K2 == 128: in-lane exchange, descending everywhere
        const u64 a = key[0], b = key[EPL - 1];
        key[0] = a > b ? a : b;
        key[EPL - 1] = a > b ? b : a;
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int g = e * 64 + lane;
            const u64 o = xchg64<J>(key[e]);
            const bool keep_max = ((g & J) == 0) == ((g & K2) == 0);
            key[e] = ((key[e] > o) == keep_max) ? key[e] : o;  // keys are unique
        }
    }
}
template <int EPL, int K2, int J>
__device__ __forceinline__ void bitonic_merge(u64 (&key)[EPL], int lane)
{
    bitonic_step<EPL, K2, J>(key, lane);
    if constexpr (J > 1) bitonic_merge<EPL, K2, J / 2>(key, lane);
}
template <int EPL, int K2>
__device__ __forceinline__ void bitonic_sort(u64 (&key)[EPL], int lane)
{
    if constexpr (K2 > 2) bitonic_sort<EPL, K2 / 2>(key, lane);
    bitonic_merge<EPL, K2, K2 / 2>(key, lane);
}

// Sort one row's buffer (c valid keys) descending, keep the best k.  Wave-wide call.  Returns the
// new tau (pd of the k-th best, or -inf while fewer than k candidates are known).
template <int CAP>
__device__ __forceinline__ float compact_row(u64 *kr, int c, int k, int lane)
{
    constexpr int EPL = CAP / 64;
    u64 key[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int g = e * 64 + lane;
        key[e] = (g < c) ? kr[g] : 0ull;
    }
    bitonic_sort<EPL, CAP>(key, lane);
#pragma unroll
    for (int e = 0; e < EPL; ++e) kr[e * 64 + lane] = key[e];
    float t = -__builtin_inff();
    if (c >= k) {
        const u64 kk = ((k - 1) >> 6) ? key[EPL - 1] : key[0];
        const unsigned hi = __shfl((unsigned)(kk >> 32), (k - 1) & 63);
        t = ordered_to_pd(hi);
    }
    return t;
}

template <int CP, int CAP>
__global__ __launch_bounds__(64, 2) void knn_kernel(const float *__restrict__ xt, const float *__restrict__ xx,
                                                    int B, int N, int Npad, int k, int32_t *__restrict__ idx)
{
    constexpr int CH = CP / 2;
    constexpr int VEC = CH >= 4 ? 4 : CH;
    constexpr int G = CH / VEC;
    typedef float vecf __attribute__((ext_vector_type(VEC)));

    __shared__ u64 keys[32][CAP];
    __shared__ int cnts[32];
    __shared__ float taus[32];

    const int lane = threadIdx.x, h = lane >> 5, j32 = lane & 31;
    // blocks are dealt round-robin over the XCDs: with B a multiple of 8 all waves of batch
    // element b run on one XCD and share its L2 copy of xt[b] (speed only)
    const int b = blockIdx.x % B, rb = blockIdx.x / B;
    const int q0 = rb * 32;
    const float *xtb = xt + (size_t)b * Npad * CP;
    const float *xxb = xx + (size_t)b * Npad;

    auto load_operand = [&](int tile, float(&dst)[CH]) {
        const vecf *p = reinterpret_cast<const vecf *>(xtb + (size_t)tile * 32 * CP) + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const vecf v = p[g * 64];
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[g * VEC + e] = v[e];
        }
    };

    float a[CH];
    load_operand(rb, a);  // the query rows are tile rb of the same array
    float xxq[16], taur[16];
    int cntr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xxq[r] = xxb[q0 + mfma32_row(r, h)];
        taur[r] = -__builtin_inff();
        cntr[r] = 0;
    }

    // one 32 x 32 tile of distances -> selection
    auto process = [&](const float(&bop)[CH], float xxc, int cand) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < CH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bop[s], acc, 0, 0, 0);

        // pass 1: append where the row's buffer has room, flag the rows that would overflow
        unsigned retry = 0u;  // bit (r + 16 * half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pd = __builtin_fmaf(2.0f, acc[r], -xxc) - xxq[r];
            const bool pass = pd > taur[r];
            const u64 m = __ballot(pass);
            if (m) {
                const int row = mfma32_row(r, h);
                const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
                const int tot = __popc(mh), c = cntr[r];
                const bool ovf = c + tot > CAP;
                const u64 mo = __ballot(ovf);
                if (!ovf) {
                    if (pass) {
                        float pdk = pd;
                        asm volatile("" : "+v"(pdk));  // keep the key arithmetic on this (rare) path
                        keys[row][c + __popc(mh & ((1u << j32) - 1u))] = make_key(pdk, cand);
                    }
                    cntr[r] = c + tot;
                }
                if (mo) {
                    if (ovf && j32 == 0) cnts[row] = c;
                    if ((unsigned)mo) retry |= 1u << r;
                    if (mo >> 32) retry |= 1u << (r + 16);
                }
            }
        }
        if (retry) {
            // compact every flagged row (runtime loop: one copy of the sort), then offer the
            // candidates of those rows again: count <= k and at most 32 newcomers always fit
            unsigned rr = retry;
            while (rr) {
                const int bit = __ffs(rr) - 1;
                rr &= rr - 1;
                const int row = mfma32_row(bit & 15, bit >> 4);
                const int c = cnts[row];
                const float tnew = compact_row<CAP>(&keys[row][0], c, k, lane);
                if (lane == 0) {
                    cnts[row] = c < k ? c : k;
                    taus[row] = tnew;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (retry & (0x10001u << r)) {
                    const int row = mfma32_row(r, h);
                    const bool mine = (retry >> (r + 16 * h)) & 1u;
                    if (mine) {
                        taur[r] = taus[row];
                        cntr[r] = cnts[row];
                    }
                    const float pd = __builtin_fmaf(2.0f, acc[r], -xxc) - xxq[r];
                    const bool pass = mine && (pd > taur[r]);
                    const u64 m = __ballot(pass);
                    const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
                    if (pass) keys[row][cntr[r] + __popc(mh & ((1u << j32) - 1u))] = make_key(pd, cand);
                    cntr[r] += __popc(mh);
                }
            }
        }
    };

    const int ntiles = Npad / 32;  // even (Npad is a multiple of 128)
    float b0[CH], b1[CH];
    load_operand(0, b0);
    float x0 = xxb[j32], x1;
    for (int t = 0; t < ntiles; t += 2) {
        load_operand(t + 1, b1);
        x1 = xxb[(t + 1) * 32 + j32];
        process(b0, x0, t * 32 + j32);
        if (t + 2 < ntiles) {
            load_operand(t + 2, b0);
            x0 = xxb[(t + 2) * 32 + j32];
        }
        process(b1, x1, (t + 1) * 32 + j32);
    }

    // ---- final sort and write-out ------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (j32 == 0) cnts[mfma32_row(r, h)] = cntr[r];
    for (int row = 0; row < 32; ++row) {
        const int c = cnts[row];
        (void)compact_row<CAP>(&keys[row][0], c, k, lane);
        const int q = q0 + row;
        if (q < N) {
            for (int g = lane; g < k; g += 64) {
                const unsigned lo = (unsigned)keys[row][g];
                idx[((size_t)b * N + q) * k + g] = (int32_t)(0xffffffffu - lo);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
static int knn_cp(int C)
{
    int cp = 4;
    while (cp < C) cp <<= 1;
    return cp;
}
static int knn_npad(int N) { return ceil_div(N, KNN_ROWPAD) * KNN_ROWPAD; }

template <int CP>
static void knn_launch_cp(int cap, int B, hipStream_t st, const float *xt, const float *xx, int N, int Npad, int k,
                          int32_t *idx)
{
    const dim3 grid(B * (Npad / 32)), block(64);
    if (cap == 64)
        hipLaunchKernelGGL((knn_kernel<CP, 64>), grid, block, 0, st, xt, xx, B, N, Npad, k, idx);
    else
        hipLaunchKernelGGL((knn_kernel<CP, 128>), grid, block, 0, st, xt, xx, B, N, Npad, k, idx);
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_knn_workspace_bytes(int B, int C, int N, int k)
{
    (void)k;
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t npad = knn_npad(N), cp = knn_cp(C);
    return align_up((size_t)B * npad * cp * sizeof(float), 256) +
           align_up((size_t)B * npad * sizeof(float), 256);
}

extern "C" int dgcnn_knn(const float *x, int B, int N, int C, int k, int32_t *idx, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(x && idx && workspace, "dgcnn_knn: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0, "dgcnn_knn: bad shape B=%d N=%d C=%d", B, N, C);
    DG_REQUIRE(C <= 128, "dgcnn_knn: C=%d > 128 unsupported", C);
    // the reference raises from topk when k > N (model.py:39)
    DG_REQUIRE(k >= 1 && k <= N, "dgcnn_knn: selected index k out of range (k=%d, N=%d)", k, N);
    DG_REQUIRE(k <= 96, "dgcnn_knn: k=%d > 96 unsupported", k);
    DG_REQUIRE(workspace_bytes >= dgcnn_knn_workspace_bytes(B, C, N, k), "dgcnn_knn: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Npad = knn_npad(N), CP = knn_cp(C);
    float *xt = (float *)workspace;
    float *xx = (float *)((char *)workspace + align_up((size_t)B * Npad * CP * sizeof(float), 256));

    hipLaunchKernelGGL(knn_norm_kernel, dim3(ceil_div(Npad, 256), B), dim3(256), 0, st, x, N, C, Npad, xx);
    const size_t nvec = (size_t)Npad * CP / (CP >= 8 ? 4 : 2);
    const dim3 tgrid((unsigned)((nvec + 255) / 256), B);
    if (CP >= 8)
        hipLaunchKernelGGL((knn_tile_kernel<4>), tgrid, dim3(256), 0, st, x, N, C, Npad, CP, xt);
    else
        hipLaunchKernelGGL((knn_tile_kernel<2>), tgrid, dim3(256), 0, st, x, N, C, Npad, CP, xt);
    DG_CHECK_LAUNCH("knn prep kernels");

    const int cap = (k <= 32) ? 64 : 128;
    switch (CP) {
        case 4: knn_launch_cp<4>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        case 8: knn_launch_cp<8>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        case 16: knn_launch_cp<16>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        case 32: knn_launch_cp<32>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        case 64: knn_launch_cp<64>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        case 128: knn_launch_cp<128>(cap, B, st, xt, xx, N, Npad, k, idx); break;
        default: DG_FAIL("dgcnn_knn: unsupported padded channel count %d", CP);
    }
    DG_CHECK_LAUNCH("knn_kernel");
    return 0;
}
