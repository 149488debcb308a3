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
//   xx_l = the same chain with both operands x_l             (knn_prep_kernel)
//   t    = fma(2, dot, -xx_l);  pd = (t - xx_i) + 0
//   order: pd descending, candidate index ascending among equal pd
//
// Structure: one workgroup = 4 waves = 128 query rows of one batch element; every wave
// owns 32 query rows (the MFMA M dimension) for the whole kernel: its A operand (the
// query features) stays in registers, its top-k state is wave-private, so the only
// workgroup-level synchronisation is the hand-over of the shared candidate tile (one
// barrier per 64 candidates, register-prefetched double buffer).
//
// Streaming top-k per query row: a buffer of CAP 64-bit keys in LDS
//   key = orderable(pd) << 32 | ~candidate      (u64 max == pd desc, index asc)
// holds the current best k plus pending candidates; a candidate is appended only if it
// beats tau = the k-th best pd at the last compaction (ballot + popcount give the slot,
// no atomics).  When a row's buffer would overflow, the wave sorts it (bitonic network
// across the 64 lanes via shuffles), keeps the best k and tightens tau.
#include "common.h"

namespace dgcnn {

constexpr int KNN_ROWPAD = 128;  // query/candidate rows are padded to a multiple of this
constexpr int KNN_BN = 64;       // candidates per staged tile

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------
// prep: point-major [B][N][C] -> de-interleaved, padded rows + squared norms.
//   xs[b][n][h * CH + s] = x[b][n][2 s + h]   (h = channel parity, CH = CP / 2)
// so that lane-half h of the 32x32x2 MFMA (which supplies k = 2 s + h at step s) reads a
// contiguous run.  Rows n >= N are zero with xx = +inf, which makes pd = -inf: a padded
// candidate can never enter a list.
__global__ void knn_prep_kernel(const float *__restrict__ x, int N, int C, int Npad, int CP,
                                float *__restrict__ xs, float *__restrict__ xx)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= Npad) return;
    const int CH = CP >> 1;
    float *dst = xs + ((size_t)b * Npad + n) * CP;
    if (n < N) {
        const float *src = x + ((size_t)b * N + n) * C;
        float acc = 0.0f;
        for (int c = 0; c < CP; ++c) {
            const float v = (c < C) ? src[c] : 0.0f;
            acc = __builtin_fmaf(v, v, acc);
            dst[(c & 1) * CH + (c >> 1)] = v;
        }
        xx[(size_t)b * Npad + n] = acc;
    } else {
        for (int c = 0; c < CP; ++c) dst[c] = 0.0f;
        xx[(size_t)b * Npad + n] = __builtin_inff();
    }
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

// bitonic sort, descending, of CAP = 64 * EPL keys held EPL per lane (element e of lane l
// has position e * 64 + l).
template <int EPL>
__device__ __forceinline__ void bitonic_desc(u64 (&key)[EPL], int lane)
{
#pragma unroll
    for (int k2 = 2; k2 <= 64 * EPL; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                // EPL == 2, j == 64, k2 == 128: in-lane exchange, descending everywhere
                const u64 a = key[0], b = key[EPL - 1];
                key[0] = a > b ? a : b;
                key[EPL - 1] = a > b ? b : a;
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int g = e * 64 + lane;
                    const u64 o = __shfl_xor(key[e], j);
                    const bool keep_max = ((g & j) == 0) == ((g & k2) == 0);
                    const u64 mx = key[e] > o ? key[e] : o;
                    const u64 mn = key[e] > o ? o : key[e];
                    key[e] = keep_max ? mx : mn;
                }
            }
        }
    }
}

// Sort one row's buffer (c valid keys), keep the best k.  Wave-wide call.  Returns the new
// tau (pd of the k-th best, or -inf while fewer than k candidates are known).
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
    bitonic_desc<EPL>(key, lane);
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

// WAVES waves per workgroup, 32 query rows each (4 for CAP = 64; 2 for CAP = 128 so that
// the key buffers still fit the 160 KB LDS).
template <int CP, int CAP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void knn_kernel(const float *__restrict__ xs,
                                                          const float *__restrict__ xx, int N,
                                                          int Npad, int k,
                                                          int32_t *__restrict__ idx)
{
    constexpr int KNN_WAVES = WAVES, KNN_THREADS = WAVES * 64, KNN_BM = WAVES * 32;
    constexpr int CH = CP / 2;
    constexpr int STR = CH + (CH >= 4 ? 4 : 0);  // +16 B: conflict-free ds_read_b128 rows
    constexpr int F4_TILE = KNN_BN * CP / 4;
    constexpr int F4_THREAD = (F4_TILE + KNN_THREADS - 1) / KNN_THREADS;

    __shared__ __attribute__((aligned(16))) float tile[2][2][KNN_BN][STR];
    __shared__ u64 keys[KNN_WAVES][32][CAP];
    __shared__ int cnts[KNN_WAVES][32];
    __shared__ float taus[KNN_WAVES][32];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j32 = lane & 31;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * KNN_BM + wave * 32;
    const float *xsb = xs + (size_t)b * Npad * CP;
    const float *xxb = xx + (size_t)b * Npad;

    if (lane < 32) {
        cnts[wave][lane] = 0;
        taus[wave][lane] = -__builtin_inff();
    }

    // A operand: lane (row j32, half h) supplies A[row][k = 2 s + h] at step s.
    float a[CH];
    {
        const float *src = xsb + (size_t)(q0 + j32) * CP + h * CH;
#pragma unroll
        for (int s = 0; s < CH; ++s) a[s] = src[s];
    }
    float xxq[16], taur[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xxq[r] = xxb[q0 + mfma32_row(r, h)];
        taur[r] = -__builtin_inff();
    }

    float4 stage[F4_THREAD];
    auto load_tile = [&](int t) {
        const float4 *g = reinterpret_cast<const float4 *>(xsb + (size_t)t * KNN_BN * CP);
#pragma unroll
        for (int i = 0; i < F4_THREAD; ++i) {
            const int q = threadIdx.x + i * KNN_THREADS;
            if (q < F4_TILE) stage[i] = g[q];
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < F4_THREAD; ++i) {
            const int q = threadIdx.x + i * KNN_THREADS;
            if (q < F4_TILE) {
                const int row = q / (CP / 4), w = q % (CP / 4);
                if constexpr (CH >= 4) {
                    const int plane = w / (CH / 4), s4 = w % (CH / 4);
                    *reinterpret_cast<float4 *>(&tile[buf][plane][row][s4 * 4]) = stage[i];
                } else {  // CP == 4: one float4 = [plane0: s0 s1 | plane1: s0 s1]
                    tile[buf][0][row][0] = stage[i].x;
                    tile[buf][0][row][1] = stage[i].y;
                    tile[buf][1][row][0] = stage[i].z;
                    tile[buf][1][row][1] = stage[i].w;
                }
            }
        }
    };

    const int ntiles = Npad / KNN_BN;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_tile(t + 1);

#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int cand = t * KNN_BN + nt * 32 + j32;
            const float xxc = xxb[cand];
            const float *brow = &tile[buf][h][nt * 32 + j32][0];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < CH; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], brow[s], acc, 0, 0, 0);

            // ---- selection epilogue -------------------------------------------------
            // todo bit (r + 16 * half): that (register, half) = query row still has to
            // offer its candidate.  Rows whose buffer would overflow are compacted and
            // retried; everything in this loop is wave-uniform control flow.
            unsigned todo = 0xffffffffu;
            while (todo) {
                unsigned retry = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (todo & (0x10001u << r)) {
                        const float tt = __builtin_fmaf(2.0f, acc[r], -xxc);
                        const float pd = tt - xxq[r];
                        const bool mine = (todo >> (r + 16 * h)) & 1u;
                        const bool pass = mine && (pd > taur[r]);
                        const u64 m = __ballot(pass);
                        if (m) {
                            const int row = mfma32_row(r, h);
                            const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
                            const int tot = __popc(mh);
                            const int c = cnts[wave][row];
                            const bool ovf = pass && (c + tot > CAP);
                            const u64 mo = __ballot(ovf);
                            const bool my_ovf = h ? (mo >> 32) != 0 : ((unsigned)mo) != 0;
                            if ((unsigned)mo) retry |= 1u << r;
                            if (mo >> 32) retry |= 1u << (r + 16);
                            if (pass && !my_ovf) {
                                const int slot = c + __popc(mh & ((1u << j32) - 1u));
                                keys[wave][row][slot] = make_key(pd, cand);
                            }
                            if (!my_ovf && j32 == 0 && tot) cnts[wave][row] = c + tot;
                        }
                    }
                }
                if (retry) {
                    // compact every flagged row (runtime loop: one copy of the sort)
                    unsigned rr = retry;
                    while (rr) {
                        const int bit = __ffs(rr) - 1;
                        rr &= rr - 1;
                        const int row = mfma32_row(bit & 15, bit >> 4);
                        const int c = cnts[wave][row];
                        const float tnew = compact_row<CAP>(&keys[wave][row][0], c, k, lane);
                        if (lane == 0) {
                            cnts[wave][row] = c < k ? c : k;
                            taus[wave][row] = tnew;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (retry & (0x10001u << r)) taur[r] = taus[wave][mfma32_row(r, h)];
                }
                todo = retry;
            }
        }

        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- final sort and write-out ------------------------------------------------------
    for (int row = 0; row < 32; ++row) {
        const int c = cnts[wave][row];
        (void)compact_row<CAP>(&keys[wave][row][0], c, k, lane);
        const int q = q0 + row;
        if (q < N) {
            for (int g = lane; g < k; g += 64) {
                const unsigned lo = (unsigned)keys[wave][row][g];
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
static void knn_launch_cp(int cap, int B, hipStream_t st, const float *xs, const float *xx,
                          int N, int Npad, int k, int32_t *idx)
{
    if (cap == 64)
        hipLaunchKernelGGL((knn_kernel<CP, 64, 4>), dim3(Npad / 128, B), dim3(256), 0, st, xs, xx, N, Npad, k, idx);
    else
        hipLaunchKernelGGL((knn_kernel<CP, 128, 2>), dim3(Npad / 64, B), dim3(128), 0, st, xs, xx, N, Npad, k, idx);
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
    float *xs = (float *)workspace;
    float *xx = (float *)((char *)workspace + align_up((size_t)B * Npad * CP * sizeof(float), 256));

    hipLaunchKernelGGL(knn_prep_kernel, dim3(ceil_div(Npad, 256), B), dim3(256), 0, st, x, N, C, Npad, CP, xs, xx);
    DG_CHECK_LAUNCH("knn_prep_kernel");

    const int cap = (k <= 32) ? 64 : 128;
    switch (CP) {
        case 4: knn_launch_cp<4>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        case 8: knn_launch_cp<8>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        case 16: knn_launch_cp<16>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        case 32: knn_launch_cp<32>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        case 64: knn_launch_cp<64>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        case 128: knn_launch_cp<128>(cap, B, st, xs, xx, N, Npad, k, idx); break;
        default: DG_FAIL("dgcnn_knn: unsupported padded channel count %d", CP);
    }
    DG_CHECK_LAUNCH("knn_kernel");
    return 0;
}
