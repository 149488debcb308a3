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
// Threshold pre-pass (knn_bound_kernel): before the exact stream, a cheap pass computes for every
// query row a LOWER bound tau0 of its k-th best pd, so that the exact stream starts with a tight
// threshold instead of -inf (about 40 appends and no compaction per row instead of about 250
// appends and 8 sorts).  Lane (j, h) keeps the running maximum of its candidate class (candidates
// with index = j mod 32, even / odd tiles: 64 classes per row); the k-th largest of the 64 class
// maxima is attained by k distinct candidates, hence is a valid lower bound.  For C >= 16 the
// pre-pass runs on fp16 MFMA (v_mfma_f32_32x32x16_f16, 1/16 of the fp32 cost) over mean-centred
// features and tau0 is loosened by an a-priori error bound; C <= 4 (the raw point features) takes the same fp16 path
// with every channel as two fp16 pieces in one 16-deep slab (knn_prep_split4_kernel); for 4 < C <= 8 it is the exact
// fp32 arithmetic loosened by one ulp.  The bound only affects speed: if a row ends the exact stream
// with fewer than k candidates (tau0 was not a lower bound), the wave repeats its stream from
// tau = -inf, so the result never depends on the pre-pass.
//
// Streaming top-k per query row: a buffer of CAP 64-bit keys in LDS
//   key = orderable(pd) << 32 | ~candidate      (u64 max == pd desc, index asc)
// holds the current best k plus pending candidates; a candidate is appended only if it
// beats tau = the k-th best pd at the last compaction (ballot + popcount give the slot, the
// per-row counts live in registers: no atomics, no LDS read on the append path).  When a
// row's buffer would overflow, the wave sorts it (bitonic network across the 64 lanes; DPP
// row operations for exchange distances 1, 2, 4, 8), keeps the best k and tightens tau.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "prof.h"

namespace dgcnn {
int knn_ld(const float *x, int ldx, int B, int N, int C, int k, int32_t *idx, void *workspace, size_t workspace_bytes,
           void *stream);  // (declared for the other translation units in edge_kernels.h)
}  // namespace dgcnn

namespace dgcnn {

constexpr int KNN_ROWPAD = 128;  // query/candidate rows are padded to a multiple of this

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------
// prep 1: squared norms with the oracle's chain order; +inf for padding rows, which makes
// pd = -inf: a padded candidate can never enter a list.
__global__ void knn_norm_kernel(const float *__restrict__ x, int ldx, int N, int C, int Npad, float *__restrict__ xx)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= Npad) return;
    float acc = __builtin_inff();
    if (n < N) {
        const float *src = x + ((size_t)b * N + n) * ldx;
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
__global__ void knn_tile_kernel(const float *__restrict__ x, int ldx, int N, int C, int Npad, int CP,
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
        out[e] = (n < N && c < C) ? x[((size_t)b * N + n) * ldx + c] : 0.0f;
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
    else if constexpr (J == 16) {
        // v_permlane16_swap(v, v): first = (r0 r0 r2 r2), second = (r1 r1 r3 r3) in rows of 16 lanes
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? p[0] : p[1];
    } else {
        // v_permlane32_swap(v, v): first = (lo lo), second = (hi hi) in halves of 32 lanes
        const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? p[0] : p[1];
    }
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
// STORE (<= CAP) = entries the row's LDS buffer can hold: the sort network always spans CAP lanes x registers,
// positions beyond the valid count are padded with the smallest key.
template <int CAP, int STORE = CAP>
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
    for (int e = 0; e < EPL; ++e)
        if (STORE == CAP || e * 64 + lane < STORE) kr[e * 64 + lane] = key[e];
    float t = -__builtin_inff();
    if (c >= k) {
        const u64 kk = ((k - 1) >> 6) ? key[EPL - 1] : key[0];
        const unsigned hi = __shfl((unsigned)(kk >> 32), (k - 1) & 63);
        t = ordered_to_pd(hi);
    }
    return t;
}

// ------------------------------------------------------------------------------------
// pre-pass helpers
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// constants and helper of the augmented K slab of the fp16 streams (see "fp16 streams" below)
constexpr float KNN_AUG_K1 = 32768.0f, KNN_AUG_K2 = 16.0f, KNN_AUG_K3 = 0.0078125f;
// per-pair slack of the fp16 dot product: 2 * 2^-10 * 1.01 = 1.973e-3, with room for the fp16 rounding of its two factors
constexpr float KNN_PAIR_SLACK = 2.1e-3f;
// the same slack for SPLIT operands (C <= 4, see knn_prep_split4_kernel): the product is hi.hi + hi.lo + lo.hi of two-piece
// fp16 operands -- the dropped lo.lo term, the rounding of lo and the fp32 accumulation are each below 2^-22 |x| |y|
// WHILE lo is a normal fp16 number.  The prep kernel applies no power-of-two scaling, so for centred coordinates below
// about 2^-3 (normalised clouds) lo is subnormal or zero and its rounding is ABSOLUTE: 2^-25 per element (fp16's subnormal
// spacing 2^-24, halved), i.e. up to 2^-25 (|x| + |y|) per product term, which for |x| below about 0.03 exceeds this relative
// slack.  That absolute term is covered by the LINEAR term of knn_f16_eps (1e-6 (|y_i| + sqrt(cmax)): 2^-25 = 3e-8 per
// element times C <= 4 elements is 30 times smaller), and it relies on the matrix instruction not flushing fp16 subnormals
// (gfx950's MFMA keeps them).  tests/test_knn_gpu.py::test_knn_small_c_filter_path_is_exact runs C <= 4 at scales 1e-1,
// 1e-2, 1e-3 without an offset for exactly this range.
constexpr float KNN_PAIR_SLACK_SPLIT = 4.0e-6f;
constexpr float KNN_PAD_NORM = 1073741824.0f;  // 2^30: the "norm" of a padding candidate (never a neighbour, no inf - inf)
constexpr float KNN_F16_CMAX = 1.0e8f;         // largest centred norm the fp16 streams accept (|T| stays below 2^31)

// v = 2^15 p1 + 2^4 p2 + 2^-7 p3 up to max(2^-33 |v|, 2^-32), every piece an fp16 number: each residual is exact in
// fp32 and is scaled back up by 2^11 before it is rounded, so no piece relies on fp16 subnormals for its precision.
// |v| >= 2^31 (and +-inf) becomes (+-inf, 0, 0).  v must not be NaN.
struct Half3 {
    _Float16 p1, p2, p3;
};
__device__ __forceinline__ Half3 knn_split3(float v)
{
    Half3 o;
    if (!(__builtin_fabsf(v) < 2147483648.0f)) {
        o.p1 = (_Float16)(v > 0.0f ? __builtin_inff() : -__builtin_inff());
        o.p2 = o.p3 = (_Float16)0.0f;
        return o;
    }
    const float s1 = v * 0x1p-15f;
    o.p1 = (_Float16)s1;
    const float s2 = (s1 - (float)o.p1) * 2048.0f;
    o.p2 = (_Float16)s2;
    const float s3 = (s2 - (float)o.p2) * 2048.0f;
    o.p3 = (_Float16)s3;
    return o;
}


// approximate centre of one batch element (any vector works: distances are translation invariant; centring only
// tightens the fp16 error bound): the mean of the first <= 64 points, computed by every workgroup of knn_prep16_kernel
// itself in the same fixed order (16 KB of L2-resident reads instead of a launch of its own)
__device__ __forceinline__ void knn_centre(const float *__restrict__ x, int ldx, int b, int N, int C, float *red /* [256] */,
                                           float *ms /* [128] */)
{
    const int G = 256 / C > 0 ? 256 / C : 1;  // C <= 128
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    const int n1 = N < 64 ? N : 64;
    float s = 0.0f;
    if (g < G)
        for (int n = g; n < n1; n += G) s += x[((size_t)b * N + n) * ldx + c];
    red[threadIdx.x] = (g < G) ? s : 0.0f;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float t = 0.0f;
        for (int i = 0; i < G; ++i) t += red[i * C + threadIdx.x];
        ms[threadIdx.x] = t / (float)n1;
    }
}

// Maximum over a batch element of the per-tile norm maxima tmax[b][which][ntiles] (uint bits of non-negative
// floats; which = 0 centred, 1 raw).  Wave-wide call, uniform result.
__device__ __forceinline__ float batch_max(const unsigned *__restrict__ tmax, int b, int which, int ntiles)
{
    const unsigned *p = tmax + ((size_t)b * 2 + which) * ntiles;
    unsigned m = 0u;
    for (int i = (int)(threadIdx.x & 63); i < ntiles; i += 64) m = m > p[i] ? m : p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(m, o);
        m = m > t ? m : t;
    }
    return __uint_as_float(m);
}

// Preparation for C >= 16 in one pass over x: one workgroup = one 32-point tile, rows staged in LDS.
//   xx   raw squared norms in the oracle's chain order (+inf for padding rows)
//   xxc  squared norms of the mean-centred features (any order: they only feed the fp16 streams and their bound)
//   xh   centred features as fp16 operand tiles of v_mfma_f32_32x32x16_f16: lane (j, h) of a 32-point tile
//        supplies, at step ks, channels 16 ks + 8 h .. + 7 of point j; xh[tile][ks][lane] is one 16-byte vector;
//        step ks = CP / 16 is the candidate side of the augmented slab (see "fp16 streams"): the three fp16 pieces
//        of the point's centred norm (2^30 for padding points) and the constants that meet the query side's pieces
//   xt   raw features as fp32 operand tiles (the layout of knn_tile_kernel<4>) for the fallback kernel
//   bmax[b][which][tile] = largest centred (which = 0) / raw (1) norm of the tile, as uint bits of non-negative
//        floats (NaN bits win the comparison, as they must); batch_max() reduces them where they are needed
//   rbflag[b][tile] = 0: the fallback flag of the row block, set later by the filter / refine kernels
// CC: the channel count when it is known at compile time (64: the index arithmetic is shifts; 0: C and CP at run time)
template <int CC>
__global__ __launch_bounds__(256) void knn_prep16_kernel(const float *__restrict__ x, int ldx,
                                                         int N, int C_, int Npad, int CP_, float *__restrict__ xx,
                                                         float *__restrict__ xxc, half8 *__restrict__ xh,
                                                         float *__restrict__ xt, unsigned *__restrict__ bmax,
                                                         int *__restrict__ rbflag)
{
    __shared__ float xs[32][129];  // C <= 128; odd stride: conflict-free column walks
    __shared__ float ms[128];
    __shared__ float cred[256];
    __shared__ float yys[32];
    const int tid = threadIdx.x;
    const int C = CC ? CC : C_, CP = CC ? CC : CP_;
    const int b = blockIdx.y, tile = blockIdx.x, q0 = tile * 32;
    // the tile's rows are on their way while the centre is computed (C <= 128: at most 16 elements per thread)
    float trow[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = tid + 256 * u;
        trow[u] = (i < 32 * C && q0 + i / C < N) ? x[((size_t)b * N + q0 + i / C) * ldx + i % C] : 0.0f;
    }
    knn_centre(x, ldx, b, N, C, cred, ms);
    if (tid == 0) rbflag[(size_t)b * (Npad / 32) + tile] = 0;  // fallback flag of this row block (set by filter / refine)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = tid + 256 * u;
        if (i < 32 * C) xs[i / C][i % C] = trow[u];
    }
    __syncthreads();
    // norms: wave 0 lanes 0..31 the raw chain, wave 1 lanes 0..31 the centred one
    if ((tid & 63) < 32 && tid < 128) {
        const int r = tid & 31;
        const bool centred = tid >= 64;
        // one fma chain in channel order (the oracle's); the LDS reads of sixteen channels are issued together, so
        // that the chain pays the fma latency per channel and not an LDS round trip (channels past C add fma(0, 0, acc))
        float acc = 0.0f;
        for (int c0 = 0; c0 < C; c0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = c0 + u < C ? xs[r][c0 + u] - (centred ? ms[c0 + u] : 0.0f) : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(v[u], v[u], acc);
        }
        const bool real = q0 + r < N;
        (centred ? xxc : xx)[(size_t)b * Npad + q0 + r] = real ? acc : __builtin_inff();
        if (centred) yys[r] = real ? acc : KNN_PAD_NORM;
        unsigned m = real ? __float_as_uint(acc) : 0u;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const unsigned t = __shfl_xor(m, o);
            m = m > t ? m : t;
        }
        if (r == 0) bmax[((size_t)b * 2 + (centred ? 0 : 1)) * (Npad / 32) + tile] = m;
    }
    // fp16 operand tiles: vector v = ks * 64 + lane; lane (j, h) holds channels 16 ks + 8 h .. + 7 of point j
    const int KS = CP / 16;
    for (int v = tid; v < KS * 64; v += 256) {
        const int lane = v & 63, ks = v >> 6, r = lane & 31, h = lane >> 5;
        half8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 8 * h + e;
            out[e] = (_Float16)((q0 + r < N && c < C) ? xs[r][c] - ms[c] : 0.0f);
        }
        xh[((size_t)b * (Npad / 32) + tile) * (KS + 1) * 64 + v] = out;
    }
    __syncthreads();  // yys
    if (tid < 64) {
        half8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = (_Float16)0.0f;
        if (tid < 32) {
            const Half3 p = knn_split3(yys[tid]);
            out[0] = p.p1;
            out[1] = p.p2;
            out[2] = p.p3;
            out[3] = (_Float16)KNN_AUG_K1;
            out[4] = (_Float16)KNN_AUG_K2;
            out[5] = (_Float16)KNN_AUG_K3;
            out[6] = (_Float16)(q0 + tid < N ? __builtin_sqrtf(yys[tid]) : 0.0f);  // candidate factor of the pair slack
        }
        xh[((size_t)b * (Npad / 32) + tile) * (KS + 1) * 64 + KS * 64 + tid] = out;
    }
    // fp32 operand tiles (knn_tile_kernel<4>): vector v = g * 64 + lane, element e = channel 2 (4 g + e) + h
    const int G = CP / 8;
    for (int v = tid; v < G * 64; v += 256) {
        const int lane = v & 63, g = v >> 6, r = lane & 31, h = lane >> 5;
        float4 out;
        float *o = reinterpret_cast<float *>(&out);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 2 * (g * 4 + e) + h;
            o[e] = (q0 + r < N && c < C) ? xs[r][c] : 0.0f;
        }
        reinterpret_cast<float4 *>(xt + (size_t)b * Npad * CP + (size_t)tile * 32 * CP)[v] = out;
    }
}

// Preparation for C <= 4 (the raw point features of the first EdgeConv block, model.py:147): the fp16 streams take the
// features as TWO fp16 pieces per channel, y' = hi + lo (|y' - y| <= 2^-22 |y|), laid out so that one 16-deep slab carries
// hi.hi + hi.lo + lo.hi (knn_stream_a<.., SPLIT>): the all-pairs passes of this block run on the matrix cores like those of
// the 64-channel blocks (their exact fp32 forms, knn_bound_kernel<4> + knn_filter32_kernel<4>, are vector-ALU passes:
// 0.49 ms of a training step at B = 32, N = 7000).  One wave = one 32-point tile (a workgroup takes four); everything
// knn_prep16_kernel leaves is left here too: xx, xxc, xh (data slab + augmented slab), xt (the layout of
// knn_tile_kernel<2>, CP = 4), bmax, rbflag.
__global__ __launch_bounds__(256) void knn_prep_split4_kernel(const float *__restrict__ x, int ldx, int N, int C, int Npad,
                                                              float *__restrict__ xx, float *__restrict__ xxc,
                                                              half8 *__restrict__ xh, float *__restrict__ xt,
                                                              unsigned *__restrict__ bmax, int *__restrict__ rbflag)
{
    __shared__ float ms[128];
    __shared__ float cred[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, tile = blockIdx.x * 4 + wave, ntiles = Npad / 32;  // (Npad is a multiple of 128)
    const int n = tile * 32 + j;
    const bool real = n < N;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (real && c < C) ? x[((size_t)b * N + n) * ldx + c] : 0.0f;
    knn_centre(x, ldx, b, N, C, cred, ms);
    __syncthreads();
    if (lane == 0) rbflag[(size_t)b * ntiles + tile] = 0;
    float raw = 0.0f, cen = 0.0f, d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        d[c] = c < C ? v[c] - ms[c] : 0.0f;
        raw = __builtin_fmaf(v[c], v[c], raw);  // (channels past C add fma(0, 0, acc): the oracle's chain over c < C)
        cen = __builtin_fmaf(d[c], d[c], cen);
    }
    if (h == 0) {
        xx[(size_t)b * Npad + n] = real ? raw : __builtin_inff();
        xxc[(size_t)b * Npad + n] = real ? cen : __builtin_inff();
    }
    unsigned mr = real ? __float_as_uint(raw) : 0u, mc = real ? __float_as_uint(cen) : 0u;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        const unsigned t0 = __shfl_xor(mr, o), t1 = __shfl_xor(mc, o);
        mr = mr > t0 ? mr : t0;
        mc = mc > t1 ? mc : t1;
    }
    if (lane == 0) {
        bmax[((size_t)b * 2 + 0) * ntiles + tile] = mc;
        bmax[((size_t)b * 2 + 1) * ntiles + tile] = mr;
    }
    half8 out;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const _Float16 hi = (_Float16)(real ? d[c] : 0.0f);
        out[c] = hi;
        out[4 + c] = (_Float16)(real ? d[c] - (float)hi : 0.0f);
    }
    half8 *xtile = xh + ((size_t)b * ntiles + tile) * 2 * 64;
    xtile[lane] = out;
    half8 aug;
#pragma unroll
    for (int e = 0; e < 8; ++e) aug[e] = (_Float16)0.0f;
    if (h == 0) {
        const float yy = real ? cen : KNN_PAD_NORM;
        const Half3 p = knn_split3(yy);
        aug[0] = p.p1;
        aug[1] = p.p2;
        aug[2] = p.p3;
        aug[3] = (_Float16)KNN_AUG_K1;
        aug[4] = (_Float16)KNN_AUG_K2;
        aug[5] = (_Float16)KNN_AUG_K3;
        aug[6] = (_Float16)(real ? __builtin_sqrtf(cen) : 0.0f);  // candidate factor of the pair slack
    }
    xtile[64 + lane] = aug;
    // fp32 operand tile of the exact fallback (knn_tile_kernel<2>, CP = 4): element e = channel 2 e + h
    float2 o2;
    o2.x = h == 0 ? v[0] : v[1];
    o2.y = h == 0 ? v[2] : v[3];
    reinterpret_cast<float2 *>(xt + (size_t)b * Npad * 4)[(size_t)tile * 64 + lane] = o2;
}

__device__ __forceinline__ unsigned f32_to_ordered(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int K2, int J>
__device__ __forceinline__ void bitonic32_merge(unsigned &key, int lane)
{
    const unsigned o = xchg32<J>(key);
    const bool keep_max = ((lane & J) == 0) == ((lane & K2) == 0);
    key = ((key > o) == keep_max) ? key : o;
    if constexpr (J > 1) bitonic32_merge<K2, J / 2>(key, lane);
}
template <int K2>
__device__ __forceinline__ void bitonic32_sort(unsigned &key, int lane)
{
    if constexpr (K2 > 2) bitonic32_sort<K2 / 2>(key, lane);
    bitonic32_merge<K2, K2 / 2>(key, lane);
}

// Bound on |pd16 - pd32| for one query row, where pd16 = 2 d16 - yy_l - yy_i is evaluated from the fp16
// centred features (d16 = fp32-accumulated fp16 MFMA dot, yy = fp32 centred norms) and pd32 is the oracle's
// fp32 value on the raw features.  Both approximate the real -||x_i - x_l||^2:
//   fp16 rounding of both operands (rel. 2^-11 each) + fp32 accumulation: |2 d16 - 2 y_i.y_l| <= 2 * 2^-10 * 1.01 * sqrt(yy_i yy_l)
//       -- this one, the largest, is a rank-1 term and rides PER PAIR in the matrix product (KNN_PAIR_SLACK sqrt(yy_i)
//       sqrt(yy_l), augmented slab): against the worst candidate's norm (sqrt(yy_i cmax), the first form) it is what
//       decides the number of survivors on data whose neighbour distances are small against its norms
//   fp16 subnormals (abs. 2^-25 per element):                              <= 1e-6 (sqrt(yy_i) + sqrt(yy_l))
//   fp32 evaluation of yy; the norms and the filter threshold ride in the matrix product (augmented slab, see
//   "fp16 streams"), i.e. in up to ~90 fp32 additions of partial sums below 4 (yy_i + yy_l) in magnitude:
//                                                                          <= 4e-5 (yy_i + yy_l)
//   their three-piece fp16 representation (knn_split3), absolute:          <= 1e-9
//   the oracle's own fp32 chain (gamma_66 = 3.9e-6 on |dot| and on xx):    <= 8e-6 (xx_i + xx_l)
// with yy_l <= cmax, xx_l <= rmax (per batch element).  Worst-case bounds, not estimates.
__device__ __forceinline__ float knn_f16_eps(float yy_i, float cmax, float xx_i, float rmax)
{
    return 1e-6f * (__builtin_sqrtf(yy_i) + __builtin_sqrtf(cmax)) + 4e-5f * (yy_i + cmax) + 8e-6f * (xx_i + rmax) + 1e-9f;
}

// Threshold pre-pass for small C (CP = 4, 8) in exact fp32.  One wave = RT x 32 query rows; xop are the fp32 tiles
// of knn_tile_kernel, xn the raw norms, and the arithmetic is exactly the one of knn_kernel.  (C >= 16 uses
// knn_bound16s_kernel on the fp16 matrix cores.)
// WIDE: 128 classes per row (candidate tile index mod 4 x 32 lanes) instead of 64, for 40 < k <= 64.
template <int CP, int RT, bool WIDE = false>
__global__ __launch_bounds__(64) void knn_bound_kernel(const void *__restrict__ xop, const float *__restrict__ xn, int B,
                                                       int Npad, int k, float *__restrict__ tau0)
{
    const int lane = threadIdx.x, h = lane >> 5, j32 = lane & 31;
    const int b = blockIdx.x % B, rb0 = (blockIdx.x / B) * RT;  // first row tile of this wave
    const float *xnb = xn + (size_t)b * Npad;
    const int ntiles = Npad / 32;  // even

    float m0[RT][16], m1[RT][16];  // class maxima of 2 dot - xx_l over even / odd candidate tiles
    float m2[RT][16], m3[RT][16];  // WIDE: tiles 4g, 4g+1, 4g+2, 4g+3 feed m0 .. m3
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m0[rt][r] = m1[rt][r] = m2[rt][r] = m3[rt][r] = -__builtin_inff();

    {
        constexpr int CH = CP / 2;
        constexpr int VEC = CH >= 4 ? 4 : CH;
        constexpr int G = CH / VEC;
        typedef float vecf __attribute__((ext_vector_type(VEC)));
        const vecf *xb = reinterpret_cast<const vecf *>(reinterpret_cast<const float *>(xop) + (size_t)b * Npad * CP) + lane;
        float a[RT][CH];
        auto load = [&](int t, float(&dst)[CH]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const vecf v = xb[((size_t)t * G + g) * 64];
#pragma unroll
                for (int e = 0; e < VEC; ++e) dst[g * VEC + e] = v[e];
            }
        };
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) load(rb0 + rt, a[rt]);
        // four candidate tiles per step, the next step's operands in flight under this step's MFMAs: with one or two
        // waves per SIMD an unprefetched loop pays a full L2 round trip per tile (Npad / 32 is a multiple of 4)
        constexpr int GT = 4;
        float bcur[GT][CH], bnxt[GT][CH], xcur[GT], xnxt[GT];
#pragma unroll
        for (int u = 0; u < GT; ++u) {
            load(u, bcur[u]);
            xcur[u] = xnb[u * 32 + j32];
        }
        for (int t = 0; t < ntiles; t += GT) {
            const int tn = t + GT < ntiles ? t + GT : t;  // last step: reload the current group (unused)
#pragma unroll
            for (int u = 0; u < GT; ++u) {
                load(tn + u, bnxt[u]);
                xnxt[u] = xnb[(tn + u) * 32 + j32];
            }
#pragma unroll
            for (int u = 0; u < GT; ++u) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                    for (int s = 0; s < CH; ++s)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][s], bcur[u][s], acc, 0, 0, 0);
                    float(&m)[16] = WIDE ? (u == 0 ? m0[rt] : u == 1 ? m1[rt] : u == 2 ? m2[rt] : m3[rt]) : ((u & 1) ? m1[rt] : m0[rt]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = fmaxf(m[r], __builtin_fmaf(2.0f, acc[r], -xcur[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < GT; ++u) {
#pragma unroll
                for (int s = 0; s < CH; ++s) bcur[u][s] = bnxt[u][s];
                xcur[u] = xnxt[u];
            }
        }
    }

    // k-th largest of the 64 class maxima of every row, minus the row's own norm and the error bound
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float other = __uint_as_float(xchg32<32>(__float_as_uint(m1[rt][r])));
            const float other3 = WIDE ? __uint_as_float(xchg32<32>(__float_as_uint(m3[rt][r]))) : 0.0f;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                // row (r, hh): its lanes hold m0 (own half) and the partner half receives that half's m1
                unsigned key = f32_to_ordered((h == hh) ? m0[rt][r] : other);
                bitonic32_sort<64>(key, lane);
                if constexpr (WIDE) {
                    // the other 64 classes, sorted too; the top 64 of the union are max(A[i], B[63 - i]), a bitonic
                    // sequence that one merge pass sorts
                    unsigned keyb = f32_to_ordered((h == hh) ? m2[rt][r] : other3);
                    bitonic32_sort<64>(keyb, lane);
                    const unsigned rev = (unsigned)__shfl((int)keyb, 63 - lane);
                    key = key > rev ? key : rev;
                    bitonic32_merge<64, 32>(key, lane);
                }
                const unsigned kth = __builtin_amdgcn_readlane(key, k - 1);
                const float vk = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7fffffffu) : ~kth);
                const int row = (rb0 + rt) * 32 + mfma32_row(r, hh);
                const float xi = xnb[row];
                const float pd = vk - xi;  // same fp32 operations as knn_kernel
                const float t = pd - fmaxf(__builtin_fabsf(pd) * 4.8e-7f, 1e-37f);
                if (lane == 0) tau0[(size_t)b * Npad + row] = (t != t) ? -__builtin_inff() : t;  // NaN -> no bound
            }
        }
    }
}

// ---- fp16 streams with operand tiles shared through LDS ------------------------------------------------
// Both fp16 all-pairs streams (threshold pre-pass and filter) are bound by L2 -> CU operand traffic when every
// wave fetches every candidate tile itself (1760 waves x 0.9 MB = 1.55 GB per pass at B = 8, N = 7000, C = 64).
// Here a workgroup = KNN_ST_WAVES waves of 32 query rows stages each chunk of KNN_ST_TILES candidate tiles once in LDS
// (already in MFMA operand order, so the ds_read_b128 of a wave is contiguous and conflict-free) and all four
// waves consume it: a quarter of the traffic, one barrier per chunk.  The chunks go through a ring of KNN_ST_RING LDS
// slots: chunk c + 2 is loaded (registers) while chunk c is consumed and written behind it, so a load has a whole
// iteration to land and the barrier at the end of an iteration only orders slot reuse.
//
// The K dimension carries ONE EXTRA 16-wide slab (the "augmented" slab) that puts the additive terms of the distance
// into the matrix product, so that the accumulator IS the quantity the epilogue tests -- no per-element fma / sub
// (packed fp32 VALU beside MFMAs is an anti-lever on gfx950, and two VALU instructions per accumulator register made the
// vector pipe, not the matrix pipe, the bound of both streams):
//   candidate side (stored in xh by knn_prep16_kernel, lanes h = 0):  { p1, p2, p3, K1, K2, K3, sqrt(yy_l), 0 }
//   query side (built by the stream kernels, lanes h = 0):             { K1, K2, K3, g1, g2, g3, +-c sqrt(yy_i), 0 }
// with K = (2^15, 2^4, 2^-7), (p1, p2, p3) = knn_split3(yy_l) the centred norm of the candidate and (g1, g2, g3) =
// knn_split3(T_i) a per-row constant (0 in the pre-pass, the filter threshold in the filter).  With the query features
// scaled by -2 (exact) the accumulator is  acc = T_i + yy_l - 2 d16(i, l) +- c sqrt(yy_i yy_l) = T_i - (t(i, l) -+ slack(i, l)).
constexpr int KNN_ST_TILES = 4;  // candidate tiles per staged chunk (Npad / 32 is a multiple of 4)
constexpr int KNN_ST_WAVES = 4;  // waves (32-row tiles) per workgroup of the fp16 streams (2 measured slower: 105 vs 98 us)
constexpr int KNN_ST_THREADS = KNN_ST_WAVES * 64;
constexpr int KNN_ST_RING = 3;   // LDS slots of the chunk ring
constexpr int KNN_ST_RT = 2;     // row tiles per wave: every staged candidate operand feeds RT MFMAs (half the L2 -> CU and LDS traffic)
constexpr int KNN_ST_ROWS = KNN_ST_WAVES * KNN_ST_RT;  // row tiles per workgroup

template <int KS1>
struct Stage16 {
    static constexpr int VEC_PER_CHUNK = KNN_ST_TILES * KS1 * 64;  // half8 vectors per chunk
    static constexpr int VPT = VEC_PER_CHUNK / KNN_ST_THREADS;     // per thread (= KS1)
    half8 reg[VPT];
    __device__ __forceinline__ void gload(const half8 *__restrict__ xbatch, int chunk, int tid)
    {
#pragma unroll
        for (int i = 0; i < VPT; ++i) reg[i] = xbatch[(size_t)chunk * VEC_PER_CHUNK + i * KNN_ST_THREADS + tid];
    }
    __device__ __forceinline__ void swrite(half8 *__restrict__ buf, int tid) const
    {
#pragma unroll
        for (int i = 0; i < VPT; ++i) buf[i * KNN_ST_THREADS + tid] = reg[i];
    }
};

// query-side operand of a wave: the row tile's features times -2 and the augmented slab with the row constant T
// cs: the row's factor of the per-pair slack, +-KNN_PAIR_SLACK sqrt(yy_i) (+: the pre-pass lowers every candidate's
// value by its slack; -: the filter raises it), 0 for padding rows
// SPLIT (C <= 4, one data slab): the candidate vector of a point is {hi(4), lo(4)} in BOTH lane halves; the query side is
// {hi, hi} in lanes h = 0 and {lo, 0} in lanes h = 1, so that the 16-deep product is hi.hi + hi.lo + lo.hi -- formed here
// from the point's own candidate vector, lane by lane.
template <int KS1, bool SPLIT = false>
__device__ __forceinline__ void knn_stream_a(const half8 *__restrict__ xbatch, int rb, int lane, float T, float cs, half8 (&a)[KS1])
{
    if (SPLIT) {
        static_assert(!SPLIT || KS1 == 2, "split operands: one data slab");
        const half8 v = xbatch[((size_t)rb * KS1) * 64 + lane];
        half8 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[e] = lane < 32 ? v[e] : v[4 + e];
            q[4 + e] = lane < 32 ? v[e] : (_Float16)0.0f;
        }
        a[0] = q * (_Float16)(-2.0f);
    } else {
#pragma unroll
        for (int ks = 0; ks < KS1 - 1; ++ks) a[ks] = xbatch[((size_t)rb * KS1 + ks) * 64 + lane] * (_Float16)(-2.0f);
    }
    half8 aug;
#pragma unroll
    for (int e = 0; e < 8; ++e) aug[e] = (_Float16)0.0f;
    if (lane < 32) {
        const Half3 g = knn_split3(T);
        aug[0] = (_Float16)KNN_AUG_K1;
        aug[1] = (_Float16)KNN_AUG_K2;
        aug[2] = (_Float16)KNN_AUG_K3;
        aug[3] = g.p1;
        aug[4] = g.p2;
        aug[5] = g.p3;
        aug[6] = (_Float16)cs;
    }
    a[KS1 - 1] = aug;
}

// k-th largest of the 64 class maxima of every row of one row tile (m0, m1: the class MINIMA of acc = -t in the
// accumulator layout), minus the row's own norm and the error bound.  The maxima are transposed through LDS (row stride
// 65 words: conflict-free both ways) so that lane (h, j) holds classes 32h .. 32h+31 of row j in registers; each lane
// sorts its 32 keys with a register bitonic network (all 32 rows of the tile at once, no cross-lane traffic), the two
// halves of a row are merged with one half-cleaner across lanes j and j + 32, and a 5-stage merge finishes the half
// that holds rank k (the upper 32 in lanes h = 0, the lower 32 in lanes h = 1).
__device__ __forceinline__ void knn_bound_finish(const float (&m0)[16], const float (&m1)[16], float *tr /* [32 * 65], this wave's */,
                                                 int b, int rb, int part, int parts, int Npad, int k, float cmax, float rmax,
                                                 const float *__restrict__ yb, const float *__restrict__ xx_raw,
                                                 unsigned *__restrict__ exch, float *__restrict__ tau0,
                                                 float *__restrict__ epsr)
{
    const int lane = threadIdx.x & 63, h = lane >> 5, j32 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, h);
        tr[row * 65 + j32] = -m0[r];
        tr[row * 65 + 32 + j32] = -m1[r];
    }
    unsigned v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = f32_to_ordered(tr[j32 * 65 + h * 32 + i]);
    auto cex = [](unsigned &hi, unsigned &lo) {  // compare-exchange: hi >= lo afterwards
        const unsigned a = hi, c = lo;
        hi = a > c ? a : c;
        lo = a > c ? c : a;
    };
#pragma unroll
    for (int size = 2; size <= 32; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int p = i ^ stride;
                if (p > i) {
                    if ((i & size) == 0 || size == 32) cex(v[i], v[p]);  // descending run
                    else cex(v[p], v[i]);
                }
            }
        }
    }
    unsigned w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const unsigned o = xchg32<32>(v[31 - i]);  // partner lane's i-th smallest
        w[i] = h == 0 ? (v[i] > o ? v[i] : o) : (v[i] < o ? v[i] : o);
    }
#pragma unroll
    for (int stride = 16; stride > 0; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if ((i ^ stride) > i) cex(w[i], w[i ^ stride]);
    }
    if (parts == 2) {
        // Candidate halves: each part publishes its 64 sorted class maxima per row (lanes h = 0 the upper 32,
        // h = 1 the lower 32) and knn_bound_merge_kernel finishes the row from both lists.
        // (k <= 32: the k-th largest of the union lies in the union of the parts' upper 32 -- the lower halves stay home)
        unsigned *mine = exch + (((size_t)b * Npad + rb * 32 + j32) * 2 + part) * 64 + h * 32;
        if (k > 32 || h == 0) {
#pragma unroll
            for (int i = 0; i < 32; i += 4)
                *reinterpret_cast<uint4 *>(mine + i) = uint4{w[i], w[i + 1], w[i + 2], w[i + 3]};
        }
        return;
    }
    const int sel = (k - 1) & 31;
    unsigned kth = w[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) kth = sel == i ? w[i] : kth;
    if (h == ((k - 1) >> 5)) {
        const float vk = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7fffffffu) : ~kth);
        const int row = rb * 32 + j32;
        const float xi = yb[row], xr = xx_raw[(size_t)b * Npad + row];
        const float eps = knn_f16_eps(xi, cmax, xr, rmax);
        const float t = (vk - xi) - eps;
        tau0[(size_t)b * Npad + row] = (t != t) ? -__builtin_inff() : t;  // NaN -> no bound
        epsr[(size_t)b * Npad + row] = eps;
    }
}

// Threshold pre-pass on the fp16 matrix cores.  The instruction stream of a wave is laid out by hand, group by group
// (a group = one staged operand vector and the RT MFMAs it feeds, fenced by sched_barrier): the operand of group g + 3
// is read from LDS before the MFMAs of group g are issued, and the class minima of the tile finished last (the other
// accumulator pair) are folded a few registers per group, so that neither an LDS latency nor the vector epilogue ever
// stands between two MFMAs of the wave.  Measured on the way (B = 8, N = 7000, C = 64; the compiler's own order of the
// same work: 70 us): MFMAs alone 36 us, + epilogue 42, + LDS reads 55, + staging 62.
template <int CP, bool SPLIT = false>
__global__ __launch_bounds__(KNN_ST_THREADS, 2) void knn_bound16s_kernel(const half8 *__restrict__ xh, const float *__restrict__ xxc,
                                                           const float *__restrict__ xx_raw,
                                                           const unsigned *__restrict__ bmax, int B, int Npad, int k,
                                                           int parts, unsigned *__restrict__ exch,
                                                           float *__restrict__ tau0, float *__restrict__ epsr)
{
    constexpr int KS1 = CP / 16 + 1, RT = KNN_ST_RT, NG = KNN_ST_TILES * KS1, PD = 3, RING = KNN_ST_RING;
    constexpr int VPC = Stage16<KS1>::VEC_PER_CHUNK;
    constexpr int SBUF_BYTES = RING * VPC * (int)sizeof(half8);
    constexpr int TR_BYTES = KNN_ST_WAVES * 32 * 65 * (int)sizeof(float);  // epilogue transpose
    static_assert(RT == 2 && KNN_ST_TILES == 4 && RING == 3, "the hand-laid stream below assumes these");
    __shared__ __attribute__((aligned(16))) char smem[SBUF_BYTES > TR_BYTES ? SBUF_BYTES : TR_BYTES];
    half8 *sbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup = (batch element, candidate part, group of KNN_ST_ROWS row tiles); with B a multiple of 8 all
    // parts of a row group run on the same XCD (blockIdx % 8)
    const int b = blockIdx.x % B, part = (blockIdx.x / B) % parts;
    const int rb0 = (blockIdx.x / B / parts) * KNN_ST_ROWS + wave * RT;  // this wave's first row tile
    const int ntiles = Npad / 32, nchunks = ntiles / KNN_ST_TILES;
    const int c0 = part * nchunks / parts, c1 = (part + 1) * nchunks / parts;  // this part's candidate chunks
    const half8 *xbatch = xh + (size_t)b * ntiles * KS1 * 64;
    const float *yb = xxc + (size_t)b * Npad;

    // (a row tile beyond the batch element -- ntiles is a multiple of 4, a workgroup covers KNN_ST_ROWS -- repeats the
    // last real one and is not written: its wave still takes part in the staging)
#ifdef KNN_STAMP
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    half8 a[RT][KS1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rbc = rb0 + rt < ntiles ? rb0 + rt : ntiles - 1;
        const float yy = yb[rbc * 32 + (lane & 31)];  // (+inf for padding rows)
        knn_stream_a<KS1, SPLIT>(xbatch, rbc, lane, 0.0f,
                                 yy < KNN_PAD_NORM ? (SPLIT ? KNN_PAIR_SLACK_SPLIT : KNN_PAIR_SLACK) * __builtin_sqrtf(yy) : 0.0f, a[rt]);
    }
    // class minima of acc = yy_l - 2 d16 = -t: even candidate tiles feed m0, odd ones m1 (64 classes per row with the
    // 32 candidate lanes)
    float m0[RT][16], m1[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) m0[rt][i] = m1[rt][i] = __builtin_inff();

    // chunk ring: iteration c consumes slot c % 3, reads ahead into slot (c + 1) % 3 (written during iteration c - 1,
    // behind that iteration's barrier), writes chunk c + 2 (registers, loaded during iteration c - 1) to slot (c + 2) % 3
    // -- last read in iteration c - 1 -- and loads chunk c + 3
    const int last = c1 - 1;
    Stage16<KS1> st;
    st.gload(xbatch, c0, tid);
    st.swrite(sbuf, tid);
    st.gload(xbatch, c0 + 1 < last ? c0 + 1 : last, tid);
    st.swrite(sbuf + VPC, tid);
    st.gload(xbatch, c0 + 2 < last ? c0 + 2 : last, tid);
    __syncthreads();
    // accumulators: even tiles in accE, odd tiles in accO; the tile pending across the barrier is the odd tile 3 of the
    // chunk before (+inf, the neutral element, in the first iteration)
    f32x16 accE[RT], accO[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accE[rt][r] = accO[rt][r] = __builtin_inff();
    half8 bq[PD + 1];
#pragma unroll
    for (int g = 0; g < PD; ++g) bq[g] = sbuf[g * 64 + lane];
    int scur = 0, snxt = 1, swr = 2;
#ifdef KNN_STAMP
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = c0; c < c1; ++c) {
        const half8 *cur = sbuf + scur * VPC, *nxt = sbuf + snxt * VPC;
        half8 *wr = sbuf + swr * VPC;
        const int cload = c + 3 < last ? c + 3 : last;  // clamped: no branch around the loads
        static_for<NG>([&](auto G) {
            constexpr int g = G.value, t = g / KS1, ks = g % KS1, gp = g + PD;
            // operand of group g + PD (the first PD groups of the next chunk at the end)
            bq[gp % (PD + 1)] = gp < NG ? cur[gp * 64 + lane] : nxt[(gp - NG) * 64 + lane];
            // staging, one vector per group: registers -> ring (groups 0 ..), then global -> registers
            if constexpr (g < KS1) wr[g * KNN_ST_THREADS + tid] = st.reg[g];
            else if constexpr (g < 2 * KS1)
                st.reg[g - KS1] = xbatch[(size_t)cload * VPC + (g - KS1) * KNN_ST_THREADS + tid];
            // the MFMAs of the group
            if constexpr ((t & 1) == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) accE[rt][r] = 0.0f;
                    }
                    accE[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accE[rt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) accO[rt][r] = 0.0f;
                    }
                    accO[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accO[rt], 0, 0, 0);
                }
            }
            // a share of the pending tile's epilogue (the other accumulator pair), none in the tile's first group (its
            // last MFMAs are still in the pipe).  Inline asm: as __builtin_fminf the compiler either joins the folds of
            // two tiles into one v_min3 -- and keeps eight accumulator tiles alive for it -- or, pinned by an empty
            // asm, canonicalises both operands first (three instructions per fold)
            if constexpr (ks > 0) {
                constexpr int e0 = (ks - 1) * (RT * 16) / (KS1 - 1), e1 = ks * (RT * 16) / (KS1 - 1);
#pragma unroll
                for (int e = e0; e < e1; ++e) {
                    const int rt = e / 16, r = e % 16;
                    if constexpr ((t & 1) == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(m1[rt][r]) : "v"(accO[rt][r]));
                    else asm volatile("v_min_f32 %0, %0, %1" : "+v"(m0[rt][r]) : "v"(accE[rt][r]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        scur = snxt;
        snxt = swr;
        swr = swr + 1 == RING ? 0 : swr + 1;
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m1[rt][r] = __builtin_fminf(m1[rt][r], accO[rt][r]);

#ifdef KNN_STAMP
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    float *tr = reinterpret_cast<float *>(smem) + wave * (32 * 65);  // all waves are past the last barrier
    const float cmax = batch_max(bmax, b, 0, Npad / 32), rmax = batch_max(bmax, b, 1, Npad / 32);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if (rb0 + rt >= ntiles) break;  // uniform over the wave
        knn_bound_finish(m0[rt], m1[rt], tr, b, rb0 + rt, part, parts, Npad, k, cmax, rmax, yb, xx_raw, exch, tau0, epsr);
    }
#ifdef KNN_STAMP
    const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x % 97) == 5 && lane == 0)
        printf("bound blk %d wave %d: start %llu prologue %llu loop %llu (%d chunks, %llu per chunk) finish %llu\n", (int)blockIdx.x, wave,
               ts0, ts1 - ts0, ts2 - ts1, c1 - c0, (ts2 - ts1) / (c1 - c0), ts3 - ts2);
#endif
}

// Threshold of a row from the two parts' sorted class maxima A, B (64 each, descending): the top 64 of the
// union is the bitonic sequence max(A[i], B[63 - i]); one cross-lane half-cleaner and the in-lane merge sort
// it.  128 classes per row instead of 64 also tighten the threshold.  One wave = 32 rows, lane (h, j) = half h
// of row j as in the epilogue of knn_bound16s_kernel.
// k <= 32: only the parts' upper halves A, B (32 each, descending) were published; the top 32 of their union is the
// bitonic sequence max(A[i], B[31 - i]), sorted by the in-lane merge.  One lane = one row.
// (one row: the body of knn_bound_merge32_kernel, and -- since round 4 -- of the prologue of knn_filter16_kernel, whose lanes
// form the thresholds of their own rows instead of reading them from a 5 us launch in front of every filter)
__device__ __forceinline__ void knn_tau_merge32(const unsigned *__restrict__ exch, const float *__restrict__ xxc,
                                                const float *__restrict__ xx_raw, float cmax, float rmax, size_t row, int k,
                                                float &tau, float &eps)
{
    const uint4 *pa = reinterpret_cast<const uint4 *>(exch + (row * 2 + 0) * 64);
    const uint4 *pb = reinterpret_cast<const uint4 *>(exch + (row * 2 + 1) * 64);
    unsigned w[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 va = pa[i], vb = pb[7 - i];  // B reversed: element 31 - (4 i + e) is vb[3 - e]
        w[4 * i + 0] = va.x > vb.w ? va.x : vb.w;
        w[4 * i + 1] = va.y > vb.z ? va.y : vb.z;
        w[4 * i + 2] = va.z > vb.y ? va.z : vb.y;
        w[4 * i + 3] = va.w > vb.x ? va.w : vb.x;
    }
#pragma unroll
    for (int stride = 16; stride > 0; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if ((i ^ stride) > i) {
                const unsigned x = w[i], y = w[i ^ stride];
                w[i] = x > y ? x : y;
                w[i ^ stride] = x > y ? y : x;
            }
        }
    }
    unsigned kth = w[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) kth = k - 1 == i ? w[i] : kth;
    const float vk = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7fffffffu) : ~kth);
    const float xi = xxc[row], xr = xx_raw[row];
    eps = knn_f16_eps(xi, cmax, xr, rmax);
    const float t = (vk - xi) - eps;
    tau = (t != t) ? -__builtin_inff() : t;  // NaN -> no bound
}
__global__ __launch_bounds__(64) void knn_bound_merge32_kernel(const unsigned *__restrict__ exch,
                                                               const float *__restrict__ xxc,
                                                               const float *__restrict__ xx_raw,
                                                               const unsigned *__restrict__ bmax, int B, int Npad, int k,
                                                               float *__restrict__ tau0, float *__restrict__ epsr)
{
    const int lane = threadIdx.x;
    const int b = blockIdx.x % B, rb = blockIdx.x / B;  // rb: group of 64 rows
    const size_t row = (size_t)b * Npad + rb * 64 + lane;
    const float cmax = batch_max(bmax, b, 0, Npad / 32), rmax = batch_max(bmax, b, 1, Npad / 32);
    float tau, eps;
    knn_tau_merge32(exch, xxc, xx_raw, cmax, rmax, row, k, tau, eps);
    tau0[row] = tau;
    epsr[row] = eps;
}

__global__ __launch_bounds__(64) void knn_bound_merge_kernel(const unsigned *__restrict__ exch,
                                                             const float *__restrict__ xxc,
                                                             const float *__restrict__ xx_raw,
                                                             const unsigned *__restrict__ bmax, int B, int Npad, int k,
                                                             float *__restrict__ tau0, float *__restrict__ epsr)
{
    const int lane = threadIdx.x, h = lane >> 5, j32 = lane & 31;
    const int b = blockIdx.x % B, rb = blockIdx.x / B;
    const size_t row = (size_t)b * Npad + rb * 32 + j32;
    const uint4 *pa = reinterpret_cast<const uint4 *>(exch + (row * 2 + 0) * 64 + h * 32);
    const uint4 *pb = reinterpret_cast<const uint4 *>(exch + (row * 2 + 1) * 64 + (1 - h) * 32);
    unsigned w[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 va = pa[i], vb = pb[7 - i];  // B reversed: element 31 - (4 i + e) is vb[3 - e]
        w[4 * i + 0] = va.x > vb.w ? va.x : vb.w;
        w[4 * i + 1] = va.y > vb.z ? va.y : vb.z;
        w[4 * i + 2] = va.z > vb.y ? va.z : vb.y;
        w[4 * i + 3] = va.w > vb.x ? va.w : vb.x;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const unsigned o = xchg32<32>(w[i]);
        w[i] = h == 0 ? (w[i] > o ? w[i] : o) : (w[i] < o ? w[i] : o);
    }
#pragma unroll
    for (int stride = 16; stride > 0; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if ((i ^ stride) > i) {
                const unsigned x = w[i], y = w[i ^ stride];
                w[i] = x > y ? x : y;
                w[i ^ stride] = x > y ? y : x;
            }
        }
    }
    const int sel = (k - 1) & 31;
    unsigned kth = w[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) kth = sel == i ? w[i] : kth;
    const float cmax = batch_max(bmax, b, 0, Npad / 32), rmax = batch_max(bmax, b, 1, Npad / 32);
    if (h == ((k - 1) >> 5)) {
        const float vk = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7fffffffu) : ~kth);
        const float xi = xxc[row], xr = xx_raw[row];
        const float eps = knn_f16_eps(xi, cmax, xr, rmax);
        const float t = (vk - xi) - eps;
        tau0[row] = (t != t) ? -__builtin_inff() : t;  // NaN -> no bound
        epsr[row] = eps;
    }
}

// LDS of the exact all-pairs pass over one 32-row block
template <int CAP>
struct KnnExactLds {
    u64 keys[32][CAP];
    int cnts[32];
    float taus[32];
};

// The exact fp32 all-pairs kNN of ONE 32-row block (b, rb) by one wave: the whole of knn_kernel below, and -- called from
// knn_refine_kernel for the row blocks the fp16 filter path has flagged (features outside the fp16 range, a survivor list
// that overflowed, a row that ended with fewer than k candidates) -- the fallback of that path, without a launch of its own
// (behind every kNN build it was an idle 4.7 us launch on unflagged inputs).
template <int CP, int CAP>
__device__ __forceinline__ void knn_exact_block(KnnExactLds<CAP> &L, const float *__restrict__ xt, const float *__restrict__ xx,
                                                const float *__restrict__ tau0, int b, int rb, int N, int Npad, int k,
                                                int32_t *__restrict__ idx)
{
    constexpr int CH = CP / 2;
    constexpr int VEC = CH >= 4 ? 4 : CH;
    constexpr int G = CH / VEC;
    typedef float vecf __attribute__((ext_vector_type(VEC)));

    auto &keys = L.keys;
    auto &cnts = L.cnts;
    auto &taus = L.taus;

    const int lane = threadIdx.x, h = lane >> 5, j32 = lane & 31;
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
    unsigned cpk[4] = {0u, 0u, 0u, 0u};  // per-row candidate counts (<= CAP <= 128), one byte per register r
    auto cnt_get = [&](int r) { return (int)((cpk[r >> 2] >> ((r & 3) * 8)) & 0xffu); };
    auto cnt_set = [&](int r, int v) {
        cpk[r >> 2] = (cpk[r >> 2] & ~(0xffu << ((r & 3) * 8))) | ((unsigned)v << ((r & 3) * 8));
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xxq[r] = xxb[q0 + mfma32_row(r, h)];
        taur[r] = tau0 ? tau0[(size_t)b * Npad + q0 + mfma32_row(r, h)] : -__builtin_inff();
    }

    // Software pipeline: the MFMA chain of the NEXT candidate tile is issued interleaved, in program
    // order, with the selection epilogue of the CURRENT tile (whose accumulators completed one
    // step earlier).  A dependent fp32 MFMA occupies the matrix pipe for 64 cycles, so the few
    // vector/scalar instructions of one register's epilogue between two MFMAs are free, and the
    // MFMA -> VALU read hazard never stalls.
    auto fused = [&](auto mf_tag, const float(&bop)[CH], f32x16 &acc_next, const f32x16 &acc, float xxc, int cand) {
        constexpr bool MF = decltype(mf_tag)::value;
        unsigned retry = 0u;  // bit (r + 16 * half): row would overflow -> compact, then offer again
        // pass 1 for accumulator register r: append where the row's buffer has room
        auto epi = [&](auto rtag) {
            constexpr int r = decltype(rtag)::value;
            const float pd = __builtin_fmaf(2.0f, acc[r], -xxc) - xxq[r];
            const bool pass = pd > taur[r];
            const u64 m = __ballot(pass);
            if (__builtin_expect(m != 0, 0)) {  // out of line: the common path must not contain taken branches
                const int row = mfma32_row(r, h);
                const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
                const int tot = __popc(mh), c = cnt_get(r);
                const bool ovf = c + tot > CAP;
                const u64 mo = __ballot(ovf);
                if (!ovf) {
                    if (pass) {
                        float pdk = pd;
                        asm volatile("" : "+v"(pdk));  // keep the key arithmetic on this (rare) path
                        keys[row][c + __popc(mh & ((1u << j32) - 1u))] = make_key(pdk, cand);
                    }
                    cpk[r >> 2] += (unsigned)tot << ((r & 3) * 8);
                }
                if (mo) {
                    if (ovf && j32 == 0) cnts[row] = c;
                    if ((unsigned)mo) retry |= 1u << r;
                    if (mo >> 32) retry |= 1u << (r + 16);
                }
            }
        };
        if constexpr (MF) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_next[r] = 0.0f;
        }
        // 16 epilogue registers spread over the CH MFMAs of the chain
        constexpr int EPI_PER = CH >= 16 ? 1 : 16 / CH;  // epilogue registers after one MFMA
        constexpr int MF_PER = CH >= 16 ? CH / 16 : 1;   // MFMAs per epilogue group
        auto group = [&](auto gtag) {
            constexpr int g = decltype(gtag)::value;  // 0 .. 16 / EPI_PER - 1
            if constexpr (MF) {
#pragma unroll
                for (int i = 0; i < MF_PER; ++i)
                    acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g * MF_PER + i], bop[g * MF_PER + i], acc_next, 0, 0, 0);
            }
            if constexpr (EPI_PER == 1) {
                epi(std::integral_constant<int, g>{});
            } else {
                static_for<EPI_PER>([&](auto e) { epi(std::integral_constant<int, g * EPI_PER + decltype(e)::value>{}); });
            }
        };
        static_for<16 / EPI_PER>(group);

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
                    int c = cnt_get(r);
                    if (mine) {
                        taur[r] = taus[row];
                        c = cnts[row];
                    }
                    const float pd = __builtin_fmaf(2.0f, acc[r], -xxc) - xxq[r];
                    const bool pass = mine && (pd > taur[r]);
                    const u64 m = __ballot(pass);
                    const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
                    if (pass) keys[row][c + __popc(mh & ((1u << j32) - 1u))] = make_key(pd, cand);
                    cnt_set(r, c + __popc(mh));
                }
            }
        }
    };
    auto chain = [&](const float(&bop)[CH], f32x16 &acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < CH; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], bop[s2], acc, 0, 0, 0);
    };

    const int ntiles = Npad / 32;  // even (Npad is a multiple of 128)
    const int need = k < N ? k : N;
    for (int attempt = 0; attempt < 2; ++attempt) {
        float bA[CH], bB[CH];
        f32x16 accA, accB;
        load_operand(0, bA);
        load_operand(1, bB);
        float xA = xxb[j32], xB = xxb[32 + j32];
        chain(bA, accA);  // tile 0
        for (int t = 0; t < ntiles; t += 2) {
            // accA = tile t (complete); bB = operands of tile t + 1
            // the prefetches are unconditional (tile index clamped) so that no branch separates them
            // from the MFMA chain: a join would make the compiler wait for vmcnt(0) there
            const bool more = t + 2 < ntiles;
            const float xcurA = xA, xcurB = xB;
            const int tA = more ? t + 2 : t, tB = more ? t + 3 : t + 1;
            load_operand(tA, bA);
            xA = xxb[tA * 32 + j32];
            fused(std::true_type{}, bB, accB, accA, xcurA, t * 32 + j32);  // MFMA tile t+1 | select tile t
            load_operand(tB, bB);
            xB = xxb[tB * 32 + j32];
            if (more)
                fused(std::true_type{}, bA, accA, accB, xcurB, (t + 1) * 32 + j32);  // MFMA tile t+2 | select tile t+1
            else
                fused(std::false_type{}, bA, accA, accB, xcurB, (t + 1) * 32 + j32);  // select the last tile
        }
        // a row that collected fewer than k candidates started from a threshold that was NOT a lower
        // bound of its k-th best: start the whole stream again from tau = -inf (never taken when the
        // pre-pass bound holds; with tau0 == nullptr the first attempt already starts from -inf)
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) bad = bad || (q0 + mfma32_row(r, h) < N && cnt_get(r) < need);
        if (!__ballot(bad)) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) taur[r] = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 4; ++i) cpk[i] = 0u;
    }

    // ---- final sort and write-out ------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (j32 == 0) cnts[mfma32_row(r, h)] = cnt_get(r);
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

template <int CP, int CAP>
__global__ __launch_bounds__(64, 2) void knn_kernel(const float *__restrict__ xt, const float *__restrict__ xx,
                                                    const float *__restrict__ tau0, const int *__restrict__ rbflag,
                                                    int B, int N, int Npad, int k, int32_t *__restrict__ idx)
{
    __shared__ KnnExactLds<CAP> L;
    // blocks are dealt round-robin over the XCDs: with B a multiple of 8 all waves of batch
    // element b run on one XCD and share its L2 copy of xt[b] (speed only)
    const int b = blockIdx.x % B, rb = blockIdx.x / B;
    if (rbflag && !rbflag[b * (Npad / 32) + rb]) return;  // row block already done by the fp16 filter + refine path
    knn_exact_block<CP, CAP>(L, xt, xx, tau0, b, rb, N, Npad, k, idx);
}

// ------------------------------------------------------------------------------------
// Exact kNN through an fp16 FILTER and an exact fp32 REFINE (C >= 16): knn_filter16_kernel + knn_refine_kernel.
//
// fp32 MFMA shares the fp32 FMA lanes with the vector ALU on gfx950 (same 64 FLOP/clk/SIMD; measured here:
// MFMA-busy and VALU-active cycles add up instead of overlapping), so the exact all-pairs kernel above pays
// 64 cycles per 32x32x2 MFMA with nothing hidden underneath.  Here the all-pairs stream runs on
// v_mfma_f32_32x32x16_f16 (1/16 of the cost) over the centred fp16 features and only FILTERS: pair (i, l)
// survives if   pd16(i, l) > tau0_i - eps_i,   where tau0_i (knn_bound_kernel) is a lower bound of row i's
// k-th best exact pd and eps_i = knn_f16_eps bounds |pd16 - pd32|.  Every true neighbour survives
// (pd32 >= k-th best >= tau0_i  =>  pd16 >= tau0_i - eps_i).  The filter kernel is a light streaming kernel
// (no top-k state, ~70 VGPRs, 1 KB LDS: many waves per SIMD keep the L2 stream busy); it appends the
// survivors (about 42 per row at N = 7000, k = 30) to a per-wave list in HBM, 64 entries per coalesced store.
// The refine kernel then computes, one survivor per lane, the exact pd in the oracle's arithmetic (sequential
// fmaf chain over the raw fp32 features, c = 0..C-1) and selects with the same key / buffer / compaction
// machinery as knn_kernel.  A row block whose list overflowed, whose features do not fit fp16, or that ends
// with fewer than k candidates is flagged and recomputed by the exact fp32 kernel (knn_kernel), so the
// result never depends on the filter.
constexpr int KNN_QCAP = 256;     // LDS staging ring of the filter kernel (entries: row << 27 | candidate)
constexpr int KNN_MAXPARTS = 4;   // candidate parts of the filter (list segments per row block)
// Balanced form of the filter's work split (wpb > 0 workgroups per batch element): the (row group, candidate chunk) units of a batch
// element, row group major, are cut into wpb equal runs -- a workgroup takes one run, i.e. a candidate range of one row group
// and, where the run crosses a group boundary, a second range at the start of the next group.  The uniform split (every row
// group cut into `parts` equal parts: 896 workgroups at B = 8, N = 7000) fills the chip's 512 workgroup slots 1.75 times,
// i.e. runs two rounds; 512 runs of 24 - 25 chunks fill them once.  The list segment of a range is its ordinal among the runs
// that touch its row group: knn_bal_owner(first unit of the group) is the first of them.
__host__ __device__ inline int knn_bal_owner(int unit, int wpb, int total)  // the run that holds `unit`: boundary(j) = floor(j total / wpb)
{
    return (int)(((long long)(unit + 1) * wpb + total - 1) / total) - 1;
}
// survivor list per (32-row block, candidate part) in HBM: 4096 entries = 128 per row, so that a row block whose
// neighbours all lie in ONE candidate part (spatially ordered points: the neighbours of consecutive rows are close in
// index) fits, with room for the filter's slack on low-dimensional data (points on a sheet: the fp16 error bound is of
// the order of the neighbour distances and about twice the usual survivors pass).  Only written entries cost traffic.
// (One list of 4096 entries cut into `parts` segments, the first form, overflowed -- and fell back to the exact fp32
// kernel -- as soon as ~1000 survivors of a block met in one quarter of the candidates.)
constexpr int KNN_LISTCAP = 4096;

template <int CP, bool SPLIT = false>
__global__ __launch_bounds__(KNN_ST_THREADS, 2) void knn_filter16_kernel(const half8 *__restrict__ xh, const float *__restrict__ xxc,
                                                           const float *__restrict__ tau0,
                                                           const float *__restrict__ epsr,
                                                           const unsigned *__restrict__ bmax, int B, int Npad,
                                                           int parts, unsigned *__restrict__ lists,
                                                           int *__restrict__ lcount, int *__restrict__ rbflag,
                                                           const unsigned *__restrict__ exch, const float *__restrict__ xx_raw,
                                                           int k, float *__restrict__ tau_out, int wpb)
{
    // exch != NULL (the pre-pass ran over two candidate halves and k <= 32): the thresholds are formed HERE from the halves'
    // sorted class maxima (knn_tau_merge32), one lane per row, instead of by knn_bound_merge32_kernel in front of this launch;
    // the workgroups of candidate part 0 leave tau0 behind for the exact pass of flagged row blocks (knn_exact_block)
    constexpr int KS1 = CP / 16 + 1, RT = KNN_ST_RT, NG = KNN_ST_TILES * KS1, PD = 3, RING = KNN_ST_RING;
    constexpr int VPC = Stage16<KS1>::VEC_PER_CHUNK;
    static_assert(RT == 2 && KNN_ST_TILES == 4 && RING == 3, "the hand-laid stream below assumes these");
    __shared__ half8 sbuf[RING * VPC];
    __shared__ unsigned queues[KNN_ST_WAVES][RT][KNN_QCAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j32 = lane & 31;
    // workgroup = (batch element, candidate part, group of row blocks) as in knn_bound16s_kernel; every
    // (row block, part) owns 1 / parts of the row block's list.  rbflag is zero-filled by the launcher.
    const int b = blockIdx.x % B;
    const int ntiles = Npad / 32, nchunks = ntiles / KNN_ST_TILES;
    const int listcap = KNN_LISTCAP;
    // the workgroup's one or two items (row group, list segment, candidate chunks [c0, c1)): see knn_bal_owner
    int nitems = 1, i_rg0, i_part0, i_c00, i_c10, i_rg1 = 0, i_part1 = 0, i_c01 = 0, i_c11 = 0;
    if (wpb == 0) {
        i_part0 = (blockIdx.x / B) % parts;
        i_rg0 = blockIdx.x / B / parts;
        i_c00 = i_part0 * nchunks / parts;
        i_c10 = (i_part0 + 1) * nchunks / parts;
    } else {
        const int j = blockIdx.x / B, total = ((ntiles + KNN_ST_ROWS - 1) / KNN_ST_ROWS) * nchunks;
        const int u0 = (int)((long long)j * total / wpb), u1 = (int)((long long)(j + 1) * total / wpb);
        i_rg0 = u0 / nchunks;
        i_c00 = u0 - i_rg0 * nchunks;
        i_c10 = i_c00 + (u1 - u0) < nchunks ? i_c00 + (u1 - u0) : nchunks;
        i_part0 = j - knn_bal_owner(i_rg0 * nchunks, wpb, total);
        if (u0 + (i_c10 - i_c00) < u1) {  // the run goes on in the next row group (the launcher keeps runs no longer than a group)
            nitems = 2;
            i_rg1 = i_rg0 + 1;
            i_c01 = 0;
            i_c11 = u1 - (u0 + (i_c10 - i_c00));
            i_part1 = j - knn_bal_owner(i_rg1 * nchunks, wpb, total);
        }
        if (u0 >= u1) nitems = 0;
    }
    for (int item = 0; item < nitems; ++item) {
    const int part = item ? i_part1 : i_part0, c0 = item ? i_c01 : i_c00, c1 = item ? i_c11 : i_c10;
    const int rb0 = (item ? i_rg1 : i_rg0) * KNN_ST_ROWS + wave * RT;  // this wave's first row block
#ifdef KNN_STAMP
    const unsigned long long tsA = __builtin_amdgcn_s_memtime();
#endif
    // centred features must fit fp16 with room for the factor -2 and the threshold must fit the augmented slab
    // (|T| < 2^31); NaN/inf norms fail the test
    if (!(batch_max(bmax, b, 0, Npad / 32) < KNN_F16_CMAX)) {  // uniform over the workgroup
        if (lane < RT && rb0 + lane < ntiles) {
            rbflag[b * ntiles + rb0 + lane] = 1;
            lcount[KNN_MAXPARTS * (b * ntiles + rb0 + lane) + part] = 0;
        }
        continue;
    }
    const half8 *xbatch = xh + (size_t)b * ntiles * KS1 * 64;
    const float *yb = xxc + (size_t)b * Npad;

    // filter threshold on t = 2 d16 - yy_l:  t > (tau0 - eps) + yy_i, lowered by the rounding of this very sum; it
    // rides in the query side of the augmented slab (row 32 rb + j32 is the A row of lanes j32), so that the accumulator
    // is T - t and a pair survives iff its sign bit is set.  A row block beyond the batch element (a workgroup covers
    // KNN_ST_ROWS of them, ntiles is a multiple of 4) gets T = +inf: nothing passes, nothing is written.
    half8 a[RT][KS1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const bool real = rb0 + rt < ntiles;
        const int rbc = real ? rb0 + rt : ntiles - 1;
        const int row = rbc * 32 + j32;
        float tau, eps_row;
        const float yy = yb[row];
        if (exch) {
            const float cmax = batch_max(bmax, b, 0, Npad / 32), rmax = batch_max(bmax, b, 1, Npad / 32);
            knn_tau_merge32(exch, xxc, xx_raw, cmax, rmax, (size_t)b * Npad + row, k, tau, eps_row);
            if (part == 0 && h == 0 && real) tau_out[(size_t)b * Npad + row] = tau;
        } else {
            tau = tau0[(size_t)b * Npad + row];
            eps_row = epsr[(size_t)b * Npad + row];
        }
        const float v = (tau - eps_row) + yy;
        const float thr = v - (__builtin_fabsf(v) + __builtin_fabsf(tau) + yy) * 2.4e-7f;
        // padding rows (yy = +inf) give NaN: nothing may pass there (T = +inf: the accumulator is +inf)
        knn_stream_a<KS1, SPLIT>(xbatch, rbc, lane, (thr != thr || !real) ? __builtin_inff() : thr,
                                 yy < KNN_PAD_NORM ? -(SPLIT ? KNN_PAIR_SLACK_SPLIT : KNN_PAIR_SLACK) * __builtin_sqrtf(yy) : 0.0f, a[rt]);
    }

    // staged survivors: a ring in LDS per row block (no data movement on flush), entries already in the HBM list
    int qhead[RT], qn[RT], lpos[RT];
    bool overflow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        qhead[rt] = qn[rt] = lpos[rt] = 0;
        overflow[rt] = false;
    }
    auto flush = [&](int rt, int n) {  // move the oldest n (<= 64) staged entries of row block rt to its list
        unsigned *list = lists + ((size_t)(b * ntiles + rb0 + rt) * KNN_MAXPARTS + part) * KNN_LISTCAP;
        if (lpos[rt] + n > listcap) {
            overflow[rt] = true;
        } else if (lane < n) {
            list[lpos[rt] + lane] = queues[wave][rt][(qhead[rt] + lane) & (KNN_QCAP - 1)];
        }
        lpos[rt] += n;
        qhead[rt] = (qhead[rt] + n) & (KNN_QCAP - 1);
        qn[rt] -= n;
    };
    // Extraction in passes: every lane hands in up to three survivors of each row block per pass; an inclusive scan over
    // the wave (DPP, both blocks' counts packed in one word) gives every lane its slots in the two staging rings, so a
    // pass costs one scan and three short predicated steps instead of a ballot / mbcnt / popcount round per survivor
    // (about 24 survivors per block and chunk over 64 lanes: one pass unless a lane holds more than three).  Bit 16 s + r
    // of a block's mask = slot s, accumulator register r; slot s holds candidates cand + 32 off(s).
    auto wave_scan = [&](unsigned v) {  // inclusive prefix sum over the 64 lanes
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
        return v;
    };
    auto rounds = [&](u64 bits0, u64 bits1, int cand, unsigned offs /* four signed bytes: off(s) */) {
        bool more = __ballot((bits0 | bits1) != 0ull) != 0ull;
        while (more) {
            const int p0 = __popcll(bits0), p1 = __popcll(bits1);
            const int n0 = p0 < 3 ? p0 : 3, n1 = p1 < 3 ? p1 : 3;
            const unsigned packed = (unsigned)n0 | ((unsigned)n1 << 16);
            const unsigned incl = wave_scan(packed);
            const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            const unsigned excl = incl - packed;
            more = __ballot(p0 > 3 || p1 > 3) != 0ull;
            // a pass adds at most 192 entries to a ring of KNN_QCAP = 256
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                while (qn[rt] > KNN_QCAP - 192) flush(rt, 64);
            const unsigned base0 = (unsigned)(qhead[0] + qn[0]) + (excl & 0xffffu), base1 = (unsigned)(qhead[1] + qn[1]) + (excl >> 16);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < n0) {
                    const int pos = __builtin_ctzll(bits0), r = pos & 15;
                    bits0 &= bits0 - 1ull;
                    queues[wave][0][(base0 + i) & (KNN_QCAP - 1)] =
                        ((unsigned)((r & 3) + 8 * (r >> 2) + 4 * h) << 27) | (unsigned)(cand + 32 * (int)(signed char)(offs >> (8 * (pos >> 4))));
                }
                if (i < n1) {
                    const int pos = __builtin_ctzll(bits1), r = pos & 15;
                    bits1 &= bits1 - 1ull;
                    queues[wave][1][(base1 + i) & (KNN_QCAP - 1)] =
                        ((unsigned)((r & 3) + 8 * (r >> 2) + 4 * h) << 27) | (unsigned)(cand + 32 * (int)(signed char)(offs >> (8 * (pos >> 4))));
                }
            }
            qn[0] += (int)(tot & 0xffffu);
            qn[1] += (int)(tot >> 16);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                while (qn[rt] >= 64) flush(rt, 64);
        }
    };

    // chunk ring and hand-laid groups as in knn_bound16s_kernel; the epilogue share of a group packs sign bits of the
    // pending tile (the other accumulator pair) with one v_alignbit per register.  Two tiles share a 32-bit word (the
    // later one in the low half): w01 = {tile 3 of the chunk before, tile 0}, w23 = {tile 1, tile 2}.
    const int last = c1 - 1;
    Stage16<KS1> st;
    st.gload(xbatch, c0, tid);
    st.swrite(sbuf, tid);
    st.gload(xbatch, c0 + 1 < last ? c0 + 1 : last, tid);
    st.swrite(sbuf + VPC, tid);
    st.gload(xbatch, c0 + 2 < last ? c0 + 2 : last, tid);
    __syncthreads();
    f32x16 accE[RT], accO[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accE[rt][r] = accO[rt][r] = 0.0f;  // +0: no survivor in the pending tile of the first chunk
    half8 bq[PD + 1];
#pragma unroll
    for (int g = 0; g < PD; ++g) bq[g] = sbuf[g * 64 + lane];
    int scur = 0, snxt = 1, swr = 2;
#ifdef KNN_STAMP
    unsigned long long tg = 0, tr = 0, tb = 0;
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = c0; c < c1; ++c) {
#ifdef KNN_STAMP
        const unsigned long long u0 = __builtin_amdgcn_s_memtime();
#endif
        const half8 *cur = sbuf + scur * VPC, *nxt = sbuf + snxt * VPC;
        half8 *wr = sbuf + swr * VPC;
        const int cload = c + 3 < last ? c + 3 : last;  // clamped: no branch around the loads
        unsigned w01[RT] = {0u, 0u}, w23[RT] = {0u, 0u};
        static_for<NG>([&](auto G) {
            constexpr int g = G.value, t = g / KS1, ks = g % KS1, gp = g + PD;
            bq[gp % (PD + 1)] = gp < NG ? cur[gp * 64 + lane] : nxt[(gp - NG) * 64 + lane];
            if constexpr (g < KS1) wr[g * KNN_ST_THREADS + tid] = st.reg[g];
            else if constexpr (g < 2 * KS1)
                st.reg[g - KS1] = xbatch[(size_t)cload * VPC + (g - KS1) * KNN_ST_THREADS + tid];
            if constexpr ((t & 1) == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) accE[rt][r] = 0.0f;
                    }
                    accE[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accE[rt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) accO[rt][r] = 0.0f;
                    }
                    accO[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accO[rt], 0, 0, 0);
                }
            }
            // a share of the pending tile's sign packing (tile t - 1; none in the tile's first group: its last MFMAs
            // are still in the pipe); register 15 first, so that bit r of the tile's half word is register r
            if constexpr (ks > 0) {
                constexpr int e0 = (ks - 1) * (RT * 16) / (KS1 - 1), e1 = ks * (RT * 16) / (KS1 - 1);
#pragma unroll
                for (int e = e0; e < e1; ++e) {
                    const int rt = e / 16, r = 15 - e % 16;
                    if constexpr (t == 0) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w01[rt]) : "v"(accO[rt][r]));   // tile 3 of the chunk before
                    else if constexpr (t == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w01[rt]) : "v"(accE[rt][r]));  // tile 0
                    else if constexpr (t == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w23[rt]) : "v"(accO[rt][r]));  // tile 1
                    else asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w23[rt]) : "v"(accE[rt][r]));                         // tile 2
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef KNN_STAMP
        const unsigned long long u1 = __builtin_amdgcn_s_memtime();
#endif
        // slots: 0 = tile 0 (+0), 1 = tile 3 of the chunk before (-32), 2 = tile 2 (+64), 3 = tile 1 (+32)
        rounds(((u64)w23[0] << 32) | w01[0], ((u64)w23[1] << 32) | w01[1], c * KNN_ST_TILES * 32 + j32, 0x0102ff00u);
        scur = snxt;
        snxt = swr;
        swr = swr + 1 == RING ? 0 : swr + 1;
#ifdef KNN_STAMP
        const unsigned long long u2 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef KNN_STAMP
        const unsigned long long u3 = __builtin_amdgcn_s_memtime();
        tg += u1 - u0; tr += u2 - u1; tb += u3 - u2;
#endif
    }
#ifdef KNN_STAMP
    if ((blockIdx.x % 197) == 5 && lane == 0)
        printf("filter blk %d wave %d: prologue %llu, %d chunks: groups %llu rounds %llu barrier %llu per chunk; lpos %d %d\n", (int)blockIdx.x, wave,
               ts1 - tsA, c1 - c0, tg / (c1 - c0), tr / (c1 - c0), tb / (c1 - c0), lpos[0] + qn[0], lpos[1] + qn[1]);
#endif
    {
        // tile 3 of the last chunk
        unsigned w[RT] = {0u, 0u};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 15; r >= 0; --r) w[rt] = __builtin_amdgcn_alignbit(w[rt], __float_as_uint(accO[rt][r]), 31);
        rounds((u64)(w[0] & 0xffffu), (u64)(w[1] & 0xffffu), (c1 - 1) * KNN_ST_TILES * 32 + 96 + j32, 0u);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        while (qn[rt] > 0) flush(rt, qn[rt] < 64 ? qn[rt] : 64);
        if (lane == 0 && rb0 + rt < ntiles) {
            const int wid = b * ntiles + rb0 + rt;
            lcount[KNN_MAXPARTS * wid + part] = overflow[rt] ? 0 : lpos[rt];
            if (overflow[rt]) rbflag[wid] = 1;
        }
    }
    }  // items
}

// The same filter for small C (CP = 4, 8: the raw coordinates of the first block), where fp16 is useless (the
// rounding of coordinates of tens of metres exceeds neighbour distances): the distances are the exact fp32
// MFMA values of knn_kernel, the threshold is tau0 of the fp32 pre-pass, and the operands (CP / 2 floats per lane
// and tile) come straight from L2.  The survivors go through knn_refine_kernel like those of the fp16 filter;
// knn_kernel did the selection inside the all-pairs loop and spent most of its time in the per-survivor path.
template <int CP>
__global__ __launch_bounds__(256) void knn_filter32_kernel(const float *__restrict__ xt, const float *__restrict__ xx,
                                                           const float *__restrict__ tau0, int B, int Npad, int parts,
                                                           unsigned *__restrict__ lists, int *__restrict__ lcount,
                                                           int *__restrict__ rbflag)
{
    constexpr int CH = CP / 2;  // MFMA steps per tile = floats per lane and tile
    typedef float vecf __attribute__((ext_vector_type(CH)));
    __shared__ unsigned queues[4][KNN_QCAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j32 = lane & 31;
    const int b = blockIdx.x % B, part = (blockIdx.x / B) % parts;
    const int rb = (blockIdx.x / B / parts) * 4 + wave;  // this wave's row block
    const int ntiles = Npad / 32, nchunks = ntiles / KNN_ST_TILES;
    if (rb >= ntiles) return;
    const int c0 = part * nchunks / parts, c1 = (part + 1) * nchunks / parts;
    const int wid = b * ntiles + rb;
    const int listcap = KNN_LISTCAP;
    const int q0 = rb * 32;
    const vecf *xtb = reinterpret_cast<const vecf *>(xt + (size_t)b * Npad * CP);
    const float *yb = xx + (size_t)b * Npad;
    unsigned *list = lists + ((size_t)wid * KNN_MAXPARTS + part) * KNN_LISTCAP;
    unsigned *queue = queues[wave];

    const vecf a = xtb[(size_t)rb * 64 + lane];
    // filter threshold on t = 2 dot - xx_l:  pd = t - xx_i > tau0  <=  t > tau0 + xx_i, lowered by the rounding of
    // this very sum (a superset passes; the refine recomputes pd exactly)
    f32x2 thrp[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = q0 + mfma32_row(r, h);
        const float tau = tau0[(size_t)b * Npad + row], yy = yb[row];
        const float v = tau + yy;
        const float thr = v - (__builtin_fabsf(v) + __builtin_fabsf(tau) + yy) * 2.4e-7f;
        thrp[r >> 1][r & 1] = thr != thr ? __builtin_inff() : thr;  // padding rows (yy = +inf): nothing passes
    }

    int qhead = 0, qn = 0, lpos = 0;
    bool overflow = false;
    auto flush = [&](int n) {  // move the oldest n (<= 64) staged entries to the list
        if (lpos + n > listcap) {
            overflow = true;
        } else if (lane < n) {
            list[lpos + lane] = queue[(qhead + lane) & (KNN_QCAP - 1)];
        }
        lpos += n;
        qhead = (qhead + n) & (KNN_QCAP - 1);
        qn -= n;
    };
    auto signs = [&](const f32x16 &acc, float y) {
        unsigned bits = 0u;
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            const f32x2 t = __builtin_elementwise_fma(f32x2{acc[2 * i], acc[2 * i + 1]}, f32x2{2.0f, 2.0f}, f32x2{-y, -y});
            const f32x2 d = thrp[i] - t;
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d[1]), 31);
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d[0]), 31);
        }
        return bits;
    };
    auto rounds = [&](u64 bits, int cand) {  // as in knn_filter16_kernel: bit 16 t + r = tile t, register r
        u64 m = __ballot(bits != 0ull);
        while (m != 0) {
            if (qn > KNN_QCAP - 64) {
                while (qn >= 64) flush(64);
            }
            if (bits != 0ull) {
                const int pos = __builtin_ctzll(bits);
                const int r = pos & 15;
                const int off = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                queue[(qhead + qn + off) & (KNN_QCAP - 1)] =
                    ((unsigned)((r & 3) + 8 * (r >> 2) + 4 * h) << 27) | (unsigned)(cand + (pos >> 4) * 32);
                bits &= bits - 1ull;
            }
            qn += __popcll(m);
            m = __ballot(bits != 0ull);
        }
        if (qn >= 64) flush(64);
    };

    // four candidate tiles per iteration, the next four tiles' operands and norms in flight
    vecf bn[KNN_ST_TILES];
    float yn[KNN_ST_TILES];
#pragma unroll
    for (int tt = 0; tt < KNN_ST_TILES; ++tt) {
        bn[tt] = xtb[(size_t)(c0 * KNN_ST_TILES + tt) * 64 + lane];
        yn[tt] = yb[(c0 * KNN_ST_TILES + tt) * 32 + j32];
    }
    for (int c = c0; c < c1; ++c) {
        vecf bc[KNN_ST_TILES];
        float yc[KNN_ST_TILES];
        const int cn = c + 1 < c1 ? c + 1 : c;  // clamped prefetch: no branch around the loads
#pragma unroll
        for (int tt = 0; tt < KNN_ST_TILES; ++tt) {
            bc[tt] = bn[tt];
            yc[tt] = yn[tt];
            bn[tt] = xtb[(size_t)(cn * KNN_ST_TILES + tt) * 64 + lane];
            yn[tt] = yb[(cn * KNN_ST_TILES + tt) * 32 + j32];
        }
        f32x16 acc[KNN_ST_TILES];
#pragma unroll
        for (int tt = 0; tt < KNN_ST_TILES; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
#pragma unroll
        for (int st = 0; st < CH; ++st)
#pragma unroll
            for (int tt = 0; tt < KNN_ST_TILES; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], bc[tt][st], acc[tt], 0, 0, 0);
        const unsigned b0 = signs(acc[0], yc[0]), b1 = signs(acc[1], yc[1]);
        const unsigned b2 = signs(acc[2], yc[2]), b3 = signs(acc[3], yc[3]);
        rounds(((u64)(b2 | (b3 << 16)) << 32) | (b0 | (b1 << 16)), c * KNN_ST_TILES * 32 + j32);
    }
    while (qn > 0) flush(qn < 64 ? qn : 64);
    if (lane == 0) {
        lcount[KNN_MAXPARTS * wid + part] = overflow ? 0 : lpos;
        if (overflow) rbflag[wid] = 1;
    }
}

// One wave = one 32-row block (CAP = 64: k <= 32) or one HALF of it (CAP = 128: k <= 64 -- 128 keys per row are 32 KB for
// 32 rows, three waves per CU, and the pass is bound by the latency of its row gathers: two waves per block, each taking
// the list entries of its 16 rows -- the other half's entries are inert lanes of a batch -- bring seven to a CU:
// knn_refine<64, 128> 1.99 -> 1.52 ms, <4, 128> 0.76 -> 0.52 ms at B = 8, N = 32768, k = 64): exact pd of every listed survivor, selection, output.
template <int CP, int CAP, int HALVES_ = (CAP == 64 ? 1 : 2)>
// (the register budget is that of the refine phase -- 150 registers up to CP = 32, 243 at CP = 64 --, pinned by the second
// launch bound: the inlined exact phase, which alone would take 170 - 256 and an occupancy step, spills instead)
__global__ __launch_bounds__(64, (CP <= 32 ? 3 : 2)) void knn_refine_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ xx,
                                                        const unsigned *__restrict__ lists,
                                                        const int *__restrict__ lcount, int *__restrict__ rbflag,
                                                        int B, int N, int C, int Npad, int k, int parts,
                                                        int32_t *__restrict__ idx, const float *__restrict__ xt,
                                                        const float *__restrict__ tau0, int wpb = 0)
{
    constexpr int XS = CP + 1;  // LDS row stride of the raw query rows (odd: conflict-free column reads)
    // key storage per row: 48 entries for k <= 32 (the filter leaves ~42 survivors per row; a fuller row is
    // compacted to its best k).  With the query rows that is 20.7 KB per wave at C = 64: 7 row blocks per CU,
    // so all 1760 row blocks of B = 8, N = 7000 are resident in one round.
    constexpr int STORE = CAP == 64 ? 48 : CAP;
    constexpr int HALVES = HALVES_, ROWS = 32 / HALVES;  // rows of this wave
    constexpr unsigned INERT = 0xffffffffu;                        // a lane without an entry of this wave's rows
    // EXACT_INLINE (k <= 32: one wave per row block): a flagged row block is recomputed here by the exact all-pairs pass
    // (knn_exact_block) instead of by a kernel launched behind this one; its LDS shares the refine's (a union: 16.6 KB
    // inside the refine's 21 KB at C = 64).  k > 32 keeps the separate launch: 33 KB of keys there would halve the residency.
    constexpr bool EXACT_INLINE = CAP == 64;
    struct RefineLds {
        u64 keys[ROWS][STORE];
        int cnts[ROWS];
        u64 tkeys[ROWS];  // key of the k-th best of a compacted row: later offers must beat it as KEYS (the list
                          // is not index-ordered inside a chunk of four tiles, so an equal pd with a lower
                          // index can arrive after the compaction and must still win)
        float xq[ROWS][XS];
        float xxq[ROWS];
    };
    union LdsU {
        RefineLds r;
        KnnExactLds<EXACT_INLINE ? CAP : 1> e;
    };
    __shared__ LdsU L;
    auto &keys = L.r.keys;
    auto &cnts = L.r.cnts;
    auto &tkeys = L.r.tkeys;
    auto &xq = L.r.xq;
    auto &xxq = L.r.xxq;

    const int lane = threadIdx.x;
    const int half = HALVES == 2 ? (int)blockIdx.x & 1 : 0, bid = (int)blockIdx.x / HALVES;
    const int b = bid % B, rb = bid / B;
    const int ntiles = Npad / 32;
    const int wid = b * ntiles + rb;
    if (rbflag[wid]) {  // flagged by the prep / filter kernels: the exact fp32 pass recomputes this row block
        if constexpr (EXACT_INLINE) knn_exact_block<CP, CAP>(L.e, xt, xx, tau0, b, rb, N, Npad, k, idx);
        return;            // (k > 32: knn_kernel, launched behind this kernel)
    }
    const int q0 = rb * 32 + half * ROWS;  // first row of this wave
    const float *xxb = xx + (size_t)b * Npad;
    const float *xraw = x + (size_t)b * N * ldx;
    // the survivor list comes in `parts` segments (one per candidate part of the filter)
    const unsigned *list = lists + (size_t)wid * KNN_MAXPARTS * KNN_LISTCAP;
    // (entry e of the concatenation lives in segment p at list[p * listcap + e - (entries before segment p)])
    const int listcap = KNN_LISTCAP;
    int ncum[KNN_MAXPARTS + 1];
    ncum[0] = 0;
    // (balanced filter, wpb > 0: the row group's segments are those of the runs that touch it, see knn_bal_owner)
    if (wpb > 0) {
        const int nchunks_ = ntiles / KNN_ST_TILES, total = ((ntiles + KNN_ST_ROWS - 1) / KNN_ST_ROWS) * nchunks_;
        const int s0 = (rb / KNN_ST_ROWS) * nchunks_;
        parts = knn_bal_owner(s0 + nchunks_ - 1, wpb, total) - knn_bal_owner(s0, wpb, total) + 1;
    }
#pragma unroll
    for (int pp = 0; pp < KNN_MAXPARTS; ++pp) ncum[pp + 1] = ncum[pp] + (pp < parts ? lcount[KNN_MAXPARTS * wid + pp] : 0);
    const int n = ncum[KNN_MAXPARTS];

    for (int e = lane; e < ROWS * CP; e += 64) {
        const int r = e / CP, c = e % CP;
        xq[r][c] = (q0 + r < N && c < C) ? xraw[(size_t)(q0 + r) * ldx + c] : 0.0f;
    }
    if (lane < ROWS) {
        xxq[lane] = xxb[q0 + lane];
        cnts[lane] = 0;
        tkeys[lane] = 0ull;
    }

    // The chain list entry -> candidate row -> arithmetic is two dependent loads deep and a wave has little
    // company on its SIMD (LDS bounds the residency), so the list entries are fetched two batches ahead and the
    // candidate rows one batch ahead of the arithmetic.
    constexpr int NV = CP / 4;  // float4 per candidate row
    auto entry = [&](int base) -> unsigned {
        const int e = base + lane;
        if (e >= n) return INERT;
        int pp = 0;
#pragma unroll
        for (int q = 1; q < KNN_MAXPARTS; ++q) pp += e >= ncum[q] ? 1 : 0;
        int before = 0;
#pragma unroll
        for (int q = 1; q < KNN_MAXPARTS; ++q) before = e >= ncum[q] ? ncum[q] : before;
        const unsigned v = list[pp * listcap + (e - before)];
        return (HALVES == 2 && (int)(v >> 31) != half) ? INERT : v;  // (bits 31..27 = the row inside the 32-row block)
    };
    auto fetch = [&](unsigned qe_, float4(&v)[NV], float &yy) {
        const unsigned qe = qe_ == INERT ? 0u : qe_;  // inert lanes read candidate 0 (one address for all of them)
        const float *xl = xraw + (size_t)(qe & 0x7ffffffu) * ldx;
        yy = xxb[qe & 0x7ffffffu];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (4 * i < C) ? *reinterpret_cast<const float4 *>(xl + 4 * i) : float4{0, 0, 0, 0};
    };
    const bool vec = (C & 3) == 0 && (ldx & 3) == 0;
    unsigned qe_c = entry(0), qe_n = entry(64);
    float4 vc[NV], vn[NV];
    float yc = 0.0f, yn = 0.0f;
    if (vec) fetch(qe_c, vc, yc);
    for (int base = 0; base < n; base += 64) {
        const bool act = qe_c != INERT;
        const unsigned qe = act ? qe_c : 0u;
        const unsigned qe_nn = entry(base + 128);
        if (vec) fetch(qe_n, vn, yn);
        const int row = (int)(qe >> 27) & (ROWS - 1), cand = (int)(qe & 0x7ffffffu);
        float dot = 0.0f, ycand;
        if (vec) {
            ycand = yc;
            yc = yn;
            // channels beyond C multiply zeros: fma(0, 0, dot) == dot
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                dot = __builtin_fmaf(xq[row][4 * i + 0], vc[i].x, dot);
                dot = __builtin_fmaf(xq[row][4 * i + 1], vc[i].y, dot);
                dot = __builtin_fmaf(xq[row][4 * i + 2], vc[i].z, dot);
                dot = __builtin_fmaf(xq[row][4 * i + 3], vc[i].w, dot);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) vc[i] = vn[i];
        } else {
            const float *xl = xraw + (size_t)cand * ldx;
            ycand = xxb[cand];
            for (int c = 0; c < C; ++c) dot = __builtin_fmaf(xq[row][c], xl[c], dot);
        }
        qe_c = qe_n;
        qe_n = qe_nn;
        const float pd = __builtin_fmaf(2.0f, dot, -ycand) - xxq[row];  // the oracle's arithmetic
        const u64 key = make_key(pd, cand);
        bool pending = act;
        for (;;) {
            const bool want = pending && (key > tkeys[row]);
            int slot = -1;
            if (want) slot = atomicAdd(&cnts[row], 1);
            if (want && slot < STORE) keys[row][slot] = key;
            const bool over = want && slot >= STORE;
            const u64 mo = __ballot(over);
            if (!mo) break;
            // rows whose buffer is full: sort, keep the best k, tighten tau; their lanes offer again
            u64 todo = mo;
            while (todo) {
                const int l0 = __ffsll((long long)todo) - 1;
                const int ru = __builtin_amdgcn_readlane(row, l0);
                (void)compact_row<CAP, STORE>(&keys[ru][0], STORE, k, lane);
                if (lane == 0) {
                    cnts[ru] = k;
                    tkeys[ru] = keys[ru][k - 1];  // (written by this wave just above)
                }
                todo &= ~__ballot(over && row == ru);
            }
            pending = over;
        }
    }

    // fewer than k candidates in some row: the filter threshold was not a lower bound (cannot happen
    // while the bounds hold) -> hand the row block to the exact kernel
    const int need = k < N ? k : N;
    const bool bad = lane < ROWS && q0 + lane < N && cnts[lane] < need;
    if (__ballot(bad)) {
        if (lane == 0) rbflag[wid] = 1;
        if constexpr (EXACT_INLINE) knn_exact_block<CP, CAP>(L.e, xt, xx, tau0, b, rb, N, Npad, k, idx);
        return;
    }
    if constexpr (CAP == 64) {
        // All 32 rows at once (k <= 32, <= 48 keys per row): lane (h, j) takes half of row j's keys into registers
        // and sorts them with a bitonic network (no cross-lane traffic); the top 32 of the row are the half-cleaner
        // max(a[i], b[31 - i]) of the two lanes' lists followed by the in-lane merge.  ~1900 instructions per wave
        // against 32 wave-wide sorts of ~250.
        static_assert(STORE == 48, "two lanes x 24 keys");
        const int h = lane >> 5, j = lane & 31;
        const int jr = j < ROWS ? j : 0;  // (two waves per block: lanes of rows this wave does not own sort zeros)
        int c = j < ROWS ? cnts[jr] : 0;
        c = c < STORE ? c : STORE;
        u64 v[32];
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = h * 24 + i < c ? keys[jr][h * 24 + i] : 0ull;
#pragma unroll
        for (int i = 24; i < 32; ++i) v[i] = 0ull;
        auto cex = [](u64 &hi, u64 &lo) {  // compare-exchange: hi >= lo afterwards
            const u64 a = hi, d = lo;
            const bool gt = a > d;
            hi = gt ? a : d;
            lo = gt ? d : a;
        };
#pragma unroll
        for (int size = 2; size <= 32; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int p = i ^ stride;
                    if (p > i) {
                        if ((i & size) == 0 || size == 32) cex(v[i], v[p]);  // descending run
                        else cex(v[p], v[i]);
                    }
                }
            }
        }
        u64 w[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const u64 o = xchg64<32>(v[31 - i]);  // the other half's i-th smallest
            w[i] = v[i] > o ? v[i] : o;
        }
#pragma unroll
        for (int stride = 16; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if ((i ^ stride) > i) cex(w[i], w[i ^ stride]);
        }
        const int q = q0 + j;
        if (h == 0 && j < ROWS && q < N) {
            int32_t *out = idx + ((size_t)b * N + q) * k;
#pragma unroll
            for (int g = 0; g < 32; ++g)
                if (g < k) out[g] = (int32_t)(0xffffffffu - (unsigned)w[g]);
        }
    } else {
        for (int row = 0; row < ROWS; ++row) {
            int c = cnts[row];
            c = c < STORE ? c : STORE;
            (void)compact_row<CAP, STORE>(&keys[row][0], c, k, lane);
            const int q = q0 + row;
            if (q < N) {
                for (int g = lane; g < k; g += 64) {
                    const unsigned lo = (unsigned)keys[row][g];
                    idx[((size_t)b * N + q) * k + g] = (int32_t)(0xffffffffu - lo);
                }
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
static void knn_launch_cp(int cap, int B, hipStream_t st, const float *xt, const float *xx, const float *tau0,
                          const int *rbflag, int N, int Npad, int k, int32_t *idx)
{
    const dim3 grid(B * (Npad / 32)), block(64);
    if (cap == 64)
        hipLaunchKernelGGL((knn_kernel<CP, 64>), grid, block, 0, st, xt, xx, tau0, rbflag, B, N, Npad, k, idx);
    else
        hipLaunchKernelGGL((knn_kernel<CP, 128>), grid, block, 0, st, xt, xx, tau0, rbflag, B, N, Npad, k, idx);
}

template <int CP>
static void knn_bound_launch(int B, hipStream_t st, const void *xop, const float *xn, int Npad, int k, float *tau0)
{
    constexpr int RT = 1;  // (2 row tiles per wave halve the L2 traffic but measured slower: fewer waves)
    if (k > 40)
        hipLaunchKernelGGL((knn_bound_kernel<CP, RT, true>), dim3(B * (Npad / (32 * RT))), dim3(64), 0, st, xop, xn, B, Npad, k,
                           tau0);
    else
        hipLaunchKernelGGL((knn_bound_kernel<CP, RT>), dim3(B * (Npad / (32 * RT))), dim3(64), 0, st, xop, xn, B, Npad, k, tau0);
}

struct KnnWs {
    float *xt, *xx, *xxc, *tau0, *eps, *mu;
    half8 *xh;
    unsigned *bmax, *lists, *exch;
    int *lcount, *rbflag;
    size_t bytes;
};
static KnnWs knn_ws(int B, int C, int N, void *base)
{
    const size_t npad = knn_npad(N), cp = knn_cp(C);
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *r = p + off;
        off += align_up(bytes, 256);
        return r;
    };
    KnnWs w;
    w.xt = (float *)take((size_t)B * npad * cp * sizeof(float));
    w.xx = (float *)take((size_t)B * npad * sizeof(float));
    w.xxc = (float *)take((size_t)B * npad * sizeof(float));
    w.tau0 = (float *)take((size_t)B * npad * sizeof(float));
    w.eps = (float *)take((size_t)B * npad * sizeof(float));
    w.mu = (float *)take((size_t)B * cp * sizeof(float));
    w.xh = (half8 *)take((size_t)B * npad * ((cp < 16 ? 16 : cp) + 16) * sizeof(_Float16));  // + the augmented slab (C <= 4: one split slab)
    w.lcount = (int *)take((size_t)B * (npad / 32) * KNN_MAXPARTS * sizeof(int));
    w.bmax = (unsigned *)take((size_t)B * 2 * (npad / 32) * sizeof(unsigned));
    w.rbflag = (int *)take((size_t)B * (npad / 32) * sizeof(int));
    w.exch = (unsigned *)take((cp >= 16 || C <= 4) ? (size_t)B * npad * 2 * 64 * sizeof(unsigned) : 0);
    w.lists = (unsigned *)take((size_t)B * (npad / 32) * KNN_MAXPARTS * KNN_LISTCAP * sizeof(unsigned));
    w.bytes = off;
    return w;
}

// Workgroups per batch element of the balanced filter (0: the uniform split into `parts` candidate parts stays).  Balanced when
// the uniform grid does not fill the chip's 512 workgroup slots (two per CU) a whole number of times AND 512 / B runs per batch
// element are no longer than a row group (a run then has at most two items) and long enough that no row group is touched by
// more than KNN_MAXPARTS of them (2 + floor((U - 2) / L) <= 4).
static int knn_filter_balance(int B, int Npad, int parts)
{
    static const bool off = getenv("DGCNN_KNN_UNIFORM_FILTER") != nullptr;   // experiments
    const int ntiles = Npad / 32, U = ntiles / KNN_ST_TILES, ngroups = ceil_div(ntiles, KNN_ST_ROWS), total = ngroups * U;
    const int slots = 512;
    if (off || (B * parts * ngroups) % slots == 0) return 0;
    const int wpb = (slots + B / 2) / B;
    if (wpb < 1) return 0;
    const int lmin = total / wpb, lmax = ceil_div(total, wpb);
    if (lmin < 1 || lmax > U || 3 * lmin <= U - 2) return 0;
    return wpb;
}

template <int CP>
static void knn_f16_launch(int cap, int parts, const KnnWs &w, const float *x, int ldx, int B, int N, int C, int Npad, int k,
                           int32_t *idx, hipStream_t st, bool merge_here)
{
    const dim3 grid(B * (Npad / 32)), block(64);
    const int wpb = cap == 64 ? knn_filter_balance(B, Npad, parts) : 0;
    prof_begin(PROF_KNN_FILTER, st);
    hipLaunchKernelGGL((knn_filter16_kernel<CP>), dim3(wpb ? B * wpb : B * parts * ceil_div(Npad / 32, KNN_ST_ROWS)), dim3(KNN_ST_THREADS), 0,
                       st, w.xh, w.xxc, w.tau0, w.eps, w.bmax, B, Npad, parts, w.lists, w.lcount, w.rbflag,
                       merge_here ? w.exch : (const unsigned *)nullptr, w.xx, k, w.tau0, wpb);
    prof_end(PROF_KNN_FILTER, st);
    prof_begin(PROF_KNN_REFINE, st);
    // (two waves per block for k <= 32 as well -- knn_refine_kernel<CP, 64, 2> -- measured: 125 us against 80 at B = 8,
    // N = 7000: at 21 KB per wave that variant is not short of resident waves, and half-inert batches double its work)
    if (cap == 64)
        hipLaunchKernelGGL((knn_refine_kernel<CP, 64>), grid, block, 0, st, x, ldx, w.xx, w.lists, w.lcount, w.rbflag, B, N, C,
                           Npad, k, parts, idx, w.xt, w.tau0, wpb);
    else  // (two waves per row block: knn_refine_kernel)
        hipLaunchKernelGGL((knn_refine_kernel<CP, 128>), dim3(2 * grid.x), block, 0, st, x, ldx, w.xx, w.lists, w.lcount, w.rbflag, B,
                           N, C, Npad, k, parts, idx, w.xt, w.tau0);
    prof_end(PROF_KNN_REFINE, st);
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_knn_workspace_bytes(int B, int C, int N, int k)
{
    (void)k;
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    return knn_ws(B, C, N, nullptr).bytes;
}

extern "C" int dgcnn_knn(const float *x, int B, int N, int C, int k, int32_t *idx, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    return knn_ld(x, 0, B, N, C, k, idx, workspace, workspace_bytes, stream);
}

// the search over rows with a stride: x [B N][ldx] with the C features of a point in its first C columns (model.hip: a
// 64-column slice of the [P][192] concatenation of the block outputs); ldx = 0: dense
int dgcnn::knn_ld(const float *x, int ldx, int B, int N, int C, int k, int32_t *idx, void *workspace, size_t workspace_bytes,
                  void *stream)
{
    DG_REQUIRE(x && idx && workspace, "dgcnn_knn: null pointer");
    if (ldx <= 0) ldx = C;
    DG_REQUIRE(ldx >= C, "dgcnn_knn: row stride %d < C = %d", ldx, C);
    DG_REQUIRE(B > 0 && N > 0 && C > 0, "dgcnn_knn: bad shape B=%d N=%d C=%d", B, N, C);
    DG_REQUIRE(C <= 128, "dgcnn_knn: C=%d > 128 unsupported", C);
    // the reference raises from topk when k > N (model.py:39)
    DG_REQUIRE(k >= 1 && k <= N, "dgcnn_knn: selected index k out of range (k=%d, N=%d)", k, N);
    DG_REQUIRE(k <= 96, "dgcnn_knn: k=%d > 96 unsupported", k);
    DG_REQUIRE(workspace_bytes >= dgcnn_knn_workspace_bytes(B, C, N, k), "dgcnn_knn: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Npad = knn_npad(N), CP = knn_cp(C);
    const KnnWs w = knn_ws(B, C, N, workspace);
    float *xt = w.xt, *xx = w.xx;

    static const bool prepass_off = getenv("DGCNN_KNN_NO_PREPASS") != nullptr;
    static const bool f16_off = getenv("DGCNN_KNN_NO_F16") != nullptr;
    // the fp16 streams split the candidates of a row group over two workgroups when there are enough chunks:
    // twice the waves per SIMD (the streams are latency-bound at 1.7) and 128 instead of 64 threshold classes
    const int parts = Npad / (32 * KNN_ST_TILES) >= 8 ? 2 : 1;
    // threshold pre-pass: the k-th largest of 64 class maxima is a useful bound up to k = 40; with 128 classes (fp16
    // path: two candidate halves; small C: knn_bound_kernel's WIDE variant) up to k = 64 (about 1.4 k survivors per row)
    const bool prepass = N >= 256 && !prepass_off && (k <= 40 || (k <= 64 && (CP < 16 || parts == 2)));
    const bool f16path = prepass && CP >= 16 && !f16_off;
    // C <= 4: the same fp16 streams over two-piece operands (knn_prep_split4_kernel); more than 40 neighbours need the 128
    // threshold classes of two candidate parts
    static const bool split4_off = getenv("DGCNN_KNN_NO_SPLIT4") != nullptr;
    const bool split4 = prepass && C <= 4 && !f16_off && !split4_off && (k <= 40 || parts == 2);
    if (split4) {
        hipLaunchKernelGGL(knn_prep_split4_kernel, dim3(Npad / 128, B), dim3(256), 0, st, x, ldx, N, C, Npad, xx, w.xxc, w.xh, xt, w.bmax,
                           w.rbflag);
    } else if (prepass && CP >= 16) {
        // one pass over x: norms, fp16 and fp32 operand tiles, norm maxima (after the centre of each batch element)
        if (C == 64 && CP == 64)
            hipLaunchKernelGGL(knn_prep16_kernel<64>, dim3(Npad / 32, B), dim3(256), 0, st, x, ldx, N, C, Npad, CP, xx, w.xxc, w.xh,
                               xt, w.bmax, w.rbflag);
        else
            hipLaunchKernelGGL(knn_prep16_kernel<0>, dim3(Npad / 32, B), dim3(256), 0, st, x, ldx, N, C, Npad, CP, xx, w.xxc, w.xh,
                               xt, w.bmax, w.rbflag);
    } else {
        hipLaunchKernelGGL(knn_norm_kernel, dim3(ceil_div(Npad, 256), B), dim3(256), 0, st, x, ldx, N, C, Npad, xx);
        const size_t nvec = (size_t)Npad * CP / (CP >= 8 ? 4 : 2);
        const dim3 tgrid((unsigned)((nvec + 255) / 256), B);
        if (CP >= 8)
            hipLaunchKernelGGL((knn_tile_kernel<4>), tgrid, dim3(256), 0, st, x, ldx, N, C, Npad, CP, xt);
        else
            hipLaunchKernelGGL((knn_tile_kernel<2>), tgrid, dim3(256), 0, st, x, ldx, N, C, Npad, CP, xt);
    }
    DG_CHECK_LAUNCH("knn prep kernels");

    // threshold pre-pass (speed only; see the header)
    const float *tau0 = nullptr;
    const int *rbflag = nullptr;
    const int cap = (k <= 32) ? 64 : 128;
    // the filter cuts finer (up to KNN_MAXPARTS candidate parts): its workgroups are the unit of load balance, and
    // 3.4 of them per CU (two parts) leave 14 % of the chip idle at the end
    static const int fparts_env = getenv("DGCNN_KNN_FPARTS") ? atoi(getenv("DGCNN_KNN_FPARTS")) : 0;  // experiments
    int fparts = Npad / (32 * KNN_ST_TILES) >= 16 ? KNN_MAXPARTS : parts;
    if (fparts_env >= 1 && fparts_env <= KNN_MAXPARTS && Npad / (32 * KNN_ST_TILES) >= fparts_env) fparts = fparts_env;
    // two candidate halves and k <= 32: the thresholds CAN be merged inside the fp16 filter's prologue (knn_tau_merge32 per
    // lane) instead of by knn_bound_merge32_kernel -- measured (round 4, B = 4 and 8, N = 7000): the filter then takes 48.1
    // instead of 38.5 us per launch at B = 4, more than the 4.9 us launch it saves: every one of the filter's four candidate
    // parts repeats the merge of its rows in front of its stream.  Kept behind DGCNN_KNN_MERGE_IN_FILTER=1 for the record.
    static const bool merge_in_filter = getenv("DGCNN_KNN_MERGE_IN_FILTER") != nullptr;
    const bool merge_here = parts == 2 && k <= 32 && (f16path || split4) && merge_in_filter;
    if (prepass) {
        // (knn_bound_kernel writes tau0 for every row, padded rows included)
        if (CP >= 16 || split4) {
            const dim3 bgrid(B * parts * ceil_div(Npad / 32, KNN_ST_ROWS)), bblock(KNN_ST_THREADS);
            prof_begin(PROF_KNN_BOUND, st);
#define DG_BOUND16(...)                                                                                                      \
    hipLaunchKernelGGL((knn_bound16s_kernel<__VA_ARGS__>), bgrid, bblock, 0, st, w.xh, w.xxc, xx, w.bmax, B, Npad, k, parts, \
                       w.exch, w.tau0, w.eps)
            switch (split4 ? 4 : CP) {
                case 4: DG_BOUND16(16, true); break;
                case 16: DG_BOUND16(16); break;
                case 32: DG_BOUND16(32); break;
                case 64: DG_BOUND16(64); break;
                default: DG_BOUND16(128); break;
            }
#undef DG_BOUND16
            if (parts == 2 && k <= 32 && merge_here) {
                // the filter's own lanes merge the two halves' class maxima into the thresholds (knn_tau_merge32)
            } else if (parts == 2 && k <= 32)  // (Npad is a multiple of 128)
                hipLaunchKernelGGL(knn_bound_merge32_kernel, dim3(B * (Npad / 64)), dim3(64), 0, st, w.exch, w.xxc, xx, w.bmax, B,
                                   Npad, k, w.tau0, w.eps);
            else if (parts == 2)
                hipLaunchKernelGGL(knn_bound_merge_kernel, dim3(B * (Npad / 32)), dim3(64), 0, st, w.exch, w.xxc, xx, w.bmax, B,
                                   Npad, k, w.tau0, w.eps);
            prof_end(PROF_KNN_BOUND, st);
        } else if (CP == 8) {
            knn_bound_launch<8>(B, st, xt, xx, Npad, k, w.tau0);
        } else {
            knn_bound_launch<4>(B, st, xt, xx, Npad, k, w.tau0);
        }
        DG_CHECK_LAUNCH("knn_bound_kernel");
        tau0 = w.tau0;
    }
    static const bool f32filter_off = getenv("DGCNN_KNN_NO_F32_FILTER") != nullptr;
    if (split4) {
        // two-piece fp16 filter + exact refine; the exact fp32 kernel below only runs the row blocks these kernels flagged
        rbflag = w.rbflag;
        prof_begin(PROF_KNN_FILTER, st);
        const int wpb = cap == 64 ? knn_filter_balance(B, Npad, fparts) : 0;
        hipLaunchKernelGGL((knn_filter16_kernel<16, true>), dim3(wpb ? B * wpb : B * fparts * ceil_div(Npad / 32, KNN_ST_ROWS)), dim3(KNN_ST_THREADS),
                           0, st, w.xh, w.xxc, w.tau0, w.eps, w.bmax, B, Npad, fparts, w.lists, w.lcount, w.rbflag,
                           merge_here ? w.exch : (const unsigned *)nullptr, xx, k, w.tau0, wpb);
        prof_end(PROF_KNN_FILTER, st);
        prof_begin(PROF_KNN_REFINE, st);
        const dim3 rgrid(B * (Npad / 32));
        if (cap == 64)
            hipLaunchKernelGGL((knn_refine_kernel<4, 64>), rgrid, dim3(64), 0, st, x, ldx, w.xx, w.lists, w.lcount, w.rbflag, B, N, C, Npad,
                               k, fparts, idx, xt, w.tau0, wpb);
        else
            hipLaunchKernelGGL((knn_refine_kernel<4, 128>), dim3(2 * rgrid.x), dim3(64), 0, st, x, ldx, w.xx, w.lists, w.lcount, w.rbflag,
                               B, N, C, Npad, k, fparts, idx, xt, w.tau0);
        prof_end(PROF_KNN_REFINE, st);
        DG_CHECK_LAUNCH("knn_filter16_kernel<16, split> / knn_refine_kernel");
    } else if (prepass && CP < 16 && !f32filter_off) {
        // small C: exact fp32 filter + exact refine (see knn_filter32_kernel); knn_kernel below only runs flagged blocks
        if (hipMemsetAsync(w.rbflag, 0, (size_t)B * (Npad / 32) * sizeof(int), st) != hipSuccess)
            DG_FAIL("dgcnn_knn: memset failed");
        rbflag = w.rbflag;
        const dim3 fgrid(B * fparts * ceil_div(Npad / 32, 4)), rgrid(B * (Npad / 32));
        prof_begin(PROF_KNN_FILTER, st);
        if (CP == 4)
            hipLaunchKernelGGL((knn_filter32_kernel<4>), fgrid, dim3(256), 0, st, xt, xx, w.tau0, B, Npad, fparts, w.lists,
                               w.lcount, w.rbflag);
        else
            hipLaunchKernelGGL((knn_filter32_kernel<8>), fgrid, dim3(256), 0, st, xt, xx, w.tau0, B, Npad, fparts, w.lists,
                               w.lcount, w.rbflag);
        prof_end(PROF_KNN_FILTER, st);
        prof_begin(PROF_KNN_REFINE, st);
#define DG_REFINE(CP_, CAP_)                                                                                          \
    hipLaunchKernelGGL((knn_refine_kernel<CP_, CAP_>), dim3(rgrid.x * (CAP_ == 64 ? 1 : 2)), dim3(64), 0, st, x, ldx, w.xx, w.lists, \
                       w.lcount, w.rbflag, B, N, C, Npad, k, fparts, idx, xt, w.tau0)
        if (CP == 4) {
            if (cap == 64) DG_REFINE(4, 64);
            else DG_REFINE(4, 128);
        } else {
            if (cap == 64) DG_REFINE(8, 64);
            else DG_REFINE(8, 128);
        }
#undef DG_REFINE
        prof_end(PROF_KNN_REFINE, st);
        DG_CHECK_LAUNCH("knn_filter32_kernel / knn_refine_kernel");
    }
    if (f16path) {
        // fp16 filter + exact refine; the exact fp32 kernel below then only runs the row blocks these
        // kernels flagged (it returns at once for the others)
        rbflag = w.rbflag;
        switch (CP) {
            case 16: knn_f16_launch<16>(cap, fparts, w, x, ldx, B, N, C, Npad, k, idx, st, merge_here); break;
            case 32: knn_f16_launch<32>(cap, fparts, w, x, ldx, B, N, C, Npad, k, idx, st, merge_here); break;
            case 64: knn_f16_launch<64>(cap, fparts, w, x, ldx, B, N, C, Npad, k, idx, st, merge_here); break;
            default: knn_f16_launch<128>(cap, fparts, w, x, ldx, B, N, C, Npad, k, idx, st, merge_here); break;
        }
        DG_CHECK_LAUNCH("knn_filter16_kernel / knn_refine_kernel");
    }

    // the exact fp32 all-pairs kernel: the whole search where no filter path ran (rbflag == NULL); behind a filter path the
    // fallback for the row blocks it flagged -- for k <= 32 the refine kernel has already recomputed those itself
    // (knn_exact_block inside knn_refine_kernel), so nothing is launched here
    if (rbflag && cap == 64) return 0;
    switch (CP) {
        case 4: knn_launch_cp<4>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        case 8: knn_launch_cp<8>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        case 16: knn_launch_cp<16>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        case 32: knn_launch_cp<32>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        case 64: knn_launch_cp<64>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        case 128: knn_launch_cp<128>(cap, B, st, xt, xx, tau0, rbflag, N, Npad, k, idx); break;
        default: DG_FAIL("dgcnn_knn: unsupported padded channel count %d", CP);
    }
    DG_CHECK_LAUNCH("knn_kernel");
    return 0;
}
