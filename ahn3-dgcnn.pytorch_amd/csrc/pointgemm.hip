// pointgemm.hip -- per-POINT fp32 contractions of the EdgeConv block on v_mfma_f32_32x32x2_f32.
//
// The first convolution of every block (model.py:116,122,128: Conv2d(2C, 64, 1)) acts on the
// edge tuple e = [x_j - x_i ; x_i] (model.py:81).  Being linear,
//     W . e = Wa . x_j + (Wb - Wa) . x_i            (W = [Wa | Wb])
// so it is evaluated once per point instead of once per edge: TP = x Wa^T, TQ = x (Wb - Wa)^T
// ([P][64] each); the edge passes then only gather-and-add rows of TP.  The same identity turns
// the first convolution's backward into two more per-point products (dgrad and wgrad) over the
// scatter-reduced edge gradients.  All three run here on the exact-fp32 MFMA.
#include <stdlib.h>

#include "edge_kernels.h"
#include "edge_mlp16.h"

namespace dgcnn {

// Forward mode of the two row kernels below (FROM_W1): W is not a ready [K][NO] matrix but the convolution weight
// w1 [64][2C]; every workgroup builds Wcat^T [C][128] = [Wa ; Wb - Wa]^T in LDS itself (32 KB of L2-resident reads: one
// launch less than a separate prep kernel), workgroup 0 also writes wcat [128][C] for the backward's dgrad and zeroes
// the counters of the statistics passes that follow in the same call.
__device__ __forceinline__ float wcat_at(const float *__restrict__ w1, int C, int o, int c)
{
    return o < EC ? w1[(size_t)o * 2 * C + c] : w1[(size_t)(o - EC) * 2 * C + C + c] - w1[(size_t)(o - EC) * 2 * C + c];
}
__device__ __forceinline__ void fwd_side_jobs(const float *__restrict__ w1, int C, const FwdSide &sd)
{
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < sd.nzero; i += 256) sd.zero_words[i] = 0;
    for (int e = threadIdx.x; e < 128 * C; e += 256) sd.wcat[e] = wcat_at(w1, C, e / C, e % C);
    // eval mode: the BatchNorm affine pairs of the block from the RUNNING statistics (no reduction; they were a launch of 64
    // threads per layer: 4.8 us each, ten per eval forward of the model)
    if ((int)threadIdx.x < 64 * sd.nbn) {
        const int l = threadIdx.x >> 6, c = threadIdx.x & 63;
        const dgcnn_bn_params &bn = sd.bn[l];
        const float mean = bn.running_mean[c], var = bn.running_var[c];
        const float rstd = (float)(1.0 / sqrt((double)var + (double)sd.eps));
        const float sc = bn.weight[c] * rstd;
        float *stats = sd.stats[l];
        stats[ST_MEAN + c] = mean;
        stats[ST_RSTD + c] = rstd;
        stats[ST_SC + c] = sc;
        stats[ST_SH + c] = bn.bias[c] - mean * sc;
    }
}

// cnt[b][chunk][m] = number of edges from the chunk's rows into m (rev_count_kernel's result), by one 256-thread workgroup whose
// four waves share the N-entry table in LDS (counting does not depend on the order the adds arrive in)
__device__ __forceinline__ void rev_count_rider(const RevCountRider &a, int block, int *table)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B_ = a.nblocks / a.chunks, b = block % B_, chunk = block / B_;
    const int N = a.N, k = a.k;
    const int r0 = chunk * a.rows_per, r1 = r0 + a.rows_per < N ? r0 + a.rows_per : N;
    const int32_t *ib = a.idx + (size_t)b * N * k;
    int *out = a.cnt + (size_t)(b * a.chunks + chunk) * N;
    if (block == 0)
        for (int i = threadIdx.x; i < a.nzero; i += 256) a.zero_words[i] = 0;
    for (int m = threadIdx.x; m < N; m += 256) table[m] = 0;
    __syncthreads();
    const int l0 = lane < k ? lane : k - 1;
    for (int r = r0 + wave * 8; r < r1; r += 32) {
        int nv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = r + u < r1 ? r + u : r1 - 1;
            nv[u] = ib[(size_t)rr * k + l0] | (r + u < r1 && lane < k ? 0 : (int)0x80000000);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (nv[u] >= 0) atomicAdd(&table[nv[u]], 1);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < N; m += 256) out[m] = table[m];
}

// One wave = one 32-row tile of points x all NO (<= 32 NT) output columns.
// MFMA operand maps (32x32x2): A lane (i, h) = A[row i][k(s, h)];  B lane (n, h) = B[k(s, h)][col n];
// C/D lane (n, h), register r = D[row mfma32_row(r, h)][col n].  The contraction order is free here
// (tolerance-gated, not bit-gated), so lane-half h takes the h-th HALF of the K range:
// k(s, h) = h * KH + s with KH = ceil(K / 2): every lane reads its row's KH values as one contiguous run
// (a handful of 16-byte loads per row instead of one strided dword per MFMA step).
constexpr int RG_KH_MAX = 64;  // K0 + K1 <= 128

constexpr int RG_LDS_FLOATS = 8192;  // W (K x NO <= 128 x 64 or 64 x 128 fp32 = 32 KB) staged once per workgroup

template <int NT, bool FROM_W1>
__global__ __launch_bounds__(256) void rowgemm_kernel(const float *__restrict__ A0, int K0,
                                                      const float *__restrict__ A1, int K1,
                                                      const float *__restrict__ W, int NO,
                                                      float *__restrict__ Y0, int NO0, float *__restrict__ Y1,
                                                      int P, const FwdSide sd, const RowLd ld)
{
    // the weights are read by every wave at every step: stage them in LDS (the kernel was bound by the
    // latency of those loads: 73 % of its wave cycles in s_waitcnt)
    __shared__ float wl[RG_LDS_FLOATS];
    if (FROM_W1 && sd.rc.cnt && (int)blockIdx.x >= sd.rc.main_blocks) {  // the counting pass of the backward's reverse lists
        rev_count_rider(sd.rc, (int)blockIdx.x - sd.rc.main_blocks, reinterpret_cast<int *>(wl));
        return;
    }
    const int K = K0 + K1, KH = (K + 1) >> 1;
    const bool staged = K * NO <= RG_LDS_FLOATS;
    if (FROM_W1) {  // K = C <= 64, NO = 128: always staged
        fwd_side_jobs(W, K, sd);
        for (int e = threadIdx.x; e < K * 128; e += 256) wl[e] = wcat_at(W, K, e & 127, e >> 7);
    } else if (staged) {
        for (int e = threadIdx.x; e < K * NO; e += 256) wl[e] = W[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= P) return;
    const int row = p0 + i32;
    const bool rv = row < P;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    // this lane's K values: k = h * KH + s, s < KH (zero beyond K or beyond P)
    auto a_at = [&](int kk) -> float {
        if (!rv || kk >= K) return 0.0f;
        return (kk < K0) ? A0[(size_t)row * ld.a0 + kk] : A1[(size_t)row * K1 + (kk - K0)];
    };
    for (int s0 = 0; s0 < KH; s0 += 8) {
        float a[8];
        const int kb = h * KH + s0;
        // fast path: 8 contiguous, 16-byte aligned values inside one source array
        const bool in0 = kb + 8 <= K0 && (ld.a0 & 3) == 0 && (kb & 3) == 0;
        const bool in1 = kb >= K0 && kb + 8 <= K && (K1 & 3) == 0 && ((kb - K0) & 3) == 0;
        if (s0 + 8 <= KH && rv && (in0 || in1)) {
            const float *src = in0 ? A0 + (size_t)row * ld.a0 + kb : A1 + (size_t)row * K1 + (kb - K0);
            const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
            a[0] = v0.x, a[1] = v0.y, a[2] = v0.z, a[3] = v0.w, a[4] = v1.x, a[5] = v1.y, a[6] = v1.z, a[7] = v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (s0 + e < KH) ? a_at(kb + e) : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = kb + e;
            const bool kv = s0 + e < KH && kk < K;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + i32;
                float b = 0.0f;
                if (kv && n < NO) b = staged ? wl[kk * NO + n] : W[(size_t)kk * NO + n];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b, acc[nt], 0, 0, 0);
            }
        }
    }
    const int NO1 = NO - NO0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + i32;
        if (n < NO) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = p0 + mfma32_row(r, h);
                if (pr < P) {
                    if (n < NO0) {
                        float *o0 = Y0 + (size_t)pr * ld.y0 + n;
                        *o0 = ld.y0_acc ? *o0 + acc[nt][r] : acc[nt][r];
                    } else
                        Y1[(size_t)pr * NO1 + (n - NO0)] = acc[nt][r];
                }
            }
        }
    }
}

// Fast path of rowgemm_kernel for the shapes of the 64-channel blocks (forward K = 64 -> NO = 128, dgrad K = 128 ->
// NO = 64): everything is compile-time or wave-uniform, so the K loop is straight-line code (the general kernel
// above guards every operand and ends up with one branch and one s_waitcnt per MFMA).
//   K = 2 KH, NO = 32 NT, K * NO <= RG_LDS_FLOATS, NO0 a multiple of 32, and either K1 == 0 or K0 == KH, so that the
//   KH values of a lane (k = h KH + s) are one contiguous run of one source row.
template <int KH, int NT, bool FROM_W1>
__global__ __launch_bounds__(256) void rowgemm_fast_kernel(const float *__restrict__ A0, int K0,
                                                           const float *__restrict__ A1, int K1,
                                                           const float *__restrict__ W, float *__restrict__ Y0, int NO0,
                                                           float *__restrict__ Y1, int P, const FwdSide sd, const RowLd ld)
{
    constexpr int K = 2 * KH, NO = 32 * NT;
    __shared__ float wl[K * NO];
    if (FROM_W1 && sd.rc.cnt && (int)blockIdx.x >= sd.rc.main_blocks) {  // the counting pass of the backward's reverse lists
        rev_count_rider(sd.rc, (int)blockIdx.x - sd.rc.main_blocks, reinterpret_cast<int *>(wl));
        return;
    }
    if (FROM_W1) {  // K = C = 64, NO = 128: wl[c][o] = Wcat[o][c]; lanes run over o (conflict-free LDS writes), each
                    // reads 16 bytes of row o of w1
        static_assert(!FROM_W1 || (K == 64 && NO == 128), "forward shape");
        fwd_side_jobs(W, K, sd);
        for (int e = threadIdx.x; e < 128 * (K / 4); e += 256) {
            const int o = e & 127, c4 = (e >> 7) * 4;
            const float *row = W + (size_t)(o & 63) * 2 * K;
            float4 v = *reinterpret_cast<const float4 *>(row + c4);
            if (o >= EC) {
                const float4 b = *reinterpret_cast<const float4 *>(row + K + c4);
                v = float4{b.x - v.x, b.y - v.y, b.z - v.z, b.w - v.w};
            }
            wl[(c4 + 0) * NO + o] = v.x;
            wl[(c4 + 1) * NO + o] = v.y;
            wl[(c4 + 2) * NO + o] = v.z;
            wl[(c4 + 3) * NO + o] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < K * NO / 4; e += 256)
            reinterpret_cast<float4 *>(wl)[e] = reinterpret_cast<const float4 *>(W)[e];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int row = p0 + i32 < P ? p0 + i32 : P - 1;  // clamped: loads are unconditional, stores guarded
    const float *src = (K1 == 0 || h == 0) ? A0 + (size_t)row * ld.a0 + h * (K1 == 0 ? KH : 0) : A1 + (size_t)row * K1;
    float4 av[KH / 4];
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) av[i] = reinterpret_cast<const float4 *>(src)[i];
    __syncthreads();
    if (p0 >= P) return;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    const float *wrow = wl + (size_t)h * KH * NO + i32;
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) {
        const float a4[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], wrow[(4 * i + e) * NO + nt * 32], acc[nt], 0, 0, 0);
    }
    const bool full = p0 + 32 <= P;
    const int NO1 = NO - NO0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + i32;
        float *dst = n < NO0 ? Y0 + n : Y1 + (n - NO0);
        const int ldo = n < NO0 ? ld.y0 : NO1;
        const bool add = n < NO0 && ld.y0_acc;  // (wave-uniform: NO0 is a multiple of 32)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = p0 + mfma32_row(r, h);
            if (full || pr < P) dst[(size_t)pr * ldo] = add ? dst[(size_t)pr * ldo] + acc[nt][r] : acc[nt][r];
        }
    }
}

// gw1 [64][2C] of the block's first convolution from the per-chunk partial products G = [dR ; dD]^T x ([128][C]):
// W = [Wa | Wb] acts as Wa x_j + (Wb - Wa) x_i, so gWa = G_R - G_D (c < C: G[o][c] - G[64 + o][c]) and gWb = G_D
// (c >= C: G[64 + o][c - C]).  NW waves take consecutive shares of the chunks (eight chunks in flight, added in chunk
// order), the NW sums are added in wave order, then one subtraction: a fixed association.  Workgroup `block` finishes 64
// elements.  The body of finish_dw1_kernel (NW = 16) and of the workgroups that ride behind the input-gradient product
// (NW = 4, rowgemm_split_kernel: the two kernels read the same [dR | dD] and neither needs the other's result).
template <int NW>
__device__ __forceinline__ void finish_dw1_body(const float *__restrict__ partial, int nchunk, int C, float *__restrict__ gw1, int block,
                                                int *__restrict__ zero_words = nullptr, int nzero = 0)
{
    // (the last launch of a backward call: its first workgroup leaves the call's counters at zero for the next call that uses
    // the same saved state -- the forward zeroed them for this one)
    if (block == 0 && zero_words)
        for (int i = threadIdx.x; i < nzero; i += 64 * NW) zero_words[i] = 0;
    __shared__ float red[NW][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = block * 64 + lane, n = 128 * C;
    const int q = (nchunk + NW - 1) / NW;
    const int c0 = w * q, c1 = c0 + q < nchunk ? c0 + q : nchunk;
    float sr = 0.0f, sd = 0.0f;
    const bool live = t < EC * 2 * C;
    const int o = live ? t / (2 * C) : 0, c = live ? t % (2 * C) : 0;
    const bool first = c < C;  // a gWa column: both sources
    const int er = o * C + (first ? c : 0), ed = (EC + o) * C + (first ? c : c - C);
    if (live) {
        int ch = c0;
        for (; ch + 8 <= c1; ch += 8) {
            float vr[8], vd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                vd[u] = partial[(size_t)(ch + u) * n + ed];
                vr[u] = first ? partial[(size_t)(ch + u) * n + er] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sd += vd[u];
                sr += vr[u];
            }
        }
        if (ch < c1) {  // the last, partial batch: loads in flight together, past-the-end chunks masked
            float vr[8], vd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cu = ch + u < c1 ? ch + u : ch;
                vd[u] = partial[(size_t)cu * n + ed];
                vr[u] = first ? partial[(size_t)cu * n + er] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sd += ch + u < c1 ? vd[u] : 0.0f;
                sr += ch + u < c1 ? vr[u] : 0.0f;
            }
        }
    }
    red[w][0][lane] = sr;
    red[w][1][lane] = sd;
    __syncthreads();
    if (w == 0 && live) {
        float tr = red[0][0][lane], td = red[0][1][lane];
#pragma unroll
        for (int i = 1; i < NW; ++i) {
            tr += red[i][0][lane];
            td += red[i][1][lane];
        }
        gw1[t] = first ? tr - td : td;
    }
}
__global__ __launch_bounds__(1024) void finish_dw1_kernel(const float *__restrict__ partial, int nchunk, int C,
                                                          float *__restrict__ gw1, int *__restrict__ zero_words, int nzero)
{
    finish_dw1_body<16>(partial, nchunk, C, gw1, (int)blockIdx.x, zero_words, nzero);
}

constexpr int WGRAD_FAST_POINTS = 256;
// The same product on the fp16 matrix cores (the fp32 kernel above is bound by its 256 matrix instructions of 64 cycles per
// wave: 21.5 us at P = 56 000 against 9.4 at the fp32 MFMA peak).  The contraction runs over the POINTS, so a lane's eight
// consecutive k of an operand are eight points of its channel: eight 4-byte loads, each a 128-byte row segment per half
// wave.  A wave takes its 256 points in four chunks of 64: the chunk's values wait in registers (32 of A, 64 of X per
// lane), their largest magnitudes give the chunk's two exact power-of-two scales (gradients of any size, no range check),
// both operands are split in two fp16 halves (edge_mlp16.h) and four steps of three v_mfma_f32_32x32x16_f16 per tile go
// to a chunk accumulator, which joins the total with 1 / (sA sB): 96 matrix instructions of 32 cycles per wave.
// (`block`: the index among the product's workgroups -- the body of wgrad_split64_kernel and of the workgroups that ride in the
// input-gradient product's launch, rowgemm_split_kernel)
__device__ __forceinline__ void wgrad_split64_body(const float *__restrict__ A0, const float *__restrict__ A1,
                                                   const float *__restrict__ X, int ldx, int P, float *__restrict__ partial, int block)
{
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int pbeg = block * WGRAD_FAST_POINTS;
    const int pend = pbeg + WGRAD_FAST_POINTS < P ? pbeg + WGRAD_FAST_POINTS : P;
    const float *A = (mt < 2 ? A0 : A1) + (mt & 1) * 32;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    for (int p0 = pbeg; p0 < pend; p0 += 64) {
        float av[32], bv[2][32];
        float amax = 0.0f, bmax = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) {  // value q of the lane: point p0 + 16 (q / 8) + 8 h + q % 8
            const int pp = p0 + 16 * (q >> 3) + 8 * h + (q & 7);
            const bool pv = pp < pend;
            const size_t ro = (size_t)(pv ? pp : pbeg) * 64 + i32;  // clamped row: unconditional loads, masked values
            const size_t rx = (size_t)(pv ? pp : pbeg) * (size_t)ldx + i32;
            const float a0 = A[ro], b0 = X[rx], b1 = X[rx + 32];
            av[q] = pv ? a0 : 0.0f;
            bv[0][q] = pv ? b0 : 0.0f;
            bv[1][q] = pv ? b1 : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            amax = fmaxf(amax, fabsf(av[q]));
            bmax = fmaxf(bmax, fmaxf(fabsf(bv[0][q]), fabsf(bv[1][q])));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            amax = fmaxf(amax, __shfl_xor(amax, o));
            bmax = fmaxf(bmax, __shfl_xor(bmax, o));
        }
        const float sA = scale_to(amax, 15), sB = scale_to(bmax, 15);
        f32x16 cacc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[nt][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned H[4], L[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair(av[8 * s + 2 * j] * sA, av[8 * s + 2 * j + 1] * sA, H[j], L[j]);
            const half8 ahi = words_as_half8(H[0], H[1], H[2], H[3]), alo = words_as_half8(L[0], L[1], L[2], L[3]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) split_pair(bv[nt][8 * s + 2 * j] * sB, bv[nt][8 * s + 2 * j + 1] * sB, H[j], L[j]);
                const half8 bhi = words_as_half8(H[0], H[1], H[2], H[3]), blo = words_as_half8(L[0], L[1], L[2], L[3]);
                cacc[nt] = mfma3(ahi, alo, bhi, blo, cacc[nt]);
            }
        }
        const float un = 1.0f / (sA * sB);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = __builtin_fmaf(cacc[nt][r], un, acc[nt][r]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            partial[((size_t)block * 128 + mt * 32 + mfma32_row(r, h)) * 64 + nt * 32 + i32] = acc[nt][r];
}
__global__ __launch_bounds__(256) void wgrad_split64_kernel(const float *__restrict__ A0, const float *__restrict__ A1,
                                                            const float *__restrict__ X, int ldx, int P, float *__restrict__ partial)
{
    wgrad_split64_body(A0, A1, X, ldx, P, partial, (int)blockIdx.x);
}

// The same two products on the fp16 matrix cores: both operands scaled by an exact power of two and split in two fp16
// halves, x' = hi + lo (edge_mlp16.h), three v_mfma_f32_32x32x16_f16 per 16 k (hi.hi + hi.lo + lo.hi, fp32 accumulate):
// 48 matrix instructions of 32 cycles per 32-row tile instead of 128 of 64 -- the fp32 kernel above is bound by exactly
// that (MFMA-dense code runs at 1.3-1.7 GHz on this chip).  The dropped lo.lo term and the rounding of lo are 2^-22
// relative to the largest element of the tile.  Scales: W' = sW W with sW from max |W| (every workgroup finds it
// itself), A' = sA A with sA from the largest magnitude of the WAVE'S OWN 32-row tile -- a local, exact scale, so that
// gradients of any magnitude go through without a range check or a fallback; the accumulators leave with 1 / (sA sW).
// W image: fp16 planes [n][k] (hi plane, lo plane), 16-byte chunks XOR-swizzled so that the row reads are conflict-free.
template <int KH, int NT, bool FROM_W1>
__global__ __launch_bounds__(256) void rowgemm_split_kernel(const float *__restrict__ A0, int K0,
                                                            const float *__restrict__ A1, int K1,
                                                            const float *__restrict__ W, float *__restrict__ Y0, int NO0,
                                                            float *__restrict__ Y1, int P, const FwdSide sd, const RowLd ld,
                                                            const Dw1Rider rd)
{
    // workgroups rd.main_blocks .. with rd.wg_x: the first convolution's WEIGHT gradient (wgrad_split64_body) beside its input
    // gradient -- both read [dR | dD], neither needs the other, and the weight gradient's 219 workgroups of two waves per SIMD
    // leave most of the chip idle on their own (16.2 + 15.3 us as two launches at B = 8, N = 7000)
    if (!FROM_W1 && KH == 64 && rd.wg_x && (int)blockIdx.x >= rd.main_blocks) {
        wgrad_split64_body(A0, A1, rd.wg_x, rd.wg_ldx, P, rd.wg_partial, (int)blockIdx.x - rd.main_blocks);
        return;
    }
    // workgroups rd.main_blocks .. : the weight gradient's fold (finish_dw1_body), riding in this launch
    if (!FROM_W1 && rd.gw1 && (int)blockIdx.x >= rd.main_blocks) {
        finish_dw1_body<4>(rd.partial, rd.nchunk, rd.C, rd.gw1, (int)blockIdx.x - rd.main_blocks, rd.zero_words, rd.nzero);
        return;
    }
    constexpr int K = 2 * KH, NO = 32 * NT, ROWB = K * 2, PLANE = NO * ROWB;
    static_assert(K * NO == 8192 && (K == 64 || K == 128), "the two shapes of the 64-channel blocks");
    __shared__ __attribute__((aligned(16))) unsigned char wp[2 * PLANE];
    __shared__ float s_red[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, i32 = lane & 31;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int row = p0 + i32 < P ? p0 + i32 : P - 1;  // clamped: loads are unconditional, stores guarded
    const float *src = (K1 == 0 || h == 0) ? A0 + (size_t)row * ld.a0 + h * (K1 == 0 ? KH : 0) : A1 + (size_t)row * K1;
    float4 av[KH / 4];
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) av[i] = reinterpret_cast<const float4 *>(src)[i];

    // ---- W: item e of a thread = four consecutive k of one output column n
    if (FROM_W1) fwd_side_jobs(W, K, sd);
    float4 wv[8];
    float wmax = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (FROM_W1) {  // K = C = 64, NO = 128: column n of Wcat^T is row n of [Wa ; Wb - Wa]
            const int n = e & 127, k4 = (e >> 7) * 4;
            const float *r = W + (size_t)(n & 63) * 2 * K;
            float4 v = *reinterpret_cast<const float4 *>(r + k4);
            if (n >= EC) {
                const float4 b = *reinterpret_cast<const float4 *>(r + K + k4);
                v = float4{b.x - v.x, b.y - v.y, b.z - v.z, b.w - v.w};
            }
            wv[u] = v;
        } else {  // W [K][NO] row-major
            const int n = e & (NO - 1), k4 = (e / NO) * 4;
            wv[u] = float4{W[(size_t)(k4 + 0) * NO + n], W[(size_t)(k4 + 1) * NO + n], W[(size_t)(k4 + 2) * NO + n],
                           W[(size_t)(k4 + 3) * NO + n]};
        }
        wmax = fmaxf(fmaxf(wmax, fmaxf(fabsf(wv[u].x), fabsf(wv[u].y))), fmaxf(fabsf(wv[u].z), fabsf(wv[u].w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_red[wave] = wmax;
    __syncthreads();
    const float sW = scale_to(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])), 12);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int n = FROM_W1 ? (e & 127) : (e & (NO - 1)), k4 = FROM_W1 ? (e >> 7) * 4 : (e / NO) * 4;
        const int swz = K == 64 ? w2_swz(n) : (n & 15);
        const int off = n * ROWB + ((((k4 >> 3)) ^ swz) << 4) + (k4 & 7) * 2;
        unsigned H0, L0, H1, L1;
        split_pair(wv[u].x * sW, wv[u].y * sW, H0, L0);
        split_pair(wv[u].z * sW, wv[u].w * sW, H1, L1);
        *reinterpret_cast<uint2 *>(wp + off) = make_uint2(H0, H1);
        *reinterpret_cast<uint2 *>(wp + PLANE + off) = make_uint2(L0, L1);
    }

    // ---- A: the tile's own scale, then the split (lane half h holds k = h KH .. h KH + KH - 1 of its row)
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < KH / 4; ++i)
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(av[i].x), fabsf(av[i].y))), fmaxf(fabsf(av[i].z), fabsf(av[i].w)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sA = scale_to(amax, 10);
    unsigned aH[KH / 2], aL[KH / 2];
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) {
        split_pair(av[i].x * sA, av[i].y * sA, aH[2 * i], aL[2 * i]);
        split_pair(av[i].z * sA, av[i].w * sA, aH[2 * i + 1], aL[2 * i + 1]);
    }
    __syncthreads();
    if (p0 >= P) return;
    const bool full = p0 + 32 <= P;
    const float un = 1.0f / (sA * sW);
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    // y0_acc (the product is the SECOND gradient that reaches Y0): what Y0 holds is fetched here, ahead of the matrix
    // instructions, and joins the result as the accumulators' starting value (in the product's scale) -- read behind the
    // product it was a dependent load in front of every store: 22 us against 8 at P = 28 000
    if (ld.y0_acc) {
        const float iun = sA * sW;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + i32;
            if (n < NO0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pr = p0 + mfma32_row(r, h);
                    acc[nt][r] = (full || pr < P) ? Y0[(size_t)pr * ld.y0 + n] * iun : 0.0f;
                }
            }
        }
    }
    // step s of lane half h covers k = h KH + 8 s .. + 7: chunk h KH / 8 + s of the image rows
#pragma unroll
    for (int s = 0; s < KH / 8; ++s) {
        const half8 ahi = words_as_half8(aH[4 * s], aH[4 * s + 1], aH[4 * s + 2], aH[4 * s + 3]);
        const half8 alo = words_as_half8(aL[4 * s], aL[4 * s + 1], aL[4 * s + 2], aL[4 * s + 3]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + i32;
            const int swz = K == 64 ? w2_swz(n) : (n & 15);
            const unsigned char *bp = wp + n * ROWB + (((h * (KH / 8) + s) ^ swz) << 4);
            const half8 bhi = *reinterpret_cast<const half8 *>(bp);
            const half8 blo = *reinterpret_cast<const half8 *>(bp + PLANE);
            acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
        }
    }
    const int NO1 = NO - NO0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + i32;
        float *dst = n < NO0 ? Y0 + n : Y1 + (n - NO0);
        const int ldo = n < NO0 ? ld.y0 : NO1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = p0 + mfma32_row(r, h);
            if (full || pr < P) dst[(size_t)pr * ldo] = acc[nt][r] * un;
        }
    }
}

constexpr int WGRAD_CHUNK = 64;  // points per wave (875 waves at P = 56 000)

// partial[chunk][m][n] = sum_p A(p, m) X[p][n]: contraction over the points (MFMA K dimension).
template <int MT, int NT>
__global__ __launch_bounds__(64) void wgrad_kernel(const float *__restrict__ A0, int M0,
                                                   const float *__restrict__ A1, int M1,
                                                   const float *__restrict__ X, int ldx, int C, int P,
                                                   float *__restrict__ partial)
{
    const int lane = threadIdx.x, h = lane >> 5, i32 = lane & 31;
    const int pbeg = blockIdx.x * WGRAD_CHUNK;
    const int pend = pbeg + WGRAD_CHUNK < P ? pbeg + WGRAD_CHUNK : P;
    const int M = M0 + M1;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // the loads of 8 points (4 MFMA steps) are issued together: the wave is alone on its SIMD and a step's
    // operands are 6 dwords per lane, so one step per round trip is latency-bound
    constexpr int UN = 4;
    for (int p = pbeg; p < pend; p += 2 * UN) {
        float a[UN][MT], b[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int pp = p + 2 * u + h;
            const bool pv = pp < pend;
            const int pc = pv ? pp : pbeg;  // clamped row: the load is unconditional, the value masked
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 32 + i32;
                float v = 0.0f;
                if (m < M) v = m < M0 ? A0[(size_t)pc * M0 + m] : A1[(size_t)pc * M1 + (m - M0)];
                a[u][mt] = pv ? v : 0.0f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + i32;
                float v = 0.0f;
                if (n < C) v = X[(size_t)pc * ldx + n];
                b[u][nt] = pv ? v : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mt], b[u][nt], acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + i32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + mfma32_row(r, h);
                if (m < M && n < C) partial[((size_t)blockIdx.x * M + m) * C + n] = acc[mt][nt][r];
            }
        }
}

// out[e] = sum_chunk partial[chunk][e] for e < n.  Block = 1024 threads = 16 waves x 64 elements: wave w
// adds its contiguous share of the chunks in order, the 16 sums are then added in wave order (fixed
// association: deterministic).
constexpr int SUM_WAVES = 16;
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float *__restrict__ partial, int nchunk, int n,
                                                            float *__restrict__ out)
{
    __shared__ float red[SUM_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int q = (nchunk + SUM_WAVES - 1) / SUM_WAVES;
    const int c0 = w * q, c1 = c0 + q < nchunk ? c0 + q : nchunk;
    float s = 0.0f;
    if (e < n) {
        int ch = c0;
        for (; ch + 8 <= c1; ch += 8) {  // eight chunks in flight, added in chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(ch + u) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        if (ch < c1) {  // the last, partial batch: loads in flight together, past-the-end chunks masked
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(ch + u < c1 ? ch + u : ch) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += ch + u < c1 ? v[u] : 0.0f;
        }
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < n) {
        float t = red[0][lane];
#pragma unroll
        for (int i = 1; i < SUM_WAVES; ++i) t += red[i][lane];
        out[e] = t;
    }
}

static bool rowgemm_force_f32()  // experiments / tests: the fp32-MFMA form of the backward products
{
    static const int v = getenv("DGCNN_ROWGEMM_F32") ? atoi(getenv("DGCNN_ROWGEMM_F32")) : 0;
    return v != 0;
}
// The FORWARD product stays on the fp32 MFMA: TP and TQ decide which edge wins max-over-k, and at 2^-22 instead of 2^-24
// the winners of near-ties (gaps of a few 1e-6 after BatchNorm and the second convolution) move away from the
// reference's -- measured: 16.4 us against 21.1, and one gradient-parity case of the two-stage block lost.  The switch
// keeps the experiment runnable.
static bool rowgemm_split_fwd()
{
    static const int v = getenv("DGCNN_ROWGEMM_SPLIT_FWD") ? atoi(getenv("DGCNN_ROWGEMM_SPLIT_FWD")) : 0;
    return v != 0;
}

bool rowgemm_fwd_carries_count(int C, int N, int k)
{
    const bool off = getenv("DGCNN_NO_FWD_LISTS") != nullptr;   // experiments / tests (read per call: a test flips it): the backward builds its lists itself
    return !off && !(C == 64 && rowgemm_split_fwd()) && N <= RG_LDS_FLOATS && k <= 64;
}

void launch_rowgemm_fwd(const float *x, int C, const float *w1, float *TP, float *TQ, const FwdSide &sd, int P, hipStream_t st,
                        int ldx)
{
    const dim3 grid(ceil_div(ceil_div(P, 32), 4)), block(256);
    const RowLd ld{ldx > 0 ? ldx : C, EC, 0};
    if (C == 64 && rowgemm_split_fwd())
        hipLaunchKernelGGL((rowgemm_split_kernel<32, 4, true>), grid, block, 0, st, x, C, nullptr, 0, w1, TP, EC, TQ, P, sd, ld, Dw1Rider{});
    else {
        // (sd.rc: the counting pass of the backward's reverse lists as workgroups behind the product's, see RevCountRider)
        FwdSide s2 = sd;
        s2.rc.main_blocks = (int)grid.x;
        const dim3 g2(grid.x + (sd.rc.cnt ? sd.rc.nblocks : 0));
        if (C == 64)
            hipLaunchKernelGGL((rowgemm_fast_kernel<32, 4, true>), g2, block, 0, st, x, C, nullptr, 0, w1, TP, EC, TQ, P, s2, ld);
        else
            hipLaunchKernelGGL((rowgemm_kernel<4, true>), g2, block, 0, st, x, C, nullptr, 0, w1, 128, TP, EC, TQ, P, s2, ld);
    }
}

void launch_rowgemm(const float *A0, int K0, const float *A1, int K1, const float *W, int NO, float *Y0, int NO0,
                    float *Y1, int P, hipStream_t st, int ldy0, int y0_accumulate)
{
    const dim3 grid(ceil_div(ceil_div(P, 32), 4)), block(256);
    const int nt = ceil_div(NO, 32);
    const int K = K0 + K1, NO1 = NO - NO0;
    const RowLd ld{K0, ldy0 > 0 ? ldy0 : NO0, y0_accumulate};
    const bool runs_ok = (K1 == 0 || K0 == K / 2) && (K0 & 3) == 0 && (K1 & 3) == 0;
    if (runs_ok && NO0 % 32 == 0 && NO1 % 32 == 0 && (NO1 == 0 || Y1)) {
        if (!rowgemm_force_f32() && ((K == 64 && NO == 128) || (K == 128 && NO == 64))) {
            if (K == 64)
                hipLaunchKernelGGL((rowgemm_split_kernel<32, 4, false>), grid, block, 0, st, A0, K0, A1, K1, W, Y0, NO0, Y1, P,
                                   FwdSide{}, ld, Dw1Rider{});
            else
                hipLaunchKernelGGL((rowgemm_split_kernel<64, 2, false>), grid, block, 0, st, A0, K0, A1, K1, W, Y0, NO0, Y1, P,
                                   FwdSide{}, ld, Dw1Rider{});
            return;
        }
        if (K == 64 && NO == 128) {
            hipLaunchKernelGGL((rowgemm_fast_kernel<32, 4, false>), grid, block, 0, st, A0, K0, A1, K1, W, Y0, NO0, Y1, P,
                               FwdSide{}, ld);
            return;
        }
        if (K == 128 && NO == 64) {
            hipLaunchKernelGGL((rowgemm_fast_kernel<64, 2, false>), grid, block, 0, st, A0, K0, A1, K1, W, Y0, NO0, Y1, P,
                               FwdSide{}, ld);
            return;
        }
    }
    if (nt <= 1)
        hipLaunchKernelGGL((rowgemm_kernel<1, false>), grid, block, 0, st, A0, K0, A1, K1, W, NO, Y0, NO0, Y1, P, FwdSide{}, ld);
    else if (nt == 2)
        hipLaunchKernelGGL((rowgemm_kernel<2, false>), grid, block, 0, st, A0, K0, A1, K1, W, NO, Y0, NO0, Y1, P, FwdSide{}, ld);
    else
        hipLaunchKernelGGL((rowgemm_kernel<4, false>), grid, block, 0, st, A0, K0, A1, K1, W, NO, Y0, NO0, Y1, P, FwdSide{}, ld);
}

// wgrad_kernel for M0 = M1 = 64, C = 64 (the 64-channel blocks): a workgroup of FOUR waves takes 256 points, wave w
// the 32 output rows w of [A0 | A1]^T X (two accumulator tiles), so a workgroup leaves ONE 128 x 64 partial per 256
// points.  (One wave per 64 points, the first form, wrote 28 MB of partials at P = 56 000 -- more than the 43 MB the
// product reads -- and finish_dw1_kernel read them back: 27 + 11 us.)
__global__ __launch_bounds__(256) void wgrad_fast64_kernel(const float *__restrict__ A0, const float *__restrict__ A1,
                                                           const float *__restrict__ X, int ldx, int P, float *__restrict__ partial)
{
    constexpr int UN = 4, DEPTH = 4;  // point pairs per step; steps in the register ring (three steps of 8 MFMAs ahead: 22 us;
                                      // a ring of two steps of 16 measured 32 - 38 us)
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    const int pbeg = blockIdx.x * WGRAD_FAST_POINTS;
    const int pend = pbeg + WGRAD_FAST_POINTS < P ? pbeg + WGRAD_FAST_POINTS : P;
    const float *A = (mt < 2 ? A0 : A1) + (mt & 1) * 32;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    float a[DEPTH][UN], b[DEPTH][UN][2];
    auto fetch = [&](int p, float(&fa)[UN], float(&fb)[UN][2]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int pp = p + 2 * u + h;
            const bool pv = pp < pend;
            const size_t ro = (size_t)(pv ? pp : pbeg) * 64 + i32;  // clamped row: unconditional loads, masked values
            const size_t rx = (size_t)(pv ? pp : pbeg) * (size_t)ldx + i32;
            const float a0 = A[ro], b0 = X[rx], b1 = X[rx + 32];
            fa[u] = pv ? a0 : 0.0f;
            fb[u][0] = pv ? b0 : 0.0f;
            fb[u][1] = pv ? b1 : 0.0f;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) fetch(pbeg + d * 2 * UN, a[d], b[d]);
    static_assert(WGRAD_FAST_POINTS % (DEPTH * 2 * UN) == 0, "whole rounds of DEPTH steps");  // ring slots static
    for (int p = pbeg; p < pend; p += DEPTH * 2 * UN) {
        static_for<DEPTH>([&](auto D) {
            constexpr int d = D.value, dn = (d + DEPTH - 1) % DEPTH;
            fetch(p + (d + DEPTH - 1) * 2 * UN, a[dn], b[dn]);  // past the chunk end: all rows masked
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][u], b[d][u][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][u], b[d][u][1], acc[1], 0, 0, 0);
            }
        });
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            partial[((size_t)blockIdx.x * 128 + mt * 32 + mfma32_row(r, h)) * 64 + nt * 32 + i32] = acc[nt][r];
}

static bool wgrad_force_f32()  // experiments / tests: the fp32-MFMA form of the first convolution's weight gradient
{
    static const int v = getenv("DGCNN_WGRAD1_F32") ? atoi(getenv("DGCNN_WGRAD1_F32")) : 0;
    return v != 0;
}

// partials of launch_wgrad (the caller sizes the buffer and passes the count to launch_finish_dw1)
int wgrad_chunks(int P, int C) { return ceil_div(P, C == 64 ? WGRAD_FAST_POINTS : WGRAD_CHUNK); }

void launch_wgrad(const float *A0, int M0, const float *A1, int M1, const float *X, int C, int P, float *partial,
                  hipStream_t st, int ldx)
{
    // (M0 = M1 = 64 always: the stage widths of the model; C = 64 takes the four-wave kernel)
    const dim3 grid(wgrad_chunks(P, C)), block(64);
    if (ldx <= 0) ldx = C;
    if (C == 64) {
        if (wgrad_force_f32()) hipLaunchKernelGGL(wgrad_fast64_kernel, grid, dim3(256), 0, st, A0, A1, X, ldx, P, partial);
        else hipLaunchKernelGGL(wgrad_split64_kernel, grid, dim3(256), 0, st, A0, A1, X, ldx, P, partial);
        return;
    }
    const int mt = ceil_div(M0 + M1, 32), nt = ceil_div(C, 32);
#define DG_WG(MT_, NT_) \
    hipLaunchKernelGGL((wgrad_kernel<MT_, NT_>), grid, block, 0, st, A0, M0, A1, M1, X, ldx, C, P, partial)
    // C <= 64 is enforced by the callers (NT <= 2)
    if (mt <= 2) {
        if (nt <= 1) DG_WG(2, 1);
        else DG_WG(2, 2);
    } else {
        if (nt <= 1) DG_WG(4, 1);
        else DG_WG(4, 2);
    }
#undef DG_WG
}

// the input gradient of a block's first convolution, gx = [dR | dD] . Wcat (launch_rowgemm), and the fold of its weight
// gradient (launch_finish_dw1) in ONE launch where the shapes allow (the 64-channel blocks): the fold's 128 workgroups ride
// behind the product's
void launch_dgrad_and_dw1(const float *dR, const float *dD, const float *wcat, int C, float *gx, int ldgx, int gx_acc, int P,
                          const float *partial, int nchunk, float *gw1, hipStream_t st, int *zero_words, int nzero)
{
    if (C == 64 && !rowgemm_force_f32()) {
        const int main_blocks = ceil_div(ceil_div(P, 32), 4);
        const Dw1Rider rd{partial, nchunk, C, gw1, main_blocks, zero_words, nzero};
        const RowLd ld{EC, ldgx > 0 ? ldgx : C, gx_acc};
        hipLaunchKernelGGL((rowgemm_split_kernel<64, 2, false>), dim3(main_blocks + ceil_div(EC * 2 * C, 64)), dim3(256), 0, st, dR, EC,
                           dD, EC, wcat, gx, C, nullptr, P, FwdSide{}, ld, rd);
        return;
    }
    launch_finish_dw1(partial, nchunk, C, gw1, st, zero_words, nzero);
    launch_rowgemm(dR, EC, dD, EC, wcat, C, gx, C, nullptr, P, st, ldgx, gx_acc);
}

// the weight gradient and the input gradient of a 64-channel block's first convolution in ONE launch (the weight gradient's
// workgroups ride behind the input gradient's), then the fold of the weight gradient's partials
void launch_dgrad_wgrad64(const float *dR, const float *dD, const float *wcat, float *gx, int ldgx, int gx_acc, int P, const float *x,
                          int ldx, float *partial, float *gw1, hipStream_t st, int *zero_words, int nzero)
{
    const int C = 64, main_blocks = ceil_div(ceil_div(P, 32), 4), nchunk = wgrad_chunks(P, C);
    Dw1Rider rd{};
    rd.main_blocks = main_blocks;
    rd.wg_x = x, rd.wg_ldx = ldx > 0 ? ldx : C, rd.wg_partial = partial;
    const RowLd ld{EC, ldgx > 0 ? ldgx : C, gx_acc};
    hipLaunchKernelGGL((rowgemm_split_kernel<64, 2, false>), dim3(main_blocks + nchunk), dim3(256), 0, st, dR, EC, dD, EC, wcat, gx, C,
                       nullptr, P, FwdSide{}, ld, rd);
    launch_finish_dw1(partial, nchunk, C, gw1, st, zero_words, nzero);
}
bool dgrad_wgrad64_fused(int C)
{
    static const bool off = getenv("DGCNN_NO_DGRAD_WGRAD_FUSION") != nullptr;   // experiments
    return C == 64 && !off && !rowgemm_force_f32() && !wgrad_force_f32();
}

void launch_finish_dw1(const float *partial, int nchunk, int C, float *gw1, hipStream_t st, int *zero_words, int nzero)
{
    hipLaunchKernelGGL(finish_dw1_kernel, dim3(ceil_div(EC * 2 * C, 64)), dim3(1024), 0, st, partial, nchunk, C, gw1, zero_words, nzero);
}

void launch_sum_partials(const float *partial, int nchunk, int n, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(n, 64)), dim3(1024), 0, st, partial, nchunk, n, out);
}

}  // namespace dgcnn
