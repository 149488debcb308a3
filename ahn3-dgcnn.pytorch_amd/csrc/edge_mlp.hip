// edge_mlp.hip -- the per-EDGE 64 -> 64 contraction of the two-stage EdgeConv blocks
// (conv2 / conv4 of the reference, model.py:119-121,125-127 applied at :149,:156), fused with everything around it.
// Forward and backward run on the fp16 matrix cores (edge_mlp_fwd16.hip, edge_mlp_bwd16.hip): both operands scaled by an
// exact power of two and split in two fp16 halves, x' = hi + lo, u2 = hi.hi + hi.lo + lo.hi; the dropped lo.lo term and the
// rounding of lo are 2^-22 relative -- inside fp32's own accumulation error over 64 terms and far inside the 1e-4
// parity bound -- for 12 v_mfma_f32_32x32x16_f16 (384 cycles) per 32 x 32 x 64 tile instead of 32
// v_mfma_f32_32x32x2_f32 (2048 cycles).  On gfx950 the fp32-input MFMA also shares the vector ALU's FMA lanes: its time
// and the gather / epilogue VALU time ADD (measured: 114 us MFMA + 56 + 27 us VALU phases = the fp32 kernel's 206 us).
// This file keeps the fp32 kernels (v_mfma_f32_32x32x2_f32), which are the guarded fallbacks, and the launchers.
//
// One wave processes one point at a time; its k edges are the M dimension of a 32-row MFMA tile
// (k = 30 -> 30 of 32 rows used; k > 32 -> two tiles):
//   gather  (lane = channel)  u1[j] = TP[n(i, j)] + TQ[i];  a1 = LReLU(BN1(u1)) -> wave-private LDS tile
//   MFMA    u2[32 edges][64]  = a1 . W2^T          A from the wave's LDS tile, B = W2 from the workgroup's LDS image
//   forward epilogue: accumulator layout has the output CHANNEL on the lane and the EDGES in the
//           16 registers, so BN2 batch sums and the max/min over k are in-lane loops plus one
//           exchange between the two lane halves; nothing per-edge is written to HBM.
//   backward: du2 from the recomputed u2; gW2 += du2^T a1 uses the du2 accumulators directly as
//           the A operand (contraction over the edge rows held in the registers); da1 = du2 W2
//           goes through one LDS transpose; dv1 = da1 * LReLU'(v1) is written to HBM ([E][64])
//           for the stage-1 streaming pass, together with the BN1 backward sums.
#include <stdlib.h>

#include "edge_kernels.h"
#include "edge_mlp_common.h"
#include "fold.h"
#include "prof.h"

namespace dgcnn {

// workgroups: the backward holds 156 KB of LDS (one workgroup of eight waves per CU); the forward 39 KB and about 220
// registers per lane: two workgroups per CU (64 slots per XCD: measured 1.25 ms per step against 1.28 at 96, 1.39 at 128)
static int mlp_slots_fwd()
{
    static const int v = getenv("DGCNN_MLP_SLOTS") ? atoi(getenv("DGCNN_MLP_SLOTS")) : 64;  // experiments
    return v;
}
int mlp_grid(int P) { return point_grid(P, MLP_BWD_WAVES, 32); }
int mlp_grid_fwd(int P) { return point_grid(P, MLP_WAVES, mlp_slots_fwd()); }

// A operand of the 32x32x2 MFMA from a [32][TSTR] tile: lane (i, h) holds row i, columns 32 h + s
__device__ __forceinline__ void load_a_operand(const float (*tile)[TSTR], int i32, int h, float (&a)[32])
{
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
        const float4 v = *reinterpret_cast<const float4 *>(&tile[i32][32 * h + 4 * s4]);
        a[4 * s4 + 0] = v.x;
        a[4 * s4 + 1] = v.y;
        a[4 * s4 + 2] = v.z;
        a[4 * s4 + 3] = v.w;
    }
}

// ---------------------------------------------------------------------------------------------
// The fp32 form of the forward (v_mfma_f32_32x32x2_f32): the FALLBACK of edge_mlp_fwd16_kernel (edge_mlp_fwd16.hip).
// Launched behind it, it returns at once unless that kernel has raised *guard (|a1| beyond the fp16 range, or a
// non-finite sum), and otherwise redoes the pass: usel2, hit2, BatchNorm-2 statistics.
__global__ __launch_bounds__(256) void edge_mlp_fwd_f32_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                               const int32_t *__restrict__ idx, int N, int k, int P,
                                                               const float *__restrict__ stats1,
                                                               const float *__restrict__ w2,
                                                               const float *__restrict__ gamma2, float slope,
                                                               float *__restrict__ usel2, uint8_t *__restrict__ hit2,
                                                               int fin_on, FoldSet fs, BnBatchStats fin, const int *__restrict__ guard)
{
    if (*guard == 0) return;
    __shared__ __attribute__((aligned(16))) float tiles[MLP_WAVES][32][TSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    float(*tileA)[TSTR] = tiles[wave];

    unsigned flip[2];  // gamma2 < 0: the activation's maximum is the pre-activation's minimum
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) flip[nt] = gamma2[nt * 32 + i32] < 0.0f ? 0x80000000u : 0u;
    const float sc1 = stats1[ST_SC + lane], sh1 = stats1[ST_SH + lane];

    double sum[2] = {0.0, 0.0}, sq[2] = {0.0, 0.0};
    const PointIter it = point_iter(P, wave, MLP_WAVES);
    // (prefetching the next point's rows as the backward once did costs 32 registers and one of the two resident
    // waves per SIMD: measured 183 us against 150)
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const int b = pt / N;
        const float *Pb = TP + (size_t)b * N * EC;
        const float q = TQ[(size_t)pt * EC + lane];
        const int32_t *ip = idx + (size_t)pt * k;
        float ext[2], ps[2] = {0.0f, 0.0f}, pq[2] = {0.0f, 0.0f};  // fp32 over the <= 32 values of a lane, fp64 across points
        int arg[2] = {0, 0};  // edge slot of the extreme (the first one on ties): the backward routes dL/dy there
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ext[nt] = -__builtin_inff();  // of the sign-flipped value
        for (int j0 = 0; j0 < k; j0 += 32) {
            const int kk = k - j0 < 32 ? k - j0 : 32;
            {
                // eight rows at a time (registers: the kernel has to stay at three waves per SIMD)
                const int nv = lane < kk ? ip[j0 + lane] : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int tt = g * 8 + t < kk ? g * 8 + t : kk - 1;
                        p[t] = Pb[(unsigned)__builtin_amdgcn_readlane(nv, tt) * (unsigned)EC + (unsigned)lane];
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const bool on = g * 8 + t < kk;
                        const float u = p[t] + q;
                        const float a = on ? lrelu(__builtin_fmaf(u, sc1, sh1), slope) : 0.0f;
                        tileA[g * 8 + t][lane] = a;
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 am;
#pragma unroll
                for (int r = 0; r < 16; ++r) am[r] = 0.0f;
                {
                    float a32[32];
                    load_a_operand(tileA, i32, h, a32);
#pragma unroll
                    for (int s = 0; s < 32; ++s)
                        am = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[s], w2[(size_t)(nt * 32 + i32) * EC + 32 * h + s], am, 0, 0, 0);
                }
                // extreme of the sign-flipped value (so that it is always a maximum), then the first slot that holds it
                float best = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float u = am[r];  // exactly 0 in the padding rows (a1 = 0)
                    ps[nt] += u;
                    pq[nt] = __builtin_fmaf(u, u, pq[nt]);
                    am[r] = mfma32_row(r, h) < kk ? __uint_as_float(__float_as_uint(u) ^ flip[nt]) : -__builtin_inff();
                    best = fmaxf(best, am[r]);
                }
                if (best > ext[nt]) {
                    int slot = 0;
#pragma unroll
                    for (int r = 15; r >= 0; --r) slot = am[r] == best ? mfma32_row(r, h) : slot;
                    ext[nt] = best;
                    arg[nt] = j0 + slot;
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            sum[nt] += (double)ps[nt];
            sq[nt] += (double)pq[nt];
            const float o = __shfl_xor(ext[nt], 32);
            const int oa = __shfl_xor(arg[nt], 32);
            const bool take = o > ext[nt] || (o == ext[nt] && oa < arg[nt]);
            if (h == 0) {
                usel2[(size_t)pt * EC + nt * 32 + i32] = __uint_as_float(__float_as_uint(take ? o : ext[nt]) ^ flip[nt]);
                hit2[(size_t)pt * EC + nt * 32 + i32] = (uint8_t)(take ? oa : arg[nt]);
            }
        }
    }
    // BN2 batch statistics (training): lane (n, h) holds channel nt * 32 + n; fold the halves -- lane l then carries
    // channel l (nt = h) -- and let the pass finish the sums itself (fold.h)
    if (fin_on) {
        const double t0 = sum[0] + shfl_xor_d(sum[0], 32), t1 = sum[1] + shfl_xor_d(sum[1], 32);
        const double q0 = sq[0] + shfl_xor_d(sq[0], 32), q1 = sq[1] + shfl_xor_d(sq[1], 32);
        const double v[2] = {h ? t1 : t0, h ? q1 : q0};
        fold_and_finalize<2, MLP_WAVES>(v, fs, fin);
    }
}

// ---------------------------------------------------------------------------------------------
// The fp32 form of the backward (v_mfma_f32_32x32x2_f32 throughout).  It is the FALLBACK of edge_mlp_bwd16_kernel
// below: launched behind it, it returns at once unless that kernel has raised *guard (an operand left the fp16 range).
__global__ __launch_bounds__(512) void edge_mlp_bwd_f32_kernel(
    const float *__restrict__ TP, const float *__restrict__ TQ, const int32_t *__restrict__ idx, int N, int k, int P,
    const float *__restrict__ stats1, const float *__restrict__ stats2, const float *__restrict__ cc2,
    const float *__restrict__ w2, float slope, int training, const float *__restrict__ gdv,
    const uint8_t *__restrict__ hit2, float *__restrict__ dv1,
    float *__restrict__ dvsum, float *__restrict__ partial_dw2, FoldSet fs, BnBackwardSums fin1,
    const int *__restrict__ guard)
{
    if (guard && *guard == 0) return;
    // W2 [co][ci] once, rows padded to 65 floats, serves as the B operand of both products that contract over a
    // channel index without bank conflicts:
    //   u2  = a1 . W2^T :  B[k = ci = 32 h + s][n = co] = w2s[nt 32 + n][32 h + s]   (lanes n: stride 65)
    //   da1 = du2 . W2  :  B[k = co = 32 h + s][n = ci] = w2s[32 h + s][nt 32 + n]   (lanes n: consecutive)
    // Two tiles per wave: du2 overwrites a1 once a1 sits in registers (A operand and accumulator layout).
    __shared__ float w2s[64][65];
    __shared__ __attribute__((aligned(16))) float tiles[MLP_BWD_WAVES][2][32][TSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    float(*tileA)[TSTR] = tiles[wave][0];  // a1 [edge][ci], then du2 [edge][co]
    float(*tileU)[TSTR] = tiles[wave][1];  // u1 [edge][ci]
    float(*tileD)[TSTR] = tileA;

    for (int t = threadIdx.x; t < 4096; t += 512) w2s[t >> 6][t & 63] = w2[t];
    __syncthreads();

    // gather-phase constants (lane = channel) and accumulator-layout constants (lane (n, h) = channel nt 32 + n)
    const float sc1g = stats1[ST_SC + lane], sh1g = stats1[ST_SH + lane];
    // du2 = cA + cB u2 + (dL/dy at the winning edge): the BatchNorm mean terms are affine in u2 (zero in eval mode)
    float mean1[2], rstd1[2], sc2[2], cA[2], cB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = nt * 32 + i32;
        mean1[nt] = stats1[ST_MEAN + c];
        rstd1[nt] = stats1[ST_RSTD + c];
        sc2[nt] = stats2[ST_SC + c];
        cB[nt] = training ? -sc2[nt] * cc2[64 + c] * stats2[ST_RSTD + c] : 0.0f;
        cA[nt] = training ? -sc2[nt] * cc2[c] - cB[nt] * stats2[ST_MEAN + c] : 0.0f;
    }

    f32x16 accW[2][2];  // gW2 tile [co tile][ci tile]
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[a_][b_][r] = 0.0f;
    double sb[2] = {0.0, 0.0}, sg[2] = {0.0, 0.0};  // BN1: sum dv1, sum dv1 * uhat1

    const PointIter it = point_iter(P, wave, MLP_BWD_WAVES);
    // index row one point ahead; the other wave of the SIMD covers the latency of the table rows
    const int kk0 = k < 32 ? k : 32;
    int nvn = it.cur < it.end && lane < kk0 ? idx[(size_t)it.cur * k + lane] : 0;
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const int b = pt / N;
        const float *Pb = TP + (size_t)b * N * EC;
        const float q = TQ[(size_t)pt * EC + lane];
        const int32_t *ip = idx + (size_t)pt * k;
        float dvs[2];
        int hit[2];  // the forward's winning edge slot of (point, channel)
        float dcol[2] = {0.0f, 0.0f};  // sum over the point's edges of dv1, per channel (for the centre term dD)
        float dhat[2] = {0.0f, 0.0f};  // ... and of dv1 * uhat1 (fp32 over the <= 32 values of a lane, fp64 across points)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            hit[nt] = hit2[(size_t)pt * EC + nt * 32 + i32];
            dvs[nt] = sc2[nt] * gdv[(size_t)pt * EC + nt * 32 + i32];  // gdv = dL/dy . LReLU'(y) (bwd_point_kernel)
        }
        for (int j0 = 0; j0 < k; j0 += 32) {
            const int kk = k - j0 < 32 ? k - j0 : 32;
            {
                float p2[32];
                const int nv = j0 == 0 ? nvn : (lane < kk ? ip[j0 + lane] : 0);
                gather_rows(Pb, nv, kk, lane, p2);
                if (j0 == 0 && pt + it.step < it.end) nvn = lane < kk0 ? idx[(size_t)(pt + it.step) * k + lane] : 0;
                stage_tile<true>(p2, q, kk, sc1g, sh1g, slope, tileA, tileU, lane);
            }

            // ---- u2 = a1 . W2^T
            float a[32];
            load_a_operand(tileA, i32, h, a);
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2s[i32][32 * h + s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2s[32 + i32][32 * h + s], acc[1], 0, 0, 0);
            }
            // a1 in the accumulator layout (B operand of the weight gradient, LReLU' mask of dv1) before du2 overwrites it
            float ba[2][16];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) ba[ct][r] = tileA[mfma32_row(r, h)][ct * 32 + i32];

            // ---- du2 (in place of u2): the whole dL/dy goes to the edge the forward selected (hit2); the rest of du2
            // are the dense BatchNorm mean terms, affine in u2 (recomputed here in fp32: it differs from the forward's
            // split-fp16 value by ~1e-6 relative, which only touches these small dense terms).
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int hitrow = hit[nt] - j0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma32_row(r, h);
                    float du = 0.0f;
                    if (row < kk) du = __builtin_fmaf(cB[nt], acc[nt][r], cA[nt]) + (row == hitrow ? dvs[nt] : 0.0f);
                    acc[nt][r] = du;
                    tileD[row][nt * 32 + i32] = du;
                }
            }

            // ---- gW2[co][ci] += sum_edges du2[edge][co] a1[edge][ci]: the du2 accumulators are the
            // A operand as they stand (k = edge row of register r), a1 (read above) is B.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int cot = 0; cot < 2; ++cot)
#pragma unroll
                    for (int cit = 0; cit < 2; ++cit)
                        accW[cot][cit] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(acc[cot][r], ba[cit][r], accW[cot][cit], 0, 0, 0);
            }

            // ---- da1 = du2 . W2 (A operand through the LDS transpose), dv1 = da1 * LReLU'(v1)
            load_a_operand(tileD, i32, h, a);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2s[32 * h + s][i32], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2s[32 * h + s][32 + i32], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma32_row(r, h);
                    if (row < kk) {
                        const float d = acc[nt][r] * (ba[nt][r] > 0.0f ? 1.0f : slope);
                        const float uhat = (tileU[row][nt * 32 + i32] - mean1[nt]) * rstd1[nt];
                        __builtin_nontemporal_store(d, &dv1[((size_t)pt * k + j0 + row) * EC + nt * 32 + i32]);
                        dcol[nt] += d;
                        dhat[nt] = __builtin_fmaf(d, uhat, dhat[nt]);
                    }
                }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            sb[nt] += (double)dcol[nt];
            sg[nt] += (double)dhat[nt];
            const float t = dcol[nt] + __shfl_xor(dcol[nt], 32);
            if (h == 0) dvsum[(size_t)pt * EC + nt * 32 + i32] = t;
        }
    }

    // ---- workgroup reduction of gW2 and of the BN1 sums (tile memory is free now)
    __syncthreads();
    float *scr = &tiles[0][0][0][0];  // 8 waves x 4096 floats fit in the tile area
#pragma unroll
    for (int cot = 0; cot < 2; ++cot)
#pragma unroll
        for (int cit = 0; cit < 2; ++cit)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[wave * 4096 + (cot * 32 + mfma32_row(r, h)) * 64 + cit * 32 + i32] = accW[cot][cit][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 512) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < MLP_BWD_WAVES; ++w) t += scr[w * 4096 + e];
        partial_dw2[(size_t)blockIdx.x * 4096 + e] = t;
    }
    __syncthreads();
    // BN1 backward sums: fold the lane halves (lane l then carries channel l), finish them in this launch (fold.h; the
    // tile area is free and serves as its scratch)
    {
        const double t0 = sb[0] + shfl_xor_d(sb[0], 32), t1 = sb[1] + shfl_xor_d(sb[1], 32);
        const double g0 = sg[0] + shfl_xor_d(sg[0], 32), g1 = sg[1] + shfl_xor_d(sg[1], 32);
        const double v[2] = {h ? t1 : t0, h ? g1 : g0};
        double(*red)[2][64] = reinterpret_cast<double(*)[2][64]>(scr);
        fold_and_finalize<2, MLP_BWD_WAVES>(v, fs, fin1, red, reinterpret_cast<int *>(scr + MLP_BWD_WAVES * 2 * 64 * 2));
    }
}


// ---------------------------------------------------------------------------------------------
void launch_edge_mlp_fwd16(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats1,
                           const float *w2, const float *gamma2, float slope, float *usel2, uint8_t *hit2, int fin_on,
                           const FoldSet &fs, const BnBatchStats &fin, int *guard, hipStream_t st);  // edge_mlp_fwd16.hip

void launch_edge_mlp_fwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                         const float *stats1, const float *w2, const float *gamma2, float slope, float *usel2,
                         uint8_t *hit2, const FoldSet &fs, const FoldSet &fs_fallback, const BnBatchStats *fin, int *guard,
                         hipStream_t st)
{
    BnBatchStats f16 = fin ? *fin : BnBatchStats{}, f32 = f16;
    f16.guard = guard;   // the fp16 kernel's finalizer stands down when the guard is up
    f32.guard = nullptr;
    prof_begin(PROF_EDGE_MLP_FWD, st);
    launch_edge_mlp_fwd16(TP, TQ, idx, N, k, P, stats1, w2, gamma2, slope, usel2, hit2, fin ? 1 : 0, fs, f16, guard, st);
    // the fp32 form of the same pass: returns at once unless the fp16 kernel raised *guard
    hipLaunchKernelGGL(edge_mlp_fwd_f32_kernel, dim3(mlp_grid_fwd(P)), dim3(256), 0, st, TP, TQ, idx, N, k, P, stats1, w2,
                       gamma2, slope, usel2, hit2, fin ? 1 : 0, fs_fallback, f32, (const int *)guard);
    prof_end(PROF_EDGE_MLP_FWD, st);
}

void launch_edge_mlp_bwd16(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats1,
                           const float *stats2, const float *cc2, const float *w2, float slope, int training,
                           const float *gdv, const uint8_t *hit2, const unsigned *gmax_bits, float *dv1,
                           float *dvsum, float *partial_dw2, const FoldSet &fs, const BnBackwardSums &fin1, int *guard,
                           int grid, hipStream_t st);  // edge_mlp_bwd16.hip

int launch_edge_mlp_bwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P,
                        const float *stats1, const float *stats2, const float *cc2, const float *w2, float slope,
                        int training, const float *gdv, const uint8_t *hit2, const unsigned *gmax_bits,
                        float *dv1, float *dvsum, float *partial_dw2, const FoldSet &fs, const FoldSet &fs_fallback,
                        const BnBackwardSums &fin1, int *guard, hipStream_t st)
{
    static const int force32 = getenv("DGCNN_MLP_BWD_F32") ? atoi(getenv("DGCNN_MLP_BWD_F32")) : 0;  // experiments / tests
    const dim3 grid(mlp_grid(P)), block(64 * MLP_BWD_WAVES);
    prof_begin(PROF_EDGE_MLP_BWD, st);
    if (force32) {
        hipLaunchKernelGGL(edge_mlp_bwd_f32_kernel, grid, block, 0, st, TP, TQ, idx, N, k, P, stats1, stats2, cc2, w2, slope,
                           training, gdv, hit2, dv1, dvsum, partial_dw2, fs, fin1, (const int *)nullptr);
    } else {
        launch_edge_mlp_bwd16(TP, TQ, idx, N, k, P, stats1, stats2, cc2, w2, slope, training, gdv, hit2, gmax_bits, dv1,
                              dvsum, partial_dw2, fs, fin1, guard, (int)grid.x, st);
        // the fp32 form of the same pass: returns at once unless the fp16 kernel raised *guard
        hipLaunchKernelGGL(edge_mlp_bwd_f32_kernel, grid, block, 0, st, TP, TQ, idx, N, k, P, stats1, stats2, cc2, w2, slope,
                           training, gdv, hit2, dv1, dvsum, partial_dw2, fs_fallback, fin1, (const int *)guard);
    }
    prof_end(PROF_EDGE_MLP_BWD, st);
    return 0;
}

}  // namespace dgcnn
