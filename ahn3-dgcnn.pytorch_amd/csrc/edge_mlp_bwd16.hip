// edge_mlp_bwd16.hip -- backward of the per-edge 64 -> 64 convolution on the fp16 matrix cores (see edge_mlp.hip for the
// pass as a whole; edge_mlp.hip keeps the forward, the fp32 form of this backward and the launchers).
#include "edge_kernels.h"
#include "edge_mlp_common.h"
#include "fold.h"

namespace dgcnn {

// ---------------------------------------------------------------------------------------------
// Backward on the fp16 matrix cores.  Same algorithm as the fp32 kernel above -- recompute u2 = a1 . W2^T, form du2,
// gW2 += du2^T a1, da1 = du2 . W2, dv1 = da1 * LReLU'(v1) -- with every operand scaled by an EXACT power of two into the
// fp16 range and split in two halves of the same scale, x' = hi + lo (hi = x' with the low 13 mantissa bits cleared:
// exactly a fp16 number; lo = x' - hi, at most 13 significant bits, rounded to fp16's 11): three
// v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi, fp32 accumulate, one accumulator) replace sixteen
// v_mfma_f32_32x32x2_f32: 72 matrix instructions of 32 cycles per point instead of 192 of 64.  Error of an operand
// element: 2^-23 relative, or 2^-25 absolute on the scaled value where lo runs into fp16's subnormals.  Scales:
//   a1' = 2^8 a1          (BatchNorm-1 output: |a1| < 253.9 is checked)
//   W2' = sW W2           sW from max |W2| (computed here)          -> max |W2'| in [2^11, 2^12)
//   du2' = sD du2         sD from max |dL/dy . LReLU'| (bwd_point_kernel leaves it in *gmax_bits) and max |sc2|
//                         -> the winners' entries stay below 2^10, the dense BatchNorm terms (|sc2| |dv|max (1 + |uhat2|))
//                         below 2^15.99 for |uhat2| < 62; max |du2'| is checked
// A value outside its range, or a non-finite partial sum, raises *guard: the fp32 kernel behind this one then redoes
// the pass (its results overwrite these).  Tiles hold the two halves of an element as ONE 32-bit word {hi | lo << 16}.
constexpr float MB_SA = 256.0f;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_split(float x)  // x already scaled
{
    const float hi = __uint_as_float(__float_as_uint(x) & 0xffffe000u);
    // v_cvt_pk_f16_f32: hi is exact, the remainder is rounded to nearest (no bias towards zero in the operands)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{hi, x - hi}, f16x2));
}
// eight packed words -> the hi halves and the lo halves as two MFMA operands (element j = word j)
__device__ __forceinline__ void words_to_operands(const unsigned (&w)[8], half8 &hi, half8 &lo)
{
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x05040100u);
        l[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);
    }
    hi = __builtin_bit_cast(half8, h);
    lo = __builtin_bit_cast(half8, l);
}
__device__ __forceinline__ void load_row_words(const unsigned (*tile)[TSTR], int row, int col, unsigned (&w)[8])
{
    const uint4 a = *reinterpret_cast<const uint4 *>(&tile[row][col]);
    const uint4 b = *reinterpret_cast<const uint4 *>(&tile[row][col + 4]);
    w[0] = a.x, w[1] = a.y, w[2] = a.z, w[3] = a.w, w[4] = b.x, w[5] = b.y, w[6] = b.z, w[7] = b.w;
}
__device__ __forceinline__ f32x16 mfma3(const half8 &ahi, const half8 &alo, const half8 &bhi, const half8 &blo, f32x16 c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, c, 0, 0, 0);
    return c;
}
// 2^e as a float, e in [-126, 127]
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
// the power of two s with  2^(top - 1) <= m s < 2^top  (m > 0; s = 1 for m == 0 or non-finite m)
__device__ __forceinline__ float scale_to(float m, int top)
{
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;  // floor(log2 m) for normal m
    if (!(m > 0.0f) || e > 126) return 1.0f;
    int s = top - 1 - e;
    s = s > 120 ? 120 : (s < -120 ? -120 : s);
    return pow2i(s);
}

// the k index of the contraction over the EDGES of a tile runs in accumulator order: MFMA step s (16 edges), lane
// half h, element j  <->  edge row mfma32_row(8 s + j, h) -- so that du2, which the u2 product leaves in accumulator
// registers (lane = channel, registers = edge rows), is the A operand of the weight gradient as it stands
__device__ __forceinline__ int edge_of(int s, int j, int h) { return mfma32_row(8 * s + j, h); }

__global__ __launch_bounds__(512) void edge_mlp_bwd16_kernel(
    const float *__restrict__ TP, const float *__restrict__ TQ, const int32_t *__restrict__ idx, int N, int k, int P,
    const float *__restrict__ stats1, const float *__restrict__ stats2, const float *__restrict__ cc2,
    const float *__restrict__ w2, float slope, int training, const float *__restrict__ gy,
    const float *__restrict__ y, const uint8_t *__restrict__ hit2, const unsigned *__restrict__ gmax_bits,
    float *__restrict__ dv1, float *__restrict__ dvsum, float *__restrict__ partial_dw2, FoldSet fs,
    BnBackwardSums fin1, int *__restrict__ guard)
{
    __shared__ __attribute__((aligned(16))) unsigned w2w[64][TSTR];  // {hi | lo} of sW W2 [co][ci]
    __shared__ __attribute__((aligned(16))) float tiles[MLP_BWD_WAVES][2][32][TSTR];
    __shared__ float s_red[MLP_BWD_WAVES];
    __shared__ float cst[4][64];  // per-channel constants of the last epilogue (rstd1, -mean1 rstd1, sc1, sh1): read there,
                                  // not held in eight registers through the three products (the kernel spills)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
    unsigned(*tileA)[TSTR] = reinterpret_cast<unsigned(*)[TSTR]>(tiles[wave][0]);  // a1' words, then du2' words
    float(*tileU)[TSTR] = tiles[wave][1];                                         // u1 [edge][ci]

    // ---- scales (the same in every workgroup: functions of the inputs only)
    float wmax = 0.0f;
    for (int t = threadIdx.x; t < 4096; t += 512) wmax = fmaxf(wmax, fabsf(w2[t]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_red[wave] = wmax;
    __syncthreads();
    wmax = s_red[0];
#pragma unroll
    for (int w = 1; w < MLP_BWD_WAVES; ++w) wmax = fmaxf(wmax, s_red[w]);
    const float sW = scale_to(wmax, 12);
    for (int t = threadIdx.x; t < 4096; t += 512) w2w[t >> 6][t & 63] = pack_split(w2[t] * sW);
    float scmax = fabsf(stats2[ST_SC + lane]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) scmax = fmaxf(scmax, __shfl_xor(scmax, o));
    const float sD = scale_to(__uint_as_float(*gmax_bits) * scmax, 10);
    if (threadIdx.x < 64) {
        const float r = stats1[ST_RSTD + threadIdx.x];
        cst[0][threadIdx.x] = r;
        cst[1][threadIdx.x] = -stats1[ST_MEAN + threadIdx.x] * r;  // uhat1 = fma(u, rstd, -mean rstd)
        cst[2][threadIdx.x] = stats1[ST_SC + threadIdx.x];
        cst[3][threadIdx.x] = stats1[ST_SH + threadIdx.x];
    }
    __syncthreads();

    // gather-phase constants (lane = channel) and accumulator-layout constants (lane (n, h) = channel nt 32 + n)
    // (LeakyReLU is positively homogeneous: a1' = 2^8 a1 = LReLU(2^8 sc1 u + 2^8 sh1))
    const float sc1g = stats1[ST_SC + lane] * MB_SA, sh1g = stats1[ST_SH + lane] * MB_SA;
    float sc2[2], cA[2], cB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = nt * 32 + i32;
        sc2[nt] = stats2[ST_SC + c];
        // du2' = sD (cA + cB u2 + [winner] dvs) with u2 = u2raw / (2^8 sW): the scales are folded into the constants
        const float cb = training ? -sc2[nt] * cc2[64 + c] * stats2[ST_RSTD + c] : 0.0f;
        const float ca = training ? -sc2[nt] * cc2[c] - cb * stats2[ST_MEAN + c] : 0.0f;
        cB[nt] = cb * sD / (MB_SA * sW);
        cA[nt] = ca * sD;
    }
    const float un_da1 = 1.0f / (sD * sW), un_da1s = un_da1 * slope;  // da1 = da1raw / (sD sW), times LReLU'(v1)

    f32x16 accW[2][2];  // gW2 tile [co tile][ci tile], scaled by sD 2^8
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[a_][b_][r] = 0.0f;
    double sb[2] = {0.0, 0.0}, sg[2] = {0.0, 0.0};  // BN1: sum dv1, sum dv1 * uhat1
    float amax = 0.0f, dmax = 0.0f;                 // range checks: max |a1|, max |du2'|

    const PointIter it = point_iter(P, wave, MLP_BWD_WAVES);
    const int kk0 = k < 32 ? k : 32;
    int nvn = it.cur < it.end && lane < kk0 ? idx[(size_t)it.cur * k + lane] : 0;
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const int b = pt / N;
        const float *Pb = TP + (size_t)b * N * EC;
        const float q = TQ[(size_t)pt * EC + lane];
        const int32_t *ip = idx + (size_t)pt * k;
        float dvs[2];
        int hit[2];
        float dcol[2] = {0.0f, 0.0f}, dhat[2] = {0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            hit[nt] = hit2[(size_t)pt * EC + nt * 32 + i32];
            dvs[nt] = sD * sc2[nt] * gy[(size_t)pt * EC + nt * 32 + i32] *
                      (y[(size_t)pt * EC + nt * 32 + i32] > 0.0f ? 1.0f : slope);
        }
        for (int j0 = 0; j0 < k; j0 += 32) {
            const int kk = k - j0 < 32 ? k - j0 : 32;
            {
                float p2[32];
                const int nv = j0 == 0 ? nvn : (lane < kk ? ip[j0 + lane] : 0);
                gather_rows(Pb, nv, kk, lane, p2);
                if (j0 == 0 && pt + it.step < it.end) nvn = lane < kk0 ? idx[(size_t)(pt + it.step) * k + lane] : 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const bool on = j < kk;
                    const float u = on ? p2[j] + q : 0.0f;
                    const float a = on ? lrelu(__builtin_fmaf(u, sc1g, sh1g), slope) : 0.0f;  // a1'
                    amax = fmaxf(amax, fabsf(a));
                    tileA[j][lane] = pack_split(a);
                    tileU[j][lane] = u;
                }
            }

            // ---- u2raw = a1' . W2'^T : A = rows of the a1' tile, B[k = ci][n = co] = rows of w2w
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                unsigned w[8];
                half8 ahi, alo, bhi, blo;
                load_row_words(tileA, i32, 16 * s + 8 * h, w);
                words_to_operands(w, ahi, alo);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    load_row_words(w2w, nt * 32 + i32, 16 * s + 8 * h, w);
                    words_to_operands(w, bhi, blo);
                    acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
                }
            }
            // a1' as the B operand of the weight gradient: B[k = edge][n = ci] in accumulator edge order (read before
            // du2' overwrites the tile)
            half8 xhi[2][2], xlo[2][2];  // [ci tile][step]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    unsigned w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = tileA[edge_of(s, j, h)][ct * 32 + i32];
                    words_to_operands(w, xhi[ct][s], xlo[ct][s]);
                }

            // ---- du2' (in place of u2raw): the whole dL/dy goes to the edge the forward selected (hit2); the rest are
            // the dense BatchNorm mean terms, affine in u2.  Split once: packed words for the LDS transpose (A operand of
            // da1), and the halves of registers 8 s .. 8 s + 7 as the A operand of the weight gradient.
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int hitrow = hit[nt] - j0;
                unsigned w[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma32_row(r, h);
                    float du = 0.0f;
                    if (row < kk) du = __builtin_fmaf(cB[nt], acc[nt][r], cA[nt]) + (row == hitrow ? dvs[nt] : 0.0f);
                    dmax = fmaxf(dmax, fabsf(du));
                    w[r] = pack_split(du);
                    tileA[row][nt * 32 + i32] = w[r];
                }
                // ---- gW2[co][ci] += sum_edges du2'[edge][co] a1'[edge][ci]   (this co tile)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    unsigned ws[8];
                    half8 dhi, dlo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ws[j] = w[8 * s + j];
                    words_to_operands(ws, dhi, dlo);
#pragma unroll
                    for (int cit = 0; cit < 2; ++cit)
                        accW[nt][cit] = mfma3(dhi, dlo, xhi[cit][s], xlo[cit][s], accW[nt][cit]);
                }
            }

            // ---- da1raw = du2' . W2' : A = rows of the du2' tile (the LDS transpose), B[k = co][n = ci] = columns of w2w
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                unsigned w[8];
                half8 ahi, alo, bhi, blo;
                load_row_words(tileA, i32, 16 * s + 8 * h, w);
                words_to_operands(w, ahi, alo);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = w2w[16 * s + 8 * h + j][nt * 32 + i32];
                    words_to_operands(w, bhi, blo);
                    acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
                }
            }
            // ---- dv1 = da1 * LReLU'(v1), BN1 sums
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float rstd1_ = cst[0][nt * 32 + i32], nmr1_ = cst[1][nt * 32 + i32];
                const float sc1_ = cst[2][nt * 32 + i32], sh1_ = cst[3][nt * 32 + i32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma32_row(r, h);
                    if (row < kk) {
                        const float u = tileU[row][nt * 32 + i32];
                        const float v = __builtin_fmaf(u, sc1_, sh1_);
                        const float d = acc[nt][r] * (v > 0.0f ? un_da1 : un_da1s);
                        const float uhat = __builtin_fmaf(u, rstd1_, nmr1_);
                        // (streaming: 430 MB of dv1 must not sweep the TP table out of the XCD's L2: 395 -> 351 us)
                        __builtin_nontemporal_store(d, &dv1[((size_t)pt * k + j0 + row) * EC + nt * 32 + i32]);
                        dcol[nt] += d;
                        dhat[nt] = __builtin_fmaf(d, uhat, dhat[nt]);
                    }
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
    const float un_w = 1.0f / (sD * MB_SA);
    bool bad = !(amax < 65000.0f) || !(dmax < 65000.0f);  // (amax is max |a1'| = 2^8 max |a1|)
#pragma unroll
    for (int cot = 0; cot < 2; ++cot)
#pragma unroll
        for (int cit = 0; cit < 2; ++cit)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[wave * 4096 + (cot * 32 + mfma32_row(r, h)) * 64 + cit * 32 + i32] = accW[cot][cit][r] * un_w;
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 512) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < MLP_BWD_WAVES; ++w) t += scr[w * 4096 + e];
        partial_dw2[(size_t)blockIdx.x * 4096 + e] = t;
        bad = bad || !(fabsf(t) <= 3.0e38f);
    }
    __syncthreads();
    {
        const double t0 = sb[0] + shfl_xor_d(sb[0], 32), t1 = sb[1] + shfl_xor_d(sb[1], 32);
        const double g0 = sg[0] + shfl_xor_d(sg[0], 32), g1 = sg[1] + shfl_xor_d(sg[1], 32);
        const double v[2] = {h ? t1 : t0, h ? g1 : g0};
        bad = bad || !(fabs(v[0]) <= 1.0e300) || !(fabs(v[1]) <= 1.0e300);
        if (__ballot(bad) && lane == 0) atomicOr(guard, 1);
        double(*red)[2][64] = reinterpret_cast<double(*)[2][64]>(scr);
        fold_and_finalize<2, MLP_BWD_WAVES>(v, fs, fin1, red, reinterpret_cast<int *>(scr + MLP_BWD_WAVES * 2 * 64 * 2));
    }
}


void launch_edge_mlp_bwd16(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats1,
                           const float *stats2, const float *cc2, const float *w2, float slope, int training,
                           const float *gy, const float *y, const uint8_t *hit2, const unsigned *gmax_bits, float *dv1,
                           float *dvsum, float *partial_dw2, const FoldSet &fs, const BnBackwardSums &fin1, int *guard,
                           int grid, hipStream_t st)
{
    hipLaunchKernelGGL(edge_mlp_bwd16_kernel, dim3(grid), dim3(64 * MLP_BWD_WAVES), 0, st, TP, TQ, idx, N, k, P, stats1, stats2,
                       cc2, w2, slope, training, gy, y, hit2, gmax_bits, dv1, dvsum, partial_dw2, fs, fin1, guard);
}

}  // namespace dgcnn
