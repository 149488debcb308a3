// edge_mlp_bwd16.hip -- backward of the per-edge 64 -> 64 convolution on the fp16 matrix cores (see edge_mlp.hip for the
// pass as a whole; edge_mlp.hip keeps the forward, the fp32 form of this backward and the launchers).
#include "edge_kernels.h"
#include "edge_mlp_common.h"
#include "edge_mlp16.h"
#include "fold.h"

namespace dgcnn {

// ---------------------------------------------------------------------------------------------
// Backward on the fp16 matrix cores.  Same algorithm as the fp32 kernel above -- recompute u2 = a1 . W2^T, form du2,
// gW2 += du2^T a1, da1 = du2 . W2, dv1 = da1 * LReLU'(v1) -- with every operand scaled by an EXACT power of two into the
// fp16 range and split in two halves of the same scale, x' = hi + lo (hi = x' rounded towards zero to fp16's 11
// significant bits; lo = x' - hi, at most 13 significant bits, rounded to fp16's 11): three
// v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi, fp32 accumulate, one accumulator) replace sixteen
// v_mfma_f32_32x32x2_f32: 72 matrix instructions of 32 cycles per point instead of 192 of 64.  Error of an operand
// element: 2^-23 relative, or 2^-25 absolute on the scaled value where lo runs into fp16's subnormals.  Scales:
//   a1' = 2^8 a1          (BatchNorm-1 output: |a1| < 253.9 is checked)
//   W2' = sW W2           sW from max |W2| (computed here)          -> max |W2'| in [2^11, 2^12)
//   du2' = sD du2         sD from max |dL/dy . LReLU'| (bwd_point_kernel leaves it in *gmax_bits) and max |sc2|
//                         -> the winners' entries stay below 2^10, the dense BatchNorm terms (|sc2| |dv|max (1 + |uhat2|))
//                         below 2^15.99 for |uhat2| < 62; max |du2'| is checked
// A value outside its range, or a non-finite partial sum, raises *guard: the fp32 kernel behind this one then redoes
// the pass (its results overwrite these).  The LDS images of the operands: edge_mlp16.h.

// The contraction over the EDGES of a tile (weight gradient) runs in accumulator order: MFMA step s (16 edges), lane
// half h, element j  <->  edge row mfma32_row(8 s + j, h) = 16 s + 8 (j >> 2) + 4 h + (j & 3) -- so that du2, which the
// u2 product leaves in accumulator registers (lane = channel, registers = edge rows), is the A operand of the weight
// gradient as it stands, and the other operand is two 8-byte row reads of the a1' image (edges 16 s + 8 u + 4 h .. + 3).
// gdv [P][64] = dL/dy . LReLU'(y) of the block's output, left dense by bwd_point_kernel (round 4: one row per point instead of
// a row of dL/dy and a row of y, which in the model are 64-column slices of 192-wide rows)
template <bool S01>
__global__ __launch_bounds__(512) void edge_mlp_bwd16_kernel(
    const float *__restrict__ TP, const float *__restrict__ TQ, const int32_t *__restrict__ idx, int N, int k, int P,
    const float *__restrict__ stats1, const float *__restrict__ stats2, const float *__restrict__ cc2,
    const float *__restrict__ w2, float slope, int training, const float *__restrict__ gdv,
    const uint8_t *__restrict__ hit2, const unsigned *__restrict__ gmax_bits,
    float *__restrict__ dv1, float *__restrict__ dvsum, float *__restrict__ partial_dw2, FoldSet fs,
    BnBackwardSums fin1, int *__restrict__ guard)
{
    __shared__ __attribute__((aligned(16))) unsigned char w2p[2 * W2_PLANE];
    __shared__ __attribute__((aligned(16))) unsigned char tiles[MLP_BWD_WAVES * WAVE_TILE];
    __shared__ float s_red[MLP_BWD_WAVES];
    __shared__ float cst[4][64];  // per-channel constants of the last epilogue (rstd1, -mean1 rstd1, sc1, sh1): read there,
                                  // not held in eight registers through the three products
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, i32 = lane & 31;
    unsigned char *tT = tiles + wave * WAVE_TILE;                   // T planes: a1', then du2'
    float *tU = reinterpret_cast<float *>(tT + 2 * T_PLANE);        // u1 [edge][ci]
    const float *tUl = tU + 4 * h * 64 + i32;                       // accumulator layout: + row0 * 64 + nt * 32
    const unsigned ldo = 4 * h * 64 + i32;                          // the same offset in a [32][64] block of dv1

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
    for (int t = threadIdx.x; t < 4096; t += 512) {
        const int co = t >> 6, ci = t & 63;
        unsigned H, L;
        split_pair(w2[t] * sW, 0.0f, H, L);
        const int off = co * 128 + (((ci >> 3) ^ w2_swz(co)) << 4) + (ci & 7) * 2;
        *reinterpret_cast<unsigned short *>(w2p + off) = (unsigned short)H;
        *reinterpret_cast<unsigned short *>(w2p + W2_PLANE + off) = (unsigned short)L;
    }
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

    // ---- LDS byte offsets of this lane (loop-invariant)
    // transposed reads: lane 4 q + p of a 16-lane group supplies row q, columns 4 p .. 4 p + 3 of its block (4 rows x
    // 16 columns); the group's lanes are columns 16 gg .. 16 gg + 15 of the operand's 32 (edges, or ci)
    const int gg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    unsigned trA[2][2];  // T image, A operand [edge][channel 16 s + 8 h + 4 t + q]: [t][s & 1]; + 2048 (s >> 1)
    unsigned trW[2][2];  // W2 image, B operand [co 16 s + 8 h + 4 t + q][ci 32 nt + ..]: [t][nt]; + 2048 s
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ch = 16 * x + 8 * h + 4 * t + q4;  // (x = s & 1)
            trA[t][x] = ch * 64 + (((4 * gg + p4) ^ t_swz(ch)) << 3);
            const int co = 8 * h + 4 * t + q4;           // (x = nt; w2_swz does not see 16 s)
            trW[t][x] = co * 128 + (((4 * x + 2 * gg + (p4 >> 1)) ^ w2_swz(co)) << 4) + (p4 & 1) * 8;
        }
    const unsigned rowW = i32 * 128 + ((h ^ w2_swz(i32)) << 4);  // W2 row co = 32 nt + i32, chunk 2 s + h: (rowW ^ (s << 5)) + 4096 nt
    const unsigned wT = lane * 64 + (t_swz(lane) << 3);          // gather layout: row ci = lane, chunk g: wT ^ (g << 3)
    const unsigned bT = i32 * 64 + ((t_swz(i32) ^ h) << 3);      // accumulator layout: row 32 x + i32, chunk 2 g + h: (bT ^ (g << 4)) + 2048 x

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
        float dvs[2];  // the winner's term
        int hit[2];
        float dcol[2] = {0.0f, 0.0f}, dhat[2] = {0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            hit[nt] = (int)hit2[(size_t)pt * EC + nt * 32 + i32] - 4 * h;  // relative to this lane half's rows
            dvs[nt] = sD * sc2[nt] * gdv[(size_t)pt * EC + nt * 32 + i32];
        }
        for (int j0 = 0; j0 < k; j0 += 32) {
            const int kk = k - j0 < 32 ? k - j0 : 32;
            {
                float p2[32];
                const int nv = j0 == 0 ? nvn : (lane < kk ? ip[j0 + lane] : 0);
                gather_rows_buf(Pb, N, nv, kk, lane, p2);
                if (j0 == 0 && pt + it.step < it.end) nvn = lane < kk0 ? idx[(size_t)(pt + it.step) * k + lane] : 0;
                // four edges at a time: their a1' halves are one 8-byte chunk of row `lane` of the image.  A group that
                // lies inside the list needs no masks (the branch is wave-uniform)
                auto stage4 = [&](auto G, auto FULL) {
                    constexpr int g = decltype(G)::value;
                    constexpr bool full = decltype(FULL)::value;
                    unsigned H[2], L[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float a[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int j = 4 * g + 2 * m + e;
                            const float u = p2[j] + q;  // (rows past the end repeat the last row: finite, never used)
                            const float v = lrelu_t<S01>(__builtin_fmaf(u, sc1g, sh1g), slope);  // a1'
                            a[e] = full || j < kk ? v : 0.0f;
                            amax = fmaxf(amax, fabsf(a[e]));
                            tU[j * 64 + lane] = u;
                        }
                        split_pair(a[0], a[1], H[m], L[m]);
                    }
                    *reinterpret_cast<uint2 *>(tT + (wT ^ (g << 3))) = make_uint2(H[0], H[1]);
                    *reinterpret_cast<uint2 *>(tT + T_PLANE + (wT ^ (g << 3))) = make_uint2(L[0], L[1]);
                };
                static_for<8>([&](auto G) {
                    if (4 * decltype(G)::value + 4 <= kk) stage4(G, std::true_type{});
                    else stage4(G, std::false_type{});
                });
            }

            // ---- u2raw = a1' . W2'^T : A = columns of the a1' image, B[k = ci][n = co] = rows of the W2 image
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned char *a0 = tT + trA[0][s & 1] + 2048 * (s >> 1), *a1 = tT + trA[1][s & 1] + 2048 * (s >> 1);
                const half8 ahi = tr_read8(a0, a1), alo = tr_read8(a0 + T_PLANE, a1 + T_PLANE);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const unsigned char *bp = w2p + (rowW ^ (s << 5)) + 4096 * nt;
                    const half8 bhi = *reinterpret_cast<const half8 *>(bp);
                    const half8 blo = *reinterpret_cast<const half8 *>(bp + W2_PLANE);
                    acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
                }
            }
            // a1' as the B operand of the weight gradient: B[k = edge][n = ci], edges in accumulator order (read before
            // du2' overwrites the image)
            half8 xhi[2][2], xlo[2][2];  // [ci tile][step]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned char *x0 = tT + (bT ^ ((4 * s) << 3)) + 2048 * ct, *x1 = tT + (bT ^ ((4 * s + 2) << 3)) + 2048 * ct;
                    const uint2 h0 = *reinterpret_cast<const uint2 *>(x0), h1 = *reinterpret_cast<const uint2 *>(x1);
                    const uint2 l0 = *reinterpret_cast<const uint2 *>(x0 + T_PLANE), l1 = *reinterpret_cast<const uint2 *>(x1 + T_PLANE);
                    xhi[ct][s] = words_as_half8(h0.x, h0.y, h1.x, h1.y);
                    xlo[ct][s] = words_as_half8(l0.x, l0.y, l1.x, l1.y);
                }

            // ---- du2' (in place of u2raw): the whole dL/dy goes to the edge the forward selected (hit2); the rest are
            // the dense BatchNorm mean terms, affine in u2.  Rows past the list end are left as they come (cA: their a1'
            // rows are zero, so they add nothing to gW2, and their da1 rows are not used).  Split once, pairwise: registers
            // 8 s .. 8 s + 7 are the A operand of the weight gradient, registers 4 g .. 4 g + 3 one chunk of the du2' image.
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int hrel = hit[nt] - j0;
                unsigned Hd[8], Ld[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float du[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * i + e;
                        // (cA and cB u2 cancel when |mean2| >> sigma2: one fma first, the winner's term after it)
                        du[e] = __builtin_fmaf(cB[nt], acc[nt][r], cA[nt]) + (mfma32_row(r, 0) == hrel ? dvs[nt] : 0.0f);
                        dmax = fmaxf(dmax, fabsf(du[e]));
                    }
                    split_pair(du[0], du[1], Hd[i], Ld[i]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<uint2 *>(tT + (bT ^ (g << 4)) + 2048 * nt) = make_uint2(Hd[2 * g], Hd[2 * g + 1]);
                    *reinterpret_cast<uint2 *>(tT + T_PLANE + (bT ^ (g << 4)) + 2048 * nt) = make_uint2(Ld[2 * g], Ld[2 * g + 1]);
                }
                // ---- gW2[co][ci] += sum_edges du2'[edge][co] a1'[edge][ci]   (this co tile)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 dhi = words_as_half8(Hd[4 * s], Hd[4 * s + 1], Hd[4 * s + 2], Hd[4 * s + 3]);
                    const half8 dlo = words_as_half8(Ld[4 * s], Ld[4 * s + 1], Ld[4 * s + 2], Ld[4 * s + 3]);
#pragma unroll
                    for (int cit = 0; cit < 2; ++cit)
                        accW[nt][cit] = mfma3(dhi, dlo, xhi[cit][s], xlo[cit][s], accW[nt][cit]);
                }
            }

            // ---- da1raw = du2' . W2' : A = columns of the du2' image, B[k = co][n = ci] = columns of the W2 image
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned char *a0 = tT + trA[0][s & 1] + 2048 * (s >> 1), *a1 = tT + trA[1][s & 1] + 2048 * (s >> 1);
                const half8 ahi = tr_read8(a0, a1), alo = tr_read8(a0 + T_PLANE, a1 + T_PLANE);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const unsigned char *b0 = w2p + trW[0][nt] + 2048 * s, *b1 = w2p + trW[1][nt] + 2048 * s;
                    const half8 bhi = tr_read8(b0, b1), blo = tr_read8(b0 + W2_PLANE, b1 + W2_PLANE);
                    acc[nt] = mfma3(ahi, alo, bhi, blo, acc[nt]);
                }
            }
            // ---- dv1 = da1 * LReLU'(v1), BN1 sums.  Rows 8 g .. 8 g + 7 of the tile belong to registers 4 g .. 4 g + 3:
            // a group inside the list runs without masks, one past its end is skipped (wave-uniform branches)
            {
                float *dvb = dv1 + ((size_t)pt * k + j0) * EC;  // (wave-uniform base; lane part: ldo)
                float c_[2][4];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) c_[nt][c] = cst[c][nt * 32 + i32];
                auto rows8 = [&](auto G, auto FULL) {
                    constexpr int g = decltype(G)::value;
                    constexpr bool full = decltype(FULL)::value;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        static_for<4>([&](auto R) {
                            constexpr int r = 4 * g + decltype(R)::value, row0 = mfma32_row(r, 0);
                            const float u = tUl[row0 * 64 + nt * 32];
                            const float v = __builtin_fmaf(u, c_[nt][2], c_[nt][3]);
                            float d = acc[nt][r] * (v > 0.0f ? un_da1 : un_da1s);
                            const float uhat = __builtin_fmaf(u, c_[nt][0], c_[nt][1]);
                            const bool ok = full || row0 + 4 * h < kk;
                            // (streaming: 430 MB of dv1 must not sweep the TP table out of the XCD's L2: 395 -> 351 us)
                            if (ok) __builtin_nontemporal_store(d, &dvb[ldo + (unsigned)(row0 * 64 + nt * 32)]);
                            d = ok ? d : 0.0f;
                            dcol[nt] += d;
                            dhat[nt] = __builtin_fmaf(d, uhat, dhat[nt]);
                        });
                };
                static_for<4>([&](auto G) {
                    if (8 * decltype(G)::value + 8 <= kk) rows8(G, std::true_type{});
                    else if (8 * decltype(G)::value < kk) rows8(G, std::false_type{});
                });
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
    float *scr = reinterpret_cast<float *>(tiles);  // 8 waves x 4096 floats: exactly the tile area
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
        if (__ballot(bad)) {  // complete before this workgroup's fold ticket is drawn (see edge_mlp_fwd16.hip)
            if (lane == 0) atomicOr(guard, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        double(*red)[2][64] = reinterpret_cast<double(*)[2][64]>(scr);
        fold_and_finalize<2, MLP_BWD_WAVES>(v, fs, fin1, red, reinterpret_cast<int *>(scr + MLP_BWD_WAVES * 2 * 64 * 2));
    }
}


void launch_edge_mlp_bwd16(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats1,
                           const float *stats2, const float *cc2, const float *w2, float slope, int training,
                           const float *gdv, const uint8_t *hit2, const unsigned *gmax_bits, float *dv1,
                           float *dvsum, float *partial_dw2, const FoldSet &fs, const BnBackwardSums &fin1, int *guard,
                           int grid, hipStream_t st)
{
    if (slope >= 0.0f && slope <= 1.0f)
        hipLaunchKernelGGL(edge_mlp_bwd16_kernel<true>, dim3(grid), dim3(64 * MLP_BWD_WAVES), 0, st, TP, TQ, idx, N, k, P, stats1,
                           stats2, cc2, w2, slope, training, gdv, hit2, gmax_bits, dv1, dvsum, partial_dw2, fs, fin1, guard);
    else
        hipLaunchKernelGGL(edge_mlp_bwd16_kernel<false>, dim3(grid), dim3(64 * MLP_BWD_WAVES), 0, st, TP, TQ, idx, N, k, P, stats1,
                           stats2, cc2, w2, slope, training, gdv, hit2, gmax_bits, dv1, dvsum, partial_dw2, fs, fin1, guard);
}

}  // namespace dgcnn
