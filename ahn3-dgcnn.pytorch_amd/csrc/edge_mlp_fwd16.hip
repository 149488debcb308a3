// edge_mlp_fwd16.hip -- forward of the per-edge 64 -> 64 convolution on the fp16 matrix cores (see edge_mlp.hip for the
// pass as a whole; edge_mlp.hip keeps the fp32 form of this kernel, which is its guarded fallback, and the launchers).
#include "edge_kernels.h"
#include "edge_mlp16.h"
#include "fold.h"

namespace dgcnn {

// u2 = a1 . W2^T with both operands scaled by an exact power of two into the fp16 range and split in two halves,
// x' = hi + lo (edge_mlp16.h): three v_mfma_f32_32x32x16_f16 per 16 input channels (hi.hi + hi.lo + lo.hi, one fp32
// accumulator); the dropped lo.lo term and the rounding of lo are 2^-22 relative.  Scales:
//   a1' = 2^8 a1    max |a1'| is checked (beyond 65000 the halves no longer carry the value)
//   W2' = sW W2     sW from max |W2| (computed here)  -> max |W2'| in [2^11, 2^12)
// The a1' tile of a point goes through the wave's T planes (written by the lane that owns the input channel, four
// edges per 8-byte chunk; read transposed as the A operand), W2' sits in the workgroup's W2 planes.  An operand
// outside its range or a non-finite sum raises *guard: the BatchNorm finalizer of this kernel then stands down and
// the fp32 kernel behind it redoes the pass (usel2, hit2, BatchNorm-2 statistics).
// Four waves per workgroup share one W2 image: 4 x 8 KB of T planes + 16 KB = 48 KB and 150 registers -- three
// workgroups per CU, three waves per SIMD (96 slots per XCD: measured 142 / 123 / 151 us at 64 / 96 / 128; sixteen
// waves around one image -- four per SIMD at 128 registers -- 130 us).
constexpr int FWD16_WAVES = 4;
int mlp_grid_fwd16(int P) { return point_grid(P, FWD16_WAVES, 96); }

template <bool S01>
__global__ __launch_bounds__(64 * FWD16_WAVES) void edge_mlp_fwd16_kernel(
    const float *__restrict__ TP, const float *__restrict__ TQ, const int32_t *__restrict__ idx, int N, int k, int P,
    const float *__restrict__ stats1, const float *__restrict__ w2, const float *__restrict__ gamma2, float slope,
    float *__restrict__ usel2, uint8_t *__restrict__ hit2, int fin_on, FoldSet fs, BnBatchStats fin,
    int *__restrict__ guard)
{
    __shared__ __attribute__((aligned(16))) unsigned char w2p[2 * W2_PLANE];
    __shared__ __attribute__((aligned(16))) unsigned char tiles[FWD16_WAVES * 2 * T_PLANE];
    __shared__ float s_red[FWD16_WAVES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, i32 = lane & 31;
    unsigned char *tT = tiles + wave * 2 * T_PLANE;

    // ---- W2' planes (the scale is the same in every workgroup: a function of the weights only)
    float wmax = 0.0f;
    for (int t = threadIdx.x; t < 4096; t += 64 * FWD16_WAVES) wmax = fmaxf(wmax, fabsf(w2[t]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_red[wave] = wmax;
    __syncthreads();
    wmax = s_red[0];
#pragma unroll
    for (int w = 1; w < FWD16_WAVES; ++w) wmax = fmaxf(wmax, s_red[w]);
    const float sW = scale_to(wmax, 12);
    for (int t = threadIdx.x; t < 4096; t += 64 * FWD16_WAVES) {
        const int co = t >> 6, ci = t & 63;
        unsigned H, L;
        // (output channels with gamma2 < 0 carry -W2: the extreme of the activation is then always a maximum of u2raw)
        split_pair(w2[t] * (gamma2[co] < 0.0f ? -sW : sW), 0.0f, H, L);
        const int off = co * 128 + (((ci >> 3) ^ w2_swz(co)) << 4) + (ci & 7) * 2;
        *reinterpret_cast<unsigned short *>(w2p + off) = (unsigned short)H;
        *reinterpret_cast<unsigned short *>(w2p + W2_PLANE + off) = (unsigned short)L;
    }
    __syncthreads();

    // ---- LDS byte offsets of this lane (edge_mlp_bwd16_kernel has the derivation)
    const int gg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    unsigned trA[2][2];  // [t][s & 1]; + 2048 (s >> 1)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ch = 16 * x + 8 * h + 4 * t + q4;
            trA[t][x] = ch * 64 + (((4 * gg + p4) ^ t_swz(ch)) << 3);
        }
    const unsigned rowW = i32 * 128 + ((h ^ w2_swz(i32)) << 4);  // W2 row 32 nt + i32, chunk 2 s + h: (rowW ^ (s << 5)) + 4096 nt
    const unsigned wT = lane * 64 + (t_swz(lane) << 3);          // T row ci = lane, chunk g: wT ^ (g << 3)

    // u2 = uns u2raw: the power-of-two scales and the sign of the W2' row leave in one exact multiplication, applied to
    // the sums and to the extreme only (scaling by a power of two commutes with every rounding in between)
    const float un = 1.0f / (MB_SA * sW);
    float uns[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) uns[nt] = gamma2[nt * 32 + i32] < 0.0f ? -un : un;
    // (LeakyReLU is positively homogeneous: a1' = 2^8 a1 = LReLU(2^8 sc1 u + 2^8 sh1))
    const float sc1g = stats1[ST_SC + lane] * MB_SA, sh1g = stats1[ST_SH + lane] * MB_SA;

    double sum[2] = {0.0, 0.0}, sq[2] = {0.0, 0.0};
    float amax = 0.0f;
    const PointIter it = point_iter(P, wave, FWD16_WAVES);
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const int b = pt / N;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(TP + (size_t)b * N * EC), 0, N * EC * 4, 0x00020000);
        const float q = TQ[(size_t)pt * EC + lane];
        const int32_t *ip = idx + (size_t)pt * k;
        float ext[2], ps[2] = {0.0f, 0.0f}, pq[2] = {0.0f, 0.0f};  // of u2raw: fp32 over the <= 32 values of a lane, fp64 across points
        int arg[2] = {0, 0};  // edge slot of the extreme (the first one on ties): the backward routes dL/dy there
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ext[nt] = -__builtin_inff();
        for (int j0 = 0; j0 < k; j0 += 32) {
            const int kk = k - j0 < 32 ? k - j0 : 32;
            {
                const int nv = lane < kk ? ip[j0 + lane] : 0;
                // eight rows in flight (scalar row offsets, rows past the list end repeat its last row); four edges
                // make one 8-byte chunk of row `lane` of the image; a group inside the list needs no masks
                auto stage4 = [&](const float (&p)[8], auto G, auto FULL) {
                    constexpr int g = decltype(G)::value;
                    constexpr bool full = decltype(FULL)::value;
                    unsigned H[2], L[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float a[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int j = 4 * g + 2 * m + e;
                            const float v = lrelu_t<S01>(__builtin_fmaf(p[j & 7] + q, sc1g, sh1g), slope);  // a1'
                            a[e] = full || j < kk ? v : 0.0f;
                            amax = fmaxf(amax, fabsf(a[e]));
                        }
                        split_pair(a[0], a[1], H[m], L[m]);
                    }
                    *reinterpret_cast<uint2 *>(tT + (wT ^ (g << 3))) = make_uint2(H[0], H[1]);
                    *reinterpret_cast<uint2 *>(tT + T_PLANE + (wT ^ (g << 3))) = make_uint2(L[0], L[1]);
                };
                static_for<4>([&](auto G8) {
                    constexpr int g8 = decltype(G8)::value;
                    float p[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int tt = g8 * 8 + t < kk ? g8 * 8 + t : kk - 1;
                        const unsigned row = (unsigned)__builtin_amdgcn_readlane(nv, tt);
                        p[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, row * (EC * 4), 0));
                    }
                    static_for<2>([&](auto M) {
                        constexpr int g = 2 * g8 + decltype(M)::value;
                        if (4 * g + 4 <= kk) stage4(p, std::integral_constant<int, g>{}, std::true_type{});
                        else stage4(p, std::integral_constant<int, g>{}, std::false_type{});
                    });
                });
            }
            // ---- u2raw = a1' . W2'^T : A = columns of the a1' image, B[k = ci][n = co] = rows of the W2 image
            half8 ahi[4], alo[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned char *a0 = tT + trA[0][s & 1] + 2048 * (s >> 1), *a1 = tT + trA[1][s & 1] + 2048 * (s >> 1);
                ahi[s] = tr_read8(a0, a1);
                alo[s] = tr_read8(a0 + T_PLANE, a1 + T_PLANE);
            }
            const int kkh = kk - 4 * h;  // row 8 g + (r & 3) + 4 h of the tile is an edge iff 8 g + (r & 3) < kkh
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned char *bp = w2p + (rowW ^ (s << 5)) + 4096 * nt;
                    const half8 bhi = *reinterpret_cast<const half8 *>(bp);
                    const half8 blo = *reinterpret_cast<const half8 *>(bp + W2_PLANE);
                    acc = mfma3(ahi[s], alo[s], bhi, blo, acc);
                }
                // statistics over all rows (exactly 0 in the padding rows: a1' = 0), then the maximum over the rows that
                // are edges, and the first slot that holds it -- all on the raw accumulators, which stay as they are (the
                // group of eight rows that holds the list end is the only one that selects; branches are wave-uniform)
                float best = -__builtin_inff();
                static_for<4>([&](auto G) {
                    constexpr int g = decltype(G)::value;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ps[nt] += acc[4 * g + e];
                        pq[nt] = __builtin_fmaf(acc[4 * g + e], acc[4 * g + e], pq[nt]);
                    }
                    if (8 * g + 8 <= kk) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) best = fmaxf(best, acc[4 * g + e]);
                    } else if (8 * g < kk) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) best = fmaxf(best, 8 * g + e < kkh ? acc[4 * g + e] : -__builtin_inff());
                    }
                });
                if (best > ext[nt]) {
                    int slot = 0;
                    static_for<4>([&](auto GR) {
                        constexpr int g = 3 - decltype(GR)::value;  // descending: the first row that holds the maximum wins
                        if (8 * g + 8 <= kk) {
#pragma unroll
                            for (int e = 3; e >= 0; --e) slot = acc[4 * g + e] == best ? mfma32_row(4 * g + e, h) : slot;
                        } else if (8 * g < kk) {
#pragma unroll
                            for (int e = 3; e >= 0; --e)
                                slot = (acc[4 * g + e] == best && 8 * g + e < kkh) ? mfma32_row(4 * g + e, h) : slot;
                        }
                    });
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
                usel2[(size_t)pt * EC + nt * 32 + i32] = (take ? o : ext[nt]) * uns[nt];
                hit2[(size_t)pt * EC + nt * 32 + i32] = (uint8_t)(take ? oa : arg[nt]);
            }
        }
    }
    // an operand beyond the fp16 range (a1' checked here; a non-finite weight or input shows in the sums of squares):
    // hand the pass to the fp32 instance behind this launch
    const bool bad = !(amax < 65000.0f) || !(sq[0] <= 1.0e300) || !(sq[1] <= 1.0e300);
    // (the raise has to be COMPLETE before any wave of this workgroup can draw its fold ticket: only wave 0 drains its
    // own memory operations in fold.h, and the last arriver's finalizer reads the guard with a relaxed load -- so every
    // raising wave waits for its atomic here, ahead of the barrier in front of the fold)
    if (__ballot(bad)) {
        if (lane == 0) atomicOr(guard, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // BN2 batch statistics (training): lane (n, h) holds channel nt * 32 + n; fold the halves -- lane l then carries
    // channel l (nt = h) -- and let the pass finish the sums itself (fold.h)
    if (fin_on) {
        const double s0 = sum[0] * (double)uns[0], s1 = sum[1] * (double)uns[1], un2 = (double)un * (double)un;
        const double t0 = s0 + shfl_xor_d(s0, 32), t1 = s1 + shfl_xor_d(s1, 32);
        const double q0 = (sq[0] + shfl_xor_d(sq[0], 32)) * un2, q1 = (sq[1] + shfl_xor_d(sq[1], 32)) * un2;
        const double v[2] = {h ? t1 : t0, h ? q1 : q0};
        __syncthreads();  // every wave is done with its tile: the tile memory is the fold's scratch
        double(*red)[2][64] = reinterpret_cast<double(*)[2][64]>(tiles);
        fold_and_finalize<2, FWD16_WAVES>(v, fs, fin, red, reinterpret_cast<int *>(tiles + FWD16_WAVES * 2 * 64 * 8));
    }
}

void launch_edge_mlp_fwd16(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats1,
                           const float *w2, const float *gamma2, float slope, float *usel2, uint8_t *hit2, int fin_on,
                           const FoldSet &fs, const BnBatchStats &fin, int *guard, hipStream_t st)
{
    if (slope >= 0.0f && slope <= 1.0f)
        hipLaunchKernelGGL(edge_mlp_fwd16_kernel<true>, dim3(mlp_grid_fwd16(P)), dim3(64 * FWD16_WAVES), 0, st, TP, TQ, idx, N, k, P,
                           stats1, w2, gamma2, slope, usel2, hit2, fin_on, fs, fin, guard);
    else
        hipLaunchKernelGGL(edge_mlp_fwd16_kernel<false>, dim3(mlp_grid_fwd16(P)), dim3(64 * FWD16_WAVES), 0, st, TP, TQ, idx, N, k, P,
                           stats1, w2, gamma2, slope, usel2, hit2, fin_on, fs, fin, guard);
}

}  // namespace dgcnn
