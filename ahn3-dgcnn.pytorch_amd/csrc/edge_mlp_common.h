// edge_mlp_common.h -- constants and device helpers shared by the per-edge 64 -> 64 kernels (edge_mlp.hip: forward and
// the fp32 backward; edge_mlp_bwd16.hip: the fp16-matrix-core backward).
#pragma once
#include "edge_common.h"

namespace dgcnn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int MLP_WAVES = 4;      // forward
constexpr int MLP_BWD_WAVES = 8;  // backward: two waves per SIMD, one in its MFMA phases while the other gathers / stages
constexpr int TSTR = 68;  // LDS tile row stride in floats (64 + 4: conflict-free b128 rows)

__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// gather one M-tile (edges j0 .. j0 + 31 of a point): the loads (rows of TP -> registers; wave-uniform table base,
// 32-bit element offsets, rows past the list end repeat its last row) are split from the consumer, so that the
// next point's rows can be in flight under the current point's MFMA phases
__device__ __forceinline__ void gather_rows(const float *__restrict__ Pb, int nv, int kk, int lane, float (&p)[32])
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g * 8 < kk) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tt = g * 8 + t < kk ? g * 8 + t : kk - 1;
                p[g * 8 + t] = (Pb + (size_t)(unsigned)__builtin_amdgcn_readlane(nv, tt) * EC)[(unsigned)lane];  // scalar row base
            }
        }
    }
}
// leave a1 (and optionally u1) of the gathered tile in LDS; rows >= kk are zero
template <bool KEEP_U>
__device__ __forceinline__ void stage_tile(const float (&p)[32], float q, int kk, float sc1, float sh1, float slope,
                                           float (*tileA)[TSTR], float (*tileU)[TSTR], int lane)
{
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const bool on = j < kk;
        const float u = on ? p[j] + q : 0.0f;
        const float a = on ? lrelu(__builtin_fmaf(u, sc1, sh1), slope) : 0.0f;
        tileA[j][lane] = a;
        if (KEEP_U) tileU[j][lane] = u;
    }
}


}  // namespace dgcnn
