// hsplit.h -- what the fused head kernels (head_fused.hip) share: power-of-two operand scales, the split of a scaled fp32
// value in two fp16 halves, the three-product MFMA step, the dropout mask.
#pragma once
#include "common.h"

namespace dgcnn {
namespace hf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H_TOP = 15;  // a scaled operand's largest magnitude lies in [2^14, 2^15) (or below, when the scale comes
                           // from an upper bound): inside fp16's range with a factor of two to spare

// the power of two s with 2^(top - 1) <= m s < 2^top (m > 0 and finite; else 1)
__device__ __forceinline__ float pow2_scale(float m, int top)
{
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    if (!(m > 0.0f) || e > 126) return 1.0f;
    int s = top - 1 - e;
    s = s > 120 ? 120 : (s < -120 ? -120 : s);
    return __uint_as_float((unsigned)(s + 127) << 23);
}

// two scaled values -> H = {hi(x0), hi(x1)}, L = {lo(x0), lo(x1)}: hi = x rounded TOWARDS ZERO to fp16 (one v_cvt_pkrtz
// for the pair), lo = fp16(x - hi) (x - hi is exact in fp32; v_fma_mix{lo,hi}_f16 takes hi as an fp16 source and rounds
// the difference to nearest): 1.5 instructions per element.  |x - (hi + lo)| <= 2^-22 |x| while lo is a normal fp16
// number, 2^-25 absolute (in scaled units) below that.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &H, unsigned &L)
{
    H = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(H), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(H), "v"(x1));
    L = l;
}
__device__ __forceinline__ void split4(const float (&x)[4], u32x2 &hi, u32x2 &lo)
{
    unsigned h0, l0, h1, l1;
    split_pair(x[0], x[1], h0, l0);
    split_pair(x[2], x[3], h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

// (a_hi + a_lo) (b_hi + b_lo) without the lo.lo term, fp32 accumulate
__device__ __forceinline__ f32x16 mfma3(const half8 &ahi, const half8 &alo, const half8 &bhi, const half8 &blo, f32x16 c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.0f ? v : v * slope; }

// Dropout mask of the classifier's input (model.py:190).  A counter-based generator: the four consecutive elements
// 4 q .. 4 q + 3 of the [P][K] activation share one 64-bit mix of (seed, q) and element e is KEPT when bits 16 e .. 16 e + 15
// of it fall below keep16 = round((1 - p) 65536) -- forward and backward regenerate the same mask from the seed, nothing is
// stored.  (splitmix64 finalizer: every output bit depends on every input bit.)
__device__ __host__ __forceinline__ unsigned long long drop_mix(unsigned long long seed, unsigned long long q)
{
    unsigned long long z = seed + (q + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ bool drop_keep(unsigned long long mix, int e, unsigned keep16)
{
    return (unsigned)((mix >> (16 * e)) & 0xffffull) < keep16;
}

}  // namespace hf
}  // namespace dgcnn
