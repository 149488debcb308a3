// edge_mlp16.h -- what the fp16-matrix-core kernels of the per-edge 64 -> 64 convolution share (edge_mlp_fwd16.hip,
// edge_mlp_bwd16.hip): the split of a scaled fp32 operand in two fp16 halves, the LDS images of the operands and
// their reads, the gather of a point's neighbour rows.
#pragma once
#include "edge_mlp_common.h"

namespace dgcnn {

constexpr float MB_SA = 256.0f;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4v lds_fp16x4v;

// LDS images (no padding; 8- or 16-byte chunks of a row are XOR-swizzled so that every access below is conflict-free):
//   W2 planes  [co 64][ci 64] fp16, hi plane then lo plane (W2_PLANE bytes each); 16-byte chunk ch of row co sits at
//              chunk ch ^ w2_swz(co).  Read by rows (ds_read_b128: B operand of u2 = a1 . W2^T) and by columns
//              (ds_read_b64_tr_b16: B operand of da1 = du2 . W2) -- one image serves both products.
//   T planes   [channel 64][edge 32] fp16 per wave, hi then lo (T_PLANE bytes each): a1' first, du2' later; 8-byte chunk
//              (4 edges) ch of row c at chunk ch ^ t_swz(c).  Written 4 edges at a time by the lane that owns the channel
//              (ds_write_b64), read by columns (tr: the A operand [edge][channel] of u2 and of da1) and by rows
//              (ds_read_b64: the B operand [edge][ci] of the weight gradient, edges in accumulator order).
//   U          [edge 32][channel 64] fp32 per wave: u1, for the last epilogue.
constexpr int W2_PLANE = 64 * 64 * 2, T_PLANE = 64 * 32 * 2, WAVE_TILE = 2 * T_PLANE + 32 * 64 * 4;
__device__ __forceinline__ constexpr int w2_swz(int co) { return (((co >> 1) & 1) << 2) | ((co >> 2) & 3); }
__device__ __forceinline__ constexpr int t_swz(int c) { return ((c >> 2) & 3) | ((((c >> 1) ^ (c >> 4)) & 1) << 2); }

// LeakyReLU; S01 (0 <= slope <= 1, the launcher checks): max(v, slope v) -- the same value bit for bit, one
// instruction less than compare + select
template <bool S01>
__device__ __forceinline__ float lrelu_t(float v, float slope)
{
    return S01 ? __builtin_fmaxf(v, v * slope) : lrelu(v, slope);
}

// two scaled values -> H = {hi(x0), hi(x1)}, L = {lo(x0), lo(x1)}: hi = x rounded TOWARDS ZERO to fp16 (one
// v_cvt_pkrtz for the pair), lo = fp16(x - hi) (x - hi is exact in fp32; v_fma_mix{lo,hi}_f16 takes hi as an fp16
// source and rounds the difference to nearest): 1.5 instructions per element
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &H, unsigned &L)
{
    H = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(H), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(H), "v"(x1));
    L = l;
}
// gather one M-tile (edges j0 .. j0 + 31 of a point) through a buffer descriptor of the batch element's TP table: the
// row offset is the instruction's scalar offset and the lane's column its one address register -- no address
// arithmetic on the vector unit and no 64-bit address pairs (rows past the list end repeat its last row)
__device__ __forceinline__ void gather_rows_buf(const float *__restrict__ Pb, int N, int nv, int kk, int lane, float (&p)[32])
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)Pb, 0, N * EC * 4, 0x00020000);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g * 8 < kk) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tt = g * 8 + t < kk ? g * 8 + t : kk - 1;
                const unsigned row = (unsigned)__builtin_amdgcn_readlane(nv, tt);
                p[g * 8 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, row * (EC * 4), 0));
            }
        }
    }
}
__device__ __forceinline__ half8 words_as_half8(unsigned a, unsigned b, unsigned c, unsigned d)
{
    return __builtin_bit_cast(half8, u32x4{a, b, c, d});
}
// eight consecutive k of an operand from two transposed reads (4 k each)
__device__ __forceinline__ half8 tr_read8(const unsigned char *p0, const unsigned char *p1)
{
    struct Two { fp16x4v a, b; };
    return __builtin_bit_cast(half8, Two{__builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)p0),
                                         __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4v *)p1)});
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

}  // namespace dgcnn
