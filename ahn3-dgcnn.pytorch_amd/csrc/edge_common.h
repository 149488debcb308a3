// edge_common.h -- helpers shared by the EdgeConv kernels (gfx950).
#pragma once
#include "common.h"

namespace dgcnn {

typedef unsigned long long u64;

constexpr int EC = 64;  // output channels of every edge convolution (model.py:116-130)

// XCD-aware distribution of points over workgroups.  Blocks are dealt round-robin over the 8
// XCDs (blockIdx % 8 labels the XCD group), so giving group g the g-th contiguous eighth of
// the points keeps one batch element's gather table (N x 64 fp32 = 1.8 MB at N = 7000) inside
// one XCD's 4 MB L2 when B == 8.  Placement only affects speed, never results.
struct PointIter {
    int cur, end, step;
};
__device__ __forceinline__ PointIter point_iter(int P, int wave, int waves_per_wg)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int per = (P + 7) >> 3;
    const int beg = xcd * per;
    PointIter it;
    it.end = beg + per < P ? beg + per : P;
    it.cur = beg + slot * waves_per_wg + wave;
    it.step = nslots * waves_per_wg;
    return it;
}
// the same split over `nblocks` workgroups of a launch that carries other workgroups too (block = index among them)
__device__ __forceinline__ PointIter point_iter_at(int P, int wave, int waves_per_wg, int block, int nblocks)
{
    const int xcd = block & 7, slot = block >> 3, nslots = nblocks >> 3;
    const int per = (P + 7) >> 3;
    const int beg = xcd * per;
    PointIter it;
    it.end = beg + per < P ? beg + per : P;
    it.cur = beg + slot * waves_per_wg + wave;
    it.step = nslots * waves_per_wg;
    return it;
}
static inline int point_grid(int P, int waves_per_wg, int max_slots)
{
    const int per = (P + 7) / 8;
    int slots = ceil_div(per, waves_per_wg);
    if (slots > max_slots) slots = max_slots;
    if (slots < 1) slots = 1;
    return 8 * slots;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }

// fp64 partial records [2][64] written per workgroup by the statistics passes: at most PART_MAX of them, followed
// in the same buffer by PART_FOLD records of scratch for the folding step of the finalize kernels
constexpr int PART_MAX = 4096, PART_FOLD = 32;

// in-edge lists longer than this are cut into segments (edge_stream.hip: hub points of the kNN graph)
constexpr int REV_SEG = 64;

// stats block of one BatchNorm stage inside `saved`: 4 rows of 64 floats
enum { ST_MEAN = 0, ST_RSTD = 64, ST_SC = 128, ST_SH = 192, ST_SIZE = 256 };

}  // namespace dgcnn
