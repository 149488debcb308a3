// prof.h -- measurement hook: HIP events around named kernel groups of the hot path, recorded on the stream the
// kernels are launched on (dgcnn_profile_* in dgcnn_hip.h).  Disabled (the default) it costs one load per launch.
#pragma once
#include <hip/hip_runtime.h>

namespace dgcnn {

enum ProfRegion {
    PROF_KNN_BOUND = 0,   // knn_bound16s_kernel (+ knn_bound_merge_kernel): fp16 threshold pass
    PROF_KNN_FILTER,      // knn_filter16_kernel: fp16 all-pairs filter
    PROF_KNN_REFINE,      // knn_refine_kernel: exact fp32 distances of the survivors, selection
    PROF_EDGE_REDUCE,     // edge_reduce_kernel: forward gather, BatchNorm statistics, max-over-k
    PROF_EDGE_RGATHER,    // edge_rgather_kernel (both launches): backward gather over the reverse kNN graph
    PROF_EDGE_MLP_FWD,    // edge_mlp_fwd16_kernel (+ its idle fp32 fallback): second 64 -> 64 convolution of a two-stage block, forward
    PROF_EDGE_MLP_BWD,    // edge_mlp_bwd16_kernel (+ its idle fp32 fallback): its backward (recompute, weight and input gradients, dv1)
    // the head, head_fused.hip:
    PROF_HEAD_CONV6,      // hgemm<RAW, COLRED>: conv6 -> bn6 statistics -> max over the points (nothing stored)
    PROF_HEAD_CONV7,      // hgemm<RAW, STATS>: u7 = cat W7c^T + add7, BatchNorm-7 column sums
    PROF_HEAD_CONV8,      // hgemm<BN, STATS>: u8 = LReLU(BN7(u7)) W8^T, BatchNorm-8 column sums
    PROF_HEAD_DZ7,        // hgemm<DU, DV>: dL/dz7 = du8 W8 -> dv7, BatchNorm-7 backward sums
    PROF_HEAD_DCAT,       // hgemm<DU, PLAIN>: dL/dcat = du7 W7c
    PROF_HEAD_DW8,        // hwgrad<DU, BN>: dW8 = du8^T z7
    PROF_HEAD_DW7,        // hwgrad<DU, RAW>: dW7c = du7^T cat
    PROF_NREGION
};

void prof_begin(int region, hipStream_t st);
void prof_end(int region, hipStream_t st);

}  // namespace dgcnn
