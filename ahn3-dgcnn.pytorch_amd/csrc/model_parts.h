// model_parts.h -- pieces of other translation units that model.hip composes (internal; the public surface is dgcnn_hip.h).
#pragma once
#include "edge_kernels.h"

namespace dgcnn {

// head_fused.hip: dgcnn_embed_bwd with a row stride of dcat and an accumulate switch (dcat_acc: dcat += ...)
int embed_bwd_ld(const float *cat, int ldc, const uint32_t *amax_cat, const float *catsum, const float *w, int ldw,
                 const int32_t *arg, const float *uwin, const float *stats, const float *gg, int B, int N, int F, int C,
                 int training, float slope, float *dcat, int lddcat, int dcat_acc, float *dw, float *gbn_w, float *gbn_b,
                 void *workspace, size_t workspace_bytes, void *stream);

}  // namespace dgcnn
