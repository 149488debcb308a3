// model.hip -- DGCNN.forward at depth 10 (reference model.py:144-193) and its backward as ONE call each on the caller's
// stream: the composition of this library's own entry points, with the framework ops a host-language composition needs in
// between (torch.cat of model.py:167, the sums of the gradients that meet in x1 / x2 / x3, the pooled-feature product of
// model.py:179 and its backward, the counters of the eight BatchNorms) replaced by row strides, accumulate switches and
// three small kernels.
//
//   forward   x [B][F][N] --to_point_major--> xpm [P][F]
//             block b = 0, 1, 2:  kNN(in_b) -> idx[b];  EdgeConv(in_b, idx[b]) -> cat[:, 64 b .. 64 b + 63]
//                                 in_0 = xpm (dense), in_b = cat[:, 64 (b - 1) ..] (row stride 192)
//             rows_stats(cat) -> max |cat|, column sums;  embed(cat) -> g [B][emb]  (conv6 -> bn6 -> LReLU -> max over points)
//             add7 = g W7[:, :emb]^T  (pooled_fwd_kernel);  head tail(cat, add7) -> logits [P][NC] --from_point_major--> caller
//   backward  dlogits -> head tail bwd -> gcat [P][192], gadd7;  pooled_bwd_kernel -> gg = gadd7 W7[:, :emb];
//             gw7_kernel -> gW7 = [gadd7^T g | gW7c];  embed bwd: gcat += dL/dcat of conv6;
//             block 2, 1, 0 backward with gy = gcat[:, 64 b ..] (+ block b + 1's dense input gradient, added by the block's
//             first kernel)
#include "model_parts.h"

namespace dgcnn {
namespace {

constexpr int CATW = 192;  // x1 | x2 | x3

struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base((char *)p), off(0) {}
    template <typename T>
    T *take(size_t n)
    {
        T *r = (T *)(base + off);
        off += align_up(n * sizeof(T), 256);
        return r;
    }
    void *bytes(size_t n)
    {
        void *r = base + off;
        off += align_up(n, 256);
        return r;
    }
};

dgcnn_edgeconv_desc block_desc(const dgcnn_model_desc *d, int b)
{
    dgcnn_edgeconv_desc e{};
    e.B = d->B, e.N = d->N, e.C = b == 0 ? d->F : EC, e.k = d->k;
    e.n_stage = b < 2 ? 2 : 1;
    e.training = d->training;
    e.eps = d->eps[2 * b], e.momentum = d->momentum[2 * b], e.slope = d->slope;
    e.w1 = d->w[2 * b], e.bn1 = d->bn[2 * b];
    if (b < 2) e.w2 = d->w[2 * b + 1], e.bn2 = d->bn[2 * b + 1];
    return e;
}

dgcnn_head_tail_desc tail_desc(const dgcnn_model_desc *d)
{
    dgcnn_head_tail_desc t{};
    t.B = d->B, t.N = d->N, t.F = CATW, t.C7 = d->c7, t.C8 = d->c8, t.NC = d->num_classes;
    t.ldcat = CATW, t.ldw7 = d->emb + CATW;
    t.training = d->training;
    t.eps7 = d->eps[6], t.eps8 = d->eps[7], t.momentum7 = d->momentum[6], t.momentum8 = d->momentum[7];
    t.slope = d->slope, t.p_drop = d->p_drop, t.seed = d->seed;
    t.w7c = d->w[6] ? d->w[6] + d->emb : nullptr, t.bn7 = d->bn[6];
    t.w8 = d->w[7], t.bn8 = d->bn[7];
    t.w9 = d->w[8];
    // (the counters of bn6 .. bn8 are bumped by pooled_fwd_kernel)
    t.bn7.num_batches_tracked = nullptr, t.bn8.num_batches_tracked = nullptr;
    return t;
}

int check_model(const dgcnn_model_desc *d, const char *who, bool params)
{
    DG_REQUIRE(d, "%s: null descriptor", who);
    DG_REQUIRE(d->B > 0 && d->B <= 64 && d->N >= 128 && d->F > 0 && d->F <= 64 && d->k > 0 && d->k <= 64 && d->k <= d->N,
               "%s: shape B=%d N=%d F=%d k=%d outside 1 <= B <= 64, N >= 128, F <= 64, k <= min(64, N)", who, d->B, d->N, d->F, d->k);
    DG_REQUIRE((size_t)d->B * d->N < ((size_t)1 << 23), "%s: B N >= 2^23 points", who);
    DG_REQUIRE(d->emb >= 32 && (d->emb & 31) == 0 && d->emb <= 8192, "%s: emb=%d must be a multiple of 32 (<= 8192)", who, d->emb);
    DG_REQUIRE(d->c7 >= 32 && (d->c7 & 31) == 0 && d->c7 <= 1024 && d->c8 >= 32 && (d->c8 & 31) == 0 && d->c8 <= 1024,
               "%s: c7=%d, c8=%d must be multiples of 32 (<= 1024)", who, d->c7, d->c8);
    DG_REQUIRE(d->num_classes >= 1 && d->num_classes <= 8, "%s: num_classes=%d outside 1..8", who, d->num_classes);
    DG_REQUIRE(d->slope >= 0.0f && d->p_drop >= 0.0f && d->p_drop < 1.0f, "%s: slope / p_drop out of range", who);
    if (!params) return 0;
    for (int i = 0; i < 9; ++i) DG_REQUIRE(d->w[i], "%s: weight %d missing", who, i + 1);
    for (int i = 0; i < 8; ++i) {
        DG_REQUIRE(d->bn[i].weight && d->bn[i].bias, "%s: BatchNorm %d parameters missing", who, i + 1);
        DG_REQUIRE(d->training || (d->bn[i].running_mean && d->bn[i].running_var), "%s: eval mode needs running statistics (bn%d)", who, i + 1);
    }
    for (int b = 0; b < 3; ++b) DG_REQUIRE(d->idx[b], "%s: idx[%d] missing", who, b);
    DG_REQUIRE(((uintptr_t)d->w[5] & 15) == 0 && ((uintptr_t)d->w[6] & 15) == 0, "%s: conv6 / conv7 weights must be 16-byte aligned", who);
    return 0;
}

struct Saved {
    float *xpm, *cat, *catsum, *g, *uwin, *stats6, *add7;
    uint32_t *amax_cat;
    int32_t *arg;
    void *edge[3], *tail;
    size_t bytes;
};
Saved saved_layout(const dgcnn_model_desc *d, void *p)
{
    Carver c(p);
    Saved s;
    const size_t P = (size_t)d->B * d->N;
    s.xpm = c.take<float>(P * d->F);
    s.cat = c.take<float>(P * CATW);
    s.amax_cat = c.take<uint32_t>(64);
    s.catsum = c.take<float>(CATW);
    s.g = c.take<float>((size_t)d->B * d->emb);
    s.arg = c.take<int32_t>((size_t)d->B * d->emb);
    s.uwin = c.take<float>((size_t)d->B * d->emb);
    s.stats6 = c.take<float>((size_t)4 * d->emb);
    s.add7 = c.take<float>((size_t)d->B * d->c7);
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        s.edge[b] = c.bytes(dgcnn_edgeconv_saved_bytes(&e));
    }
    const dgcnn_head_tail_desc t = tail_desc(d);
    s.tail = c.bytes(dgcnn_head_tail_saved_bytes(&t));
    s.bytes = c.off;
    return s;
}

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

struct FwdWs {
    float *logits_pm;
    void *scratch;
    size_t scratch_bytes, bytes;
};
FwdWs fwd_ws(const dgcnn_model_desc *d, void *p)
{
    Carver c(p);
    FwdWs w;
    const size_t P = (size_t)d->B * d->N;
    w.logits_pm = c.take<float>(P * d->num_classes);
    size_t m = 0;
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        m = max_sz(m, dgcnn_edgeconv_fwd_workspace_bytes(&e));
        m = max_sz(m, dgcnn_knn_workspace_bytes(d->B, e.C, d->N, d->k));
    }
    m = max_sz(m, dgcnn_rows_stats_workspace_bytes((int)P, CATW));
    m = max_sz(m, dgcnn_embed_fwd_workspace_bytes(d->B, d->N, CATW, d->emb));
    const dgcnn_head_tail_desc t = tail_desc(d);
    m = max_sz(m, dgcnn_head_tail_fwd_workspace_bytes(&t));
    w.scratch_bytes = m;
    w.scratch = c.bytes(m);
    w.bytes = c.off;
    return w;
}

struct BwdWs {
    float *gl_pm, *gcat, *gadd7, *gg, *gw7c;
    float *gxd;          // [P][64]: a block's input gradient (dense); the previous block's first kernel adds it to its gcat slice
    // the reverse neighbour lists of the three blocks, built in one go (launch_rev_build_merged)
    int *rzero, *rcnt, *rindeg, *rrowptr, *rsegbase, *rpcnt, *rrev, *rextra;
    int rextra_stride;
    void *scratch;
    size_t scratch_bytes, bytes;
};
BwdWs bwd_ws(const dgcnn_model_desc *d, void *p)
{
    Carver c(p);
    BwdWs w;
    const size_t P = (size_t)d->B * d->N;
    w.gl_pm = c.take<float>(P * d->num_classes);
    w.gcat = c.take<float>(P * CATW);
    w.gadd7 = c.take<float>((size_t)d->B * d->c7);
    w.gg = c.take<float>((size_t)d->B * d->emb);
    w.gw7c = c.take<float>((size_t)d->c7 * CATW);
    w.gxd = c.take<float>(P * EC);
    {
        const size_t E = P * d->k;
        w.rextra_stride = (int)(2 * (E / REV_SEG + 1));
        w.rzero = c.take<int>(3 * Z_WORDS);
        w.rcnt = c.take<int>((size_t)3 * d->B * rev_chunks(3 * d->B, d->N, d->k) * d->N);
        w.rindeg = c.take<int>(3 * P);
        w.rrowptr = c.take<int>(3 * P);
        w.rsegbase = c.take<int>(3 * P);
        w.rpcnt = c.take<int>(3 * P);
        w.rrev = c.take<int>(3 * E);
        w.rextra = c.take<int>((size_t)3 * w.rextra_stride);
    }
    size_t m = 0;
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        m = max_sz(m, dgcnn_edgeconv_bwd_workspace_bytes(&e));
    }
    m = max_sz(m, dgcnn_embed_bwd_workspace_bytes(d->B, d->N, CATW, d->emb));
    const dgcnn_head_tail_desc t = tail_desc(d);
    m = max_sz(m, dgcnn_head_tail_bwd_workspace_bytes(&t));
    w.scratch_bytes = m;
    w.scratch = c.bytes(m);
    w.bytes = c.off;
    return w;
}

// x [B][F][N] by element strides that IS point-major memory already (the permuted view train.py:176 / test.py:119 hand over):
// then the rows are read in place
bool is_point_major(const dgcnn_model_desc *d, int64_t sb, int64_t sc, int64_t sn)
{
    return sc == 1 && sn == d->F && sb == (int64_t)d->N * d->F;
}

// ---- the pooled feature's share of conv7 (model.py:179-184: cat(repeat(g), x1, x2, x3) through conv7) ------------------
// add7[b][o] = sum_{c < emb} g[b][c] W7[o][c]: one wave per output, the wave's 64 partial sums folded by a butterfly (fixed
// order).  Workgroup 0's first thread also bumps the counters of bn6 .. bn8 (training mode; the launches that finish their
// statistics serve several callers, this one only the model).
__global__ __launch_bounds__(256) void pooled_fwd_kernel(const float *__restrict__ g, const float *__restrict__ W7, int ldw, int B,
                                                         int emb, int c7, float *__restrict__ add7, int64_t *nbt6, int64_t *nbt7,
                                                         int64_t *nbt8)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (nbt6) *nbt6 += 1;
        if (nbt7) *nbt7 += 1;
        if (nbt8) *nbt8 += 1;
    }
    const int lane = threadIdx.x & 63, out = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (out >= B * c7) return;
    const int b = out / c7, o = out - b * c7;
    const float *gr = g + (size_t)b * emb, *wr = W7 + (size_t)o * ldw;
    float s = 0.0f;
    for (int c = lane * 4; c < emb; c += 256) {  // (emb is a multiple of 4; rows 16-byte aligned)
        const float4 a = *reinterpret_cast<const float4 *>(gr + c), w = *reinterpret_cast<const float4 *>(wr + c);
        s = __builtin_fmaf(a.x, w.x, s);
        s = __builtin_fmaf(a.y, w.y, s);
        s = __builtin_fmaf(a.z, w.z, s);
        s = __builtin_fmaf(a.w, w.w, s);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) add7[out] = s;
}

// gg[b][c] = sum_o gadd7[b][o] W7[o][c] (c < emb).  Workgroup = (32 columns, one batch element): eight groups of 32 lanes take
// an eighth of the o range each (128-byte row segments of W7, eight loads in flight), their partial sums meet in LDS and are
// added in group order (fixed association).  (One thread per (b, c) walking all c7 rows: 59 us at B = 4.)
__global__ __launch_bounds__(256) void pooled_bwd_kernel(const float *__restrict__ gadd7, const float *__restrict__ W7, int ldw, int B,
                                                         int emb, int c7, float *__restrict__ gg)
{
    __shared__ float part[8][32];
    const int b = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    const int per = (c7 + 7) / 8, o0 = grp * per, o1 = o0 + per < c7 ? o0 + per : c7;
    const float *ga = gadd7 + (size_t)b * c7;
    const int cc = c < emb ? c : 0;
    float s = 0.0f;
    for (int o = o0; o < o1; o += 8) {
        float w[8], a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int oo = o + u < o1 ? o + u : o1 - 1;   // clamped: unconditional loads, masked values
            w[u] = W7[(size_t)oo * ldw + cc];
            a[u] = ga[oo];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = __builtin_fmaf(o + u < o1 ? a[u] : 0.0f, w[u], s);
    }
    part[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && c < emb) {
        float t = part[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += part[i][threadIdx.x];
        gg[(size_t)b * emb + c] = t;
    }
}

// gW7 [c7][emb + F]: columns < emb = gadd7^T g (a sum over the B batch elements, in order), the others = the head tail's gW7c
__global__ __launch_bounds__(256) void gw7_kernel(const float *__restrict__ gadd7, const float *__restrict__ g,
                                                  const float *__restrict__ gw7c, int B, int emb, int c7, int F, float *__restrict__ gw7)
{
    const int ld = emb + F;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)c7 * ld) return;
    const int o = (int)(t / ld), c = (int)(t - (size_t)o * ld);
    float s;
    if (c < emb) {
        s = 0.0f;
        for (int b = 0; b < B; ++b) s = __builtin_fmaf(gadd7[(size_t)b * c7 + o], g[(size_t)b * emb + c], s);
    } else {
        s = gw7c[(size_t)o * F + (c - emb)];
    }
    gw7[t] = s;
}

}  // namespace
}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_model_saved_bytes(const dgcnn_model_desc *d)
{
    return check_model(d, "dgcnn_model_saved_bytes", false) == 0 ? saved_layout(d, nullptr).bytes : 0;
}
extern "C" size_t dgcnn_model_fwd_workspace_bytes(const dgcnn_model_desc *d)
{
    return check_model(d, "dgcnn_model_fwd_workspace_bytes", false) == 0 ? fwd_ws(d, nullptr).bytes : 0;
}
extern "C" size_t dgcnn_model_bwd_workspace_bytes(const dgcnn_model_desc *d)
{
    if (check_model(d, "dgcnn_model_bwd_workspace_bytes", false)) return 0;
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        if (dgcnn_edgeconv_bwd_workspace_bytes(&e) == 0) return 0;  // (its own message: N above the reverse-list limit)
    }
    return bwd_ws(d, nullptr).bytes;
}

// the backward builds the reverse lists of the three graphs in one go (see dgcnn_model_bwd) -- where the blocks' training
// forwards do not prepare them (edgeconv.hip, lists_in_forward: then the list passes ride in launches that run anyway and the
// backward has nothing to build).  The forward must know the answer too (EdgeLd::no_fwd_lists).
static bool lists_merged(const dgcnn_model_desc *d)
{
    const size_t En = (size_t)d->B * d->N * d->k;
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        if (edgeconv_lists_in_forward(&e)) return false;
    }
    return d->idx[1] == d->idx[0] + En && d->idx[2] == d->idx[1] + En && (size_t)3 * En < ((size_t)1 << 31) &&
           (size_t)3 * En * sizeof(int) <= (size_t)8 * 3584 * 1024;
}

extern "C" int dgcnn_model_fwd(const dgcnn_model_desc *d, const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn, float *logits,
                               int64_t l_sb, int64_t l_sc, int64_t l_sn, void *saved, void *workspace, size_t workspace_bytes,
                               void *stream)
{
    if (check_model(d, "dgcnn_model_fwd", true)) return 1;
    DG_REQUIRE(x && logits && saved && workspace, "dgcnn_model_fwd: null pointer");
    const FwdWs w = fwd_ws(d, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_model_fwd: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    const Saved s = saved_layout(d, saved);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, N = d->N, P = B * N, emb = d->emb;

    const float *xpm = x;
    if (!is_point_major(d, x_sb, x_sc, x_sn)) {
        if (dgcnn_to_point_major(x, B, d->F, N, x_sb, x_sc, x_sn, s.xpm, stream)) return 1;
        xpm = s.xpm;
    }
    const int nolists = lists_merged(d) ? 1 : 0;
    for (int b = 0; b < 3; ++b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        const float *in = b == 0 ? xpm : s.cat + EC * (b - 1);
        const int ldin = b == 0 ? 0 : CATW;
        if (!d->idx_given && knn_ld(in, ldin, B, N, e.C, d->k, d->idx[b], w.scratch, w.scratch_bytes, stream)) return 1;
        if (edgeconv_fwd_ld(&e, EdgeLd{ldin, CATW, 0, 0, 0, nullptr, nolists}, in, d->idx[b], s.cat + EC * b, s.edge[b], w.scratch, w.scratch_bytes, stream))
            return 1;
    }
    // max |cat| (the operand scale of every product that reads cat) and its column sums (conv6's backward)
    if (dgcnn_rows_stats(s.cat, CATW, P, CATW, s.amax_cat, s.catsum, w.scratch, w.scratch_bytes, stream)) return 1;
    dgcnn_bn_params bn6 = d->bn[5];
    bn6.num_batches_tracked = nullptr;
    if (dgcnn_embed_fwd(s.cat, CATW, d->w[5], CATW, B, N, CATW, emb, s.amax_cat, &bn6, d->training, d->eps[5], d->momentum[5], d->slope,
                        s.g, s.arg, s.uwin, s.stats6, w.scratch, w.scratch_bytes, stream))
        return 1;
    const bool bump = d->training != 0;
    hipLaunchKernelGGL(pooled_fwd_kernel, dim3(ceil_div(B * d->c7, 4)), dim3(256), 0, st, s.g, d->w[6], emb + CATW, B, emb, d->c7, s.add7,
                       bump ? d->bn[5].num_batches_tracked : nullptr, bump ? d->bn[6].num_batches_tracked : nullptr,
                       bump ? d->bn[7].num_batches_tracked : nullptr);
    DG_CHECK_LAUNCH("pooled_fwd_kernel");
    const dgcnn_head_tail_desc t = tail_desc(d);
    if (dgcnn_head_tail_fwd(&t, s.cat, s.amax_cat, s.add7, w.logits_pm, s.tail, w.scratch, w.scratch_bytes, stream)) return 1;
    return dgcnn_from_point_major(w.logits_pm, B, d->num_classes, N, logits, l_sb, l_sc, l_sn, stream);
}

extern "C" int dgcnn_model_bwd(const dgcnn_model_desc *d, const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn,
                               const float *glogits, int64_t g_sb, int64_t g_sc, int64_t g_sn, const void *saved,
                               const dgcnn_model_grads *g, void *workspace, size_t workspace_bytes, int phases, void *stream)
{
    // phases: bit 0 = the head (everything up to dL/dcat: afterwards the gradients of conv6 .. conv9 and bn6 .. bn8 are
    // final), bit 1 = the three EdgeConv blocks.  A data-parallel caller runs them as two calls on the same workspace and starts
    // the all-reduce of the head's 97 % of the gradient bytes in between, under the blocks' backward.
    DG_REQUIRE(phases >= 1 && phases <= 3, "dgcnn_model_bwd: phases must be 1 (head), 2 (blocks) or 3 (both)");
    if (check_model(d, "dgcnn_model_bwd", true)) return 1;
    DG_REQUIRE(x && glogits && saved && g && workspace, "dgcnn_model_bwd: null pointer");
    for (int i = 0; i < 9; ++i) DG_REQUIRE(g->gw[i], "dgcnn_model_bwd: gradient output gw[%d] missing", i);
    for (int i = 0; i < 8; ++i) DG_REQUIRE(g->gbn_w[i] && g->gbn_b[i], "dgcnn_model_bwd: gradient output of bn%d missing", i + 1);
    const size_t need = dgcnn_model_bwd_workspace_bytes(d);
    if (need == 0) return 1;  // (the workspace query has left its message)
    DG_REQUIRE(workspace_bytes >= need, "dgcnn_model_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    const BwdWs w = bwd_ws(d, workspace);
    const Saved s = saved_layout(d, const_cast<void *>(saved));
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, N = d->N, emb = d->emb, c7 = d->c7;
    const float *xpm = is_point_major(d, x_sb, x_sc, x_sn) ? x : s.xpm;

    // the reverse neighbour lists of all three blocks (they depend on the graphs alone) in three launches instead of nine, when
    // the caller keeps the three graphs in one array (ops.dgcnn_full does; graphs injected one by one take the per-block build)
    const size_t Pn = (size_t)B * N, En = Pn * d->k;
    // ... and when the lists under construction fit the L2s: the placement pass merges its scattered 4-byte stores into whole
    // lines behind ONE XCD's L2 per batch element (edge_stream.hip); a merged build has 3 B / 8 list regions of N k ints in
    // flight per XCD.  At B = 32 that is 10 MB per 4 MB L2 and the merged placement took 285 us against 3 x 34 (measured,
    // profiles/r4_90_train_step_kernel_stats.csv); up to ~3.5 MB per XCD the merged form wins (B = 4: 101 -> 68 us for the
    // three builds).
    const bool merged = lists_merged(d);
    if (phases & 1) {
    if (merged && launch_rev_build_merged(d->idx[0], 3, B, N, d->k, w.rcnt, w.rindeg, w.rrowptr, w.rsegbase, w.rpcnt, w.rrev, w.rextra,
                                          w.rextra_stride, w.rzero, Z_WORDS, Z_ALLOC, st))
        return 1;
    if (dgcnn_to_point_major(glogits, B, d->num_classes, N, g_sb, g_sc, g_sn, w.gl_pm, stream)) return 1;
    const dgcnn_head_tail_desc t = tail_desc(d);
    dgcnn_head_tail_grads tg{};
    tg.gcat = w.gcat, tg.ldgcat = CATW, tg.gadd7 = w.gadd7, tg.gw7c = w.gw7c;
    tg.gbn7_w = g->gbn_w[6], tg.gbn7_b = g->gbn_b[6], tg.gw8 = g->gw[7], tg.gbn8_w = g->gbn_w[7], tg.gbn8_b = g->gbn_b[7];
    tg.gw9 = g->gw[8];
    if (dgcnn_head_tail_bwd(&t, s.cat, s.amax_cat, w.gl_pm, s.tail, &tg, w.scratch, w.scratch_bytes, stream)) return 1;
    hipLaunchKernelGGL(pooled_bwd_kernel, dim3(ceil_div(emb, 32), B), dim3(256), 0, st, w.gadd7, d->w[6], emb + CATW, B, emb, c7, w.gg);
    hipLaunchKernelGGL(gw7_kernel, dim3((unsigned)(((size_t)c7 * (emb + CATW) + 255) / 256)), dim3(256), 0, st, w.gadd7, s.g, w.gw7c, B,
                       emb, c7, CATW, g->gw[6]);
    DG_CHECK_LAUNCH("pooled_bwd_kernel / gw7_kernel");
    if (embed_bwd_ld(s.cat, CATW, s.amax_cat, s.catsum, d->w[5], CATW, s.arg, s.uwin, s.stats6, w.gg, B, N, CATW, emb, d->training,
                     d->slope, w.gcat, CATW, 1, g->gw[5], g->gbn_w[5], g->gbn_b[5], w.scratch, w.scratch_bytes, stream))
        return 1;
    }  // phase 1
    if (!(phases & 2)) return 0;
    for (int b = 2; b >= 0; --b) {
        const dgcnn_edgeconv_desc e = block_desc(d, b);
        dgcnn_edgeconv_grads eg{};
        eg.gx = b > 0 ? w.gxd : nullptr;  // dense; block b - 1 adds it to its slice of gcat in its first kernel (bwd_point)
        eg.gw1 = g->gw[2 * b], eg.gbn1_w = g->gbn_w[2 * b], eg.gbn1_b = g->gbn_b[2 * b];
        if (b < 2) eg.gw2 = g->gw[2 * b + 1], eg.gbn2_w = g->gbn_w[2 * b + 1], eg.gbn2_b = g->gbn_b[2 * b + 1];
        const float *in = b == 0 ? xpm : s.cat + EC * (b - 1);
        const EdgeLd ld{b == 0 ? 0 : CATW, CATW, CATW, 0, 0, b < 2 ? w.gxd : nullptr, merged ? 1 : 0};
        const RevPrebuilt pre{w.rzero + b * Z_WORDS, w.rindeg + b * Pn, w.rrowptr + b * Pn, w.rsegbase + b * Pn, w.rpcnt + b * Pn,
                              w.rrev + b * En,   w.rextra + (size_t)b * w.rextra_stride};
        if (edgeconv_bwd_ld(&e, ld, in, d->idx[b], s.cat + EC * b, w.gcat + EC * b, s.edge[b], &eg, w.scratch, w.scratch_bytes, stream,
                            merged ? &pre : nullptr))
            return 1;
    }
    return 0;
}
