// edgeconv.hip -- C ABI of the EdgeConv block: validates the descriptor, carves the caller's
// workspace and enqueues the kernel sequence on the caller's stream (no allocation, no sync).
//
// Forward (reference model.py:147-150 / 154-157 / 161-163), P = B N points, E = P k edges; training mode:
//   rowgemm_fwd      TP = x Wa^T, TQ = x (Wb - Wa)^T  (MFMA, per point; Wcat = [Wa ; Wb - Wa] built in LDS from W1 = [Wa | Wb];
//                    also zeroes the counters of the statistics passes below)
//   edge_reduce      u = TP[n] + TQ[i]: BN1 batch sums -- finished in the same launch (fold.h): mean / rstd / affine,
//                    running statistics -- and, for a one-stage block, the selected extreme over k
//   2 stages: edge_mlp_fwd  a1 = LReLU(BN1(u)); u2 = W2 a1 (MFMA, per edge); BN2 sums (finished in the launch) + extreme
//   edge_out         y = LReLU(BN(extreme))      == max over k of the activations, exactly
// (eval mode: the affine pairs come from the running statistics, written by workgroup 0 of the per-point product.)
//
// Backward (autograd nodes listed in SURVEY.md 2.1 "Bw"; formulas Appendix C.5):
//   rev_count, rev_slots, rev_place   reverse adjacency of the neighbour graph, lists sorted by edge id (the first
//                    kernel zeroes the counters used below)
//   bwd_point        last BN: gbn_w, gbn_b and the two mean terms, finished in the same launch
//   2 stages: edge_mlp_bwd  recompute a1, u2; du2; gW2 partials; dv1 = (W2^T du2) * LReLU'; BN1 sums finished in the launch
//             sum_partials  gW2
//   edge_point_bwd   dD (sum over k), per-edge hit records and G (one-stage)
//   edge_rgather     dR (gather over the reverse lists: fixed order, no atomics)
//   wgrad + finish   gW1 = [ (dR - dD)^T x | dD^T x ]                           (MFMA, per point)
//   rowgemm          gx = dR Wa + dD (Wb - Wa)                                   (MFMA, per point)
#include <stdlib.h>

#include "edge_kernels.h"

namespace dgcnn {


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
};

struct SavedLayout {
    float *TP, *TQ, *usel, *usum, *stats1, *stats2, *wcat;
    uint8_t *hitj;
    // lists_in_forward: the first-slot tables of the reverse lists, their per-point arrays (list start, in-degree, first extra
    // segment, arrival counter), the extra (point, segment) pairs, and the backward's counters (zeroed by the forward)
    int *cnt, *bzero, *rowptr, *indeg, *segbase, *pcnt, *extra;
    size_t bytes;
};
// Training forward of a 64-channel block: the first two passes of the backward's reverse neighbour lists ride in the forward --
// the counting pass in the per-point product's launch (pointgemm.hip, rev_count_rider: a kernel bound by its matrix
// instructions next to one bound by its LDS atomics), the list-space pass in the gather / reduce pass's (rev_slots_pass) -- and
// the third, the placement, rides in the backward's first launch (bwd_point).  What the forward leaves waits in the saved
// state: the first-slot tables [B rev_chunks][N], the per-point arrays, the extra-segment pairs and the backward's counters
// (zeroed by the forward; the list allocation pair keeps its values, the rest is zeroed again by the backward's last launch,
// the arrival counters by their last arrivers: a second backward from the same saved state finds everything as the first did).
// A function of the descriptor alone: the two calls must agree.
static bool lists_in_forward(const dgcnn_edgeconv_desc *d)
{
    return d->training && rowgemm_fwd_carries_count(d->C, d->N, d->k);
}
bool edgeconv_lists_in_forward(const dgcnn_edgeconv_desc *d) { return lists_in_forward(d); }
constexpr int LISTS_RIDE_BELOW = 16;  // batch elements from which on the list-space and placement passes run as launches of their own
static SavedLayout saved_layout(const dgcnn_edgeconv_desc *d, void *p)
{
    Carver c(p);
    SavedLayout s;
    const size_t P = (size_t)d->B * d->N;
    s.TP = c.take<float>(P * EC);
    s.TQ = c.take<float>(P * EC);
    s.usel = c.take<float>(P * EC);
    s.usum = c.take<float>(P * EC);   // sum over k of the stage-1 pre-activation (backward, training mode)
    s.hitj = c.take<uint8_t>(P * EC);  // slot of the edge that wins max-over-k (of the block's last stage)
    s.stats1 = c.take<float>(ST_SIZE);
    s.stats2 = c.take<float>(ST_SIZE);
    s.wcat = c.take<float>((size_t)128 * d->C);  // [Wa ; Wb - Wa] of the forward, reused by the backward's dgrad
    if (lists_in_forward(d)) {
        s.cnt = c.take<int>((size_t)d->B * rev_chunks(d->B, d->N, d->k) * d->N);
        s.bzero = c.take<int>(Z_WORDS);
        s.rowptr = c.take<int>(P);
        s.indeg = c.take<int>(P);
        s.segbase = c.take<int>(P);
        s.pcnt = c.take<int>(P);
        s.extra = c.take<int>(2 * (P * d->k / REV_SEG + 1));
    } else {
        s.cnt = s.bzero = s.rowptr = s.indeg = s.segbase = s.pcnt = s.extra = nullptr;
    }
    s.bytes = c.off;
    return s;
}

struct FoldBufs {
    double *rec, *folded;
};
template <typename C>
static FoldBufs take_fold(C &c)
{
    FoldBufs f;
    f.rec = c.template take<double>((size_t)PART_MAX * 2 * 64);
    f.folded = c.template take<double>((size_t)FOLD_GROUPS * 2 * 64);
    return f;
}

struct FwdWs {
    int *zero;  // [Z_WORDS]
    FoldBufs f1, f2;
    size_t bytes;
};
static FwdWs fwd_ws(const dgcnn_edgeconv_desc *d, void *p)
{
    Carver c(p);
    FwdWs w;
    w.zero = c.take<int>(Z_WORDS);
    w.f1 = take_fold(c);
    w.f2 = d->n_stage == 2 ? take_fold(c) : FoldBufs{nullptr, nullptr};
    w.bytes = c.off;
    return w;
}

struct BwdWs {
    int *zero;  // [Z_WORDS]: fold counters, list allocation
    FoldBufs fa, fb;
    float *cc1, *cc2, *dR, *dD, *wg_partial, *dv1, *dvsum, *dw2_partial, *G, *part0, *partx, *gdv;
    int *cnt, *indeg, *rowptr, *segbase, *pcnt, *rev, *extra;  // reverse adjacency of the neighbour graph
    void *hits;
    size_t bytes;
};
static BwdWs bwd_ws(const dgcnn_edgeconv_desc *d, void *p)
{
    Carver c(p);
    BwdWs w;
    const size_t P = (size_t)d->B * d->N;
    w.zero = c.take<int>(Z_WORDS);
    w.fa = take_fold(c);
    w.fb = d->n_stage == 2 ? take_fold(c) : FoldBufs{nullptr, nullptr};
    w.cc1 = c.take<float>(128);
    w.cc2 = c.take<float>(128);
    w.dR = c.take<float>(P * EC);
    w.dD = c.take<float>(P * EC);
    w.gdv = c.take<float>(P * EC);   // dL/dy . LReLU'(y), left by bwd_point for every kernel behind it
    w.wg_partial = c.take<float>((size_t)wgrad_chunks((int)P, d->C) * 128 * d->C);
    w.cnt = c.take<int>((size_t)d->B * rev_chunks(d->B, d->N, d->k) * d->N);
    w.indeg = c.take<int>(P);
    w.rowptr = c.take<int>(P);
    w.segbase = c.take<int>(P);
    w.pcnt = c.take<int>(P);
    w.rev = c.take<int>(P * d->k);
    w.extra = c.take<int>(2 * (P * d->k / REV_SEG + 1));
    w.part0 = c.take<float>(P * 128);
    w.partx = c.take<float>((P * d->k / REV_SEG + 1) * 128);
    w.hits = d->n_stage == 1 ? (void *)c.take<u64>(P * d->k * 4) : nullptr;  // 32 bytes per edge
    w.G = nullptr;   // (the per-point copy sc . gdv of the sparse backward: the gather reads gdv itself since round 4)
    if (d->n_stage == 2) {
        w.dv1 = c.take<float>(P * d->k * EC);
        w.dvsum = c.take<float>(P * EC);
        w.dw2_partial = c.take<float>((size_t)PART_MAX * EC * EC);
    } else {
        w.dv1 = nullptr;
        w.dvsum = nullptr;
        w.dw2_partial = nullptr;
    }
    w.bytes = c.off;
    return w;
}

// shape limits of the block (the workspace queries check these only: they are asked before any parameter exists)
static int check_shape(const dgcnn_edgeconv_desc *d, const char *who, bool backward)
{
    DG_REQUIRE(d, "%s: null descriptor", who);
    DG_REQUIRE(d->B > 0 && d->N > 0 && d->C > 0 && d->k > 0, "%s: bad shape B=%d N=%d C=%d k=%d", who, d->B, d->N,
               d->C, d->k);
    DG_REQUIRE(d->C <= 64, "%s: C=%d > 64 unsupported", who, d->C);
    DG_REQUIRE(d->k <= 96, "%s: k=%d > 96 unsupported", who, d->k);
    DG_REQUIRE((size_t)d->B * d->N * d->k < ((size_t)1 << 31), "%s: more than 2^31 edges", who);
    DG_REQUIRE((size_t)d->B * d->N < ((size_t)1 << 24), "%s: more than 2^24 points", who);  // 32-bit byte offsets of rows
    DG_REQUIRE(d->n_stage == 1 || d->n_stage == 2, "%s: n_stage must be 1 or 2", who);
    DG_REQUIRE(d->n_stage == 1 || d->k <= 64, "%s: two-stage blocks support k <= 64 (k=%d)", who, d->k);
    // every backward (training or eval) gathers over the reverse neighbour lists, whose kernels hold N ints in LDS
    DG_REQUIRE(!backward || d->N <= REV_MAX_N, "%s: N=%d > %d unsupported in the backward", who, d->N, REV_MAX_N);
    return 0;
}

static int check_desc(const dgcnn_edgeconv_desc *d, const char *who, bool backward = false)
{
    if (check_shape(d, who, backward)) return 1;
    // (the block takes max over k BEFORE BatchNorm's affine map and the activation, on the side the sign of gamma
    // selects: the activation has to be non-decreasing)
    DG_REQUIRE(d->slope >= 0.0f, "%s: LeakyReLU slope %g < 0 unsupported", who, (double)d->slope);
    DG_REQUIRE(d->w1 && d->bn1.weight && d->bn1.bias, "%s: stage-1 parameters missing", who);
    DG_REQUIRE(d->training || (d->bn1.running_mean && d->bn1.running_var), "%s: eval mode needs running statistics",
               who);
    if (d->n_stage == 2) {
        DG_REQUIRE(d->w2 && d->bn2.weight && d->bn2.bias, "%s: stage-2 parameters missing", who);
        DG_REQUIRE(d->training || (d->bn2.running_mean && d->bn2.running_var),
                   "%s: eval mode needs running statistics", who);
    }
    return 0;
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_edgeconv_saved_bytes(const dgcnn_edgeconv_desc *d)
{
    return d ? saved_layout(d, nullptr).bytes : 0;
}
extern "C" size_t dgcnn_edgeconv_fwd_workspace_bytes(const dgcnn_edgeconv_desc *d)
{
    return d ? fwd_ws(d, nullptr).bytes : 0;
}
static int bwd_query_ok(const dgcnn_edgeconv_desc *d) { return check_shape(d, "dgcnn_edgeconv_bwd_workspace_bytes", true); }
extern "C" size_t dgcnn_edgeconv_bwd_workspace_bytes(const dgcnn_edgeconv_desc *d)
{
    // 0 = unsupported shape (dgcnn_last_error says which limit): the backward itself would refuse it
    return bwd_query_ok(d) == 0 ? bwd_ws(d, nullptr).bytes : 0;
}

extern "C" int dgcnn_edgeconv_fwd(const dgcnn_edgeconv_desc *d, const float *x, const int32_t *idx, float *y,
                                  void *saved, void *workspace, size_t workspace_bytes, void *stream)
{
    return edgeconv_fwd_ld(d, EdgeLd{0, 0, 0, 0, 0, nullptr}, x, idx, y, saved, workspace, workspace_bytes, stream);
}

// the forward with row strides: x and y may be 64-column slices of wider rows (model.hip: the [P][192] concatenation of the
// block outputs, which the blocks then write and read in place -- the torch.cat of model.py:167 disappears)
int dgcnn::edgeconv_fwd_ld(const dgcnn_edgeconv_desc *d, const EdgeLd &ld, const float *x, const int32_t *idx, float *y,
                           void *saved, void *workspace, size_t workspace_bytes, void *stream)
{
    if (check_desc(d, "dgcnn_edgeconv_fwd")) return 1;
    DG_REQUIRE((ld.x == 0 || ld.x >= d->C) && (ld.y == 0 || ld.y >= EC) && (ld.y & 3) == 0, "dgcnn_edgeconv_fwd: bad row strides");
    DG_REQUIRE(x && idx && y && saved && workspace, "dgcnn_edgeconv_fwd: null pointer");
    const FwdWs w = fwd_ws(d, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_edgeconv_fwd: workspace too small (%zu < %zu)", workspace_bytes,
               w.bytes);
    const SavedLayout s = saved_layout(d, saved);
    hipStream_t st = (hipStream_t)stream;
    const int P = d->B * d->N, N = d->N, C = d->C, k = d->k;
    const double E = (double)P * k;

    FwdSide sd{};
    sd.wcat = s.wcat, sd.zero_words = w.zero, sd.nzero = Z_WORDS;
    if (!d->training) {  // BatchNorm constants from the running statistics: written by workgroup 0 of the per-point product
        sd.nbn = d->n_stage, sd.eps = d->eps;
        sd.bn[0] = d->bn1, sd.stats[0] = s.stats1;
        sd.bn[1] = d->bn2, sd.stats[1] = s.stats2;
    }
    const bool lists = lists_in_forward(d) && !ld.no_fwd_lists;
    if (lists) {
        const int chunks = rev_chunks(d->B, N, k);
        sd.rc = RevCountRider{idx, N, k, chunks, ceil_div(N, chunks), d->B * chunks, 0, s.cnt, s.bzero, Z_WORDS};
    }
    // (the list-space pass rides below sixteen batch elements only: at B = 32 its riders cost the gather / reduce pass 23 us per
    // launch, more than the 18 us launch they replace; the backward then runs it from the forward's counts)
    const RevSlotsArgs slots_ = rev_slots_args(d->B, N, k, s.cnt, s.indeg, s.rowptr, s.segbase, s.pcnt, s.bzero + Z_ALLOC, s.extra);
    const RevSlotsArgs *slots = lists && d->B < LISTS_RIDE_BELOW ? &slots_ : nullptr;   // (training mode: both forms of the block run edge_reduce)
    launch_rowgemm_fwd(x, C, d->w1, s.TP, s.TQ, sd, P, st, ld.x);
    const FoldSet fs1{w.f1.rec, w.f1.folded, w.zero + Z_FOLD_A}, fs2{w.f2.rec, w.f2.folded, w.zero + Z_FOLD_B};
    const BnBatchStats fin1{d->bn1, E, d->eps, d->momentum, s.stats1}, fin2{d->bn2, E, d->eps, d->momentum, s.stats2};
    if (d->n_stage == 1) {
        launch_edge_reduce(s.TP, s.TQ, idx, N, k, P, d->bn1.weight, s.usel, s.usum, s.hitj, fs1, d->training ? &fin1 : nullptr,
                           st, slots);
        launch_edge_out(s.usel, s.stats1, d->slope, P, y, st, ld.y);
    } else {
        if (d->training) launch_edge_reduce(s.TP, s.TQ, idx, N, k, P, d->bn1.weight, nullptr, s.usum, nullptr, fs1, &fin1, st, slots);
        const FoldSet fs3{w.f2.rec, w.f2.folded, w.zero + Z_FOLD_C};
        launch_edge_mlp_fwd(s.TP, s.TQ, idx, N, k, P, s.stats1, d->w2, d->bn2.weight, d->slope, s.usel, s.hitj, fs2, fs3,
                            d->training ? &fin2 : nullptr, w.zero + Z_GUARD, st);
        launch_edge_out(s.usel, s.stats2, d->slope, P, y, st, ld.y);
    }
    DG_CHECK_LAUNCH("dgcnn_edgeconv_fwd");
    return 0;
}

extern "C" int dgcnn_edgeconv_bwd(const dgcnn_edgeconv_desc *d, const float *x, const int32_t *idx, const float *y,
                                  const float *gy, const void *saved, const dgcnn_edgeconv_grads *g, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    return edgeconv_bwd_ld(d, EdgeLd{0, 0, 0, 0, 0, nullptr}, x, idx, y, gy, saved, g, workspace, workspace_bytes, stream, nullptr);
}

// the backward with row strides (x, y, gy, gx as 64-column slices of wider rows) and, with ld.gx_acc, gx ADDED to what the
// slice holds: the second gradient that reaches a block's output (model.py:167 / :180 feed x1, x2 to the head as well)
int dgcnn::edgeconv_bwd_ld(const dgcnn_edgeconv_desc *d, const EdgeLd &ld, const float *x, const int32_t *idx, const float *y,
                           const float *gy, const void *saved, const dgcnn_edgeconv_grads *g, void *workspace,
                           size_t workspace_bytes, void *stream, const RevPrebuilt *pre)
{
    if (check_desc(d, "dgcnn_edgeconv_bwd", true)) return 1;
    DG_REQUIRE((ld.x == 0 || ld.x >= d->C) && (ld.y == 0 || ld.y >= EC) && (ld.gy == 0 || ld.gy >= EC) && (ld.gx == 0 || ld.gx >= d->C),
               "dgcnn_edgeconv_bwd: bad row strides");
    DG_REQUIRE(x && idx && y && gy && saved && g && workspace, "dgcnn_edgeconv_bwd: null pointer");
    DG_REQUIRE(g->gw1 && g->gbn1_w && g->gbn1_b, "dgcnn_edgeconv_bwd: stage-1 gradient outputs missing");
    if (d->n_stage == 2) DG_REQUIRE(g->gw2 && g->gbn2_w && g->gbn2_b, "dgcnn_edgeconv_bwd: stage-2 gradient outputs missing");
    BwdWs w = bwd_ws(d, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_edgeconv_bwd: workspace too small (%zu < %zu)", workspace_bytes,
               w.bytes);
    const SavedLayout s = saved_layout(d, const_cast<void *>(saved));
    hipStream_t st = (hipStream_t)stream;
    const int P = d->B * d->N, N = d->N, C = d->C, k = d->k;
    const double E = (double)P * k;
    if (pre) {  // the lists exist (launch_rev_build_merged, which has also zeroed the call's counters)
        w.zero = pre->zero, w.indeg = pre->indeg, w.rowptr = pre->rowptr, w.segbase = pre->segbase, w.pcnt = pre->pcnt;
        w.rev = pre->rev, w.extra = pre->extra;
    }

    // (measured: building the lists on a side stream, concurrently with the BatchNorm-backward sums and the
    // per-point pass, costs more in cross-stream barriers than it hides: 0.776 ms against 0.725 ms per step)
    // The reverse adjacency: its counting pass first (it also zeroes the call's counters), its list-space pass inside the
    // launch of bwd_point (two short passes that leave most of the chip idle on their own), then the placement.
    // Large batches (>= 16 batch elements): the list passes that run in this call back to back, in front of bwd_point.  Riding
    // in bwd_point's launch, the placement pass meets the 57 MB of dL/dy . LReLU'(y) that bwd_point streams at B = 32: its
    // scattered stores then run into that write-back (110 us against 80 per launch).  Small batches ride.
    const bool lists_first = !pre && d->B >= LISTS_RIDE_BELOW;
    // the forward has counted (lists_in_forward) and, below sixteen batch elements, allotted the lists: then only the placement
    // is left -- in bwd_point's launch; from sixteen on the list-space pass and the placement run here, from the forward's counts
    const bool counted = !pre && lists_in_forward(d) && !ld.no_fwd_lists;
    const bool allotted = counted && !lists_first;
    if (allotted) {
        w.zero = s.bzero, w.cnt = s.cnt, w.indeg = s.indeg, w.rowptr = s.rowptr, w.segbase = s.segbase, w.pcnt = s.pcnt;
        w.extra = s.extra;
    } else if (counted) {
        w.zero = s.bzero;
        if (launch_rev_slots_place(idx, d->B, N, k, s.cnt, w.cnt, w.indeg, w.rowptr, w.segbase, w.pcnt, w.rev, w.zero + Z_ALLOC, w.extra, st))
            return 1;
    } else if (!pre) {
        if (launch_rev_build(idx, d->B, N, k, w.cnt, w.indeg, w.rowptr, w.segbase, w.pcnt, w.rev, w.zero + Z_ALLOC, w.extra, w.zero,
                             Z_WORDS, lists_first, lists_first, st))
            return 1;
    }
    const RevSlotsArgs slots_ = rev_slots_args(d->B, N, k, w.cnt, w.indeg, w.rowptr, w.segbase, w.pcnt, w.zero + Z_ALLOC, w.extra);
    const RevSlotsArgs *slots = (pre || lists_first || counted) ? nullptr : &slots_;
    const bool place_later = !pre && !lists_first && !counted;
    const int rchunks = rev_chunks(d->B, N, k);
    const RevPlaceArgs place_{idx, N, k, rchunks, ceil_div(N, rchunks), d->B * rchunks, w.cnt, w.rev, d->B, 0};
    const RevPlaceArgs *place = allotted ? &place_ : nullptr;
    const FoldSet fsa{w.fa.rec, w.fa.folded, w.zero + Z_FOLD_A}, fsb{w.fb.rec, w.fb.folded, w.zero + Z_FOLD_B};
    if (d->n_stage == 1) {
        launch_bwd_point(gy, s.usel, s.stats1, d->slope, P, E, d->training, fsa, g->gbn1_w, g->gbn1_b, w.cc1, nullptr, slots, st,
                         ld.gy, ld.gy2, w.gdv, place);
        if (place_later && launch_rev_place(idx, d->B, N, k, w.cnt, w.rev, st)) return 1;
        if (launch_edge_bwd(s.TP, s.TQ, d->B, N, k, s.stats1, w.cc1, d->slope, w.gdv, s.hitj, s.usum, nullptr, nullptr,
                            w.rowptr, w.indeg, w.segbase, w.pcnt, w.rev, w.zero + Z_ALLOC, w.extra, w.hits, w.gdv, w.part0,
                            w.partx, w.dR, w.dD, st))
            return 1;
    } else {
        launch_bwd_point(gy, s.usel, s.stats2, d->slope, P, E, d->training, fsa, g->gbn2_w, g->gbn2_b, w.cc2,
                         reinterpret_cast<unsigned *>(w.zero + Z_GMAX), slots, st, ld.gy, ld.gy2, w.gdv, place);
        if (place_later && launch_rev_place(idx, d->B, N, k, w.cnt, w.rev, st)) return 1;
        const BnBackwardSums fin1{E, d->training, g->gbn1_w, g->gbn1_b, w.cc1};
        const FoldSet fsc{w.fb.rec, w.fb.folded, w.zero + Z_FOLD_C};
        if (launch_edge_mlp_bwd(s.TP, s.TQ, idx, N, k, P, s.stats1, s.stats2, w.cc2, d->w2, d->slope, d->training, w.gdv,
                                s.hitj, reinterpret_cast<const unsigned *>(w.zero + Z_GMAX), w.dv1, w.dvsum, w.dw2_partial, fsb,
                                fsc, fin1, w.zero + Z_GUARD, st))
            return 1;
        // (the fold of the gW2 partials rides in the per-point pass's launch)
        const SumRider fold_gw2{w.dw2_partial, mlp_grid(P), EC * EC, g->gw2, 0};
        if (launch_edge_bwd(s.TP, s.TQ, d->B, N, k, s.stats1, w.cc1, d->slope, w.gdv, nullptr, s.usum, w.dv1, w.dvsum,
                            w.rowptr, w.indeg, w.segbase, w.pcnt, w.rev, w.zero + Z_ALLOC, w.extra, nullptr, nullptr,
                            w.part0, w.partx, w.dR, w.dD, st, &fold_gw2))
            return 1;
    }
    if (g->gx && dgrad_wgrad64_fused(C)) {   // the two products of the first convolution's backward in one launch, then the fold
        launch_dgrad_wgrad64(w.dR, w.dD, s.wcat, g->gx, ld.gx, ld.gx_acc, P, x, ld.x, w.wg_partial, g->gw1, st, counted ? s.bzero : nullptr,
                             allotted ? Z_ALLOC : Z_WORDS);
        DG_CHECK_LAUNCH("dgcnn_edgeconv_bwd");
        return 0;
    }
    launch_wgrad(w.dR, EC, w.dD, EC, x, C, P, w.wg_partial, st, ld.x);
    // (the counters in the saved state are left at zero for the next backward; allotted lists keep their allocation pair)
    int *rezero = counted ? s.bzero : nullptr;
    const int nrezero = allotted ? Z_ALLOC : Z_WORDS;
    if (g->gx)
        launch_dgrad_and_dw1(w.dR, w.dD, s.wcat, C, g->gx, ld.gx, ld.gx_acc, P, w.wg_partial, wgrad_chunks(P, C), g->gw1, st, rezero, nrezero);
    else
        launch_finish_dw1(w.wg_partial, wgrad_chunks(P, C), C, g->gw1, st, rezero, nrezero);
    DG_CHECK_LAUNCH("dgcnn_edgeconv_bwd");
    return 0;
}

extern "C" int dgcnn_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k, float *out,
                                       void *stream)
{
    DG_REQUIRE(x && idx && out, "dgcnn_graph_feature_fwd: null pointer");
    DG_REQUIRE(B > 0 && N > 0 && C > 0 && k > 0, "dgcnn_graph_feature_fwd: bad shape");
    launch_graph_feature_fwd(x, idx, B, N, C, k, out, (hipStream_t)stream);
    DG_CHECK_LAUNCH("graph_feature_fwd_kernel");
    return 0;
}

struct GfWs {
    int *zero, *cnt, *indeg, *rowptr, *segbase, *pcnt, *rev, *extra;
    size_t bytes;
};
static GfWs gf_ws(int B, int N, int k, void *p)
{
    Carver c(p);
    GfWs w;
    const size_t P = (size_t)B * N;
    w.zero = c.take<int>(Z_WORDS);
    w.cnt = c.take<int>((size_t)B * rev_chunks(B, N, k) * N);
    w.indeg = c.take<int>(P);
    w.rowptr = c.take<int>(P);
    w.segbase = c.take<int>(P);
    w.pcnt = c.take<int>(P);
    w.rev = c.take<int>(P * k);
    w.extra = c.take<int>(2 * (P * k / REV_SEG + 1));
    w.bytes = c.off;
    return w;
}

static int gf_bwd_shape_ok(int B, int N, int C, int k, const char *who)
{
    DG_REQUIRE(B > 0 && N > 0 && C > 0 && k > 0, "%s: bad shape", who);
    DG_REQUIRE(k <= 128, "%s: k=%d > 128 unsupported", who, k);
    DG_REQUIRE(N <= REV_MAX_N, "%s: N=%d > %d unsupported (reverse neighbour lists)", who, N, REV_MAX_N);
    DG_REQUIRE((size_t)B * N * k < ((size_t)1 << 31), "%s: more than 2^31 edges", who);
    return 0;
}
extern "C" size_t dgcnn_graph_feature_bwd_workspace_bytes(int B, int N, int C, int k)
{
    return gf_bwd_shape_ok(B, N, C, k, "dgcnn_graph_feature_bwd_workspace_bytes") == 0 ? gf_ws(B, N, k, nullptr).bytes : 0;
}

extern "C" int dgcnn_graph_feature_bwd(const float *gout, const int32_t *idx, int B, int N, int C, int k, float *gx,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    DG_REQUIRE(gout && idx && gx && workspace, "dgcnn_graph_feature_bwd: null pointer");
    if (gf_bwd_shape_ok(B, N, C, k, "dgcnn_graph_feature_bwd")) return 1;
    const GfWs w = gf_ws(B, N, k, workspace);
    DG_REQUIRE(workspace_bytes >= w.bytes, "dgcnn_graph_feature_bwd: workspace too small (%zu < %zu)", workspace_bytes,
               w.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (launch_rev_build(idx, B, N, k, w.cnt, w.indeg, w.rowptr, w.segbase, w.pcnt, w.rev, w.zero + Z_ALLOC, w.extra, w.zero,
                         Z_WORDS, true, true, st))
        return 1;
    launch_graph_feature_bwd(gout, w.rowptr, w.indeg, w.rev, B, N, C, k, gx, st);
    DG_CHECK_LAUNCH("graph_feature_bwd_kernel");
    return 0;
}
