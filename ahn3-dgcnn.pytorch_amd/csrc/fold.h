// fold.h -- deterministic in-launch reduction of per-workgroup fp64 records ("the last arriver finishes").
//
// Every workgroup of a statistics pass leaves one record [NV][64] of fp64 partial sums (channel = lane).  Instead of a
// fold kernel and a finalize kernel behind the pass, the pass finishes the sums itself, in two levels:
//   group  : records are cut into groups of FOLD_GS = 64 consecutive workgroups; the workgroup that arrives LAST at its
//            group's counter adds the group's records in record order (wave w takes a contiguous share, the wave sums are
//            added in wave order) and leaves one folded record;
//   final  : the workgroup that arrives last at the final counter adds the <= 64 folded records the same way and calls
//            `fin(total, channel)` on its first wave.
// The association is a function of the record count only -- WHICH workgroup does the adding does not enter -- so the
// result is bit-stable from run to run and independent of dispatch order or XCD placement.
//
// Hand-off (cdna_hip_programming.md section 6, Guideline 16, the counter form with write-through payload): records are
// stored with sc1 (agent-scope, write-through) stores by ONE wave, drained with s_waitcnt vmcnt(0) before that wave's
// lane 0 adds to the counter (agent-scope atomic), and every load of a record is an sc1 load -- so no release / acquire
// fence (an agent release writes back the whole L2 of the XCD: microseconds per workgroup) is needed.
// Any OTHER word a finalizer reads (the fp16-range guard of the second-conv kernels, BnBatchStats::guard) obeys the same
// rule on the writer's side: a wave that raises it waits for its atomic (s_waitcnt vmcnt(0)) ahead of the workgroup
// barrier that precedes the ticket, so the raise is in L2 before ANY ticket of that workgroup can make another
// workgroup the last arriver.
// The counters must be ZERO when the kernel starts: the kernel ahead of the pass in the same C-ABI call zeroes them
// (edgeconv.hip), i.e. they are re-initialised on every call (and on every replay of a captured graph).
#pragma once
#include "edge_common.h"

namespace dgcnn {

constexpr int FOLD_GS = 64;        // records per group
constexpr int FOLD_GROUPS = 64;    // at most PART_MAX / FOLD_GS groups
constexpr int FOLD_CTR_STRIDE = 32; // ints between two counters: every counter on a 128-byte line of its own -- the
                                    // tickets of ALL groups are drawn within a microsecond of each other, and on one line
                                    // they would queue at one memory channel
constexpr int FOLD_CTR_WORDS = (FOLD_GROUPS + 1) * FOLD_CTR_STRIDE;  // counters of one fold: the groups' and the final one
static_assert(PART_MAX <= FOLD_GS * FOLD_GROUPS, "record groups");

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store((gu64 *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int ticket(int *ctr)
{
    return (int)__hip_atomic_fetch_add((gu32 *)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FoldSet {
    double *rec;     // [gridDim.x rounded up to a multiple of FOLD_GS][NV][64] (allocated: PART_MAX records)
    double *folded;  // [FOLD_GROUPS][NV][64]
    int *ctr;        // [FOLD_CTR_WORDS], zero at kernel start
};

// add `n` consecutive records [NV][64] starting at src (n <= FOLD_GS): wave w takes records w PER .. w PER + PER - 1 in
// order, the NW wave sums are added in wave order by wave 0.  Returns the sum in wave 0's `out` (channel = lane).
template <int NV, int NW>
__device__ __forceinline__ void fold_span(const double *src, int n, double (*red)[NV][64], double (&out)[NV])
{
    constexpr int PER = FOLD_GS / NW, BATCH = PER < 8 ? PER : 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = 0.0;
    // BATCH records in flight per wave (the loads are what the tail waits for; more in flight would raise the register
    // allocation of the whole kernel and cost the statistics pass its occupancy)
#pragma unroll 1
    for (int u0 = 0; u0 < PER; u0 += BATCH) {
        // (unconditional loads at constant offsets from one base: the span lies inside the record buffer, which
        // holds FOLD_GS records per group, whatever n is; records past n are dropped by the select, never added)
        const double *base = src + ((size_t)(wave * PER + u0) * NV) * 64 + lane;
        double t[BATCH][NV];
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) t[u][i] = ld_sc1(base + (u * NV + i) * 64);
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] += wave * PER + u0 + u < n ? t[u][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave][i][lane] = a[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = red[0][i][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][i][lane];
            out[i] = s;
        }
    }
}

// Called by ALL threads of every workgroup (NW waves) at the end of a statistics pass; v = this thread's partial sums
// for channel `lane`.  fin(total, channel) runs once per launch, on the first wave of the last workgroup to arrive.
// `red` ([NW][NV][64] doubles) and `role` (one int) are LDS scratch of the caller (free at that point).
// nrec_in: number of workgroups that take part (the first nrec_in of the launch) when the launch also carries
// workgroups of another pass; < 0: the whole grid.
template <int NV, int NW, typename Fin>
__device__ __forceinline__ void fold_and_finalize(const double (&v)[NV], const FoldSet fs, Fin fin,
                                                  double (*red)[NV][64], int *role, int nrec_in = -1)
{
    static_assert(FOLD_GS % NW == 0, "waves per workgroup");
    int &s_role = *role;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrec = nrec_in < 0 ? (int)gridDim.x : nrec_in, ngroups = (nrec + FOLD_GS - 1) / FOLD_GS, g = (int)blockIdx.x / FOLD_GS;
    const int members = nrec - g * FOLD_GS < FOLD_GS ? nrec - g * FOLD_GS : FOLD_GS;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave][i][lane] = v[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = red[0][i][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][i][lane];
            st_sc1(fs.rec + ((size_t)blockIdx.x * NV + i) * 64 + lane, s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record has left this wave before the ticket is drawn
        if (lane == 0) s_role = ticket(fs.ctr + g * FOLD_CTR_STRIDE) == members - 1 ? 1 : 0;
    }
    __syncthreads();
    if (s_role == 0) return;  // (uniform over the workgroup)

    double tot[NV];
    fold_span<NV, NW>(fs.rec + (size_t)g * FOLD_GS * NV * 64, members, red, tot);
    if (ngroups > 1) {
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) st_sc1(fs.folded + ((size_t)g * NV + i) * 64 + lane, tot[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) s_role = ticket(fs.ctr + FOLD_GROUPS * FOLD_CTR_STRIDE) == ngroups - 1 ? 2 : 1;
        }
        __syncthreads();
        if (s_role != 2) return;
        fold_span<NV, NW>(fs.folded, ngroups, red, tot);
    }
    if (wave == 0) fin(tot, lane);
}

template <int NV, int NW, typename Fin>
__device__ __forceinline__ void fold_and_finalize(const double (&v)[NV], const FoldSet fs, Fin fin, int nrec_in = -1)
{
    __shared__ double red[NW][NV][64];
    __shared__ int role;
    fold_and_finalize<NV, NW, Fin>(v, fs, fin, red, &role, nrec_in);
}

// ---- the two finalizers of the EdgeConv block ------------------------------------------------------------------------
// BatchNorm2d batch statistics of one stage from the folded sums (sum u, sum u^2 over all `count` edge values): mean /
// rstd / the affine pair (sc, sh) into `stats`, and the running-statistics update (the biased variance normalises, the
// unbiased one goes into running_var: PyTorch's rule).
struct BnBatchStats {
    dgcnn_bn_params bn;
    double count;
    float eps, momentum;
    float *stats;
    const int *guard = nullptr;  // non-NULL and raised: another launch redoes the pass -- leave the statistics alone
    __device__ __forceinline__ void operator()(const double (&t)[2], int c) const
    {
        if (guard && __hip_atomic_load((gu32 *)guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        const double m = t[0] / count;
        double v = t[1] / count - m * m;
        if (v < 0.0) v = 0.0;
        const float mean = (float)m, var = (float)v;
        if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;  // (one thread of one workgroup finishes channel 0)
        if (bn.running_mean) {
            const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
            bn.running_mean[c] = (1.0f - momentum) * bn.running_mean[c] + momentum * mean;
            bn.running_var[c] = (1.0f - momentum) * bn.running_var[c] + momentum * (float)unbiased;
        }
        const float rstd = (float)(1.0 / sqrt((double)var + (double)eps));
        const float sc = bn.weight[c] * rstd;
        stats[ST_MEAN + c] = mean;
        stats[ST_RSTD + c] = rstd;
        stats[ST_SC + c] = sc;
        stats[ST_SH + c] = bn.bias[c] - mean * sc;
    }
};

// BatchNorm backward sums: gbn_b = sum dv, gbn_w = sum dv * uhat; cc[0..63] = gbn_b / count, cc[64..127] = gbn_w / count
// (the two mean terms of the BatchNorm input gradient; zero in eval mode, where the statistics are constants).
struct BnBackwardSums {
    double count;
    int training;
    float *gbn_w, *gbn_b, *cc;
    __device__ __forceinline__ void operator()(const double (&t)[2], int c) const
    {
        gbn_b[c] = (float)t[0];
        gbn_w[c] = (float)t[1];
        cc[c] = training ? (float)(t[0] / count) : 0.0f;
        cc[64 + c] = training ? (float)(t[1] / count) : 0.0f;
    }
};

}  // namespace dgcnn
