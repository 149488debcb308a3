// edge_stream.hip -- the edge-streaming passes of the EdgeConv block (gather, reduce, scatter).
//
// Mapping for every pass: one wave = one point at a time, lane = output channel (64).  A
// neighbour's row of TP is 64 fp32 = 256 contiguous bytes, so each gather is one fully coalesced
// wave-load and each scatter one 256-byte wave-atomic (the shape that global float atomics run
// at full rate on MI355X).  The neighbour list of the point sits in one VGPR (lane j holds
// idx[i][j]) and is broadcast with v_readlane.
//
// What the passes compute (reference lines in /root/reference/model.py):
//   u[i][j]  = TP[n(i, j)] + TQ[i]           the first conv on the edge tuple (model.py:74-81 + :116/:122/:128)
//   BN stats = sum u, sum u^2 over all edges  (BatchNorm2d batch statistics, model.py:107-111), in fp64
//   usel[i]  = the extreme of u over the k neighbours that survives max-over-k (model.py:150,157,163):
//              BN-affine + LeakyReLU is monotone per channel (increasing if gamma >= 0, else
//              decreasing), and every fp32 rounding in it is monotone too, so
//              max_j LReLU(BN(u_j)) == LReLU(BN(max_j u_j))  (min_j if gamma < 0)  bit for bit.
#include <stdlib.h>

#include "edge_kernels.h"
#include "prof.h"

namespace dgcnn {

constexpr int EDGE_WAVES = 4;  // waves per workgroup in the streaming passes

// Workgroups per streaming pass (speed only).  The passes are bound by the latency of the row gathers, so the
// right number is "as many resident waves as the registers allow": measured at B = 8, N = 7000, k = 30
// (us at 64 / 128 / 256 / 512 slots per XCD): edge_reduce 194 / 115 / 109 / 92, edge_rgather 211 / 152 / 128 / 131,
// edge_point_bwd 43 / 29 / 24 / 24.
static int slots_override()
{
    static const int v = getenv("DGCNN_EDGE_SLOTS") ? atoi(getenv("DGCNN_EDGE_SLOTS")) : 0;  // experiments
    return v;
}
int edge_grid(int P)  // edge_reduce / edge_bwd: also the number of fp64 partial records (<= PART_MAX)
{
    return point_grid(P, EDGE_WAVES, slots_override() ? slots_override() : 512);
}
static int stream_grid(int P)  // passes without partial records
{
    return point_grid(P, EDGE_WAVES, slots_override() ? slots_override() : 256);
}

// sum the per-lane doubles of the workgroup's 4 waves and write partial[blockIdx][which][64]
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *__restrict__ partial)
{
    __shared__ double red[EDGE_WAVES][NV][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave][i][lane] = v[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = red[0][i][lane];
#pragma unroll
            for (int w = 1; w < EDGE_WAVES; ++w) s += red[w][i][lane];
            partial[((size_t)blockIdx.x * NV + i) * 64 + lane] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Also written for the backward: usum[i] = sum_j u[i][j] (fp32) and, with WANT_EXT, hitj[i] = the slot j (0..k-1)
// of the FIRST edge attaining the extreme (PyTorch's max routes the gradient to the first maximum).
template <bool WANT_EXT>
__global__ __launch_bounds__(256) void edge_reduce_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                          const int32_t *__restrict__ idx, int N, int k, int P,
                                                          const float *__restrict__ gamma, float *__restrict__ usel,
                                                          float *__restrict__ usum, uint8_t *__restrict__ hitj,
                                                          double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool neg = WANT_EXT ? (gamma[lane] < 0.0f) : false;
    double acc[2] = {0.0, 0.0};
    const PointIter it = point_iter(P, wave, EDGE_WAVES);
    // The pass is latency-bound (index row -> neighbour rows -> arithmetic), so every point's loads are issued
    // as one batch of up to 32 rows, and the next point's index row and TQ row are fetched one point ahead.
    int pt = it.cur;
    int nv_next = 0;
    float q_next = 0.0f;
    if (pt < it.end) {
        nv_next = lane < k ? __builtin_nontemporal_load(&idx[(size_t)pt * k + lane]) : 0;
        q_next = __builtin_nontemporal_load(&TQ[(size_t)pt * EC + lane]);
    }
    for (; pt < it.end; pt += it.step) {
        const int b = pt / N;
        // wave-uniform table base + 32-bit element offsets: each row load is v_readlane, one v_lshl_add and a
        // global_load with scalar base (64-bit address arithmetic per row costs ~10 scalar instructions)
        const float *Pb = TP + (size_t)b * N * EC;
        const float q = q_next;
        int nv = nv_next;  // neighbours 0..63 of this point (lane j holds idx[pt][j])
        const int32_t *ip = idx + (size_t)pt * k;
        const int ptn = pt + it.step;
        if (ptn < it.end) {
            nv_next = lane < k ? __builtin_nontemporal_load(&idx[(size_t)ptn * k + lane]) : 0;
            q_next = __builtin_nontemporal_load(&TQ[(size_t)ptn * EC + lane]);
        }
        // the extreme is tracked on sign-flipped values where gamma < 0 (exact negation), so one comparison serves
        // both directions
        const unsigned sgn = neg ? 0x80000000u : 0u;
        float ext = -__builtin_inff();
        float us = 0.0f;
        int hit = 0;
        auto row = [&](float pv, int slot) {
            const float u = pv + q;
            acc[0] += (double)u;
            acc[1] = __builtin_fma((double)u, (double)u, acc[1]);
            us += u;
            if (WANT_EXT) {
                const float uf = __uint_as_float(__float_as_uint(u) ^ sgn);
                const bool better = uf > ext;  // strict: the first edge attaining the extreme wins
                ext = better ? uf : ext;
                hit = better ? slot : hit;
            }
        };
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int kk = k - j0 < 64 ? k - j0 : 64;
            if (j0 > 0) nv = lane < kk ? ip[j0 + lane] : 0;
            for (int j = 0; j < kk; j += 32) {
                // all loads of up to 32 rows are issued before the first is consumed, in groups of 8 behind one
                // uniform branch each; rows past the list end repeat its last row (loaded, never used)
                const int nrow = kk - j < 32 ? kk - j : 32;
                float p[32];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g * 8 < nrow) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int tt = j + g * 8 + t < kk ? j + g * 8 + t : kk - 1;
                            const unsigned n = (unsigned)__builtin_amdgcn_readlane(nv, tt);
                            p[g * 8 + t] = Pb[n * (unsigned)EC + (unsigned)lane];
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g * 8 + 8 <= nrow) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) row(p[g * 8 + t], j0 + j + g * 8 + t);
                    } else if (g * 8 < nrow) {
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (g * 8 + t < nrow) row(p[g * 8 + t], j0 + j + g * 8 + t);
                    }
                }
            }
        }
        // streaming outputs bypass the cache hierarchy's retention: the gather table must stay in L2
        __builtin_nontemporal_store(us, &usum[(size_t)pt * EC + lane]);
        if (WANT_EXT) {
            __builtin_nontemporal_store(__uint_as_float(__float_as_uint(ext) ^ sgn), &usel[(size_t)pt * EC + lane]);
            __builtin_nontemporal_store((uint8_t)hit, &hitj[(size_t)pt * EC + lane]);
        }
    }
    block_reduce_store<2>(acc, partial);
}

// Sum of fp64 partial records [nparts][2][64] for channel c.  1024 threads = 16 waves: wave w
// adds records w, w + 16, ... in order, then the 16 wave sums are added in wave order: a fixed
// association, so the result is deterministic (and identical for every launch geometry).
constexpr int FIN_WAVES = 16;
__device__ __forceinline__ void reduce_records(const double *__restrict__ partial, int nparts, double &s0,
                                               double &s1, int first = 0)
{
    partial += (size_t)first * 2 * 64;
    __shared__ double red[FIN_WAVES][2][64];
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0;
    int g = w;
    for (; g + 3 * FIN_WAVES < nparts; g += 4 * FIN_WAVES) {  // four records in flight, added in record order
        double v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v0[u] = partial[((size_t)(g + u * FIN_WAVES) * 2 + 0) * 64 + c];
            v1[u] = partial[((size_t)(g + u * FIN_WAVES) * 2 + 1) * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 += v0[u];
            a1 += v1[u];
        }
    }
    for (; g < nparts; g += FIN_WAVES) {
        a0 += partial[((size_t)g * 2 + 0) * 64 + c];
        a1 += partial[((size_t)g * 2 + 1) * 64 + c];
    }
    red[w][0][c] = a0;
    red[w][1][c] = a1;
    __syncthreads();
    s0 = 0.0;
    s1 = 0.0;
    for (int i = 0; i < FIN_WAVES; ++i) {
        s0 += red[i][0][c];
        s1 += red[i][1][c];
    }
}

// A single workgroup reads its records at one CU's bandwidth (4096 records = 4 MB: ~60 us), so large record
// sets are first folded to PART_FOLD records by PART_FOLD workgroups (fixed ranges and order: deterministic).
__global__ __launch_bounds__(1024) void fold_records_kernel(const double *__restrict__ partial, int nparts,
                                                            double *__restrict__ folded)
{
    const int per = (nparts + PART_FOLD - 1) / PART_FOLD;
    const int first = blockIdx.x * per;
    int n = nparts - first;
    n = n < 0 ? 0 : (n < per ? n : per);
    double s0, s1;
    reduce_records(partial, n, s0, s1, first);
    if (threadIdx.x < 64) {
        folded[((size_t)blockIdx.x * 2 + 0) * 64 + threadIdx.x] = s0;
        folded[((size_t)blockIdx.x * 2 + 1) * 64 + threadIdx.x] = s1;
    }
}
// returns the record set the finalize kernel should read
static const double *fold_records(double *partial, int &nparts, hipStream_t st)
{
    if (nparts <= 4 * PART_FOLD) return partial;
    double *folded = partial + (size_t)PART_MAX * 2 * 64;
    hipLaunchKernelGGL(fold_records_kernel, dim3(PART_FOLD), dim3(1024), 0, st, partial, nparts, folded);
    nparts = PART_FOLD;
    return folded;
}

// one workgroup of 1024 threads; the first wave finishes channel c = lane.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double *__restrict__ partial, int nparts,
                                                           double count, int training, dgcnn_bn_params bn, float eps,
                                                           float momentum, float *__restrict__ stats)
{
    double s = 0.0, ss = 0.0;
    if (training) reduce_records(partial, nparts, s, ss);
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    float mean, var;
    if (training) {
        const double m = s / count;
        double v = ss / count - m * m;  // biased variance, as BatchNorm normalises with
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        if (bn.running_mean) {
            const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
            bn.running_mean[c] = (1.0f - momentum) * bn.running_mean[c] + momentum * mean;
            bn.running_var[c] = (1.0f - momentum) * bn.running_var[c] + momentum * (float)unbiased;
        }
    } else {
        mean = bn.running_mean[c];
        var = bn.running_var[c];
    }
    const float rstd = (float)(1.0 / sqrt((double)var + (double)eps));
    const float sc = bn.weight[c] * rstd;
    stats[ST_MEAN + c] = mean;
    stats[ST_RSTD + c] = rstd;
    stats[ST_SC + c] = sc;
    stats[ST_SH + c] = bn.bias[c] - mean * sc;
}

__global__ void edge_out_kernel(const float *__restrict__ usel, const float *__restrict__ stats, float slope,
                                size_t n4, float *__restrict__ y)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const int c = (int)(t & 15) * 4;
    const float4 u = reinterpret_cast<const float4 *>(usel)[t];
    const float4 sc = *reinterpret_cast<const float4 *>(stats + ST_SC + c);
    const float4 sh = *reinterpret_cast<const float4 *>(stats + ST_SH + c);
    float4 o;
    o.x = lrelu(__builtin_fmaf(u.x, sc.x, sh.x), slope);
    o.y = lrelu(__builtin_fmaf(u.y, sc.y, sh.y), slope);
    o.z = lrelu(__builtin_fmaf(u.z, sc.z, sh.z), slope);
    o.w = lrelu(__builtin_fmaf(u.w, sc.w, sh.w), slope);
    reinterpret_cast<float4 *>(y)[t] = o;
}

// ---------------------------------------------------------------------------------------------
// Backward of the LAST BatchNorm of the block.  Only the arg-max edge of each (point, channel)
// carries gradient through max-over-k, so  sum dv  and  sum dv * uhat  over all edges reduce to
// sums over points of quantities at the selected extreme.
__global__ __launch_bounds__(256) void bwd_point_kernel(const float *__restrict__ gy, const float *__restrict__ usel,
                                                        const float *__restrict__ stats, float slope, int P,
                                                        double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane];
    const float sc = stats[ST_SC + lane], sh = stats[ST_SH + lane];
    double acc[2] = {0.0, 0.0};
    for (int pt = blockIdx.x * EDGE_WAVES + wave; pt < P; pt += gridDim.x * EDGE_WAVES) {
        const float us = usel[(size_t)pt * EC + lane];
        const float v = __builtin_fmaf(us, sc, sh);
        const float dv = gy[(size_t)pt * EC + lane] * (v > 0.0f ? 1.0f : slope);
        const float uhat = (us - mean) * rstd;
        acc[0] += (double)dv;
        acc[1] = __builtin_fma((double)dv, (double)uhat, acc[1]);
    }
    block_reduce_store<2>(acc, partial);
}

// gbn_b = sum dv, gbn_w = sum dv * uhat;  cc[0..63] = gbn_b / count, cc[64..127] = gbn_w / count
// (the two mean terms of the BatchNorm input gradient; zero in eval mode, where the statistics
// are constants).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double *__restrict__ partial, int nparts,
                                                               double count, int training, float *__restrict__ gbn_w,
                                                               float *__restrict__ gbn_b, float *__restrict__ cc)
{
    double sb, sg;
    reduce_records(partial, nparts, sb, sg);
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    gbn_b[c] = (float)sb;
    gbn_w[c] = (float)sg;
    cc[c] = training ? (float)(sb / count) : 0.0f;
    cc[64 + c] = training ? (float)(sg / count) : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Stage-1 backward over the edges.  With du = sc * (dv - c1 - c2 * uhat)   (Appendix C.5)
//   dD[i]        = sum_j du[i][j]                 (gradient of TQ[i]: RepeatBackward0 + Sub/CatBackward0)
//   dR[n(i, j)] += du[i][j]                       (gradient of TP rows: IndexBackward0, a scatter-add)
// DENSE = false: dv is non-zero only at the first edge j* whose activation equals y[i] (PyTorch's
// max routes the whole gradient to the first maximum); the dense mean terms are streamed for all
// edges and the single dv term is added once per (point, channel).
// DENSE = true : dv[i][j] comes from the second stage's backward ([E][64] in HBM).
template <bool DENSE>
__global__ __launch_bounds__(256) void edge_bwd_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                       const int32_t *__restrict__ idx, int N, int k, int P,
                                                       const float *__restrict__ stats, const float *__restrict__ cc,
                                                       float slope, int training, const float *__restrict__ gy,
                                                       const float *__restrict__ y, const float *__restrict__ dv,
                                                       float *__restrict__ dR, float *__restrict__ dD)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane];
    const float sc = stats[ST_SC + lane], sh = stats[ST_SH + lane];
    const float c1 = cc[lane], c2 = cc[64 + lane];
    const PointIter it = point_iter(P, wave, EDGE_WAVES);
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const int b = pt / N;
        const float *Pb = TP + (size_t)b * N * EC + lane;
        float *Rb = dR + (size_t)b * N * EC + lane;
        const float q = TQ[(size_t)pt * EC + lane];
        const int32_t *ip = idx + (size_t)pt * k;
        float yv = 0.0f, dvs = 0.0f;
        if (!DENSE) {
            yv = y[(size_t)pt * EC + lane];
            dvs = gy[(size_t)pt * EC + lane] * (yv > 0.0f ? 1.0f : slope);
        }
        int hit = -1;  // neighbour index of the first edge attaining the maximum activation
        float dsum = 0.0f;
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int kk = k - j0 < 64 ? k - j0 : 64;
            const int nv = lane < kk ? ip[j0 + lane] : 0;
            for (int j = 0; j < kk; j += 8) {
                float p[8], d[8];
                int n[8];
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (j + t < kk) {
                        n[t] = __builtin_amdgcn_readlane(nv, j + t);
                        p[t] = Pb[(size_t)n[t] * EC];
                        if (DENSE) d[t] = dv[((size_t)pt * k + j0 + j + t) * EC + lane];
                    }
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (j + t < kk) {
                        const float u = p[t] + q;
                        if (!DENSE) {
                            const float a = lrelu(__builtin_fmaf(u, sc, sh), slope);
                            if (hit < 0 && a == yv) hit = n[t];
                        }
                        if (DENSE || training) {
                            const float uhat = (u - mean) * rstd;
                            const float du = sc * ((DENSE ? d[t] : 0.0f) - c1 - c2 * uhat);
                            dsum += du;
                            atomicAdd(Rb + (size_t)n[t] * EC, du);
                        }
                    }
            }
        }
        if (!DENSE) {
            const float du = sc * dvs;
            dsum += du;
            if (hit >= 0) atomicAdd(Rb + (size_t)hit * EC, du);
        }
        dD[(size_t)pt * EC + lane] = dsum;
    }
}

// ---------------------------------------------------------------------------------------------
// Training-mode stage-1 backward as a GATHER over the reverse adjacency (CSR by destination) of the kNN graph.
// With alpha = sc c2 rstd and A[i] = -sc (c1 + c2 rstd (TQ[i] - mean)) = a0 - alpha TQ[i] (the part of du that is
// the same for all k edges of i) the edge gradient is du[i][j] = sc dv[i][j] + A[i] - alpha TP[n(i, j)], hence
//   dR[m] = indeg(m) a0 - alpha (sum_{i in in(m)} TQ[i] + indeg(m) TP[m]) + sum_{e in in(m)} sc dv[e]
//   dD[i] = sum_j sc dv[i][j] + k A[i] - alpha (usum[i] - k TQ[i])                (usum = sum_j u[i][j], forward)
// The scatter form of dR costs one 256-byte wave-atomic per edge, and global atomics run at ~20 G 64-byte
// segments per second on MI355X whatever their width (1.68 M edges x 4 segments: 0.4 ms at B = 8, N = 7000,
// k = 30).  The gather needs one coalesced 256-byte row load per in-edge and no atomic; the lists are built
// with LDS atomics.  Sparse dv (single-stage block: dL/dy of (i, c) belongs to the one edge that won
// max-over-k, slot hitj[i][c] from the forward): every edge carries a 64-bit channel mask and a compact
// record of its first 4 hit values, so the gather adds them without reading a second row (edges with more
// than 4 hits read the row of G).  The order of a point's in-edges follows the LDS atomics of rev_place, so
// the fp32 sum order can differ between runs (as it did with float atomics).
// Build, per batch element b (its in-edges occupy [b N k, (b + 1) N k) of `rev`): the sources of a batch element
// are cut into `chunks` contiguous ranges, one workgroup each, and all counting happens with LDS atomics on an
// N-entry table (scattered global atomics run at ~20 G/s on MI355X whatever their width: 1.68 M of them cost
// as much as the whole gather).
//   rev_count: cnt[b][chunk][m] = edges of the chunk that point at m
//   rev_degree / rev_scan / rev_slots: indeg[m] = sum over chunks; rowptr = exclusive scan of indeg;
//              cnt[b][chunk][m] <- first slot of that chunk's edges into m
//   rev_place: rev[slot++] = edge id, slot from the LDS copy of cnt[b][chunk][.]
constexpr int REV_THREADS = 1024;
constexpr int REV_MAX_N = 40000;  // N ints of LDS per workgroup (160 KB per CU)

__global__ __launch_bounds__(REV_THREADS) void rev_count_kernel(const int32_t *__restrict__ idx, int N, int k,
                                                               int chunks, int *__restrict__ cnt)
{
    extern __shared__ int table[];
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    for (int m = threadIdx.x; m < N; m += REV_THREADS) table[m] = 0;
    __syncthreads();
    const int ne = N * k, per = (ne + chunks - 1) / chunks;
    const int beg = chunk * per, end = beg + per < ne ? beg + per : ne;
    const int32_t *ib = idx + (size_t)b * ne;
    for (int e = beg + threadIdx.x; e < end; e += REV_THREADS) atomicAdd(&table[ib[e]], 1);
    __syncthreads();
    int *out = cnt + (size_t)blockIdx.x * N;
    for (int m = threadIdx.x; m < N; m += REV_THREADS) out[m] = table[m];
}

// indeg[b][m] = sum over chunks of cnt[b][chunk][m]
__global__ __launch_bounds__(256) void rev_degree_kernel(const int *__restrict__ cnt, int N, int chunks, int P,
                                                         int *__restrict__ indeg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const int b = g / N, m = g - b * N;
    const int *cb = cnt + (size_t)b * chunks * N + m;
    int s = 0;
    for (int c = 0; c < chunks; ++c) s += cb[(size_t)c * N];
    indeg[g] = s;
}

// rowptr = exclusive scan of indeg; one workgroup = one batch element
__global__ __launch_bounds__(REV_THREADS) void rev_scan_kernel(const int *__restrict__ indeg, int N, int k,
                                                              int *__restrict__ rowptr, int *__restrict__ nextra)
{
    __shared__ int part[REV_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0 && tid == 0) *nextra = 0;  // counter of rev_slots_kernel
    const int *deg = indeg + (size_t)b * N;
    const int per = (N + REV_THREADS - 1) / REV_THREADS;
    const int beg = tid * per < N ? tid * per : N, end = beg + per < N ? beg + per : N;
    int s = 0;
    for (int i = beg; i < end; ++i) s += deg[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < REV_THREADS; off <<= 1) {  // inclusive Hillis-Steele scan
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = b * N * k + part[tid] - s;
    for (int i = beg; i < end; ++i) {
        rowptr[(size_t)b * N + i] = run;
        run += deg[i];
    }
    if (b == (int)gridDim.x - 1 && tid == REV_THREADS - 1) rowptr[(size_t)gridDim.x * N] = (int)gridDim.x * N * k;
}

// cnt[b][chunk][m] <- first slot of chunk's edges into m; also lists the extra segments of long lists:
// (point, segment) pairs for segments 1 .. (deg - 1) / REV_SEG, allocated with one atomic per workgroup
__global__ __launch_bounds__(256) void rev_slots_kernel(const int *__restrict__ rowptr, int N, int chunks, int P,
                                                        int *__restrict__ cnt, int *__restrict__ nextra,
                                                        int2 *__restrict__ extra)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int nseg = 0;
    if (g < P) {
        const int b = g / N, m = g - b * N;
        int *cb = cnt + (size_t)b * chunks * N + m;
        int slot = rowptr[g];
        nseg = (rowptr[g + 1] - slot - 1) / REV_SEG;  // deg <= REV_SEG (also 0): none
        nseg = nseg < 0 ? 0 : nseg;
        for (int c = 0; c < chunks; ++c) {
            const int n = cb[(size_t)c * N];
            cb[(size_t)c * N] = slot;
            slot += n;
        }
    }
    // workgroup-aggregated allocation (one atomicAdd on the global counter per workgroup: per-wave atomics on one
    // address cost ~9 us here): inclusive scan of nseg over the lanes, wave totals through LDS
    __shared__ int wtot[4], wbase[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = nseg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t0 = wtot[0], t1 = wtot[1], t2 = wtot[2], t3 = wtot[3];
        const int total = t0 + t1 + t2 + t3;
        const int b0 = total ? atomicAdd(nextra, total) : 0;
        wbase[0] = b0;
        wbase[1] = b0 + t0;
        wbase[2] = b0 + t0 + t1;
        wbase[3] = b0 + t0 + t1 + t2;
    }
    __syncthreads();
    const int base = wbase[wave] + incl - nseg;
    for (int sgm = 0; sgm < nseg; ++sgm) extra[base + sgm] = int2{g, sgm + 1};
}

__global__ __launch_bounds__(REV_THREADS) void rev_place_kernel(const int32_t *__restrict__ idx, int N, int k,
                                                               int chunks, const int *__restrict__ cnt,
                                                               int32_t *__restrict__ rev)
{
    extern __shared__ int table[];
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int *first = cnt + (size_t)blockIdx.x * N;
    for (int m = threadIdx.x; m < N; m += REV_THREADS) table[m] = first[m];
    __syncthreads();
    const int ne = N * k, per = (ne + chunks - 1) / chunks;
    const int beg = chunk * per, end = beg + per < ne ? beg + per : ne;
    const int32_t *ib = idx + (size_t)b * ne;
    for (int e = beg + threadIdx.x; e < end; e += REV_THREADS) rev[atomicAdd(&table[ib[e]], 1)] = b * ne + e;
}

constexpr int HIT_REC = 4;  // hit values stored per edge
// what the sparse backward keeps per edge (one 32-byte sector, read once by the gather): the channels whose
// max-over-k picked this edge and the gradient of the first HIT_REC of them (measured: records written in
// reverse-list order make the gather's read coalesced but cost more as scattered writes: 101 + 60 us against
// 113 + 27 us)
struct __attribute__((aligned(16))) HitRec {
    u64 mask;
    float v[HIT_REC];
    u64 pad;
};
static_assert(sizeof(HitRec) == 32, "HitRec is one 32-byte sector");

// kNN graphs in feature space have hubs (at B = 8, N = 7000, k = 30 on Gaussian features: median in-degree 12,
// maximum 1372), and one wave walking a hub's list alone would set the kernel time.  A point's list is therefore
// cut into segments of REV_SEG in-edges (64: measured 76 / 69 / 67 / 70 us at 32 / 48 / 64 / 96): rev_slots_kernel lists
// the (point, segment) pairs beyond segment 0, and edge_rgather_kernel runs workgroups for the points (segment 0) and
// for those pairs side by side; a point with more than one segment is assembled with one 256-byte wave-atomic per
// segment (its row is zeroed beforehand).
// sums over the list entries [beg, end) (at most REV_SEG; ev_first = entries beg .. beg + 63 in the lanes):
// sq = sum of the sources' TQ rows, sd = sum of their dv rows (DENSE); sparse hits go to sacc[channel]
template <bool DENSE>
__device__ __forceinline__ void rgather_segment(const float *__restrict__ TQ, const int32_t *__restrict__ rev, int k,
                                                const float *__restrict__ dv, const HitRec *__restrict__ hits,
                                                const float *__restrict__ G, int beg, int end, int ev_first,
                                                float *sacc, int lane, float &sq, float &sd)
{
    float s0 = 0.0f, s1 = 0.0f, d0 = 0.0f, d1 = 0.0f;
    for (int e0 = beg; e0 < end; e0 += 64) {
        const int cnt = end - e0 < 64 ? end - e0 : 64;
        const int ev = e0 == beg ? ev_first : (lane < cnt ? rev[e0 + lane] : 0);
        const int src = (int)((unsigned)ev / (unsigned)k);
        // sparse dv: lane t owns in-edge t here -- its channel mask and hit record; the few hits of an edge
        // (64 per centre over its k edges) go to the channel sums with LDS float atomics, so the cost is per
        // hit and not per (edge, channel)
        u64 hm = 0;
        float4 rec = float4{0.0f, 0.0f, 0.0f, 0.0f};
        if (!DENSE && lane < cnt) {
            const HitRec *h = hits + ev;
            hm = h->mask;
            rec = *reinterpret_cast<const float4 *>(h->v);
        }
        // 8 rows in flight per wave (16, as two groups, measured slower: most lists are short)
        for (int t = 0; t < cnt; t += 8) {
            float q[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = t + u < cnt ? t + u : t;  // clamped: no branch around the loads
                const int i = __builtin_amdgcn_readlane(src, tt);
                q[u] = TQ[(unsigned)i * (unsigned)EC + (unsigned)lane];  // scalar base + 32-bit offset
                if (DENSE) d[u] = dv[(size_t)__builtin_amdgcn_readlane(ev, tt) * EC + lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = t + u < cnt;
                if (u & 1) {
                    s1 += on ? q[u] : 0.0f;
                    if (DENSE) d1 += on ? d[u] : 0.0f;
                } else {
                    s0 += on ? q[u] : 0.0f;
                    if (DENSE) d0 += on ? d[u] : 0.0f;
                }
            }
        }
        if (!DENSE) {
            const float rv[HIT_REC] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
            for (int r = 0; r < HIT_REC; ++r) {
                if (hm) {
                    atomicAdd(&sacc[__builtin_ctzll(hm)], rv[r]);
                    hm &= hm - 1ull;
                }
            }
            while (__ballot(hm != 0)) {  // rare: more hits on one edge than its record holds
                if (hm) {
                    const int c = __builtin_ctzll(hm);
                    atomicAdd(&sacc[c], G[(size_t)src * EC + c]);
                    hm &= hm - 1ull;
                }
            }
        }
    }
    sq = s0 + s1;
    sd = d0 + d1;
}

template <bool DENSE>
__global__ __launch_bounds__(256) void edge_rgather_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                           const int *__restrict__ rowptr,
                                                           const int32_t *__restrict__ rev, int k, int P,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ cc, const float *__restrict__ dv,
                                                           const HitRec *__restrict__ hits,
                                                           const float *__restrict__ G, const int *__restrict__ nextra,
                                                           const int2 *__restrict__ extra, int main_blocks,
                                                           float *__restrict__ dR)
{
    // Workgroups 0 .. main_blocks - 1 walk the points (segment 0 of every list), the others the extra segments of
    // the long lists, concurrently (one launch; dispatching the extra workgroups first was measured slower): every
    // contribution to a point with more than one segment is an atomic add into its row of dR, which
    // edge_point_bwd_kernel has zeroed.
    // DENSE: dv [E][64] rows by edge id.  Sparse: hits[e] by edge id and G [P][64] = sc dL/dv of the centre's
    // surviving edge per channel, both written by edge_point_bwd_kernel.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane], sc = stats[ST_SC + lane];
    const float c1 = cc[lane], c2 = cc[64 + lane];
    const float alpha = sc * c2 * rstd, a0 = -sc * (c1 - c2 * rstd * mean);
    __shared__ float sacc_all[EDGE_WAVES][64];  // sparse dv: hit gradients of the current point, by channel
    float *sacc = sacc_all[wave];
    if (!DENSE) sacc[lane] = 0.0f;
    if ((int)blockIdx.x >= main_blocks) {
        const int n = *nextra;
        for (int w = ((int)blockIdx.x - main_blocks) * EDGE_WAVES + wave; w < n;
             w += ((int)gridDim.x - main_blocks) * EDGE_WAVES) {
            const int2 ps = extra[w];
            const int pbeg = __builtin_amdgcn_readfirstlane(rowptr[ps.x]), pend = __builtin_amdgcn_readfirstlane(rowptr[ps.x + 1]);
            const int beg = pbeg + __builtin_amdgcn_readfirstlane(ps.y) * REV_SEG;
            const int end = beg + REV_SEG < pend ? beg + REV_SEG : pend;
            const int ev_first = lane < end - beg ? rev[beg + lane] : 0;
            float sq, sd;
            rgather_segment<DENSE>(TQ, rev, k, dv, hits, G, beg, end, ev_first, sacc, lane, sq, sd);
            float r = -alpha * sq;
            if (DENSE) {
                r += sc * sd;
            } else {
                __builtin_amdgcn_wave_barrier();
                r += sacc[lane];
                sacc[lane] = 0.0f;
                __builtin_amdgcn_wave_barrier();
            }
            atomicAdd(dR + (size_t)__builtin_amdgcn_readfirstlane(ps.x) * EC + lane, r);
        }
        return;
    }
    PointIter it;  // point_iter() over the first main_blocks workgroups only
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = main_blocks >> 3;
        const int per = (P + 7) >> 3, first = xcd * per;
        it.end = first + per < P ? first + per : P;
        it.cur = first + slot * EDGE_WAVES + wave;
        it.step = nslots * EDGE_WAVES;
    }
    // The chain list bounds -> list entries -> rows is three dependent loads deep, so it is software-pipelined
    // across the points of a wave: the bounds are fetched two points ahead, the first 64 list entries one.
    int pt = it.cur;
    int beg_n = 0, end_n = 0, ev_n = 0;  // bounds (uniform) and first entries of the point about to be processed
    int b0_nn = 0, b1_nn = 0;            // bounds of the point after it, in flight
    if (pt < it.end) {
        beg_n = __builtin_amdgcn_readfirstlane(rowptr[pt]);
        end_n = __builtin_amdgcn_readfirstlane(rowptr[pt + 1]);
        ev_n = lane < end_n - beg_n ? rev[beg_n + lane] : 0;
        if (pt + it.step < it.end) {
            b0_nn = rowptr[pt + it.step];
            b1_nn = rowptr[pt + it.step + 1];
        }
    }
    for (; pt < it.end; pt += it.step) {
        // (wave-uniform values are made visibly so: lane selects of v_readlane must be scalars)
        const int beg = beg_n, end = end_n;
        const int ev_first = ev_n;
        if (pt + it.step < it.end) {
            beg_n = __builtin_amdgcn_readfirstlane(b0_nn);
            end_n = __builtin_amdgcn_readfirstlane(b1_nn);
            ev_n = lane < end_n - beg_n ? rev[beg_n + lane] : 0;
        }
        if (pt + 2 * it.step < it.end) {
            b0_nn = rowptr[pt + 2 * it.step];
            b1_nn = rowptr[pt + 2 * it.step + 1];
        }
        float sq, sd;
        rgather_segment<DENSE>(TQ, rev, k, dv, hits, G, beg, beg + REV_SEG < end ? beg + REV_SEG : end, ev_first, sacc,
                               lane, sq, sd);
        const float deg = (float)(end - beg);
        float r = deg * a0 - alpha * (sq + deg * TP[(size_t)pt * EC + lane]);
        if (DENSE) {
            r += sc * sd;
        } else {
            __builtin_amdgcn_wave_barrier();
            r += sacc[lane];
            sacc[lane] = 0.0f;
            __builtin_amdgcn_wave_barrier();
        }
        if (end - beg > REV_SEG) atomicAdd(dR + (size_t)pt * EC + lane, r);  // shared with the extra segments
        else dR[(size_t)pt * EC + lane] = r;
    }
}

// dD[i] = sum_j sc dv[i][j] + k A[i] - alpha (usum[i] - k TQ[i]).  Dense dv: `dv` here is its per-point sum over the
// k edges [P][64].  Sparse dv has one non-zero per (i, c), on the
// edge in slot hitj[i][c]: the kernel also writes G[i][c] = sc dv of that edge and the HitRec of every edge of
// the point.  The records of a point are assembled in LDS:
// one 64-bit atomic-or per channel lane builds the masks, then lane j owns edge j.
template <bool DENSE>
__global__ __launch_bounds__(256) void edge_point_bwd_kernel(const float *__restrict__ TQ, int N, int k, int P,
                                                             const float *__restrict__ stats,
                                                             const float *__restrict__ cc, float slope,
                                                             const float *__restrict__ gy, const float *__restrict__ y,
                                                             const uint8_t *__restrict__ hitj,
                                                             const float *__restrict__ usum,
                                                             const float *__restrict__ dv,
                                                             HitRec *__restrict__ hits, float *__restrict__ G,
                                                             const int *__restrict__ rowptr, float *__restrict__ dR,
                                                             float *__restrict__ dD)
{
    __shared__ u64 smask[EDGE_WAVES][128];          // by slot (k <= 96)
    __shared__ float sval[EDGE_WAVES][128][HIT_REC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane], sc = stats[ST_SC + lane];
    const float c1 = cc[lane], c2 = cc[64 + lane];
    const float alpha = sc * c2 * rstd;
    u64 *sm = smask[wave];
    const PointIter it = point_iter(P, wave, EDGE_WAVES);
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const size_t o = (size_t)pt * EC + lane;
        const float q = TQ[o];
        const float A = -sc * (c1 + c2 * rstd * (q - mean));
        float dsum;
        if (DENSE) {
            dsum = sc * dv[o];  // dv = the per-point sums of dv1 over the k edges (edge_mlp_bwd_kernel)
        } else {
            const float yv = y[o];
            dsum = sc * gy[o] * (yv > 0.0f ? 1.0f : slope);
            G[o] = dsum;
            const int hj = hitj[o];
            sm[lane] = 0;
            sm[64 + lane] = 0;
            __builtin_amdgcn_wave_barrier();
            atomicOr(&sm[hj], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            const int rank = __popcll(sm[hj] & ((1ull << lane) - 1ull));
            if (rank < HIT_REC) sval[wave][hj][rank] = dsum;
            __builtin_amdgcn_wave_barrier();
            if (lane < k) {
                HitRec *h = hits + (size_t)pt * k + lane;
                h->mask = sm[lane];
                *reinterpret_cast<float4 *>(h->v) = *reinterpret_cast<const float4 *>(sval[wave][lane]);
            }
            if (64 + lane < k) {
                HitRec *h = hits + (size_t)pt * k + 64 + lane;
                h->mask = sm[64 + lane];
                *reinterpret_cast<float4 *>(h->v) = *reinterpret_cast<const float4 *>(sval[wave][64 + lane]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        dD[o] = dsum + (float)k * A - alpha * (usum[o] - (float)k * q);
        // rows of dR that edge_rgather_kernel assembles with atomics (lists of more than one segment) start at zero
        if (rowptr[pt + 1] - rowptr[pt] > REV_SEG) dR[o] = 0.0f;
    }
}

int rev_chunks(int B, int N, int k)
{
    // about one workgroup per CU, at least ~4 edges per thread
    int c = ceil_div(128, B);
    const int cap = ceil_div(N * k, 4 * REV_THREADS);
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}

int launch_rev_build(const int32_t *idx, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int32_t *rev, int *extra,
                     hipStream_t st)
{
    DG_REQUIRE(N <= REV_MAX_N, "dgcnn_edgeconv_bwd: N=%d > %d unsupported in training mode", N, REV_MAX_N);
    const int P = B * N, chunks = rev_chunks(B, N, k);
    const size_t lds = (size_t)N * sizeof(int);
    static bool attr_done = false;
    if (!attr_done) {  // more than the default 64 KB of dynamic LDS for large N
        (void)hipFuncSetAttribute((const void *)rev_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, REV_MAX_N * 4);
        (void)hipFuncSetAttribute((const void *)rev_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, REV_MAX_N * 4);
        attr_done = true;
    }
    int *nextra = extra;  // [0]: counter, then (point, segment) pairs
    int2 *elist = reinterpret_cast<int2 *>(extra + 2);
    hipLaunchKernelGGL(rev_count_kernel, dim3(B * chunks), dim3(REV_THREADS), lds, st, idx, N, k, chunks, cnt);
    hipLaunchKernelGGL(rev_degree_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, cnt, N, chunks, P, indeg);
    hipLaunchKernelGGL(rev_scan_kernel, dim3(B), dim3(REV_THREADS), 0, st, indeg, N, k, rowptr, nextra);
    hipLaunchKernelGGL(rev_slots_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, rowptr, N, chunks, P, cnt, nextra, elist);
    hipLaunchKernelGGL(rev_place_kernel, dim3(B * chunks), dim3(REV_THREADS), lds, st, idx, N, k, chunks, cnt, rev);
    return 0;
}

int launch_edge_bwd_nogather(const float *TP, const float *TQ, int B, int N, int k, const float *stats, const float *cc,
                             float slope, const float *gy, const float *y, const uint8_t *hitj, const float *usum,
                             const float *dv, const float *dvsum, const int *rowptr, const int32_t *rev, const int *extra,
                             void *hits,
                             float *G, float *dR, float *dD, hipStream_t st)
{
    const int P = B * N;
    const dim3 grid(stream_grid(P)), block(256);
    const int *nextra = extra;
    const int2 *elist = reinterpret_cast<const int2 *>(extra + 2);
    if (dv)
        hipLaunchKernelGGL((edge_point_bwd_kernel<true>), grid, block, 0, st, TQ, N, k, P, stats, cc, slope, gy, y, hitj, usum,
                           dvsum, (HitRec *)hits, G, rowptr, dR, dD);
    else
        hipLaunchKernelGGL((edge_point_bwd_kernel<false>), grid, block, 0, st, TQ, N, k, P, stats, cc, slope, gy, y, hitj, usum,
                           nullptr, (HitRec *)hits, G, rowptr, dR, dD);
    prof_begin(PROF_EDGE_RGATHER, st);
    // one launch: the point workgroups and, behind them, workgroups for the extra segments of the long lists
    const int main_blocks = (int)grid.x, extra_blocks = main_blocks / 2;
    const dim3 ggrid(main_blocks + extra_blocks);
    if (dv)
        hipLaunchKernelGGL((edge_rgather_kernel<true>), ggrid, block, 0, st, TP, TQ, rowptr, rev, k, P, stats, cc, dv,
                           (const HitRec *)hits, G, nextra, elist, main_blocks, dR);
    else
        hipLaunchKernelGGL((edge_rgather_kernel<false>), ggrid, block, 0, st, TP, TQ, rowptr, rev, k, P, stats, cc, dv,
                           (const HitRec *)hits, G, nextra, elist, main_blocks, dR);
    prof_end(PROF_EDGE_RGATHER, st);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Materialised edge features for the public get_graph_feature API (model.py:64-81).
// thread = one float of out [B][N][k][2C].
__global__ void graph_feature_fwd_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx, int N, int C,
                                         int k, size_t total, float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c2 = (int)(t % (2 * C));
    const size_t e = t / (2 * C);  // edge (b, i, j)
    const size_t pt = e / k;       // point (b, i)
    const size_t b = pt / N;
    const float ctr = x[pt * C + (c2 < C ? c2 : c2 - C)];
    float v = ctr;
    if (c2 < C) v = x[(b * N + idx[e]) * C + c2] - ctr;
    out[t] = v;
}

// gx[b][i] += sum_j (g_ctr[i][j] - g_diff[i][j]);  gx[b][n(i, j)] += g_diff[i][j]
__global__ void graph_feature_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx, int N, int C,
                                         int k, size_t npc, float *__restrict__ gx)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (point, channel)
    if (t >= npc) return;
    const int c = (int)(t % C);
    const size_t pt = t / C, b = pt / N;
    float s = 0.0f;
    for (int j = 0; j < k; ++j) {
        const size_t e = pt * k + j;
        const float gd = gout[e * 2 * C + c], gc = gout[e * 2 * C + C + c];
        s += gc - gd;
        atomicAdd(&gx[(b * N + idx[e]) * C + c], gd);
    }
    atomicAdd(&gx[t], s);
}

// ---------------------------------------------------------------------------------------------
void launch_edge_reduce(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *gamma,
                        float *usel, float *usum, uint8_t *hitj, double *partial, hipStream_t st)
{
    const dim3 grid(edge_grid(P)), block(256);
    prof_begin(PROF_EDGE_REDUCE, st);
    if (usel)
        hipLaunchKernelGGL((edge_reduce_kernel<true>), grid, block, 0, st, TP, TQ, idx, N, k, P, gamma, usel, usum, hitj,
                           partial);
    else
        hipLaunchKernelGGL((edge_reduce_kernel<false>), grid, block, 0, st, TP, TQ, idx, N, k, P, gamma, usel, usum, hitj,
                           partial);
    prof_end(PROF_EDGE_REDUCE, st);
}

void launch_bn_finalize(double *partial, int nparts, double count, int training, const dgcnn_bn_params &bn,
                        float eps, float momentum, float *stats, hipStream_t st)
{
    const double *rec = fold_records(partial, nparts, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, st, rec, nparts, count, training, bn, eps,
                       momentum, stats);
}

void launch_edge_out(const float *usel, const float *stats, float slope, int P, float *y, hipStream_t st)
{
    const size_t n4 = (size_t)P * EC / 4;
    hipLaunchKernelGGL(edge_out_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, usel, stats, slope, n4, y);
}

void launch_bn_bwd_finalize(double *partial, int nparts, double count, int training, float *gbn_w, float *gbn_b,
                            float *cc, hipStream_t st)
{
    const double *rec = fold_records(partial, nparts, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, rec, nparts, count, training, gbn_w,
                       gbn_b, cc);
}

void launch_bwd_point(const float *gy, const float *usel, const float *stats, float slope, int P, double count,
                      int training, double *partial, float *gbn_w, float *gbn_b, float *cc, hipStream_t st)
{
    int grid = ceil_div(P, EDGE_WAVES * 8);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(bwd_point_kernel, dim3(grid), dim3(256), 0, st, gy, usel, stats, slope, P, partial);
    launch_bn_bwd_finalize(partial, grid, count, training, gbn_w, gbn_b, cc, st);
}

void launch_edge_bwd(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *stats,
                     const float *cc, float slope, int training, const float *gy, const float *y, const float *dv,
                     float *dR, float *dD, hipStream_t st)
{
    const dim3 grid(edge_grid(P)), block(256);
    if (dv)
        hipLaunchKernelGGL((edge_bwd_kernel<true>), grid, block, 0, st, TP, TQ, idx, N, k, P, stats, cc, slope,
                           training, gy, y, dv, dR, dD);
    else
        hipLaunchKernelGGL((edge_bwd_kernel<false>), grid, block, 0, st, TP, TQ, idx, N, k, P, stats, cc, slope,
                           training, gy, y, dv, dR, dD);
}

void launch_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k, float *out,
                              hipStream_t st)
{
    const size_t total = (size_t)B * N * k * 2 * C;
    hipLaunchKernelGGL(graph_feature_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, idx, N, C,
                       k, total, out);
}

void launch_graph_feature_bwd(const float *gout, const int32_t *idx, int B, int N, int C, int k, float *gx,
                              hipStream_t st)
{
    const size_t npc = (size_t)B * N * C;
    hipLaunchKernelGGL(graph_feature_bwd_kernel, dim3((unsigned)((npc + 255) / 256)), dim3(256), 0, st, gout, idx, N,
                       C, k, npc, gx);
}

}  // namespace dgcnn
