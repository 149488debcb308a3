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
#include "fold.h"
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
int edge_grid(int P)  // edge_reduce: also the number of fp64 partial records (<= PART_MAX)
{
    static const int v = getenv("DGCNN_REDUCE_SLOTS") ? atoi(getenv("DGCNN_REDUCE_SLOTS")) : 0;  // experiments
    // (measured at B = 8, N = 7000, k = 30 with the in-launch fold: 66.8 / 64.5 / 64.5 / 65.2 us at 512 / 384 / 256 / 192 slots)
    return point_grid(P, EDGE_WAVES, v ? v : (slots_override() ? slots_override() : 256));
}
static int stream_grid(int P)  // passes without partial records
{
    return point_grid(P, EDGE_WAVES, slots_override() ? slots_override() : 256);
}

// ---------------------------------------------------------------------------------------------
// Also written for the backward: usum[i] = sum_j u[i][j] (fp32) and, with WANT_EXT, hitj[i] = the slot j (0..k-1)
// of the FIRST edge attaining the extreme (PyTorch's max routes the gradient to the first maximum).
// FIN: the pass also finishes the BatchNorm batch statistics (training mode, fold.h); without it the sums are dropped.
template <int NB>
__device__ __forceinline__ void rev_slots_pass(const RevSlotsArgs &a, int block);
template <bool WANT_EXT, bool FIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void edge_reduce_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                          const int32_t *__restrict__ idx, int N, int k, int P,
                                                          const float *__restrict__ gamma, float *__restrict__ usel,
                                                          float *__restrict__ usum, uint8_t *__restrict__ hitj,
                                                          FoldSet fs, BnBatchStats fin, int nmain, RevSlotsArgs slots)
{
    // workgroups nmain .. (training forward of a block whose per-point product carried the counting pass of the backward's
    // reverse lists): the list-space pass, which needs those counts and nothing of this pass
    // (four loads in flight per thread instead of sixteen: the rider must fit the register budget of THIS pass, whose eight
    // waves per SIMD hide its gathers -- with the sixteen-deep form the kernel went from 62 to 73 registers, six waves, and
    // lost 6 us at B = 8 and 63 us at B = 32; the rider has the whole gather pass to finish in)
    if (FIN && (int)blockIdx.x >= nmain) {
        rev_slots_pass<4>(slots, (int)blockIdx.x - nmain);
        return;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (visibly uniform)
    const bool neg = WANT_EXT ? (gamma[lane] < 0.0f) : false;
    // the extreme is tracked on sign-flipped values where gamma < 0: s p + s q is the exact negation of u = p + q (one
    // fma with s = +-1), so one comparison serves both directions; the sums are flipped back at the end
    const float sgn = neg ? -1.0f : 1.0f;
    double acc[2] = {0.0, 0.0};
    const PointIter it = point_iter_at(P, wave, EDGE_WAVES, (int)blockIdx.x, nmain);
    // The pass is latency-bound (index row -> neighbour rows -> arithmetic), so every point's loads are issued
    // as one batch of up to 32 rows, and the next point's index row and TQ row are fetched one point ahead.
    // Once the loads are batched the vector instructions per gathered row bound it (measured: 12.6 per row, 62 % VALU
    // issue): now eight -- v_readlane for the row index (the address is scalar arithmetic on it plus the lane offset),
    // the fma above, three for the fp64 BatchNorm sums, compare + two selects for the first extreme; usum falls out of the fp64 sum (difference of the running sum across the point).
    int pt = it.cur;
    int nv_next = 0;
    float q_next = 0.0f;
    if (pt < it.end) {
        nv_next = lane < k ? __builtin_nontemporal_load(&idx[(size_t)pt * k + lane]) : 0;
        q_next = __builtin_nontemporal_load(&TQ[(size_t)pt * EC + lane]);
    }
    for (; pt < it.end; pt += it.step) {
        const int b = pt / N;
        // the batch element's table behind a buffer descriptor: a gather is then one instruction with the row offset
        // in a scalar register and the lane's column in ONE address register (no 64-bit address per row in flight)
        const __amdgpu_buffer_rsrc_t Pb = __builtin_amdgcn_make_buffer_rsrc((void *)(TP + (size_t)b * N * EC), 0, N * EC * 4, 0x00020000);
        const float qs = q_next * sgn;
        int nv = nv_next;  // neighbours 0..63 of this point (lane j holds idx[pt][j])
        const int32_t *ip = idx + (size_t)pt * k;
        const int ptn = pt + it.step;
        if (ptn < it.end) {
            nv_next = lane < k ? __builtin_nontemporal_load(&idx[(size_t)ptn * k + lane]) : 0;
            q_next = __builtin_nontemporal_load(&TQ[(size_t)ptn * EC + lane]);
        }
        float ext = -__builtin_inff();
        int hit = 0;
        const double acc0_before = acc[0];
        auto row = [&](float pv, int slot) {
            const float u = __builtin_fmaf(pv, sgn, qs);
            acc[0] += (double)u;
            acc[1] = __builtin_fma((double)u, (double)u, acc[1]);
            if (WANT_EXT) {
                const bool keep = !(u > ext);  // strict: the first edge attaining the extreme wins
                ext = keep ? ext : u;
                hit = keep ? hit : slot;
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
                            p[g * 8 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Pb, lane * 4, n * (EC * 4), 0));
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
        __builtin_nontemporal_store((float)(acc[0] - acc0_before) * sgn, &usum[(size_t)pt * EC + lane]);
        if (WANT_EXT) {
            __builtin_nontemporal_store(ext * sgn, &usel[(size_t)pt * EC + lane]);
            __builtin_nontemporal_store((uint8_t)hit, &hitj[(size_t)pt * EC + lane]);
        }
    }
    acc[0] *= (double)sgn;
    if (FIN) fold_and_finalize<2, EDGE_WAVES>(acc, fs, fin, nmain);
}

__global__ void edge_out_kernel(const float *__restrict__ usel, const float *__restrict__ stats, float slope,
                                size_t n4, float *__restrict__ y, int ldy)
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
    *reinterpret_cast<float4 *>(y + (t >> 4) * (size_t)ldy + c) = o;  // (ldy: y may be a 64-column slice of a wider row)
}

// alloc[0]: list space handed out so far, alloc[1]: number of extra (point, segment) pairs (both zeroed by
// rev_count_kernel).  Per point: rowptr (list start), indeg, segbase (index of its first pair in `extra`), pcnt = 0 (the
// arrival counter of its segments in edge_rgather_kernel).
// (A device function: the pass rides in the launch of bwd_point_kernel -- both are short passes that leave most of the
// chip idle on their own -- as the workgroups behind that kernel's own; `block` = index among the pass's workgroups.)
// NB: loads in flight per thread over the chunks of a point's count column
template <int NB>
__device__ __forceinline__ void rev_slots_pass(const RevSlotsArgs &a, int block)
{
    // a merged build (nblk graphs of Pblk points each, side by side: the three blocks of the model in one launch): a workgroup
    // stays inside ONE graph, which has its own list space (alloc, rev) and its own extra-pair list; the per-point arrays and
    // the count table are indexed by the merged point number
    const int N = a.N, chunks = a.chunks;
    const int wgs = (a.Pblk + 255) / 256, blk = block / wgs, gl = (block - blk * wgs) * 256 + (int)threadIdx.x;
    const bool live = gl < a.Pblk;
    int *__restrict__ cnt = a.cnt, *__restrict__ rowptr = a.rowptr, *__restrict__ indeg = a.indeg;
    int *__restrict__ segbase = a.segbase, *__restrict__ pcnt = a.pcnt, *__restrict__ alloc = a.alloc + (size_t)blk * a.alloc_stride;
    int2 *__restrict__ extra = a.extra + (size_t)blk * a.extra_stride;
    const int g = blk * a.Pblk + gl;
    int deg = 0;
    int *cb = nullptr, *co = nullptr;
    if (live) {
        const int b = g / N, m = g - b * N;
        cb = cnt + (size_t)b * chunks * N + m;
        co = (a.cnt_out ? a.cnt_out : cnt) + (size_t)b * chunks * N + m;
        int c = 0;
        for (; c + NB <= chunks; c += NB) {  // NB loads in flight (alone in a launch: one workgroup per CU, nothing else hides them)
            int v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = cb[(size_t)(c + u) * N];
#pragma unroll
            for (int u = 0; u < NB; ++u) deg += v[u];
        }
        for (; c < chunks; ++c) deg += cb[(size_t)c * N];
    }
    const int nseg = deg > REV_SEG ? (deg - 1) / REV_SEG : 0;  // segments beyond the first
    // exclusive scans of deg and nseg over the workgroup; ONE atomic each on the global counters per workgroup
    __shared__ int wtot[4][2], wbase[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc_d = deg, inc_s = nseg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int td = __shfl_up(inc_d, o), ts = __shfl_up(inc_s, o);
        if (lane >= o) {
            inc_d += td;
            inc_s += ts;
        }
    }
    if (lane == 63) {
        wtot[wave][0] = inc_d;
        wtot[wave][1] = inc_s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int td = 0, ts = 0;
        for (int w = 0; w < 4; ++w) {
            wbase[w][0] = td;
            wbase[w][1] = ts;
            td += wtot[w][0];
            ts += wtot[w][1];
        }
        const int bd = td ? atomicAdd(&alloc[0], td) : 0;
        const int bs = ts ? atomicAdd(&alloc[1], ts) : 0;
        for (int w = 0; w < 4; ++w) {
            wbase[w][0] += bd;
            wbase[w][1] += bs;
        }
    }
    __syncthreads();
    if (live) {
        int slot = wbase[wave][0] + inc_d - deg;
        const int sb = wbase[wave][1] + inc_s - nseg;
        rowptr[g] = slot;
        indeg[g] = deg;
        segbase[g] = sb;
        pcnt[g] = 0;
        int c = 0;
        for (; c + NB <= chunks; c += NB) {
            int v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = cb[(size_t)(c + u) * N];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                co[(size_t)(c + u) * N] = slot;
                slot += v[u];
            }
        }
        for (; c < chunks; ++c) {
            const int n = cb[(size_t)c * N];
            co[(size_t)c * N] = slot;
            slot += n;
        }
        for (int sg = 0; sg < nseg; ++sg) extra[sb + sg] = int2{gl, sg + 1};  // (the point's number inside its graph)
    }
}


// ---------------------------------------------------------------------------------------------
// Backward of the LAST BatchNorm of the block.  Only the arg-max edge of each (point, channel)
// carries gradient through max-over-k, so  sum dv  and  sum dv * uhat  over all edges reduce to
// sums over points of quantities at the selected extreme.  The pass finishes
// the sums itself (fold.h): gbn_b, gbn_w and the two mean terms cc.
// gy2 != NULL: dL/dy = gy + gy2 (gy2 dense [P][64]: the second gradient that reaches the block's output -- the next block's
// input gradient: one extra stream here instead of a read-modify-write of the strided slice in the product that makes gy2:
// 8 -> 22 us there at P = 28 000).  gdv [P][64] <- dL/dy . LReLU'(y) (y > 0 <=> sc usel + sh > 0): what every kernel behind
// this one needs of (dL/dy, y) -- they read ONE dense row per point instead of two rows that may be column slices of wider ones.
template <bool K2>
__device__ __forceinline__ void rev_place_pass(const RevPlaceArgs &a, int block, int *table);
__global__ __launch_bounds__(256) void bwd_point_kernel(const float *__restrict__ gy, int ldgy, const float *__restrict__ gy2,
                                                        float *__restrict__ gdv, const float *__restrict__ usel,
                                                        const float *__restrict__ stats, float slope, int P,
                                                        FoldSet fs, BnBackwardSums fin, unsigned *__restrict__ gmax_bits,
                                                        int nbp, RevSlotsArgs slots, RevPlaceArgs place, int nslots)
{
    // workgroups nbp .. : the list-space pass of the reverse adjacency (rev_slots_pass), which only needs rev_count_kernel
    // before it and nothing of this pass
    // workgroups nbp + nslots .. : the placement pass of the reverse lists when the forward has counted and allotted them (one wave
    // of each workgroup works; the pass is bound by its LDS atomics, this one by its streams: 21.0 + 19.2 us as two launches,
    // 27.3 as one at B = 8, N = 7000, k = 30)
    if ((int)blockIdx.x >= nbp + nslots) {
        __shared__ int table[REV_RIDER_MAX_N];
        if (threadIdx.x < 64) rev_place_pass<false>(place, (int)blockIdx.x - nbp - nslots, table);
        return;
    }
    if ((int)blockIdx.x >= nbp) {
        rev_slots_pass<16>(slots, (int)blockIdx.x - nbp);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane];
    const float sc = stats[ST_SC + lane], sh = stats[ST_SH + lane];
    double acc[2] = {0.0, 0.0};
    float gmax = 0.0f;
    // sixteen points in flight per wave (a wave owns 27 points at B = 8, ~100 at B = 32: one load pair per round trip
    // was 66 us at B = 32, four pairs 19 us at B = 8 -- seven dependent round trips)
    constexpr int INF = 16;
    const int stride = nbp * EDGE_WAVES;
    for (int pt = blockIdx.x * EDGE_WAVES + wave; pt < P; pt += INF * stride) {
        float us[INF], g4[INF], g2[INF];
#pragma unroll
        for (int u = 0; u < INF; ++u) {
            const int p = pt + u * stride < P ? pt + u * stride : pt;  // clamped: the load is unconditional, the value masked
            us[u] = usel[(size_t)p * EC + lane];
            g4[u] = gy[(size_t)p * ldgy + lane];
            // (its own register: added below, behind ALL the loads -- as `g4[u] += gy2[..]` here the sum waited for its two loads
            // before the next point's were issued: 100 us against 51 without the second addend at B = 32)
            g2[u] = gy2 ? gy2[(size_t)p * EC + lane] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < INF; ++u) g4[u] += g2[u];
#pragma unroll
        for (int u = 0; u < INF; ++u) {
            if (pt + u * stride < P) {  // (uniform; added in point order)
                const float v = __builtin_fmaf(us[u], sc, sh);
                const float dv = g4[u] * (v > 0.0f ? 1.0f : slope);
                // (streamed: 57 MB at B = 32 that nothing reads before the list kernels behind this one have run -- kept in
                // L2 they crowd out the merging of the placement pass's scattered stores: rev_place 87 -> 111 us per launch)
                __builtin_nontemporal_store(dv, &gdv[(size_t)(pt + u * stride) * EC + lane]);
                const float uhat = (us[u] - mean) * rstd;
                acc[0] += (double)dv;
                acc[1] = __builtin_fma((double)dv, (double)uhat, acc[1]);
                gmax = fmaxf(gmax, fabsf(dv));
            }
        }
    }
    // max |dv| for the operand scale of edge_mlp_bwd16_kernel: non-negative floats order like their bits, and a
    // maximum does not depend on the order of its arguments
    if (gmax_bits) {  // (one atomic per workgroup, and only while it still raises the maximum: same-address atomics queue)
        __shared__ float s_gmax[EDGE_WAVES];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        if (lane == 0) s_gmax[wave] = gmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = s_gmax[0];
#pragma unroll
            for (int w = 1; w < EDGE_WAVES; ++w) m = fmaxf(m, s_gmax[w]);
            const unsigned bits = __float_as_uint(m);
            if (bits > __hip_atomic_load(gmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gmax_bits, bits);
        }
    }
    fold_and_finalize<2, EDGE_WAVES>(acc, fs, fin, nbp);
}

// ---------------------------------------------------------------------------------------------
// Stage-1 backward as a GATHER over the reverse adjacency (CSR by destination) of the neighbour graph.
// With du = sc (dv - c1 - c2 uhat) (Appendix C.5; c1 = c2 = 0 in eval mode), alpha = sc c2 rstd and
// A[i] = -sc (c1 + c2 rstd (TQ[i] - mean)) = a0 - alpha TQ[i] (the part of du that is the same for all k edges of i)
// the edge gradient is du[i][j] = sc dv[i][j] + A[i] - alpha TP[n(i, j)], hence
//   dR[m] = indeg(m) a0 - alpha (sum_{i in in(m)} TQ[i] + indeg(m) TP[m]) + sum_{e in in(m)} sc dv[e]   (IndexBackward0)
//   dD[i] = sum_j sc dv[i][j] + k A[i] - alpha (usum[i] - k TQ[i])          (Repeat/Sub/CatBackward0; usum from the forward)
// The scatter form of dR costs one 256-byte wave-atomic per edge (global float atomics: ~1.3 TB/s of added bytes on
// MI355X: 0.4 ms for 1.68 M edges) and sums in arrival order.  The gather needs one coalesced 256-byte row load per
// in-edge, no float atomic on global memory, and -- the lists being SORTED by edge id -- adds in a fixed order: the
// backward is bit-stable from run to run.
//
// Reverse lists, deterministic by construction.  The sources of a batch element are cut into `chunks` contiguous row
// ranges; ONE WAVE owns a range and walks its rows in order, lane j holding neighbour j of the row.  The neighbours of
// a row are distinct (kNN), so the 64 lanes of one step touch distinct entries of the wave's N-entry LDS table and a
// step needs no ordering among its lanes; steps execute in program order.  Hence
//   rev_count : cnt[b][chunk][m]  = number of edges from the chunk's rows into m               (ds_add, in LDS)
//   rev_slots : indeg[m] = sum over chunks; list space for m is allocated (one global atomic per 256 points: WHERE a
//               list lies is arbitrary, its contents are not); cnt[b][chunk][m] <- first slot of the chunk in m's list;
//               the (point, segment) pairs beyond the first REV_SEG entries of long lists are listed as extra work
//   rev_place : rev[table[m]++] = edge id, rows in order (ds_add_rtn)  ->  every list ascends in edge id.
// (Rows with REPEATED neighbours -- no kNN graph has them -- are still handled correctly; only the order of the repeated
// pair inside the list is then left to the LDS unit.)
// (REV_MAX_N, edge_kernels.h: N ints of LDS per wave, 160 KB per CU)
constexpr int REV_ROWS = 8;       // rows per step; two steps are in flight (the next step's index rows are loaded
                                  // while this step's go through the LDS: two waves per CU, nothing else hides a load)

// index rows r .. r + REV_ROWS - 1 of one batch element, lane j = neighbour j (K2: also neighbours 64 .. k - 1); rows
// past r1 and lanes past k read a clamped address and come back negative (no branch around a load: the waits stay counted)
template <bool K2>
struct RevRows {
    int n0[REV_ROWS], n1[K2 ? REV_ROWS : 1];
};
template <bool K2>
__device__ __forceinline__ void rev_load_rows(const int32_t *__restrict__ ib, int k, int r, int r1, int lane, RevRows<K2> &v)
{
    const int l0 = lane < k ? lane : k - 1, l1 = 64 + lane < k ? 64 + lane : k - 1;
#pragma unroll
    for (int u = 0; u < REV_ROWS; ++u) {
        const int rr = r + u < r1 ? r + u : r1 - 1;
        // (the sign bit marks an invalid entry; or-ed in, so that the load is needed on every path and the compiler
        // cannot sink it under a branch with a full wait behind it)
        v.n0[u] = ib[(size_t)rr * k + l0] | (r + u < r1 && lane < k ? 0 : (int)0x80000000);
        if (K2) v.n1[u] = ib[(size_t)rr * k + l1] | (r + u < r1 && 64 + lane < k ? 0 : (int)0x80000000);
    }
}

template <bool K2>
__global__ __launch_bounds__(64) void rev_count_kernel(const int32_t *__restrict__ idx, int N, int k, int chunks,
                                                       int rows_per, int *__restrict__ cnt,
                                                       int *__restrict__ zero_words, int nzero)
{
    extern __shared__ int table[];
    const int lane = threadIdx.x;
    // workgroups are dealt round-robin over the 8 XCDs: batch element = blockIdx % B keeps all writers of a batch
    // element's lists (scattered 4-byte stores into one region of `rev`) behind ONE L2, where they merge into whole lines
    const int B_ = (int)gridDim.x / chunks, b = (int)blockIdx.x % B_, chunk = (int)blockIdx.x / B_;
    const int row = b * chunks + chunk;  // this chunk's row of cnt
    // the counters of the kernels behind this one in the same call (fold.h tickets, list allocation) start at zero
    if (blockIdx.x == 0)
        for (int i = lane; i < nzero; i += 64) zero_words[i] = 0;
    const int r0 = chunk * rows_per, r1 = r0 + rows_per < N ? r0 + rows_per : N;
    const int32_t *ib = idx + (size_t)b * N * k;
    int *out = cnt + (size_t)row * N;
    if (r0 >= r1) {  // (an empty chunk: only when N is not a multiple of the chunk size)
        for (int m = lane; m < N; m += 64) out[m] = 0;
        return;
    }
    RevRows<K2> ra, rb;
    rev_load_rows<K2>(ib, k, r0, r1, lane, ra);
    for (int m = lane; m < N; m += 64) table[m] = 0;
    auto count = [&](const RevRows<K2> &v) {
#pragma unroll
        for (int u = 0; u < REV_ROWS; ++u) {
            if (v.n0[u] >= 0) atomicAdd(&table[v.n0[u]], 1);
            if (K2 && v.n1[u] >= 0) atomicAdd(&table[v.n1[u]], 1);
        }
    };
    for (int r = r0; r < r1; r += 2 * REV_ROWS) {  // ping-pong: the other buffer's rows are in flight
        rev_load_rows<K2>(ib, k, r + REV_ROWS, r1, lane, rb);
        count(ra);
        rev_load_rows<K2>(ib, k, r + 2 * REV_ROWS, r1, lane, ra);
        count(rb);
    }
    for (int m0 = 0; m0 < N; m0 += 64 * 16) {  // 16 LDS reads in flight, then 16 stores
        int v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = m0 + u * 64 + lane < N ? table[m0 + u * 64 + lane] : 0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (m0 + u * 64 + lane < N) out[m0 + u * 64 + lane] = v[u];
    }
}

// (A device function of one wave; block = index among the pass's B chunks workgroups.)
template <bool K2>
__device__ __forceinline__ void rev_place_pass(const RevPlaceArgs &a, int block, int *table)
{
    const int32_t *__restrict__ idx = a.idx;
    const int N = a.N, k = a.k, chunks = a.chunks, rows_per = a.rows_per;
    const int *__restrict__ cnt = a.cnt;
    const int lane = threadIdx.x & 63;
    // workgroups are dealt round-robin over the 8 XCDs: batch element = block % B keeps all writers of a batch
    // element's lists (scattered 4-byte stores into one region of `rev`) behind ONE L2, where they merge into whole lines
    const int B_ = a.nblocks / chunks, b = block % B_, chunk = block / B_;
    const int row = b * chunks + chunk;  // this chunk's row of cnt
    const int *first = cnt + (size_t)row * N;
    const int r0 = chunk * rows_per, r1 = r0 + rows_per < N ? r0 + rows_per : N;
    if (r0 >= r1) return;
    const int32_t *ib = idx + (size_t)b * N * k;
    // (a merged build: batch element b belongs to graph b / Bblk, whose lists live in their own E-entry range of rev and hold
    // edge ids local to the graph)
    const int blk = b / a.Bblk, bl = b - blk * a.Bblk;
    int32_t *__restrict__ rev = a.rev + (size_t)blk * a.Eblk;
    // the first slots of this chunk's edges in every list: 16 loads in flight, then 16 LDS writes
    auto load_table = [&]() {
        for (int m0 = 0; m0 < N; m0 += 64 * 16) {
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = first[m0 + u * 64 + lane < N ? m0 + u * 64 + lane : 0];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (m0 + u * 64 + lane < N) table[m0 + u * 64 + lane] = v[u];
        }
    };
    // (measured, round 4: a ring of four row groups instead of two, and a branch-free form for k <= 64 -- lanes past k sitting out,
    // whole groups only, the eight LDS atomics of a group in flight together and its eight stores in one batch: 107 -> 109 us per
    // launch at B = 32 with the index rows coming from memory, the rider in bwd_point 76 -> 86 us per step at B = 4.  What the
    // pass follows is the latency of its index rows when they are not in L2: 80 us per launch at B = 32 directly behind the
    // counting pass, which leaves them there, 107 us since that pass moved into the forward; a probe kernel that reads the rows
    // ahead brings the 80 back and costs the 26 it saves.)
    RevRows<K2> ra, rb;
    rev_load_rows<K2>(ib, k, r0, r1, lane, ra);
    load_table();
    auto place = [&](const RevRows<K2> &v, int r) {
#pragma unroll
        for (int u = 0; u < REV_ROWS; ++u) {
            const int e = (bl * N + r + u) * k + lane;  // edge id = (source point within its graph) * k + slot
            if (v.n0[u] >= 0) rev[atomicAdd(&table[v.n0[u]], 1)] = e;
            if (K2 && v.n1[u] >= 0) rev[atomicAdd(&table[v.n1[u]], 1)] = e + 64;
        }
    };
    for (int r = r0; r < r1; r += 2 * REV_ROWS) {  // ping-pong: the other buffer's rows are in flight
        rev_load_rows<K2>(ib, k, r + REV_ROWS, r1, lane, rb);
        place(ra, r);
        rev_load_rows<K2>(ib, k, r + 2 * REV_ROWS, r1, lane, ra);
        place(rb, r + REV_ROWS);
    }
}

template <bool K2>
__global__ __launch_bounds__(64) void rev_place_kernel(RevPlaceArgs a)
{
    extern __shared__ int table[];
    rev_place_pass<K2>(a, (int)blockIdx.x, table);
}

int rev_chunks(int B, int N, int k)
{
    (void)k;
    // one wave per chunk, 4 N bytes of LDS each: about two waves per CU (512 per launch), as many as fit for large N,
    // at least 16 rows per chunk.  The count table cnt [B chunks][N] ints is written and read three times: 14 MB at
    // 512 chunks of N = 7000.
    int per_cu = (160 * 1024) / (4 * N + 64);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    static const int mult = getenv("DGCNN_REV_MULT") ? atoi(getenv("DGCNN_REV_MULT")) : 0;  // experiments
    if (mult > 0) per_cu = mult;
    int c = ceil_div(256 * per_cu, B);
    const int cap = ceil_div(N, 16);
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}

RevSlotsArgs rev_slots_args(int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase, int *pcnt, int *alloc,
                            int *extra)
{
    return RevSlotsArgs{N, rev_chunks(B, N, k), B * N, cnt, rowptr, indeg, segbase, pcnt, alloc, reinterpret_cast<int2 *>(extra), 1, B * N,
                        0, 0};
}
__global__ __launch_bounds__(256) void rev_slots_kernel(RevSlotsArgs a) { rev_slots_pass<16>(a, (int)blockIdx.x); }

int launch_rev_build(const int32_t *idx, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase, int *pcnt,
                     int32_t *rev, int *alloc, int *extra, int *zero_words, int nzero, bool with_slots, bool with_place,
                     hipStream_t st)
{
    DG_REQUIRE(N <= REV_MAX_N, "dgcnn: N=%d > %d unsupported in the backward", N, REV_MAX_N);
    const int P = B * N, chunks = rev_chunks(B, N, k), rows_per = ceil_div(N, chunks);
    const size_t lds = (size_t)N * sizeof(int);
    if (lds > 64 * 1024) {
        // more than the default 64 KB of dynamic LDS: the attribute belongs to the (kernel, device) pair, so it is set
        // once per device of the process (the flags are only ever raised; a second thread that finds one still down sets
        // the same attribute again, which is harmless)
        static bool attr_done[64] = {};
        int dev = 0;
        DG_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "dgcnn: no current HIP device");
        if (!attr_done[dev]) {
            const void *fns[4] = {(const void *)rev_count_kernel<false>, (const void *)rev_count_kernel<true>,
                                  (const void *)rev_place_kernel<false>, (const void *)rev_place_kernel<true>};
            for (const void *fn : fns)
                DG_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, REV_MAX_N * 4) == hipSuccess,
                           "dgcnn: cannot reserve %d bytes of dynamic LDS for the reverse-list kernels", REV_MAX_N * 4);
            attr_done[dev] = true;
        }
    }
    if (k > 64)
        hipLaunchKernelGGL((rev_count_kernel<true>), dim3(B * chunks), dim3(64), lds, st, idx, N, k, chunks, rows_per, cnt,
                           zero_words, nzero);
    else
        hipLaunchKernelGGL((rev_count_kernel<false>), dim3(B * chunks), dim3(64), lds, st, idx, N, k, chunks, rows_per, cnt,
                           zero_words, nzero);
    if (!with_slots) return 0;  // (the caller carries the list-space pass in its next launch: launch_bwd_point)
    hipLaunchKernelGGL(rev_slots_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, rev_slots_args(B, N, k, cnt, indeg, rowptr, segbase, pcnt, alloc, extra));
    if (!with_place) return 0;
    return launch_rev_place(idx, B, N, k, cnt, rev, st);
}

// The list-space pass with riders that READ the index rows of every placement chunk (results dropped): the placement pass
// behind this launch follows the latency of its index rows, and since the counting pass runs in the forward nothing has left
// them in the L2 of the XCD that will place them (107 us per launch at B = 32 against 80 directly behind the counting pass).
// A rider wave reads the rows of the placement workgroup with the same blockIdx % 8 -- the same XCD under the round-robin
// dispatch; a different placement would only leave the rows in another L2, as before.
__global__ __launch_bounds__(256) void rev_slots_warm_kernel(RevSlotsArgs a, int nslots, const int32_t *__restrict__ idx, int k, int rows_per,
                                                             int nplace, int *__restrict__ sink)
{
    if ((int)blockIdx.x < nslots) {
        rev_slots_pass<16>(a, (int)blockIdx.x);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (int)blockIdx.x - nslots, x = (int)blockIdx.x & 7;
    // the placement blocks p with p % 8 == x, four per rider workgroup
    const int p = x + 8 * (4 * (r >> 3) + wave);
    if (p >= nplace) return;
    const int B_ = nplace / a.chunks, b = p % B_, chunk = p / B_;
    const int r0 = chunk * rows_per, r1 = r0 + rows_per < a.N ? r0 + rows_per : a.N;
    const int32_t *ib = idx + (size_t)b * a.N * k;
    int acc = 0;
    for (int row = r0; row < r1; row += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ib[(size_t)(row + u < r1 ? row + u : r1 - 1) * k + (lane < k ? lane : 0)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= v[u];
    }
    if (acc == (int)0x80000000) sink[0] = acc;   // (never: indices are non-negative; keeps the loads alive)
}
int launch_rev_slots_place(const int32_t *idx, int B, int N, int k, const int *counts, int *slots, int *indeg, int *rowptr, int *segbase,
                           int *pcnt, int32_t *rev, int *alloc, int *extra, hipStream_t st)
{
    DG_REQUIRE(N <= REV_MAX_N && k <= 64, "dgcnn: reverse lists from the forward's counts take N <= %d, k <= 64", REV_MAX_N);
    RevSlotsArgs sa = rev_slots_args(B, N, k, const_cast<int *>(counts), indeg, rowptr, segbase, pcnt, alloc, extra);
    sa.cnt_out = slots;
    const int nslots = ceil_div(B * N, 256), chunks = rev_chunks(B, N, k), nplace = B * chunks;
    // (eight consecutive riders cover the eight residues of blockIdx % 8 once, wherever the riders start: rider r takes the
    // (r / 8)-th group of four placement blocks of its residue)
    hipLaunchKernelGGL(rev_slots_warm_kernel, dim3(nslots + 8 * ceil_div(ceil_div(nplace, 8), 4)), dim3(256), 0, st, sa, nslots, idx, k,
                       ceil_div(N, chunks), nplace, pcnt);
    return launch_rev_place(idx, B, N, k, slots, rev, st);
}

// The reverse adjacencies of nblk graphs over the same B x N points (idx [nblk][B][N][k], contiguous) in three launches: the
// graphs ride side by side as nblk B batch elements.  Graph g gets its own list space rev + g E, its own allocation counters at
// zero_words + g zero_stride + alloc_off (the first kernel zeroes all nblk zero_stride words) and its own extra-pair list at
// extra + g extra_stride ints; the per-point arrays are [nblk][P].
int launch_rev_build_merged(const int32_t *idx, int nblk, int B, int N, int k, int *cnt, int *indeg, int *rowptr, int *segbase,
                            int *pcnt, int32_t *rev, int *extra, int extra_stride, int *zero_words, int zero_stride, int alloc_off,
                            hipStream_t st)
{
    DG_REQUIRE(N <= REV_MAX_N, "dgcnn: N=%d > %d unsupported in the backward", N, REV_MAX_N);
    DG_REQUIRE(k <= 64, "dgcnn: merged reverse lists take k <= 64");
    const int Bm = nblk * B, chunks = rev_chunks(Bm, N, k), rows_per = ceil_div(N, chunks);
    const size_t lds = (size_t)N * sizeof(int);
    if (lds > 64 * 1024) {
        static bool attr_done[64] = {};
        int dev = 0;
        DG_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "dgcnn: no current HIP device");
        if (!attr_done[dev]) {
            const void *fns[2] = {(const void *)rev_count_kernel<false>, (const void *)rev_place_kernel<false>};
            for (const void *fn : fns)
                DG_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, REV_MAX_N * 4) == hipSuccess,
                           "dgcnn: cannot reserve %d bytes of dynamic LDS for the reverse-list kernels", REV_MAX_N * 4);
            attr_done[dev] = true;
        }
    }
    hipLaunchKernelGGL((rev_count_kernel<false>), dim3(Bm * chunks), dim3(64), lds, st, idx, N, k, chunks, rows_per, cnt, zero_words,
                       nblk * zero_stride);
    const RevSlotsArgs sa{N, chunks, Bm * N, cnt, rowptr, indeg, segbase, pcnt, zero_words + alloc_off, reinterpret_cast<int2 *>(extra),
                          nblk, B * N, zero_stride, extra_stride / 2};
    hipLaunchKernelGGL(rev_slots_kernel, dim3(nblk * ceil_div(B * N, 256)), dim3(256), 0, st, sa);
    const RevPlaceArgs pa{idx, N, k, chunks, rows_per, Bm * chunks, cnt, rev, B, B * N * k};
    hipLaunchKernelGGL((rev_place_kernel<false>), dim3(Bm * chunks), dim3(64), lds, st, pa);
    return 0;
}

int launch_rev_place(const int32_t *idx, int B, int N, int k, const int *cnt, int32_t *rev, hipStream_t st)
{
    const int chunks = rev_chunks(B, N, k), rows_per = ceil_div(N, chunks);
    const size_t lds = (size_t)N * sizeof(int);
    const RevPlaceArgs a{idx, N, k, chunks, rows_per, B * chunks, cnt, rev, B, 0};
    if (k > 64) hipLaunchKernelGGL((rev_place_kernel<true>), dim3(B * chunks), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((rev_place_kernel<false>), dim3(B * chunks), dim3(64), lds, st, a);
    return 0;
}

// ---------------------------------------------------------------------------------------------
constexpr int HIT_REC = 4;  // hit values stored per edge
// what the sparse backward keeps per edge (one 32-byte sector, read once by the gather): the channels whose
// max-over-k picked this edge and the gradient of the first HIT_REC of them
struct __attribute__((aligned(16))) HitRec {
    u64 mask;
    float v[HIT_REC];
    u64 pad;
};
static_assert(sizeof(HitRec) == 32, "HitRec is one 32-byte sector");

// Per-point part of the stage-1 backward:  dD[i] = sum_j sc dv[i][j] + k A[i] - alpha (usum[i] - k TQ[i]).
// DENSE: `dv` here is the per-point sum of dv1 over the k edges [P][64] (edge_mlp_bwd_kernel).
// Sparse (single-stage block): dL/dy of (i, c) belongs to the ONE edge that won max-over-k, slot hitj[i][c] from the
// forward.  The kernel writes the HitRec of every edge of the point (values sc . gdv; an edge with more hits than its record
// holds sends the gather to gdv itself: `G` there IS gdv since round 4, the per-point copy sc . gdv is not written any more); the records of a
// point are assembled in LDS: one 64-bit atomic-or per channel lane builds the masks, then lane j owns edge j.
template <bool DENSE>
__global__ __launch_bounds__(256) void edge_point_bwd_kernel(const float *__restrict__ TQ, int k, int P,
                                                             const float *__restrict__ stats,
                                                             const float *__restrict__ cc, float slope,
                                                             const float *__restrict__ gdv,
                                                             const uint8_t *__restrict__ hitj,
                                                             const float *__restrict__ usum,
                                                             const float *__restrict__ dv, HitRec *__restrict__ hits,
                                                             float *__restrict__ G, float *__restrict__ dD, const SumRider sr)
{
    __shared__ u64 smask[EDGE_WAVES][128];  // by slot (k <= 96)
    __shared__ float sval[EDGE_WAVES][128][HIT_REC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroups sr.main_blocks .. (two-stage blocks): the fold of the second convolution's weight-gradient partials
    // (out[e] = sum over the chunks, in chunk order), which nothing in this call reads: it rides here instead of in a launch
    // of its own.  Four waves take consecutive shares of the chunks, eight in flight; the four sums are added in wave order.
    if (DENSE && sr.out && (int)blockIdx.x >= sr.main_blocks) {
        float *red = reinterpret_cast<float *>(&sval[0][0][0]);  // [4][64]
        const int e = ((int)blockIdx.x - sr.main_blocks) * 64 + lane;
        const int q = (sr.nchunk + EDGE_WAVES - 1) / EDGE_WAVES;
        const int c0 = wave * q, c1 = c0 + q < sr.nchunk ? c0 + q : sr.nchunk;
        float s = 0.0f;
        if (e < sr.n) {
            int ch = c0;
            for (; ch + 8 <= c1; ch += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = sr.partial[(size_t)(ch + u) * sr.n + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            if (ch < c1) {  // the last, partial batch: loads in flight together, past-the-end chunks masked
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = sr.partial[(size_t)(ch + u < c1 ? ch + u : ch) * sr.n + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += ch + u < c1 ? v[u] : 0.0f;
            }
        }
        red[wave * 64 + lane] = s;
        __syncthreads();
        if (wave == 0 && e < sr.n) sr.out[e] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        return;
    }
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane], sc = stats[ST_SC + lane];
    const float c1 = cc[lane], c2 = cc[64 + lane];
    const float alpha = sc * c2 * rstd;
    u64 *sm = smask[wave];
    // (the launch may carry the fold's workgroups behind this pass's: the split over the POINT workgroups only)
    const PointIter it = point_iter_at(P, wave, EDGE_WAVES, (int)blockIdx.x, DENSE && sr.out ? sr.main_blocks : (int)gridDim.x);
    for (int pt = it.cur; pt < it.end; pt += it.step) {
        const size_t o = (size_t)pt * EC + lane;
        const float q = TQ[o];
        const float A = -sc * (c1 + c2 * rstd * (q - mean));
        float dsum;
        if (DENSE) {
            dsum = sc * dv[o];
        } else {
            dsum = sc * gdv[o];  // gdv = dL/dy . LReLU'(y), left by bwd_point_kernel
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
        // (eval mode: alpha = 0 and the forward has not written usum)
        const float mterm = alpha != 0.0f ? alpha * (usum[o] - (float)k * q) : 0.0f;
        dD[o] = dsum + (float)k * A - mterm;
    }
}

// Neighbour graphs in feature space have hubs (at B = 8, N = 7000, k = 30 on Gaussian features: median in-degree 12,
// maximum 1372), and one wave walking a long list alone would set the kernel time.  A list is therefore cut into
// segments of REV_SEG in-edges (64: measured 76 / 69 / 67 / 70 us at 32 / 48 / 64 / 96): rev_slots_kernel lists the
// (point, segment) pairs beyond segment 0, and edge_rgather_kernel runs workgroups for the points (segment 0) and for
// those pairs side by side.  A point with ONE segment is finished by the wave that walks it.  A point with more leaves
// one partial [sq | sd] per segment (written through, sc1) and an arrival count; the wave that arrives LAST adds the
// partials in SEGMENT order and writes the row (fold.h's hand-off: payload sc1 + drained, ticket, sc1 loads) -- which
// wave that is does not enter the sum, so the row is bit-stable, and no float atomic touches global memory.
//
// one segment: the entries [0, cnt) held one per lane (ev = edge id): sq = sum of the sources' TQ rows, sd = sum of
// their dv rows (DENSE, by edge id) or of the hit gradients (sparse: lane t owns in-edge t -- its channel mask and hit
// record; the few hits of an edge go to the channel sums in LDS (sacc), so the cost is per hit and not per (edge,
// channel); LDS float atomics of ONE wave-instruction on one address are served in a fixed lane order, and the
// instructions run in program order, so sacc is a function of the (sorted) list alone).
template <bool DENSE>
__device__ __forceinline__ void rgather_segment(const __amdgpu_buffer_rsrc_t TQb, int k, const float *__restrict__ dv,
                                                const HitRec *__restrict__ hits, const float *__restrict__ G, int cnt,
                                                int ev, u64 hm, float4 rec, float *sacc, int lane, float &sq, float &sd,
                                                const float *__restrict__ sc_tab)
{
    float s0 = 0.0f, s1 = 0.0f, d0 = 0.0f, d1 = 0.0f;
    const int src = (int)((unsigned)ev / (unsigned)k);
    // 8 rows in flight per wave (16, as two groups, measured slower: most lists are short).  Whole groups of 8 run
    // without masks or clamps (a row costs v_readlane, one address instruction and the add; the masked form cost 8.7
    // vector instructions per row and the pass sat at 53 % VALU issue); only the last, partial group selects.
    const int full = cnt & ~7;
    for (int t = 0; t < full; t += 8) {
        float q[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = __builtin_amdgcn_readlane(src, t + u);
#ifdef DGCNN_ABL_RGATHER_NOTQ   // ablation (times only, results invalid): the pass without its TQ row gathers
            q[u] = (float)i;
#else
            q[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(TQb, lane * 4, (int)((unsigned)i * (unsigned)(EC * 4)), 0));  // scalar row offset
#endif
            if (DENSE) d[u] = __builtin_nontemporal_load(&(dv + (size_t)(unsigned)__builtin_amdgcn_readlane(ev, t + u) * EC)[(unsigned)lane]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) {
                s1 += q[u];
                if (DENSE) d1 += d[u];
            } else {
                s0 += q[u];
                if (DENSE) d0 += d[u];
            }
        }
    }
    if (full < cnt) {
        const int t = full;
        float q[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + u < cnt ? t + u : t;  // clamped: no branch around the loads
            const int i = __builtin_amdgcn_readlane(src, tt);
#ifdef DGCNN_ABL_RGATHER_NOTQ
            q[u] = (float)i;
#else
            q[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(TQb, lane * 4, (int)((unsigned)i * (unsigned)(EC * 4)), 0));
#endif
            // (dv1 rows are read once: streamed past the L2 that holds the TQ table)
            if (DENSE) d[u] = __builtin_nontemporal_load(&(dv + (size_t)(unsigned)__builtin_amdgcn_readlane(ev, tt) * EC)[(unsigned)lane]);
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
    sq = s0 + s1;
    sd = d0 + d1;
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
                atomicAdd(&sacc[c], sc_tab[c] * G[(size_t)src * EC + c]);   // (= the value a record would hold: sc . gdv, the same product)
                hm &= hm - 1ull;
            }
        }
        __builtin_amdgcn_wave_barrier();
        sd = sacc[lane];
        sacc[lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
    }
    (void)hits;
}

__device__ __forceinline__ void st_sc1_f32(float *p, float v)
{
    __hip_atomic_store((gu32 *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1_f32(const float *p)
{
    return __uint_as_float(__hip_atomic_load((gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct RgatherCtx {
    const float *TP;
    float *dR;
    const int *indeg, *segbase;
    int *pcnt;
    float *part0;  // [P][2][64]: partial of segment 0 of a multi-segment point
    float *partx;  // [pairs][2][64]: partials of the extra (point, segment) pairs, by pair index
    float alpha, a0, dscale;
};
// a wave has finished segment `seg` of the multi-segment point pt (deg in-edges): publish its partial, draw a ticket;
// the last arriver adds the partials in segment order and writes dR[pt]
__device__ __forceinline__ void rgather_publish(const RgatherCtx &cx, int pt, int seg, int deg, float sq, float sd, int lane)
{
    const int nseg = (deg + REV_SEG - 1) / REV_SEG;
    const int sb = cx.segbase[pt];
    float *dst = seg == 0 ? cx.part0 + (size_t)pt * 128 : cx.partx + (size_t)(sb + seg - 1) * 128;
    st_sc1_f32(dst + lane, sq);
    st_sc1_f32(dst + 64 + lane, sd);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has left this wave before the ticket is drawn
    int last = 0;
    if (lane == 0) last = ticket(cx.pcnt + pt) == nseg - 1 ? 1 : 0;
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    // (the counter is zero again for the next backward that walks these lists: they may come from the forward, built once)
    if (lane == 0) __hip_atomic_store((gu32 *)(cx.pcnt + pt), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float tq = ld_sc1_f32(cx.part0 + (size_t)pt * 128 + lane), td = ld_sc1_f32(cx.part0 + (size_t)pt * 128 + 64 + lane);
    const float *px = cx.partx + (size_t)sb * 128;
    for (int s0 = 0; s0 < nseg - 1; s0 += 8) {  // 8 partials in flight, added in segment order
        float vq[8], vd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sx = s0 + u < nseg - 1 ? s0 + u : s0;
            vq[u] = ld_sc1_f32(px + (size_t)sx * 128 + lane);
            vd[u] = ld_sc1_f32(px + (size_t)sx * 128 + 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = s0 + u < nseg - 1;
            tq += on ? vq[u] : 0.0f;
            td += on ? vd[u] : 0.0f;
        }
    }
    const float fdeg = (float)deg;
    cx.dR[(size_t)pt * EC + lane] = fdeg * cx.a0 - cx.alpha * (tq + fdeg * cx.TP[(size_t)pt * EC + lane]) + cx.dscale * td;
}

template <bool DENSE>
__global__ __launch_bounds__(256) void edge_rgather_kernel(const float *__restrict__ TP, const float *__restrict__ TQ,
                                                           const int *__restrict__ rowptr, const int *__restrict__ indeg,
                                                           const int *__restrict__ segbase, int *__restrict__ pcnt,
                                                           const int32_t *__restrict__ rev, int k, int P,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ cc, const float *__restrict__ dv,
                                                           const HitRec *__restrict__ hits, const float *__restrict__ G,
                                                           const int *__restrict__ nextra, const int2 *__restrict__ extra,
                                                           float *__restrict__ part0, float *__restrict__ partx,
                                                           int main_blocks, float *__restrict__ dR)
{
    // Workgroups 0 .. main_blocks - 1 walk the points (segment 0 of every list), the others the extra segments of the
    // long lists, concurrently (one launch; dispatching the extra workgroups first was measured slower).
    // DENSE: dv [E][64] rows by edge id (sc applied here).  Sparse: hits[e] by edge id (edge_point_bwd_kernel) and G [P][64] =
    // gdv, dL/dv of the centre's surviving edge per channel as bwd_point left it (times sc where a record overflows).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mean = stats[ST_MEAN + lane], rstd = stats[ST_RSTD + lane], sc = stats[ST_SC + lane];
    const float c1 = cc[lane], c2 = cc[64 + lane];
    // the TQ table behind a buffer descriptor: a row gather is one instruction with a scalar row offset (P EC 4 < 2^32
    // is checked by the launcher)
    const __amdgpu_buffer_rsrc_t TQb = __builtin_amdgcn_make_buffer_rsrc((void *)TQ, 0, (int)((unsigned)P * (unsigned)(EC * 4)), 0x00020000);
    RgatherCtx cx;
    cx.TP = TP;
    cx.dR = dR;
    cx.indeg = indeg;
    cx.segbase = segbase;
    cx.pcnt = pcnt;
    cx.part0 = part0;
    cx.partx = partx;
    cx.alpha = sc * c2 * rstd;
    cx.a0 = -sc * (c1 - c2 * rstd * mean);
    cx.dscale = DENSE ? sc : 1.0f;
    __shared__ float sacc_all[EDGE_WAVES][64];  // sparse dv: hit gradients of the current segment, by channel
    float *sacc = sacc_all[wave];
    if (!DENSE) sacc[lane] = 0.0f;
    if ((int)blockIdx.x >= main_blocks) {
        const int n = *nextra;
        for (int w = ((int)blockIdx.x - main_blocks) * EDGE_WAVES + wave; w < n;
             w += ((int)gridDim.x - main_blocks) * EDGE_WAVES) {
            const int2 ps = extra[w];
            const int pt = __builtin_amdgcn_readfirstlane(ps.x), seg = __builtin_amdgcn_readfirstlane(ps.y);
            const int pbeg = __builtin_amdgcn_readfirstlane(rowptr[pt]), deg = __builtin_amdgcn_readfirstlane(indeg[pt]);
            const int e0 = seg * REV_SEG;
            const int cnt = deg - e0 < REV_SEG ? deg - e0 : REV_SEG;
            const int ev = lane < cnt ? rev[pbeg + e0 + lane] : 0;
            u64 hm = 0;
            float4 rec = float4{0.0f, 0.0f, 0.0f, 0.0f};
#ifndef DGCNN_ABL_RGATHER_NOREC   // ablation (times only, results invalid): the pass without its hit-record reads
            if (!DENSE && lane < cnt) {
                const HitRec *h = hits + ev;
                hm = h->mask;
                rec = *reinterpret_cast<const float4 *>(h->v);
            }
#endif
            float sq, sd;
            rgather_segment<DENSE>(TQb, k, dv, hits, G, cnt, ev, hm, rec, sacc, lane, sq, sd, stats + ST_SC);
            rgather_publish(cx, pt, seg, deg, sq, sd, lane);
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
    int deg_n = 0, ev_n = 0;    // degree (uniform) and first entries of the point about to be processed
    int b_nn = 0, d_nn = 0;     // bounds of the point after it, in flight
    if (pt < it.end) {
        const int beg_n = __builtin_amdgcn_readfirstlane(rowptr[pt]);
        deg_n = __builtin_amdgcn_readfirstlane(indeg[pt]);
        ev_n = lane < deg_n ? rev[beg_n + lane] : 0;
        if (pt + it.step < it.end) {
            b_nn = rowptr[pt + it.step];
            d_nn = indeg[pt + it.step];
        }
    }
    for (; pt < it.end; pt += it.step) {
        // (wave-uniform values are made visibly so: lane selects of v_readlane must be scalars)
        const int deg = deg_n;
        const int ev = ev_n;
        if (pt + it.step < it.end) {
            const int beg_n = __builtin_amdgcn_readfirstlane(b_nn);
            deg_n = __builtin_amdgcn_readfirstlane(d_nn);
            ev_n = lane < deg_n ? rev[beg_n + lane] : 0;
        }
        if (pt + 2 * it.step < it.end) {
            b_nn = rowptr[pt + 2 * it.step];
            d_nn = indeg[pt + 2 * it.step];
        }
        const int cnt = deg < REV_SEG ? deg : REV_SEG;
        u64 hm = 0;
        float4 rec = float4{0.0f, 0.0f, 0.0f, 0.0f};
#ifndef DGCNN_ABL_RGATHER_NOREC
        if (!DENSE && lane < cnt) {
            const HitRec *h = hits + ev;
            hm = h->mask;
            rec = *reinterpret_cast<const float4 *>(h->v);
        }
#endif
        float sq, sd;
        rgather_segment<DENSE>(TQb, k, dv, hits, G, cnt, ev, hm, rec, sacc, lane, sq, sd, stats + ST_SC);
        if (deg <= REV_SEG) {
            const float fdeg = (float)deg;
            dR[(size_t)pt * EC + lane] = fdeg * cx.a0 - cx.alpha * (sq + fdeg * TP[(size_t)pt * EC + lane]) + cx.dscale * sd;
        } else {
            rgather_publish(cx, pt, 0, deg, sq, sd, lane);
        }
    }
}

int launch_edge_bwd(const float *TP, const float *TQ, int B, int N, int k, const float *stats, const float *cc, float slope,
                    const float *gdv, const uint8_t *hitj, const float *usum, const float *dv,
                    const float *dvsum, const int *rowptr, const int *indeg, const int *segbase, int *pcnt,
                    const int32_t *rev, const int *alloc, const int *extra, void *hits, float *G, float *part0,
                    float *partx, float *dR, float *dD, hipStream_t st, const SumRider *rider)
{
    const int P = B * N;
    DG_REQUIRE((size_t)P * EC * 4 < ((size_t)1 << 32), "edge backward: B * N = %d points exceed the 4 GB row-offset range of the TQ gather", P);
    const dim3 grid(stream_grid(P)), block(256);
    // (measured: the placement pass of the reverse adjacency as extra workgroups of this launch -- as the list-space
    // pass rides with bwd_point -- gains nothing: 47.9 us against 28.0 + 17.3; its N-int LDS table is then reserved by
    // every workgroup of the launch and thins out the point pass)
    if (dv)
    {
        SumRider sr = rider ? *rider : SumRider{nullptr, 0, 0, nullptr, 0};
        sr.main_blocks = (int)grid.x;
        hipLaunchKernelGGL((edge_point_bwd_kernel<true>), dim3(grid.x + (sr.out ? ceil_div(sr.n, 64) : 0)), block, 0, st, TQ, k, P, stats,
                           cc, slope, gdv, hitj, usum, dvsum, (HitRec *)hits, G, dD, sr);
    }
    else
        hipLaunchKernelGGL((edge_point_bwd_kernel<false>), grid, block, 0, st, TQ, k, P, stats, cc, slope, gdv, hitj, usum,
                           nullptr, (HitRec *)hits, G, dD, SumRider{nullptr, 0, 0, nullptr, 0});
    prof_begin(PROF_EDGE_RGATHER, st);
    // one launch: the point workgroups and, behind them, workgroups for the extra segments of the long lists
    const int main_blocks = (int)grid.x, extra_blocks = main_blocks / 2;
    const dim3 ggrid(main_blocks + extra_blocks);
    const int2 *elist = reinterpret_cast<const int2 *>(extra);
    if (dv)
        hipLaunchKernelGGL((edge_rgather_kernel<true>), ggrid, block, 0, st, TP, TQ, rowptr, indeg, segbase, pcnt, rev, k, P,
                           stats, cc, dv, (const HitRec *)hits, G, alloc + 1, elist, part0, partx, main_blocks, dR);
    else
        hipLaunchKernelGGL((edge_rgather_kernel<false>), ggrid, block, 0, st, TP, TQ, rowptr, indeg, segbase, pcnt, rev, k, P,
                           stats, cc, dv, (const HitRec *)hits, G, alloc + 1, elist, part0, partx, main_blocks, dR);
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

// gx[b][m] = sum_j (g_ctr[m][j] - g_diff[m][j]) + sum_{e in in(m)} g_diff[e]: the neighbour term as a gather over
// the reverse lists (sorted by edge id: a fixed order of additions; autograd's IndexBackward0 of model.py:74 scatters
// with atomics).  thread = (point, channel); gx is overwritten.
__global__ void graph_feature_bwd_kernel(const float *__restrict__ gout, const int *__restrict__ rowptr,
                                         const int *__restrict__ indeg, const int32_t *__restrict__ rev, int C, int k,
                                         size_t npc, float *__restrict__ gx)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npc) return;
    const int c = (int)(t % C);
    const size_t pt = t / C;
    float s = 0.0f;
    for (int j = 0; j < k; ++j) {
        const size_t e = pt * k + j;
        s += gout[e * 2 * C + C + c] - gout[e * 2 * C + c];
    }
    const int beg = rowptr[pt], deg = indeg[pt];
    for (int i = 0; i < deg; ++i) s += gout[(size_t)rev[beg + i] * 2 * C + c];
    gx[t] = s;
}

// ---------------------------------------------------------------------------------------------
void launch_edge_reduce(const float *TP, const float *TQ, const int32_t *idx, int N, int k, int P, const float *gamma,
                        float *usel, float *usum, uint8_t *hitj, const FoldSet &fs, const BnBatchStats *fin, hipStream_t st,
                        const RevSlotsArgs *slots)
{
    const int nmain = edge_grid(P);
    const dim3 block(256);
    const BnBatchStats f = fin ? *fin : BnBatchStats{};
    const RevSlotsArgs none{0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0, 0};
    // (slots: only with fin -- the training forward -- see edge_reduce_kernel)
    const dim3 grid(nmain + (slots && fin ? slots->nblk * ceil_div(slots->Pblk, 256) : 0));
    const RevSlotsArgs sa = slots && fin ? *slots : none;
    prof_begin(PROF_EDGE_REDUCE, st);
    if (usel) {
        if (fin)
            hipLaunchKernelGGL((edge_reduce_kernel<true, true>), grid, block, 0, st, TP, TQ, idx, N, k, P, gamma, usel, usum,
                               hitj, fs, f, nmain, sa);
        else
            hipLaunchKernelGGL((edge_reduce_kernel<true, false>), grid, block, 0, st, TP, TQ, idx, N, k, P, gamma, usel, usum,
                               hitj, fs, f, nmain, sa);
    } else {
        hipLaunchKernelGGL((edge_reduce_kernel<false, true>), grid, block, 0, st, TP, TQ, idx, N, k, P, gamma, usel, usum,
                           hitj, fs, f, nmain, sa);
    }
    prof_end(PROF_EDGE_REDUCE, st);
}

void launch_edge_out(const float *usel, const float *stats, float slope, int P, float *y, hipStream_t st, int ldy)
{
    const size_t n4 = (size_t)P * EC / 4;
    hipLaunchKernelGGL(edge_out_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, usel, stats, slope, n4, y,
                       ldy > 0 ? ldy : EC);
}

void launch_bwd_point(const float *gy, const float *usel, const float *stats, float slope, int P, double count,
                      int training, const FoldSet &fs, float *gbn_w, float *gbn_b, float *cc, unsigned *gmax_bits,
                      const RevSlotsArgs *slots, hipStream_t st, int ldgy, const float *gy2, float *gdv, const RevPlaceArgs *place)
{
    int grid = ceil_div(P, EDGE_WAVES * 8);
    if (grid > 512) grid = 512;
    const BnBackwardSums fin{count, training, gbn_w, gbn_b, cc};
    // (slots: the list-space pass of the reverse adjacency rides behind this pass's workgroups, see rev_slots_pass)
    const RevSlotsArgs none{0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0, 0};
    const int extra_wgs = slots ? slots->nblk * ceil_div(slots->Pblk, 256) : 0;
    hipLaunchKernelGGL(bwd_point_kernel, dim3(grid + extra_wgs + (place ? place->nblocks : 0)), dim3(256), 0, st, gy, ldgy > 0 ? ldgy : EC, gy2, gdv, usel, stats,
                       slope, P, fs, fin, gmax_bits, grid, slots ? *slots : none, place ? *place : RevPlaceArgs{}, extra_wgs);
}

void launch_graph_feature_fwd(const float *x, const int32_t *idx, int B, int N, int C, int k, float *out,
                              hipStream_t st)
{
    const size_t total = (size_t)B * N * k * 2 * C;
    hipLaunchKernelGGL(graph_feature_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, idx, N, C,
                       k, total, out);
}

void launch_graph_feature_bwd(const float *gout, const int *rowptr, const int *indeg, const int32_t *rev, int B, int N,
                              int C, int k, float *gx, hipStream_t st)
{
    const size_t npc = (size_t)B * N * C;
    hipLaunchKernelGGL(graph_feature_bwd_kernel, dim3((unsigned)((npc + 255) / 256)), dim3(256), 0, st, gout, rowptr, indeg,
                       rev, C, k, npc, gx);
}

}  // namespace dgcnn
