// dev microbenchmark: the scatter-add of per-edge [64] rows into a per-batch-element table [N][64] -- the shape of
// dR[n(i, j)] += dv1[i][j] in the EdgeConv backward -- with 64-bit INTEGER atomics (fixed point: integer addition is
// associative, so the result does not depend on the order the adds arrive in: a deterministic scatter), with 32-bit integer
// atomics and with float atomics, against the time the streaming write of the same rows takes.  B = 8, N = 7000, k = 30
// (1.68 M rows of 64 values).  One wave = one point at a time, lane = channel; workgroups of a batch element run on one XCD
// (blockIdx % 8), as the EdgeConv passes place them.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/int_atomic_scatter scripts/microbench/int_atomic_scatter.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int MODE>  // 0: u64 atomic add, 1: u32 atomic add, 2: float atomic add, 3: streaming store of the row (no scatter)
__global__ __launch_bounds__(256) void scatter(const int *__restrict__ idx, int N, int k, int P, void *table, float *rows)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int per = (P + 7) >> 3, beg = xcd * per, end = beg + per < P ? beg + per : P;
    for (int pt = beg + slot * 4 + wave; pt < end; pt += nslots * 4) {
        const int b = pt / N;
        const int nv = lane < k ? idx[(size_t)pt * k + lane] : 0;
        for (int j = 0; j < k; ++j) {
            const int n = __builtin_amdgcn_readlane(nv, j);
            const size_t row = (size_t)b * N + n;
            const float v = (float)(pt + j + lane) * 1e-3f;
            if (MODE == 0) {
                atomicAdd((unsigned long long *)table + row * 64 + lane, (unsigned long long)(long long)(v * 1048576.0f));
            } else if (MODE == 1) {
                atomicAdd((unsigned *)table + row * 64 + lane, (unsigned)(int)(v * 1024.0f));
            } else if (MODE == 2) {
                atomicAdd((float *)table + row * 64 + lane, v);
            } else {
                __builtin_nontemporal_store(v, &rows[((size_t)pt * k + j) * 64 + lane]);
            }
        }
    }
}

template <int MODE>
void run(const char *name, int slots, const int *idx, int N, int k, int P, void *table, size_t table_bytes, float *rows)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(table, 0, table_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((scatter<MODE>), dim3(8 * slots), dim3(256), 0, 0, idx, N, k, P, table, rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double rowsn = (double)P * k;
    printf("%-28s slots/XCD %4d: %8.1f us   %.2f G rows/s   %.2f TB/s of 256-byte rows\n", name, slots, best * 1e3, rowsn / best / 1e6,
           rowsn * 256.0 / best / 1e9);
}

int main()
{
    const int B = 8, N = 7000, k = 30, P = B * N;
    std::vector<int> h((size_t)P * k);
    srand(1);
    for (int p = 0; p < P; ++p)
        for (int j = 0; j < k; ++j) h[(size_t)p * k + j] = rand() % N;
    int *idx;
    void *table;
    float *rows;
    const size_t tb = (size_t)P * 64 * 8;
    hipMalloc(&idx, h.size() * 4);
    hipMalloc(&table, tb);
    hipMalloc(&rows, (size_t)P * k * 64 * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int slots : {64, 128, 256}) {
        run<0>("u64 atomic add (L2 table 3.6MB)", slots, idx, N, k, P, table, tb, rows);
        run<1>("u32 atomic add (L2 table 1.8MB)", slots, idx, N, k, P, table, tb, rows);
        run<2>("f32 atomic add", slots, idx, N, k, P, table, tb, rows);
        run<3>("streaming store of the rows", slots, idx, N, k, P, table, tb, rows);
    }
    return 0;
}
