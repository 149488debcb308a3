// dev microbenchmark: rate of random 256-byte row gathers (lane = channel, row index wave-uniform) from a table
// that fits the XCD's L2, by loads in flight per wave and waves per SIMD.  Variant V: 0 = dword per lane (one
// row per wave-load), 1 = float4 per lane (4 rows per wave-load: 16 lanes per row).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int UNROLL, int V>
__global__ __launch_bounds__(256) void k(const float *__restrict__ table, const int *__restrict__ idx, int nrows_per_xcd,
                                         int per_wave, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const float *tb = table + (size_t)xcd * nrows_per_xcd * 64;
    const int *ib = idx + ((size_t)(blockIdx.x * 4 + wave)) * per_wave;
    float s = 0.f;
    for (int j0 = 0; j0 < per_wave; j0 += 64) {
        const int nv = ib[j0 + lane];
        if (V == 0) {
            for (int j = 0; j < 64; j += UNROLL) {
                float v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = tb[(size_t)__builtin_amdgcn_readlane(nv, j + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) s += v[u];
            }
        } else {
            // lane l: row slot l / 16 of each group of 4 rows, channels 4 (l % 16) .. +3
            for (int j = 0; j < 64; j += 4 * UNROLL) {
                float4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int r = __shfl(nv, j + 4 * u + (lane >> 4));
                    v[u] = *reinterpret_cast<const float4 *>(tb + (size_t)r * 64 + 4 * (lane & 15));
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int UNROLL, int V>
void run(int slots, const float *table, const int *idx, int nrows, float *out, int per_wave = 2048 /* rows per wave */)
{
    const int nblk = 8 * slots;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<UNROLL, V>), dim3(nblk), dim3(256), 0, 0, table, idx, nrows, per_wave, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)nblk * 4 * per_wave * 256.0;
    printf("variant %d unroll %2d slots/XCD %4d (waves/SIMD %.1f) rows/wave %4d: %.3f ms  %.2f TB/s\n", V, UNROLL, slots,
           slots * 4 / 128.0, per_wave, ms, bytes / ms / 1e9);
}

int main()
{
    const int nrows = 7000;  // 1.8 MB per XCD
    const int maxblk = 8 * 512;
    float *table, *out; int *idx;
    hipMalloc(&table, (size_t)8 * nrows * 64 * 4); hipMemset(table, 0, (size_t)8 * nrows * 64 * 4);
    hipMalloc(&out, (size_t)maxblk * 256 * 4);
    std::vector<int> h((size_t)maxblk * 4 * 2048);
    srand(1);
    for (auto &v : h) v = rand() % nrows;
    hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<8, 0>(128, table, idx, nrows, out); run<8, 0>(256, table, idx, nrows, out);
    run<16, 0>(128, table, idx, nrows, out); run<16, 0>(256, table, idx, nrows, out);
    run<32, 0>(128, table, idx, nrows, out); run<32, 0>(256, table, idx, nrows, out); run<32, 0>(512, table, idx, nrows, out);
    run<4, 1>(128, table, idx, nrows, out); run<4, 1>(256, table, idx, nrows, out);
    run<8, 1>(128, table, idx, nrows, out); run<8, 1>(256, table, idx, nrows, out); run<8, 1>(512, table, idx, nrows, out);
    // short waves, as the passes of the step have them (~200 rows per wave, 8 waves per SIMD: 1.68 M rows in all)
    run<32, 0>(256, table, idx, nrows, out, 192); run<32, 0>(256, table, idx, nrows, out, 256);
    run<32, 0>(256, table, idx, nrows, out, 512); run<8, 0>(256, table, idx, nrows, out, 192);
    return 0;
}
