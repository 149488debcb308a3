// dev microbenchmark: does a knn_kernel-like resource footprint (256 VGPRs, 16.9 KB LDS, a little
// scratch) keep 2 waves per SIMD resident?  1760 single-wave workgroups x 7040 dependent MFMAs each.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS, bool BIGV, bool SCR>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters, float a0, float b0, int dyn)
{
    __shared__ float pad[LDS ? 4224 : 1];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    if (LDS) pad[threadIdx.x] = a;
    if (BIGV) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    float loc[16];
    if (SCR) { for (int i = 0; i < 16; ++i) loc[i] = a + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (LDS) s += pad[(threadIdx.x + 1) & 63];
    if (SCR) s += loc[dyn & 15];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <bool LDS, bool BIGV, bool SCR>
void run(const char *name)
{
    const int nblk = 1760, iters = 220;
    float *out; (void)hipMalloc(&out, nblk * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<LDS, BIGV, SCR>), dim3(nblk), dim3(64), 0, 0, out, iters, 1.0f, 2.0f, rep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s %.3f ms  (ideal at 2 waves/SIMD, perfectly balanced: %.3f ms)\n", name, ms,
           1760.0 * 220 * 32 * 64 / 1024 / 2.4e6);
    (void)hipFree(out);
}

int main()
{
    run<false, false, false>("plain");
    run<true, false, false>("lds 16.9K");
    run<false, true, false>("vgpr 256");
    run<true, true, false>("lds + vgpr");
    run<true, true, true>("lds + vgpr + scratch");
    return 0;
}
