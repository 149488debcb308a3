// dev microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 for dependent chains,
// by waves per SIMD and independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int lds_bytes)
{
    const int nblk = 256 * 4 * waves_per_simd;
    float *out; hipMalloc(&out, nblk * 64 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(nblk), dim3(64), lds_bytes, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)nblk * iters * 32 * NACC / 1024.0;
    const double flops = (double)nblk * iters * 32 * NACC * 4096.0;
    printf("waves/SIMD %d nacc %d lds %6d: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n",
           waves_per_simd, NACC, lds_bytes, ms, flops / ms / 1e9, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    run<1>(1, 0); run<2>(1, 0); run<4>(1, 0);
    run<1>(2, 0); run<2>(2, 0);
    run<1>(4, 0);
    run<1>(2, 17000);  // LDS footprint of knn_kernel: caps residency at 9 workgroups per CU
    run<1>(3, 0);
    return 0;
}
