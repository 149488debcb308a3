// dev microbenchmark: issue cost of the fp16 MFMA shapes on gfx950 -- is the legacy K = 8 form (v_mfma_f32_32x32x8_f16)
// half the cost of v_mfma_f32_32x32x16_f16?  (the kNN streams carry a 7-slot "augmented" K slab that would fit K = 8)
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_shapes mfma16_shapes.hip && ./mfma16_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a0)
{
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    }
    half8 a8, b8;
    half4 a4, b4;
    for (int e = 0; e < 8; ++e) a8[e] = (_Float16)(a0 + threadIdx.x + e), b8[e] = (_Float16)(a0 * 0.5f + e);
    for (int e = 0; e < 4; ++e) a4[e] = a8[e], b4[e] = b8[e];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
                if (SHAPE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
                if (SHAPE == 2) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc4[i], 0, 0, 0);
                if (SHAPE == 3) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int SHAPE>
void run(const char *name, double flops_per)
{
    const int nblk = 256 * 4 * 2;
    float *out;
    hipMalloc(&out, nblk * 64 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(nblk), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double n = (double)nblk * iters * 64;  // MFMAs issued
    printf("%-28s %.3f ms  %7.1f TFLOP/s  %.2f ns per MFMA per SIMD\n", name, ms, n * flops_per / ms / 1e9, ms * 1e6 / (n / 1024.0));
    hipFree(out);
}

int main()
{
    run<0>("v_mfma_f32_32x32x16_f16", 32768.0);
    run<1>("v_mfma_f32_32x32x8_f16", 16384.0);
    run<2>("v_mfma_f32_16x16x32_f16", 16384.0);
    run<3>("v_mfma_f32_16x16x16_f16", 8192.0);
    return 0;
}
