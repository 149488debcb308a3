// dev microbenchmark: does VALU work issue in the shadow of v_mfma_f32_32x32x16_f16 on gfx950?
// Each wave runs chains of MFMAs with NV independent VALU instructions of one kind between two MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV, int MF>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a0)
{
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + threadIdx.x); b[i] = (_Float16)(a0 * 0.5f); }
    float v[8]; f32x2 p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = a0 + i + threadIdx.x; p[i] = f32x2{a0 + i, a0 - i}; u[i] = threadIdx.x * 77 + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MF) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
                if (KIND == 1) p[j & 7] = __builtin_elementwise_fma(p[j & 7], f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
                if (KIND == 2) u[j & 7] = __builtin_amdgcn_alignbit(u[j & 7], u[(j + 1) & 7], 31);
                if (KIND == 3) u[j & 7] = u[j & 7] > u[(j + 3) & 7] ? u[j & 7] : u[(j + 3) & 7] + 1;
            }
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND, int NV, int MF>
void run(int wps, const char *name)
{
    const int nblk = 256 * 4 * wps;
    float *out; hipMalloc(&out, nblk * 64 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(nblk), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD and per "slot" (2 MFMA + NV VALU), in cycles at 2.4 GHz
    const double slots = (double)wps * iters * 8;
    printf("%-10s NV %2d MFMA %d waves/SIMD %d: %.3f ms  %.1f cycles per slot per SIMD (2 MFMA = 64, NV VALU = %d)\n", name, NV, MF, wps, ms,
           ms * 1e-3 * 2.4e9 / slots, NV * 4);
    hipFree(out);
}

int main()
{
    run<0, 0, 1>(1, "mfma only"); run<0, 0, 1>(2, "mfma only"); run<0, 0, 1>(4, "mfma only");
    run<0, 16, 0>(1, "fma only"); run<0, 16, 0>(2, "fma only");
    run<1, 16, 0>(2, "pkfma only"); run<2, 16, 0>(2, "alignb only"); run<3, 16, 0>(2, "umax only");
    run<0, 8, 1>(1, "fma"); run<0, 8, 1>(2, "fma"); run<0, 8, 1>(4, "fma");
    run<0, 16, 1>(1, "fma"); run<0, 16, 1>(2, "fma"); run<0, 16, 1>(4, "fma");
    run<1, 16, 1>(2, "pkfma"); run<1, 16, 1>(4, "pkfma");
    run<2, 16, 1>(2, "alignbit"); run<2, 16, 1>(4, "alignbit");
    run<3, 16, 1>(2, "umax"); run<3, 16, 1>(4, "umax");
    return 0;
}
