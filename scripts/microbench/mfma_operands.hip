// dev microbenchmark: dependent v_mfma_f32_32x32x2_f32 chain, 1760 single-wave workgroups (2 waves/SIMD),
// 7040 MFMAs per wave.  Variants: operands in 1 or 32+32 different VGPRs; nothing / s_nop / VALU between MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NREG, int FILL>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters, float a0, float b0)
{
    float a[32], b[32];
    for (int s = 0; s < 32; ++s) { a[s] = a0 + (float)(s % NREG) * threadIdx.x; b[s] = b0 - (float)(s % NREG) * threadIdx.x; }
    for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(a[s]), "+v"(b[s]));  // pin them in (distinct) VGPRs
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float t = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[NREG == 1 ? 0 : s], b[NREG == 1 ? 0 : s], acc, 0, 0, 0);
            if (FILL == 1) asm volatile("s_nop 0");
            if (FILL == 2) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(t));
            if (FILL == 3) asm volatile("s_waitcnt vmcnt(0)");
        }
    }
    float sum = t;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
}

// knn_kernel-like stream: two accumulator sets, every 32-MFMA chain starts from zero and its result is read
// by VALU code one chain later.  ZERO = 0: chains continue (no reset); 1: reset through v_mov; 2: srcC = 0.
template <int ZERO>
__global__ __launch_bounds__(64, 2) void k2(float *out, int iters, float a0, float b0)
{
    float a[32], b[32];
    for (int s = 0; s < 32; ++s) { a[s] = a0 + (float)s * threadIdx.x; b[s] = b0 - (float)s * threadIdx.x; }
    for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(a[s]), "+v"(b[s]));
    f32x16 accA, accB;
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.f;
    float sum = 0.f;
    for (int it = 0; it < iters; it += 2) {
        asm volatile("" : "+v"(b[0]));  // loop-variant operand: nothing can be hoisted
        if (ZERO) for (int r = 0; r < 16; ++r) accB[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], accB, 0, 0, 0);
            if ((s & 1) == 1) sum += accA[s >> 1];
        }
        asm volatile("" : "+v"(b[1]));
        if (ZERO) for (int r = 0; r < 16; ++r) accA[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], accA, 0, 0, 0);
            if ((s & 1) == 1) sum += accB[s >> 1];
        }
    }
    for (int r = 0; r < 16; ++r) sum += accA[r] + accB[r];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
}

template <int ZERO>
void run2(const char *name)
{
    const int nblk = 1760, iters = 220;
    float *out; (void)hipMalloc(&out, nblk * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k2<ZERO>), dim3(nblk), dim3(64), 0, 0, out, iters, 1.0f, 2.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-52s %.3f ms\n", name, ms);
    (void)hipFree(out);
}

template <int NREG, int FILL>
void run(const char *name)
{
    const int nblk = 1760, iters = 220;
    float *out; (void)hipMalloc(&out, nblk * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NREG, FILL>), dim3(nblk), dim3(64), 0, 0, out, iters, 1.0f, 2.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-52s %.3f ms\n", name, ms);
    (void)hipFree(out);
}

int main()
{
    run<1, 0>("same operand registers, nothing between");
    run<32, 0>("32+32 operand registers, nothing between");
    run<1, 1>("same operand registers, s_nop between");
    run<32, 1>("32+32 operand registers, s_nop between");
    run<32, 2>("32+32 operand registers, 3 v_add between");
    run<32, 3>("32+32 operand registers, s_waitcnt between");
    run<2, 0>("2+2 operand registers, nothing between");
    run<4, 0>("4+4 operand registers, nothing between");
    run2<0>("two alternating accumulator sets, no reset");
    run2<1>("two alternating accumulator sets, reset per chain");
    return 0;
}
