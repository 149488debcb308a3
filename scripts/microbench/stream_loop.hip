// dev microbenchmark: the inner loop shape of the fp16 kNN streams (groups of 1 ds_read_b128 + 2 MFMAs + NV v_min_f32)
// without global memory or barriers: how much of the LDS reads and of the vector epilogue hides under the MFMAs on
// gfx950, with one and with two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NV, bool LDSREAD, bool MFMA, int PD, bool BARRIER, bool STAGE, bool RESET = true>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float a0, unsigned long long *cyc, const half8 *__restrict__ gsrc)
{
    __shared__ half8 sbuf[3 * 1280];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 1280; i += 256) {
        half8 v;
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)(a0 * 0.001f * ((i * 7 + e) % 13));
        sbuf[i] = v;
    }
    __syncthreads();
    half8 a[2][5];
    for (int rt = 0; rt < 2; ++rt)
        for (int ks = 0; ks < 5; ++ks)
            for (int e = 0; e < 8; ++e) a[rt][ks][e] = (_Float16)(a0 * 0.01f * ((lane + ks + e + rt) % 7));
    float m0[2][16], m1[2][16];
    f32x16 accE[2], accO[2];
    for (int rt = 0; rt < 2; ++rt)
        for (int r = 0; r < 16; ++r) { m0[rt][r] = m1[rt][r] = 1e30f; accE[rt][r] = accO[rt][r] = 0.f; }
    half8 sreg[5];
    for (int g = 0; g < 5; ++g) sreg[g] = gsrc[g * 256 + tid];
    half8 bq[PD + 1];
    for (int g = 0; g < PD; ++g) bq[g] = sbuf[g * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const half8 *cur = sbuf + (it % 3) * 1280;
        half8 *wr = sbuf + ((it + 2) % 3) * 1280;
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            const int t = g / 5, ks = g % 5, gp = g + PD;
            if (LDSREAD) bq[gp % (PD + 1)] = cur[(gp % 20) * 64 + lane];
            if (STAGE) {
                if (g < 5) wr[g * 256 + tid] = sreg[g];
                else if (g < 10) sreg[g - 5] = gsrc[(size_t)((it + blockIdx.x) & 63) * 1280 + (g - 5) * 256 + tid];
            }
            if (MFMA) {
                if ((t & 1) == 0) {
                    for (int rt = 0; rt < 2; ++rt) {
                        if (RESET && ks == 0) for (int r = 0; r < 16; ++r) accE[rt][r] = 0.f;
                        accE[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accE[rt], 0, 0, 0);
                    }
                } else {
                    for (int rt = 0; rt < 2; ++rt) {
                        if (RESET && ks == 0) for (int r = 0; r < 16; ++r) accO[rt][r] = 0.f;
                        accO[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][ks], bq[g % (PD + 1)], accO[rt], 0, 0, 0);
                    }
                }
            }
            if (ks > 0) {
                const int e0 = (ks - 1) * NV / 16, e1 = ks * NV / 16;  // NV folds per iteration (4 tiles)
#pragma unroll
                for (int e = e0; e < e1; ++e) {
                    const int rt = (e / 16) & 1, r = e % 16;
                    if ((t & 1) == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(m1[rt][r]) : "v"(accO[rt][r]));
                    else asm volatile("v_min_f32 %0, %0, %1" : "+v"(m0[rt][r]) : "v"(accE[rt][r]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && tid == 0) cyc[0] = t1 - t0;
    float s = 0.f;
    for (int rt = 0; rt < 2; ++rt)
        for (int r = 0; r < 16; ++r) s += m0[rt][r] + m1[rt][r] + accE[rt][r] + accO[rt][r];
    for (int g = 0; g <= PD; ++g) s += (float)bq[g][0];
    for (int g = 0; g < 5; ++g) s += (float)sreg[g][1];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NV, bool LDSREAD, bool MFMA, int PD, bool BARRIER, bool STAGE, bool RESET = true>
void run(int wgs_per_cu, const char *name)
{
    const int nblk = 256 * wgs_per_cu;
    float *out; hipMalloc(&out, nblk * 256 * 4);
    unsigned long long *cyc; hipMalloc(&cyc, 8); unsigned long long hc = 0;
    half8 *gsrc; hipMalloc(&gsrc, 64 * 1280 * 16); hipMemset(gsrc, 0x11, 64 * 1280 * 16);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, LDSREAD, MFMA, PD, BARRIER, STAGE, RESET>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f, cyc, gsrc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    // cycles per SIMD per iteration-of-one-wave at 2.4 GHz (nominal): an iteration = 40 MFMAs = 1280 MFMA cycles
    printf("%-28s NV %3d lds %d mfma %d pd %d bar %d stage %d wgs/CU %d: %.3f ms  %.0f ns per wave-iteration, %.0f nominal cycles per SIMD-iteration (MFMA floor %d); s_memtime: %.0f cycles per wave-iteration, clock %.2f GHz\n",
           name, NV, (int)LDSREAD, (int)MFMA, PD, (int)BARRIER, (int)STAGE, wgs_per_cu, ms, ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / iters, 1280 * wgs_per_cu, (double)hc / iters, (double)hc / (ms * 1e6));
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<0, false, true, 3, false, false, false>(w, "pure mfma (no reset)");
        run<0, true, true, 3, false, false, false>(w, "mfma + reads (no reset)");
        run<0, true, true, 3, true, true, false>(w, "mfma + reads + bar + stage");
        run<32, true, true, 3, true, true, false>(w, "same + 32 v_min");
        run<64, true, true, 3, true, true, false>(w, "same + 64 v_min");
        run<128, false, true, 3, false, false>(w, "mfma + 128 v_min");
        run<128, true, true, 3, false, false>(w, "+ reads");
        run<128, true, true, 3, true, false>(w, "+ reads + barrier");
        run<128, true, true, 3, false, true>(w, "+ reads + stage");
        run<128, true, true, 3, true, true>(w, "+ reads + barrier + stage");
        run<64, true, true, 3, true, true>(w, "64 v_min, all");
        run<128, true, false, 3, true, true>(w, "no mfma, all");
    }
    return 0;
}
