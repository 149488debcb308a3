/*
 * oracle/knn_exact.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's kNN graph build with a *specified* fp32
 * evaluation order, so that the HIP kernel can be gated bit-exactly.
 *
 * Follows /root/reference/model.py:25-40 (`knn`):
 *     inner = -2 * x^T x                       (model.py:35)
 *     xx    = sum(x**2, dim=1)                 (model.py:36)
 *     pd    = -xx - inner - xx^T               (model.py:37)
 *     idx   = pd.topk(k, dim=-1)[1]            (model.py:39)
 *
 * The reference leaves two things implementation-defined: the summation order
 * of `xx` / the dot product (MKL sgemm) and the order of equal values in
 * `topk`.  This oracle pins both:
 *
 *   xx_l  = fma chain over c = 0..C-1 starting from +0:  acc = fmaf(x[c,l], x[c,l], acc)
 *   dot   = fma chain over c = 0..C-1 starting from +0:  acc = fmaf(x[c,i], x[c,l], acc)
 *   t     = fl(2*dot - xx_l)        (2*dot is exact, one rounding)
 *   pd    = fl(t - xx_i) + 0.0f     (the +0 only canonicalises -0 to +0)
 *   order = pd descending, candidate index ascending among equal pd
 *
 * which is algebraically the reference's  (-xx_l - (-2 dot)) - xx_i.
 * Parity status: pinned against the imported reference by tests/golden/knn_*.npz
 * (tie-free inputs: identical index lists) -- see oracle/gen_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* x: [B][C][N] contiguous fp32 (the reference's layout, model.py:35).
 * idx_out: [B][N][k] int32, neighbour indices local to the batch element.
 * pd_out : optional [B][N][k] fp32 with the pd value of each selected neighbour
 *          (may be NULL).  Returns 0 on success, 1 on bad arguments. */
int knn_exact_f32(const float *x, int B, int C, int N, int k,
                  int32_t *idx_out, float *pd_out)
{
    if (!x || !idx_out || B <= 0 || C <= 0 || N <= 0 || k <= 0 || k > N)
        return 1;

    float *xx = (float *)malloc((size_t)B * N * sizeof(float));
    if (!xx) return 1;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * C * N;
        float *xxb = xx + (size_t)b * N;
        for (int l = 0; l < N; ++l) xxb[l] = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float *xc = xb + (size_t)c * N;
            for (int l = 0; l < N; ++l)
                xxb[l] = __builtin_fmaf(xc[l], xc[l], xxb[l]);
        }
    }

    int err = 0;
#pragma omp parallel
    {
        float *acc = (float *)malloc((size_t)N * sizeof(float));
        float *best_pd = (float *)malloc((size_t)k * sizeof(float));
        int32_t *best_ix = (int32_t *)malloc((size_t)k * sizeof(int32_t));
        if (!acc || !best_pd || !best_ix) {
#pragma omp atomic write
            err = 1;
        } else {
#pragma omp for schedule(dynamic, 16) collapse(2)
            for (int b = 0; b < B; ++b) {
                for (int i = 0; i < N; ++i) {
                    const float *xb = x + (size_t)b * C * N;
                    const float *xxb = xx + (size_t)b * N;
                    for (int l = 0; l < N; ++l) acc[l] = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        const float *xc = xb + (size_t)c * N;
                        const float q = xc[i];
                        /* each acc[l] is its own sequential chain over c:
                         * vectorising over l does not change any rounding */
#pragma omp simd
                        for (int l = 0; l < N; ++l)
                            acc[l] = __builtin_fmaf(q, xc[l], acc[l]);
                    }
                    const float xxi = xxb[i];
                    int cnt = 0;
                    for (int l = 0; l < N; ++l) {
                        float t = __builtin_fmaf(2.0f, acc[l], -xxb[l]);
                        float pd = (t - xxi) + 0.0f;
                        /* candidates arrive in ascending l, so a strict '>'
                         * keeps the lower index among equal pd */
                        if (cnt < k) {
                            int p = cnt++;
                            while (p > 0 && pd > best_pd[p - 1]) {
                                best_pd[p] = best_pd[p - 1];
                                best_ix[p] = best_ix[p - 1];
                                --p;
                            }
                            best_pd[p] = pd;
                            best_ix[p] = l;
                        } else if (pd > best_pd[k - 1]) {
                            int p = k - 1;
                            while (p > 0 && pd > best_pd[p - 1]) {
                                best_pd[p] = best_pd[p - 1];
                                best_ix[p] = best_ix[p - 1];
                                --p;
                            }
                            best_pd[p] = pd;
                            best_ix[p] = l;
                        }
                    }
                    int32_t *o = idx_out + ((size_t)b * N + i) * k;
                    memcpy(o, best_ix, (size_t)k * sizeof(int32_t));
                    if (pd_out)
                        memcpy(pd_out + ((size_t)b * N + i) * k, best_pd,
                               (size_t)k * sizeof(float));
                }
            }
        }
        free(acc);
        free(best_pd);
        free(best_ix);
    }
    free(xx);
    return err;
}

/* pd value of one (query, candidate) pair under the same arithmetic; used by the
 * tests to explain index disagreements with PyTorch-CPU by distance gaps. */
float knn_exact_pd(const float *x, int C, int N, int i, int l)
{
    float dot = 0.0f, xxl = 0.0f, xxi = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float *xc = x + (size_t)c * N;
        dot = __builtin_fmaf(xc[i], xc[l], dot);
        xxl = __builtin_fmaf(xc[l], xc[l], xxl);
        xxi = __builtin_fmaf(xc[i], xc[i], xxi);
    }
    float t = __builtin_fmaf(2.0f, dot, -xxl);
    return (t - xxi) + 0.0f;
}

int knn_exact_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
