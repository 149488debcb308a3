// seg.hip -- the steps either side of DGCNN.forward in the reference's loops, on the device (SURVEY.md 8f rows N2-N4;
// "next" rows after the EdgeConv path, built to the same parity bar):
//   N2  loss + metric of a training / validation step: softmax over the classes (train.py:180, :267), the optional
//       training mask with its extra "ignored" class (train.py:182-196), label-smoothed cross entropy OF THE
//       PROBABILITIES (cal_loss, util.py:17-34, applies log_softmax to the softmax output) and the arg-max
//       prediction (train.py:201, :271) -- one pass over the logits, which also leaves dLoss/dlogits behind;
//       Adam over one flat parameter buffer (torch.optim.Adam as constructed at train.py:118).
//   N3  float64 [B][N][F] blocks -> fp32 (the `.float()` of train.py:176 / test.py:119).
//   N4  inference post-step: softmax, max, confidence threshold -> UNCLASSIFIED (test.py:123-127, :308-311).
// All are HBM-bound streaming kernels (a few bytes per point and class); the loss is reduced through fp64
// per-workgroup partials in a fixed order.
#include "common.h"

namespace dgcnn {

constexpr int SEG_THREADS = 256;

// logits element (b, c, n) at b * sb + c * sc + n * sn (elements): [B, cls, N] as DGCNN.forward returns it, or a
// point-major [B, N, cls] buffer, without a copy
struct SegView {
    long long sb, sc, sn;
    int N, P, cls;
};

template <int MAXC>
__device__ __forceinline__ void seg_softmax(const float *__restrict__ z, const SegView &vw, int t, float (&p)[MAXC])
{
    const int b = t / vw.N, n = t - b * vw.N;
    const float *zp = z + (long long)b * vw.sb + (long long)n * vw.sn;
    float m = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < vw.cls) {
            p[c] = zp[(long long)c * vw.sc];
            m = fmaxf(m, p[c]);
        }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < vw.cls) {
            p[c] = expf(p[c] - m);
            s += p[c];
        }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < vw.cls) p[c] = p[c] / s;
}

// one thread per point.  gz (same element strides as z) = d(mean loss)/dz; pred [P]; partial[blockIdx] = fp64 sum of
// the workgroup's per-point losses.  With a mask the rows have cls + 1 columns: (p, 0) with the point's label where
// mask != 0, (0, ..., 0, 1) with label `cls` elsewhere (constant rows: they count in the mean, carry no gradient).
template <int MAXC>
__global__ __launch_bounds__(SEG_THREADS) void seg_loss_kernel(const float *__restrict__ z, SegView vw,
                                                               const void *__restrict__ labels, int labels_f64,
                                                               const uint8_t *__restrict__ mask, float smoothing,
                                                               float *__restrict__ gz, int64_t *__restrict__ pred,
                                                               double *__restrict__ partial)
{
    __shared__ double red[SEG_THREADS / 64];
    const int t = blockIdx.x * SEG_THREADS + threadIdx.x;
    double loss = 0.0;
    if (t < vw.P) {
        float p[MAXC], v[MAXC + 1];
        seg_softmax<MAXC>(z, vw, t, p);
        const int nn = vw.cls + (mask ? 1 : 0);
        const bool in = mask ? mask[t] != 0 : true;
        // (labels as the reference's loader hands them over -- float64, data.py:169-182 -- or int64)
        const int lab = in ? (labels_f64 ? (int)((const double *)labels)[t] : (int)((const int64_t *)labels)[t]) : vw.cls;
#pragma unroll
        for (int c = 0; c < MAXC + 1; ++c) {
            float x = 0.0f;
            if (c < vw.cls)
                x = in ? p[c < MAXC ? c : 0] : 0.0f;
            else if (c == vw.cls && mask)
                x = in ? 0.0f : 1.0f;
            v[c] = x;
        }
        // log_softmax of the row v (util.py:30), first arg-max (train.py:201)
        float m = -__builtin_inff();
        int arg = 0;
#pragma unroll
        for (int c = 0; c < MAXC + 1; ++c)
            if (c < nn && v[c] > m) {
                m = v[c];
                arg = c;
            }
        float e[MAXC + 1], s = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC + 1; ++c)
            if (c < nn) {
                e[c] = expf(v[c] - m);
                s += e[c];
            }
        const float ls = logf(s);
        const float off = smoothing / (float)(nn - 1), on = 1.0f - smoothing;
        float acc = 0.0f, dot = 0.0f;
        const float inv = 1.0f / (float)vw.P;
        float gp[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC + 1; ++c)
            if (c < nn) {
                const float oh = c == lab ? on : off;
                acc += oh * ((v[c] - m) - ls);
                if (c < MAXC && c < vw.cls) {
                    gp[c] = in ? (e[c] / s - oh) * inv : 0.0f;  // d/dv of the mean loss (the smoothed targets sum to 1)
                    dot = __builtin_fmaf(p[c], gp[c], dot);
                }
            }
        // a label outside [0, classes) has no target column (the reference's scatter fails on it): poison the loss
        loss = (lab >= 0 && lab < nn) ? -(double)acc : (double)__builtin_nanf("");
        pred[t] = arg;
        const int b = t / vw.N, n = t - b * vw.N;
        float *gp_out = gz + (long long)b * vw.sb + (long long)n * vw.sn;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < vw.cls) gp_out[(long long)c * vw.sc] = p[c] * (gp[c] - dot);  // through the first softmax
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int lo = __double2loint(loss), hi = __double2hiint(loss);
        lo = __shfl_xor(lo, o);
        hi = __shfl_xor(hi, o);
        loss += __hiloint2double(hi, lo);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = (sum of the partials, in index order) / P
__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const double *__restrict__ partial, int n, int P,
                                                                float *__restrict__ loss)
{
    __shared__ double red[256];
    double s = 0.0;
    const int per = (n + 255) / 256;
    for (int i = threadIdx.x * per; i < (threadIdx.x + 1) * per && i < n; ++i) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)P);
}

// pred = arg max of softmax(z), replaced by `unclassified` where the maximum probability is below min_conf
template <int MAXC>
__global__ __launch_bounds__(SEG_THREADS) void seg_predict_kernel(const float *__restrict__ z, SegView vw, float min_conf,
                                                                  int unclassified, int64_t *__restrict__ pred,
                                                                  float *__restrict__ prob)
{
    const int t = blockIdx.x * SEG_THREADS + threadIdx.x;
    if (t >= vw.P) return;
    float p[MAXC];
    seg_softmax<MAXC>(z, vw, t, p);
    float m = -__builtin_inff();
    int arg = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < vw.cls && p[c] > m) {
            m = p[c];
            arg = c;
        }
    pred[t] = m < min_conf ? unclassified : arg;
    if (prob) prob[t] = m;
}

// torch.optim.Adam (no amsgrad; weight decay added to the gradient) on flat buffers; n4 = n / 4 vector part
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, size_t n, float lr_c1, float beta1, float beta2, float eps,
                            float wd, float rsqrt_c2)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float pv = p[t];
    const float gv = __builtin_fmaf(wd, pv, g[t]);
    const float mv = __builtin_fmaf(gv - m[t], 1.0f - beta1, m[t]);  // exp_avg.lerp_(grad, 1 - beta1)
    const float vv = beta2 * v[t] + (1.0f - beta2) * gv * gv;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[t] = mv;
    v[t] = vv;
    p[t] = pv - lr_c1 * (mv / (sqrtf(vv) * rsqrt_c2 + eps));      // denom = sqrt(v) / sqrt(1 - beta2^t) + eps
}

__global__ void cast_f64_f32_kernel(const double *__restrict__ x, size_t n, float *__restrict__ y)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = (float)x[t];
}

static int seg_check(const char *who, const void *z, int B, int N, int cls)
{
    DG_REQUIRE(z, "%s: null pointer", who);
    DG_REQUIRE(B > 0 && N > 0 && cls > 0, "%s: bad shape B=%d N=%d classes=%d", who, B, N, cls);
    DG_REQUIRE(cls <= 32, "%s: at most 32 classes are supported (got %d)", who, cls);
    DG_REQUIRE((size_t)B * N < ((size_t)1 << 31), "%s: more than 2^31 points", who);
    return 0;
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" size_t dgcnn_seg_loss_workspace_bytes(int B, int N)
{
    if (B <= 0 || N <= 0) return 0;
    return align_up((size_t)ceil_div(B * N, SEG_THREADS) * sizeof(double), 256);
}

extern "C" int dgcnn_seg_loss(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                              int64_t stride_n, const int64_t *labels, const uint8_t *mask, float smoothing,
                              float *loss, float *grad_logits, int64_t *pred, void *workspace, size_t workspace_bytes,
                              void *stream)
{
    return dgcnn_seg_loss_labels(logits, B, N, classes, stride_b, stride_c, stride_n, labels, 0, mask, smoothing, loss, grad_logits,
                                 pred, workspace, workspace_bytes, stream);
}

extern "C" int dgcnn_seg_loss_labels(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                                     int64_t stride_n, const void *labels, int labels_f64, const uint8_t *mask, float smoothing,
                                     float *loss, float *grad_logits, int64_t *pred, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    if (seg_check("dgcnn_seg_loss", logits, B, N, classes)) return 1;
    DG_REQUIRE(labels && loss && grad_logits && pred && workspace, "dgcnn_seg_loss: null pointer");
    DG_REQUIRE(classes + (mask ? 1 : 0) >= 2, "dgcnn_seg_loss: label smoothing needs at least two classes");
    DG_REQUIRE(workspace_bytes >= dgcnn_seg_loss_workspace_bytes(B, N), "dgcnn_seg_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    SegView vw = {stride_b, stride_c, stride_n, N, B * N, classes};
    const int grid = ceil_div(vw.P, SEG_THREADS);
    double *partial = (double *)workspace;
    if (classes <= 8)
        hipLaunchKernelGGL(seg_loss_kernel<8>, dim3(grid), dim3(SEG_THREADS), 0, st, logits, vw, labels, labels_f64, mask, smoothing,
                           grad_logits, pred, partial);
    else
        hipLaunchKernelGGL(seg_loss_kernel<32>, dim3(grid), dim3(SEG_THREADS), 0, st, logits, vw, labels, labels_f64, mask,
                           smoothing, grad_logits, pred, partial);
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, grid, vw.P, loss);
    DG_CHECK_LAUNCH("dgcnn_seg_loss");
    return 0;
}

extern "C" int dgcnn_seg_predict(const float *logits, int B, int N, int classes, int64_t stride_b, int64_t stride_c,
                                 int64_t stride_n, float min_confidence, int unclassified, int64_t *pred, float *prob,
                                 void *stream)
{
    if (seg_check("dgcnn_seg_predict", logits, B, N, classes)) return 1;
    DG_REQUIRE(pred, "dgcnn_seg_predict: null pointer");
    SegView vw = {stride_b, stride_c, stride_n, N, B * N, classes};
    const int grid = ceil_div(vw.P, SEG_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (classes <= 8)
        hipLaunchKernelGGL(seg_predict_kernel<8>, dim3(grid), dim3(SEG_THREADS), 0, st, logits, vw, min_confidence,
                           unclassified, pred, prob);
    else
        hipLaunchKernelGGL(seg_predict_kernel<32>, dim3(grid), dim3(SEG_THREADS), 0, st, logits, vw, min_confidence,
                           unclassified, pred, prob);
    DG_CHECK_LAUNCH("dgcnn_seg_predict");
    return 0;
}

extern "C" int dgcnn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, void *stream)
{
    DG_REQUIRE(param && grad && exp_avg && exp_avg_sq, "dgcnn_adam_step: null pointer");
    DG_REQUIRE(step >= 1, "dgcnn_adam_step: step counts from 1 (got %d)", step);
    if (n == 0) return 0;
    const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, (float)((double)lr / c1), beta1, beta2, eps, weight_decay,
                       (float)(1.0 / sqrt(c2)));
    DG_CHECK_LAUNCH("dgcnn_adam_step");
    return 0;
}

extern "C" int dgcnn_cast_f64_f32(const double *x, size_t n, float *y, void *stream)
{
    DG_REQUIRE(x && y, "dgcnn_cast_f64_f32: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, y);
    DG_CHECK_LAUNCH("dgcnn_cast_f64_f32");
    return 0;
}
