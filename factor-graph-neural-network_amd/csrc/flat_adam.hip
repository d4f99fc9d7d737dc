// flat_adam.hip — Adam (torch.optim.Adam's update rule, no amsgrad; the optimizer of every reference training script,
// e.g. /root/reference/train_ldpc.py:160-166) over ONE flat f32 parameter buffer in ONE pass:
//     g' = g * gscale + wd * p ;  m = b1 m + (1-b1) g' ;  v = b2 v + (1-b2) g'^2 ;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// and, when asked, the bf16 mirror of the parameters the node-wise GEMMs read is refreshed in the same pass
// (dp.FlatAdam used eight elementwise torch kernels + one cast kernel per step).  gscale = 1 / world folds the mean of the
// all-reduced gradient into the update.  Memory-bound: 4 f32 streams read, 3 written (+ 2 B/param mirror).
#include "fgnn_common.h"
#include <math.h>

struct FaParams {
    float* p;
    const float* g;
    float* m;
    float* v;
    uint16_t* mirror;     // or NULL
    int64_t n4;           // float4 chunks
    int64_t n;            // elements (tail handled by the last thread)
    float lr_bc1, inv_sqrt_bc2, eps, b1, b2, wd, gscale;
    const float* coef;    // or NULL; {lr / bc1, 1 / sqrt(bc2)} in device memory (the capturable form: flat_adam_tick_kernel writes it)
};

// The capturable step's first kernel: advance the device-resident step count and form this step's two bias-corrected
// coefficients from the device-resident learning rate, in double as the host form does.  One thread.
__global__ void flat_adam_tick_kernel(int64_t* step, const float* lr, float b1, float b2, float* coef) {
    const int64_t t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    coef[0] = (float)((double)*lr / bc1);
    coef[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void flat_adam_kernel(const FaParams a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float lr_bc1 = a.coef ? a.coef[0] : a.lr_bc1, inv_sqrt_bc2 = a.coef ? a.coef[1] : a.inv_sqrt_bc2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += stride) {
        f32x4 p = reinterpret_cast<const f32x4*>(a.p)[i], g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<const f32x4*>(a.m)[i], v = reinterpret_cast<const f32x4*>(a.v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = fmaf(a.wd, p[e], g[e] * a.gscale);
            m[e] = fmaf(1.f - a.b1, ge - m[e], m[e]);                 // lerp, as torch's exp_avg.lerp_(grad, 1 - beta1)
            v[e] = fmaf(a.b2, v[e], (1.f - a.b2) * ge * ge);
            const float denom = sqrtf(v[e]) * inv_sqrt_bc2 + a.eps;
            p[e] = p[e] - lr_bc1 * (m[e] / denom);
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.mirror) {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            const b4 h = {(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3]};
            reinterpret_cast<b4*>(a.mirror)[i] = h;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t j = a.n4 * 4; j < a.n; ++j) {                     // < 4 tail elements
            const float ge = fmaf(a.wd, a.p[j], a.g[j] * a.gscale);
            const float m = fmaf(1.f - a.b1, ge - a.m[j], a.m[j]);
            const float v = fmaf(a.b2, a.v[j], (1.f - a.b2) * ge * ge);
            a.m[j] = m; a.v[j] = v;
            const float pn = a.p[j] - lr_bc1 * (m / (sqrtf(v) * inv_sqrt_bc2 + a.eps));
            a.p[j] = pn;
            if (a.mirror) { const __bf16 h = (__bf16)pn; a.mirror[j] = __builtin_bit_cast(uint16_t, h); }
        }
    }
}

static int flat_adam_check(const float* param, const float* grad, const float* exp_avg, const float* exp_avg_sq,
                           const void* bf16_mirror, int64_t n) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) FGNN_FAIL(FGNN_EINVAL, "flat_adam: null buffer");
    if (n < 0) FGNN_FAIL(FGNN_EINVAL, "flat_adam: n >= 0");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        FGNN_FAIL(FGNN_EINVAL, "flat_adam: buffers must be 16-byte aligned");
    if (bf16_mirror && ((uintptr_t)bf16_mirror & 7)) FGNN_FAIL(FGNN_EINVAL, "flat_adam: mirror must be 8-byte aligned");
    return FGNN_OK;
}

static int flat_adam_launch(FaParams& a, hipStream_t stream) {
    int64_t g = (a.n4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "flat_adam launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// step = 1-based step count (bias corrections 1 - beta^step are formed in double on the host)
extern "C" int fgnn_flat_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_mirror,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float grad_scale, int64_t step, fgnn_stream_t stream) {
    if (int rc = flat_adam_check(param, grad, exp_avg, exp_avg_sq, bf16_mirror, n)) return rc;
    if (step < 1) FGNN_FAIL(FGNN_EINVAL, "flat_adam: step >= 1");
    if (n == 0) return FGNN_OK;
    FaParams a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.mirror = (uint16_t*)bf16_mirror;
    a.n = n; a.n4 = n / 4; a.coef = nullptr;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.lr_bc1 = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.eps = eps; a.b1 = beta1; a.b2 = beta2; a.wd = weight_decay; a.gscale = grad_scale;
    return flat_adam_launch(a, (hipStream_t)stream);
}

// The same update with NOTHING step-dependent in the launch arguments, so that the launch can be recorded into a hipGraph and
// replayed: the step count (*step_dev, int64, starts at 0, advanced by one per call) and the learning rate (*lr_dev, f32; a
// scheduler overwrites it between replays) live in device memory; coef_dev is two floats of scratch.  Two launches.
extern "C" int fgnn_flat_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_mirror,
                                  int64_t n, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                  float grad_scale, int64_t* step_dev, float* coef_dev, fgnn_stream_t stream) {
    if (int rc = flat_adam_check(param, grad, exp_avg, exp_avg_sq, bf16_mirror, n)) return rc;
    if (!lr_dev || !step_dev || !coef_dev) FGNN_FAIL(FGNN_EINVAL, "flat_adam_dev: null lr / step / coefficient buffer");
    hipLaunchKernelGGL(flat_adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, coef_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "flat_adam_tick launch: %s", hipGetErrorString(e));
    if (n == 0) return FGNN_OK;
    FaParams a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.mirror = (uint16_t*)bf16_mirror;
    a.n = n; a.n4 = n / 4; a.coef = coef_dev; a.lr_bc1 = 0.f; a.inv_sqrt_bc2 = 0.f;
    a.eps = eps; a.b1 = beta1; a.b2 = beta2; a.wd = weight_decay; a.gscale = grad_scale;
    return flat_adam_launch(a, (hipStream_t)stream);
}
