// instnorm.hip — per-sample InstanceNorm (+ optional ReLU) over the node axis of channel-fastest
// activations [B][N][C]: the normalisation of the reference's node-wise maps `iid_mapping_in`
// (/root/reference/lib/model/mpnn/base_model.py:82-90: Conv2d 1x1 + InstanceNorm2d + ReLU; used for the
// v2v / f2f maps and the classifier head, factor_mpnn_sp.py:64-77,104-108) — SURVEY §8f rank 1.
//
//   y[b,n,c] = act( (x[b,n,c] - mean_n x[b,:,c]) * rsqrt(var_n x[b,:,c] + eps) )     biased variance, eps 1e-5
//
// PyTorch needs ~6 elementwise / reduction launches over f32 copies for this (profiles/r01); here one
// workgroup owns (sample, 64-channel block): 64 channels x 4 node groups of threads, two passes over a
// tile that stays in L1/L2, statistics in f32, nothing but x is kept for the backward
//   gx = rstd * ( g - mean_n g - xhat * mean_n (g * xhat) ),   g = gy * [y > 0].
#include "fgnn_common.h"

#define IN_THREADS 256
#define IN_CH 64
#define IN_EPS 1e-5f

struct InParams {
    const void* x;
    const void* y;      // backward: forward output (ReLU mask) or NULL when relu == 0
    const void* gy;
    void* out;          // forward: y; backward: gx
    int B, N, C, relu;
};

__device__ __forceinline__ float in_reduce4(float v, float* red, int cg, int ng) {
    // sum over the 4 node groups of one channel
    red[ng * IN_CH + cg] = v;
    __syncthreads();
    const float s = red[cg] + red[IN_CH + cg] + red[2 * IN_CH + cg] + red[3 * IN_CH + cg];
    __syncthreads();
    return s;
}

template <typename T>
__global__ __launch_bounds__(IN_THREADS) void instnorm_fwd_kernel(const InParams p) {
    __shared__ float red[4 * IN_CH];
    const int cg = threadIdx.x & (IN_CH - 1), ng = threadIdx.x >> 6;
    const int nblk = (p.C + IN_CH - 1) / IN_CH;
    const int b = blockIdx.x / nblk, c = (blockIdx.x - b * nblk) * IN_CH + cg;
    const bool ok = c < p.C;
    const T* xb = static_cast<const T*>(p.x) + (int64_t)b * p.N * p.C + c;
    T* yb = static_cast<T*>(p.out) + (int64_t)b * p.N * p.C + c;
    // two-pass statistics (mean, then centred second moment): no E[x^2]-mean^2 cancellation; the tile is
    // L1/L2-resident for the re-reads
    float s = 0.f, ss = 0.f;
    if (ok)
        for (int n = ng; n < p.N; n += 4) s += fgnn_ld(xb + (int64_t)n * p.C);
    const float mean = in_reduce4(s, red, cg, ng) / (float)p.N;
    if (ok)
        for (int n = ng; n < p.N; n += 4) { const float v = fgnn_ld(xb + (int64_t)n * p.C) - mean; ss += v * v; }
    const float var = in_reduce4(ss, red, cg, ng) / (float)p.N;
    const float rstd = rsqrtf(var + IN_EPS);
    if (ok)
        for (int n = ng; n < p.N; n += 4) {
            float v = (fgnn_ld(xb + (int64_t)n * p.C) - mean) * rstd;
            if (p.relu) v = fmaxf(v, 0.f);
            fgnn_st(yb + (int64_t)n * p.C, v);
        }
}

template <typename T>
__global__ __launch_bounds__(IN_THREADS) void instnorm_bwd_kernel(const InParams p) {
    __shared__ float red[4 * IN_CH];
    const int cg = threadIdx.x & (IN_CH - 1), ng = threadIdx.x >> 6;
    const int nblk = (p.C + IN_CH - 1) / IN_CH;
    const int b = blockIdx.x / nblk, c = (blockIdx.x - b * nblk) * IN_CH + cg;
    const bool ok = c < p.C;
    const int64_t base = (int64_t)b * p.N * p.C + c;
    const T* xb = static_cast<const T*>(p.x) + base;
    const T* gb = static_cast<const T*>(p.gy) + base;
    T* ob = static_cast<T*>(p.out) + base;
    // two-pass statistics (mean, then centred second moment): no E[x^2]-mean^2 cancellation; the tile is
    // L1/L2-resident for the re-reads
    float s = 0.f, ss = 0.f;
    if (ok)
        for (int n = ng; n < p.N; n += 4) s += fgnn_ld(xb + (int64_t)n * p.C);
    const float mean = in_reduce4(s, red, cg, ng) / (float)p.N;
    if (ok)
        for (int n = ng; n < p.N; n += 4) { const float v = fgnn_ld(xb + (int64_t)n * p.C) - mean; ss += v * v; }
    const float var = in_reduce4(ss, red, cg, ng) / (float)p.N;
    const float rstd = rsqrtf(var + IN_EPS);
    float sg = 0.f, sgx = 0.f;
    if (ok)
        for (int n = ng; n < p.N; n += 4) {
            const float xh = (fgnn_ld(xb + (int64_t)n * p.C) - mean) * rstd;
            float g = fgnn_ld(gb + (int64_t)n * p.C);
            if (p.relu && xh <= 0.f) g = 0.f;          // y = relu(xhat): y > 0  <=>  xhat > 0
            sg += g;
            sgx += g * xh;
        }
    sg = in_reduce4(sg, red, cg, ng) / (float)p.N;
    sgx = in_reduce4(sgx, red, cg, ng) / (float)p.N;
    if (ok)
        for (int n = ng; n < p.N; n += 4) {
            const float xh = (fgnn_ld(xb + (int64_t)n * p.C) - mean) * rstd;
            float g = fgnn_ld(gb + (int64_t)n * p.C);
            if (p.relu && xh <= 0.f) g = 0.f;
            fgnn_st(ob + (int64_t)n * p.C, rstd * (g - sg - xh * sgx));
        }
}

static int in_check(const void* a, const void* b, int B, int N, int C, int dtype) {
    if (!a || !b) FGNN_FAIL(FGNN_EINVAL, "instnorm: null pointer");
    if (B < 0 || N < 1 || C < 1) FGNN_FAIL(FGNN_EINVAL, "instnorm: bad sizes B=%d N=%d C=%d", B, N, C);
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "instnorm: unknown dtype %d", dtype);
    return FGNN_OK;
}

extern "C" int fgnn_instnorm_forward(const void* x, void* y, int B, int N, int C, int dtype, int relu,
                                     fgnn_stream_t stream) {
    int rc = in_check(x, y, B, N, C, dtype);
    if (rc) return rc;
    if (B == 0) return FGNN_OK;
    InParams p = {x, nullptr, nullptr, y, B, N, C, relu};
    const int grid = B * ((C + IN_CH - 1) / IN_CH);
    if (dtype == FGNN_F32) hipLaunchKernelGGL(instnorm_fwd_kernel<float>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(instnorm_fwd_kernel<bf16_t>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "instnorm forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

extern "C" int fgnn_instnorm_backward(const void* x, const void* gy, void* gx, int B, int N, int C, int dtype,
                                      int relu, fgnn_stream_t stream) {
    int rc = in_check(x, gy, B, N, C, dtype);
    if (rc) return rc;
    if (!gx) FGNN_FAIL(FGNN_EINVAL, "instnorm: null pointer");
    if (B == 0) return FGNN_OK;
    InParams p = {x, nullptr, gy, gx, B, N, C, relu};
    const int grid = B * ((C + IN_CH - 1) / IN_CH);
    if (dtype == FGNN_F32) hipLaunchKernelGGL(instnorm_bwd_kernel<float>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(instnorm_bwd_kernel<bf16_t>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "instnorm backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
