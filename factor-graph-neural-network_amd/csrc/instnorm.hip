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
#include <stdlib.h>

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

// ----------------------------------------------------------------------------------------
// Vectorised variants for C % 64 == 0 and N <= 128 (every InstanceNorm of the LDPC model): one workgroup owns
// (sample, 64-channel block); a thread owns one 16-byte channel chunk of up to MAXR rows and keeps it in
// registers, so x (and gy) are read from memory exactly once — the kernels above re-read the tile 3-4 times
// with 2-byte loads and ran at ~1.8 TB/s.  Per-channel sums fold over the row groups of a wave with xor-shuffles
// and over the 4 waves through LDS.
// ----------------------------------------------------------------------------------------
template <typename T> struct InChunk;
template <> struct InChunk<float> {
    static constexpr int EPC = 4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct InChunk<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static void load(const bf16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
        v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
    }
    __device__ static void store(bf16_t* p, const float (&v)[8]) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        uint4 t;
        b2 q;
        q[0] = (__bf16)v[0]; q[1] = (__bf16)v[1]; t.x = __builtin_bit_cast(unsigned, q);
        q[0] = (__bf16)v[2]; q[1] = (__bf16)v[3]; t.y = __builtin_bit_cast(unsigned, q);
        q[0] = (__bf16)v[4]; q[1] = (__bf16)v[5]; t.z = __builtin_bit_cast(unsigned, q);
        q[0] = (__bf16)v[6]; q[1] = (__bf16)v[7]; t.w = __builtin_bit_cast(unsigned, q);
        *reinterpret_cast<uint4*>(p) = t;
    }
};

// Sums of a[e] and b[e] over all row groups of the workgroup, per channel; every thread gets the totals of its chunk.
template <int EPC, int CPR, int CH>
__device__ __forceinline__ void in_fold2(float (&a)[EPC], float (&b)[EPC], float* red, int cg, int wave) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int m = CPR; m < 64; m <<= 1) { a[e] += __shfl_xor(a[e], m); b[e] += __shfl_xor(b[e], m); }   // row groups inside the wave
    }
    if ((threadIdx.x & 63) < CPR) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { red[wave * CH + cg * EPC + e] = a[e]; red[(4 + wave) * CH + cg * EPC + e] = b[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = cg * EPC + e;
        a[e] = (red[c] + red[CH + c]) + (red[2 * CH + c] + red[3 * CH + c]);
        b[e] = (red[4 * CH + c] + red[5 * CH + c]) + (red[6 * CH + c] + red[7 * CH + c]);
    }
    __syncthreads();
}

template <typename T, bool BWD, int CH, int MAXR>
__global__ __launch_bounds__(IN_THREADS) void instnorm_vec_kernel(const InParams p) {
    constexpr int EPC = InChunk<T>::EPC, CPR = CH / EPC, RG = IN_THREADS / CPR;    // MAXR >= ceil(N / RG) rows per thread
    __shared__ float red[8 * CH];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int cg = tid & (CPR - 1), rg = tid / CPR;
    const int nblk = p.C / CH;
    const int b = blockIdx.x / nblk, c0 = (blockIdx.x - b * nblk) * CH + cg * EPC;
    const int64_t base = (int64_t)b * p.N * p.C + c0;
    const T* xb = static_cast<const T*>(p.x) + base;
    const T* gb = static_cast<const T*>(p.gy) + base;
    T* ob = static_cast<T*>(p.out) + base;
    const int N = p.N;
    float x[MAXR][EPC], g[BWD ? MAXR : 1][EPC], K[EPC];
    InChunk<T>::load(xb, K);                               // row 0 of the sample: the shift of the one-pass variance
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int n = rg + i * RG;
        if (n < N) {
            InChunk<T>::load(xb + (int64_t)n * p.C, x[i]);
            if constexpr (BWD) InChunk<T>::load(gb + (int64_t)n * p.C, g[i]);
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { x[i][e] = K[e]; if constexpr (BWD) g[i][e] = 0.f; }
        }
    }
    // shifted one-pass statistics: sums of (x - K) and (x - K)^2 with K a member of the population, so
    // |mean - K| ~ std and var = E[(x-K)^2] - (E[x-K])^2 loses at most a few bits (padding rows contribute 0)
    const float invn = 1.0f / (float)N;
    float mean[EPC], rstd[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXR; ++i) { const float dv = x[i][e] - K[e]; x[i][e] = dv; s += dv; ss = fmaf(dv, dv, ss); }
        mean[e] = s; rstd[e] = ss;
    }
    in_fold2<EPC, CPR, CH>(mean, rstd, red, cg, wave);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float m = mean[e] * invn;
        const float var = fmaxf(rstd[e] * invn - m * m, 0.f);
        mean[e] = m;                                       // mean of (x - K)
        rstd[e] = rsqrtf(var + IN_EPS);
    }
    if constexpr (!BWD) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int n = rg + i * RG;
            if (n < N) {
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float v = (x[i][e] - mean[e]) * rstd[e]; o[e] = p.relu ? fmaxf(v, 0.f) : v; }
                InChunk<T>::store(ob + (int64_t)n * p.C, o);
            }
        }
    } else {
        float sg[EPC], sgx[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sg[e] = 0.f; sgx[e] = 0.f;
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                const float xh = (x[i][e] - mean[e]) * rstd[e];
                x[i][e] = xh;                              // xhat from here on
                if (p.relu && xh <= 0.f) g[i][e] = 0.f;    // y = relu(xhat): y > 0  <=>  xhat > 0
                sg[e] += g[i][e];                          // padding rows carry g = 0
                sgx[e] = fmaf(g[i][e], xh, sgx[e]);
            }
        }
        in_fold2<EPC, CPR, CH>(sg, sgx, red, cg, wave);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int n = rg + i * RG;
            if (n < N) {
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = rstd[e] * (g[i][e] - sg[e] * invn - x[i][e] * sgx[e] * invn);
                InChunk<T>::store(ob + (int64_t)n * p.C, o);
            }
        }
    }
}

static bool in_wide() {
    return true;
}

static bool in_vec_ok(const void* a, const void* b, const void* c, int N, int C) {
    return C % IN_CH == 0 && N <= 128 && !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15);
}

static int in_check(const void* a, const void* b, int B, int N, int C, int dtype) {
    if (!a || !b) FGNN_FAIL(FGNN_EINVAL, "instnorm: null pointer");
    if (B < 0 || N < 1 || C < 1) FGNN_FAIL(FGNN_EINVAL, "instnorm: bad sizes B=%d N=%d C=%d", B, N, C);
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "instnorm: unknown dtype %d", dtype);
    return FGNN_OK;
}

// rows a thread holds = ceil(N / row groups), rounded up to an instantiated count: fewer registers -> more workgroups
// per CU -> more bytes in flight (N = 96 at 64 channels needs 3 of the 4 rows N = 128 would)
template <typename T, bool BWD, int CH>
static void in_launch_rows(int grid, int N, hipStream_t st, const InParams& p) {
    constexpr int RG = IN_THREADS / (CH / InChunk<T>::EPC), FULL = 128 / RG;
    const int need = (N + RG - 1) / RG;
    if (need * 4 <= FULL) hipLaunchKernelGGL((instnorm_vec_kernel<T, BWD, CH, (FULL / 4 > 0 ? FULL / 4 : 1)>), dim3(grid), dim3(IN_THREADS), 0, st, p);
    else if (need * 2 <= FULL) hipLaunchKernelGGL((instnorm_vec_kernel<T, BWD, CH, (FULL / 2 > 0 ? FULL / 2 : 1)>), dim3(grid), dim3(IN_THREADS), 0, st, p);
    else if (need * 4 <= FULL * 3) hipLaunchKernelGGL((instnorm_vec_kernel<T, BWD, CH, (FULL * 3 / 4 > 0 ? FULL * 3 / 4 : 1)>), dim3(grid), dim3(IN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((instnorm_vec_kernel<T, BWD, CH, FULL>), dim3(grid), dim3(IN_THREADS), 0, st, p);
}

template <bool BWD>
static void in_launch_vec(int grid64, int N, int C, int dtype, hipStream_t st, const InParams& p) {
    // 128 channels per workgroup where the width allows: 256-byte row pieces and every thread busy at N = 48
    if (C % 128 == 0 && in_wide()) {
        if (dtype == FGNN_F32) in_launch_rows<float, BWD, 128>(grid64 / 2, N, st, p);
        else in_launch_rows<bf16_t, BWD, 128>(grid64 / 2, N, st, p);
    } else if (dtype == FGNN_F32) in_launch_rows<float, BWD, 64>(grid64, N, st, p);
    else in_launch_rows<bf16_t, BWD, 64>(grid64, N, st, p);
}

extern "C" int fgnn_instnorm_forward(const void* x, void* y, int B, int N, int C, int dtype, int relu,
                                     fgnn_stream_t stream) {
    int rc = in_check(x, y, B, N, C, dtype);
    if (rc) return rc;
    if (B == 0) return FGNN_OK;
    InParams p = {x, nullptr, nullptr, y, B, N, C, relu};
    const int grid = B * ((C + IN_CH - 1) / IN_CH);
    if (in_vec_ok(x, y, nullptr, N, C)) {
        in_launch_vec<false>(grid, N, C, dtype, (hipStream_t)stream, p);
    } else if (dtype == FGNN_F32) hipLaunchKernelGGL(instnorm_fwd_kernel<float>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
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
    if (in_vec_ok(x, gy, gx, N, C)) {
        in_launch_vec<true>(grid, N, C, dtype, (hipStream_t)stream, p);
    } else if (dtype == FGNN_F32) hipLaunchKernelGGL(instnorm_bwd_kernel<float>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(instnorm_bwd_kernel<bf16_t>, dim3(grid), dim3(IN_THREADS), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "instnorm backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// ----------------------------------------------------------------------------------------
// The classifier's closing pair as ONE pass (round 5): InstanceNorm + ReLU + the 128 -> 1 map that reads it
// (/root/reference/lib/model/mpnn/factor_mpnn_sp.py:104-108: Conv2d(dim,128,1) -> InstanceNorm2d -> ReLU -> Dropout/Identity ->
// Conv2d(128, 1, 1); the LDPC decoder's logit per variable).  Staged, the normalised [B][N][128] tensor is written, read back by a
// one-column library GEMM (hipBLASLt MT1x4x256: 79 us for 101 MB), and in the backward its gradient — the outer product
// gout[b,n] * w[c] — is written by another GEMM and read by the InstanceNorm backward, with a third pass for the map's weight
// gradient (profiles/r05/train_step_sequence.csv: 115 us forward + 150 us backward on the step's critical path).  Here:
//   forward : out[b,n] = bias + sum_c w[c] * relu(xhat[b,n,c])                                   reads x once, writes B*N values
//   backward: g = gout[b,n] * w[c] * [xhat > 0] formed in registers -> gx as in instnorm_vec_kernel;  gw[c] += sum gout * relu(xhat),
//             gbias += sum gout (per-workgroup partial rows, folded in fixed order by instnorm_dot_reduce_kernel)
// One workgroup owns whole samples (C == 128 == one channel block); the backward's workgroups walk the batch with a grid
// stride so that the weight-gradient partials are a few hundred rows.  y = relu(xhat) and g stay f32 (the staged path rounds
// both to the activations' dtype in between).
// ----------------------------------------------------------------------------------------
#define IND_CH 128
#define IND_PSTRIDE 132         // floats per partial row: 128 weight-gradient slots, the bias gradient, padding
#define IND_MAXGRID 512

struct IndParams {
    const void* x;
    const float* w;             // [128]
    const float* bias;          // [1] or NULL
    const void* gout;           // backward: [B][N]
    void* out;                  // forward: [B][N]; backward: gx [B][N][128]
    float* partials;            // backward: [grid][IND_PSTRIDE]
    int B, N;
};

template <typename T, bool BWD, int MAXR>
__global__ __launch_bounds__(IN_THREADS) void instnorm_dot_kernel(const IndParams p) {
    constexpr int CH = IND_CH, EPC = InChunk<T>::EPC, CPR = CH / EPC, RG = IN_THREADS / CPR;
    __shared__ float red[8 * CH];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int cg = tid & (CPR - 1), rg = tid / CPR;
    const int N = p.N;
    const float invn = 1.0f / (float)N;
    float w[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) w[e] = p.w[cg * EPC + e];
    float gwacc[EPC], gbacc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { gwacc[e] = 0.f; gbacc[e] = 0.f; }
    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        const T* xb = static_cast<const T*>(p.x) + (int64_t)b * N * CH + cg * EPC;
        float x[MAXR][EPC], go[BWD ? MAXR : 1], K[EPC];
        InChunk<T>::load(xb, K);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int n = rg + i * RG;
            if (n < N) {
                InChunk<T>::load(xb + (int64_t)n * CH, x[i]);
                if constexpr (BWD) go[i] = fgnn_ld(static_cast<const T*>(p.gout) + (int64_t)b * N + n);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) x[i][e] = K[e];
                if constexpr (BWD) go[i] = 0.f;
            }
        }
        float mean[EPC], rstd[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int i = 0; i < MAXR; ++i) { const float dv = x[i][e] - K[e]; x[i][e] = dv; s += dv; ss = fmaf(dv, dv, ss); }
            mean[e] = s; rstd[e] = ss;
        }
        in_fold2<EPC, CPR, CH>(mean, rstd, red, cg, wave);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float m = mean[e] * invn;
            const float var = fmaxf(rstd[e] * invn - m * m, 0.f);
            mean[e] = m;
            rstd[e] = rsqrtf(var + IN_EPS);
        }
        if constexpr (!BWD) {
            const float bias = p.bias ? p.bias[0] : 0.f;
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) d = fmaf(fmaxf((x[i][e] - mean[e]) * rstd[e], 0.f), w[e], d);
#pragma unroll
                for (int m = 1; m < CPR; m <<= 1) d += __shfl_xor(d, m);       // the row's 128 channels: CPR neighbouring lanes
                const int n = rg + i * RG;
                if (cg == 0 && n < N) fgnn_st(static_cast<T*>(p.out) + (int64_t)b * N + n, d + bias);
            }
        } else {
            T* ob = static_cast<T*>(p.out) + (int64_t)b * N * CH + cg * EPC;
            float sg[EPC], sgx[EPC], g[MAXR][EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                sg[e] = 0.f; sgx[e] = 0.f;
#pragma unroll
                for (int i = 0; i < MAXR; ++i) {
                    const float xh = (x[i][e] - mean[e]) * rstd[e];
                    x[i][e] = xh;
                    const bool on = xh > 0.f;
                    gwacc[e] = fmaf(go[i], on ? xh : 0.f, gwacc[e]);            // padding rows carry gout = 0
                    const float gv = on ? go[i] * w[e] : 0.f;
                    g[i][e] = gv;
                    sg[e] += gv;
                    sgx[e] = fmaf(gv, xh, sgx[e]);
                }
            }
#pragma unroll
            for (int i = 0; i < MAXR; ++i) gbacc[0] += go[i];
            in_fold2<EPC, CPR, CH>(sg, sgx, red, cg, wave);
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                const int n = rg + i * RG;
                if (n < N) {
                    float o[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] = rstd[e] * (g[i][e] - sg[e] * invn - x[i][e] * sgx[e] * invn);
                    InChunk<T>::store(ob + (int64_t)n * CH, o);
                }
            }
        }
    }
    if constexpr (BWD) {
        in_fold2<EPC, CPR, CH>(gwacc, gbacc, red, cg, wave);          // over the workgroup's row groups, per channel
        if (rg == 0) {
            float* pr = p.partials + (int64_t)blockIdx.x * IND_PSTRIDE;
#pragma unroll
            for (int e = 0; e < EPC; ++e) pr[cg * EPC + e] = gwacc[e];
            if (cg == 0) pr[CH] = gbacc[0];                            // (column 0 of the second fold: the sum of gout over the rows)
        }
    }
}

// gw[c] += sum over the partial rows (fixed order: 7 interleaved row groups, then the groups), gbias likewise
__global__ __launch_bounds__(1024) void instnorm_dot_reduce_kernel(const float* partials, int nrows, float* gw, float* gbias) {
    constexpr int COLS = IND_CH + 1, GROUPS = 7;
    __shared__ float red[GROUPS * COLS];
    const int tid = threadIdx.x, rgp = tid / COLS, c = tid - rgp * COLS;
    if (rgp < GROUPS) {
        float a = 0.f;
        for (int r = rgp; r < nrows; r += GROUPS) a += partials[(int64_t)r * IND_PSTRIDE + c];
        red[rgp * COLS + c] = a;
    }
    __syncthreads();
    if (tid < COLS) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) a += red[q * COLS + tid];
        if (tid < IND_CH) gw[tid] += a;
        else if (gbias) gbias[0] += a;
    }
}

static int ind_check(const void* x, const float* w, const void* o, int B, int N, int C, int dtype) {
    if (!x || !w || !o) FGNN_FAIL(FGNN_EINVAL, "instnorm_dot: null pointer");
    if (B < 0 || N < 1 || C < 1) FGNN_FAIL(FGNN_EINVAL, "instnorm_dot: bad sizes B=%d N=%d C=%d", B, N, C);
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "instnorm_dot: unknown dtype %d", dtype);
    if (C != IND_CH || N > 128 || N < 2) FGNN_FAIL(FGNN_EUNSUPPORTED, "instnorm_dot: C=%d N=%d (C == 128, 2 <= N <= 128)", C, N);
    if ((uintptr_t)x & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "instnorm_dot: x is not 16-byte aligned");
    return FGNN_OK;
}

template <typename T, bool BWD>
static void ind_launch(int grid, int N, hipStream_t st, const IndParams& p) {
    constexpr int RG = IN_THREADS / (IND_CH / InChunk<T>::EPC), FULL = 128 / RG;
    const int need = (N + RG - 1) / RG;
    if (need * 2 <= FULL) hipLaunchKernelGGL((instnorm_dot_kernel<T, BWD, FULL / 2>), dim3(grid), dim3(IN_THREADS), 0, st, p);
    else if (need * 4 <= FULL * 3) hipLaunchKernelGGL((instnorm_dot_kernel<T, BWD, FULL * 3 / 4>), dim3(grid), dim3(IN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((instnorm_dot_kernel<T, BWD, FULL>), dim3(grid), dim3(IN_THREADS), 0, st, p);
}

extern "C" int fgnn_instnorm_dot_forward(const void* x, const float* w, const float* bias, void* out, int B, int N, int C,
                                         int dtype, fgnn_stream_t stream) {
    int rc = ind_check(x, w, out, B, N, C, dtype);
    if (rc) return rc;
    if (B == 0) return FGNN_OK;
    IndParams p = {x, w, bias, nullptr, out, nullptr, B, N};
    if (dtype == FGNN_F32) ind_launch<float, false>(B, N, (hipStream_t)stream, p);
    else ind_launch<bf16_t, false>(B, N, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "instnorm_dot forward launch: %s", hipGetErrorString(e));
    fgnn_note_kernel("instnorm_dot_kernel<fwd>");
    return FGNN_OK;
}

extern "C" int64_t fgnn_instnorm_dot_workspace_bytes(int B) {
    const int grid = B < IND_MAXGRID ? (B > 0 ? B : 1) : IND_MAXGRID;
    return (int64_t)grid * IND_PSTRIDE * (int64_t)sizeof(float);
}

extern "C" int fgnn_instnorm_dot_backward(const void* x, const float* w, const void* gout, void* gx, float* gw, float* gbias,
                                          int B, int N, int C, int dtype, void* ws, int64_t ws_bytes, fgnn_stream_t stream) {
    int rc = ind_check(x, w, gx, B, N, C, dtype);
    if (rc) return rc;
    if (!gout || !gw) FGNN_FAIL(FGNN_EINVAL, "instnorm_dot: null pointer");
    if ((uintptr_t)gx & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "instnorm_dot: gx is not 16-byte aligned");
    if (B == 0) return FGNN_OK;
    if (!ws || ws_bytes < fgnn_instnorm_dot_workspace_bytes(B)) FGNN_FAIL(FGNN_EINVAL, "instnorm_dot backward: workspace too small");
    const int grid = B < IND_MAXGRID ? B : IND_MAXGRID;
    IndParams p = {x, w, nullptr, gout, gx, static_cast<float*>(ws), B, N};
    if (dtype == FGNN_F32) ind_launch<float, true>(grid, N, (hipStream_t)stream, p);
    else ind_launch<bf16_t, true>(grid, N, (hipStream_t)stream, p);
    hipLaunchKernelGGL(instnorm_dot_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, static_cast<const float*>(ws), grid, gw, gbias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "instnorm_dot backward launch: %s", hipGetErrorString(e));
    fgnn_note_kernel("instnorm_dot_kernel<bwd>");
    return FGNN_OK;
}
