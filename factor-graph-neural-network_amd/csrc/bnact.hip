// bnact.hip — train-mode BatchNorm fused with the (Leaky)ReLU that follows it, on dense channel-fastest
// activations x[R][C], R = B*N rows — the normalisation of the reference's conv1 / conv2 blocks
// (`Conv2d 1x1 + BatchNorm2d + LeakyReLU`, /root/reference/lib/model/mpnn/mp_nn_residual.py:25-35), of
// `mp_conv_v2.bn` + ReLU (mp_nn.py:57-58,170-173) and of `iid_mapping_bn` (base_model.py:62-79); SURVEY §8f-1.
//
//   forward : per-channel batch statistics over the R rows (biased variance for the normalisation, running
//             statistics updated with the unbiased one, momentum 0.1 — torch.nn.BatchNorm2d semantics), then
//             y = act(x * scale + shift),  scale = gamma * invstd,  shift = beta - mean * scale
//   backward: g = gy * act'(pre);  dgamma = sum g * xhat;  dbeta = sum g;
//             gx = gamma * invstd * (g - dbeta / R - xhat * dgamma / R)
//
// Four streaming kernels (16-byte loads, each thread owns a fixed 16-byte channel group so per-channel
// partial sums live in registers) + two tiny finalisers; partials of the ~1000 workgroups go through a
// workspace, never through atomics.  Replaces 4 ATen kernels + 2 activation kernels per layer per step.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>

#define BN_THREADS 256
#define BN_GRID 512         // workgroups of the reducing kernels (= partials the finalisers fold)
#define BN_MAXPART 1024     // workspace rows: the node-wise map's statistics epilogue may bring more partials
#define BN_APPLY_GRID 4096  // workgroups of the element-wise passes: 16 per CU keep enough loads in flight (512: -25 % on 200 MB tensors)

struct BnParams {
    const void* x;
    const void* gy;
    void* out;
    float* ws;                 // [grid][2][C]
    const float* a;            // forward apply: scale   | backward: mean
    const float* b;            // forward apply: shift   | backward: invstd
    const float* gamma;
    const float* beta;
    const float* ref;          // per-channel shift K for the variance accumulation (row 0 of x), or NULL = 0
    const void* addend;        // forward apply: y = act(...) + addend (+ addend2 + addend3), each y's layout or NULL
    const void* addend2;
    const void* addend3;
    int aperiod[3];            // rows of y per addend row (1 = a tensor of y's shape; m = one row per m consecutive rows: a per-sample
                               // vector broadcast over the sample's m nodes)
    int64_t R;
    int C, cshift;             // cshift = log2(C / EPC)
    int rows_per_wg;
    float slope;               // LeakyReLU slope (0 = ReLU, 1 = identity)
    float dsum_scale;          // backward apply: 1 / R
    const float* dgamma;
    const float* dbeta;
    // the reducing kernels finalise their own sums in the last workgroup (fgnn_gridfold.h) when fold.tickets != NULL
    FgnnFold fold;
    fgnn_bn_final fin;         // MODE 0: the forward statistics
    float* dsum;               // MODE 1: [2][C] dbeta, dgamma out
    float* gweight;            // MODE 1: ACCUMULATED into, or NULL
    float* gbias;
};

template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int EPC = 4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    __device__ static void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    }
};
template <> struct Chunk<bf16_t> {
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

// Two per-channel sums over this workgroup's rows:
//   MODE 0 (forward statistics): s0 = sum (x - K), s1 = sum (x - K)^2          K = row 0 of x
//   MODE 1 (backward reduce)   : s0 = sum g,       s1 = sum g * xhat           g = gy * act'(pre)
template <typename T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(const BnParams p) {
    constexpr int EPC = Chunk<T>::EPC;
    __shared__ __attribute__((aligned(16))) float red[BN_THREADS * 2 * EPC];
    const int tid = threadIdx.x;
    const int cpr = 1 << p.cshift;                        // chunks per row
    const int cg = tid & (cpr - 1), rg = tid >> p.cshift; // channel group, row group
    const int rstep = BN_THREADS >> p.cshift;
    const int c0 = cg * EPC;
    const T* xg = static_cast<const T*>(p.x);
    const T* gg = static_cast<const T*>(p.gy);
    float s0[EPC], s1[EPC], ka[EPC], kb[EPC], kc[EPC], kd[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        s0[e] = 0.f; s1[e] = 0.f;
        if (MODE == 0) { ka[e] = p.ref ? p.ref[c0 + e] : 0.f; kb[e] = 0.f; kc[e] = 0.f; kd[e] = 0.f; }
        else {
            ka[e] = p.a[c0 + e];                          // mean
            kb[e] = p.b[c0 + e];                          // invstd
            kc[e] = p.gamma[c0 + e] * kb[e];              // scale
            kd[e] = p.beta[c0 + e] - ka[e] * kc[e];       // shift
        }
    }
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    const int64_t r_end = r_begin + p.rows_per_wg < p.R ? r_begin + p.rows_per_wg : p.R;
#pragma unroll 4
    for (int64_t r = r_begin + rg; r < r_end; r += rstep) {
        float v[EPC];
        Chunk<T>::load(xg + r * p.C + c0, v);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float dlt = v[e] - ka[e]; s0[e] += dlt; s1[e] = fmaf(dlt, dlt, s1[e]); }
        } else {
            float g[EPC];
            Chunk<T>::load(gg + r * p.C + c0, g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float pre = fmaf(v[e], kc[e], kd[e]);
                const float ge = pre > 0.f ? g[e] : g[e] * p.slope;
                s0[e] += ge;
                s1[e] = fmaf(ge, (v[e] - ka[e]) * kb[e], s1[e]);
            }
        }
    }
    // fold the row groups of each channel group through LDS
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[(tid * 2) * EPC + e] = s0[e]; red[(tid * 2 + 1) * EPC + e] = s1[e]; }
    __syncthreads();
    if (rg == 0) {
        for (int q = 1; q < rstep; ++q) {
            const int t2 = q * cpr + cg;
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s0[e] += red[(t2 * 2) * EPC + e]; s1[e] += red[(t2 * 2 + 1) * EPC + e]; }
        }
        float* w = p.ws + (int64_t)blockIdx.x * 2 * p.C;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { fgnn_fold_store(w + c0 + e, s0[e]); fgnn_fold_store(w + p.C + c0 + e, s1[e]); }
    }
    if (p.fold.tickets) {                                 // no finaliser launch: the last workgroup folds all rows and finalises
        double* sums = reinterpret_cast<double*>(red);    // (red is free: the fold starts with a barrier)
        if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) {
            if (MODE == 0) fgnn_bn_final_apply(p.fin, p.C, sums);
            else fgnn_bn_bwd_final_apply(p.C, sums, nullptr, nullptr, p.dsum, p.gweight, p.gbias);
        }
    }
}

// Sum of the workgroup partials of one channel: 16 channels x 16 partial groups per block, folded in LDS.
// 4 channels x 64 partial groups per block (grid = C/4 blocks: enough workgroups in flight to cover the
// cross-die latency of partials written on all 8 XCDs); folded in f64 through LDS in a fixed order.
#define BN_FC 4
__device__ __forceinline__ void bn_fold(const float* ws, int nwg, int C, int c, int pg, bool ok, double& s0, double& s1) {
    __shared__ double red0[256], red1[256];
    double a = 0.0, b = 0.0;
    if (ok) {
#pragma unroll 16
        for (int w = pg; w < nwg; w += 256 / BN_FC) { a += (double)ws[(int64_t)w * 2 * C + c]; b += (double)ws[(int64_t)w * 2 * C + C + c]; }
    }
    red0[threadIdx.x] = a;
    red1[threadIdx.x] = b;
    __syncthreads();
    const int cc = threadIdx.x & (BN_FC - 1);
    a = 0.0; b = 0.0;
    if (pg < 8)                                           // two-level fold: 8 x 8 partial groups
        for (int q = pg * 8; q < pg * 8 + 8; ++q) { a += red0[q * BN_FC + cc]; b += red1[q * BN_FC + cc]; }
    __syncthreads();
    if (pg < 8) { red0[threadIdx.x] = a; red1[threadIdx.x] = b; }
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (pg == 0)
        for (int q = 0; q < 8; ++q) { s0 += red0[q * BN_FC + cc]; s1 += red1[q * BN_FC + cc]; }
}

// forward finaliser (the default; also serves partials produced outside this library): mean / invstd / scale / shift and
// the running statistics — the arithmetic of fgnn_bn_final_apply (fgnn_gridfold.h), 4 channels per workgroup
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* ws, int nwg, int C, const fgnn_bn_final fin) {
    if (fin.num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *fin.num_batches_tracked += 1;      // BatchNorm2d.num_batches_tracked
    const int c = blockIdx.x * BN_FC + (threadIdx.x & (BN_FC - 1)), pg = threadIdx.x / BN_FC;
    double s0, s1;
    bn_fold(ws, nwg, C, c, pg, c < C, s0, s1);
    if (pg != 0 || c >= C) return;
    const double n = (double)fin.count;
    const double m0 = s0 / n;                             // mean of (x - K)
    double var = s1 / n - m0 * m0;
    if (var < 0.0) var = 0.0;
    const float mu = (float)(m0 + (fin.shift_k ? (double)fin.shift_k[c] : 0.0));
    const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
    fin.mean[c] = mu;
    fin.invstd[c] = is;
    const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
    fin.scale[c] = g * is;
    fin.shift[c] = b - mu * g * is;
    if (fin.running_mean) {
        const double np = (double)(fin.population > 0 ? fin.population : fin.count);
        const double unbiased = np > 1.0 ? var * np / (np - 1.0) : var;
        fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * mu;
        fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
    }
}

// backward finaliser: dgamma = sum g xhat, dbeta = sum g (also written to the caller's gradient buffers)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* ws, int nwg, int C, float* dsum,
                                                           float* gweight, float* gbias) {
    const int c = blockIdx.x * BN_FC + (threadIdx.x & (BN_FC - 1)), pg = threadIdx.x / BN_FC;
    double s0, s1;
    bn_fold(ws, nwg, C, c, pg, c < C, s0, s1);
    if (pg != 0 || c >= C) return;
    dsum[c] = (float)s0;     // dbeta
    dsum[C + c] = (float)s1; // dgamma
    if (gbias) gbias[c] += (float)s0;
    if (gweight) gweight[c] += (float)s1;
}

// MODE 0: y = act(x * scale + shift)
// MODE 1: gx = gamma * invstd * (g - dbeta/R - xhat * dgamma/R),  g = gy * act'(pre)
template <typename T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const BnParams p) {
    constexpr int EPC = Chunk<T>::EPC;
    const int tid = threadIdx.x;
    const int cpr = 1 << p.cshift;
    const int cg = tid & (cpr - 1), rg = tid >> p.cshift;
    const int rstep = BN_THREADS >> p.cshift;
    const int c0 = cg * EPC;
    const T* xg = static_cast<const T*>(p.x);
    const T* gg = static_cast<const T*>(p.gy);
    T* og = static_cast<T*>(p.out);
    float ka[EPC], kb[EPC], kc[EPC], kd[EPC], ke[EPC], kf[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        if (MODE == 0) { ka[e] = p.a[c0 + e]; kb[e] = p.b[c0 + e]; kc[e] = kd[e] = ke[e] = kf[e] = 0.f; }
        else {
            ka[e] = p.a[c0 + e];                          // mean
            kb[e] = p.b[c0 + e];                          // invstd
            kc[e] = p.gamma[c0 + e] * kb[e];              // scale
            kd[e] = p.beta[c0 + e] - ka[e] * kc[e];       // shift
            ke[e] = p.dbeta[c0 + e] * p.dsum_scale;       // dbeta / R
            kf[e] = p.dgamma[c0 + e] * p.dsum_scale;      // dgamma / R
        }
    }
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    const int64_t r_end = r_begin + p.rows_per_wg < p.R ? r_begin + p.rows_per_wg : p.R;
#pragma unroll 4
    for (int64_t r = r_begin + rg; r < r_end; r += rstep) {
        float v[EPC], o[EPC];
        Chunk<T>::load(xg + r * p.C + c0, v);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float pre = fmaf(v[e], ka[e], kb[e]); o[e] = pre > 0.f ? pre : pre * p.slope; }
            const void* const ads[3] = {p.addend, p.addend2, p.addend3};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (ads[a]) {
                    float ad[EPC];
                    const int64_t ar = p.aperiod[a] > 1 ? r / p.aperiod[a] : r;
                    Chunk<T>::load(static_cast<const T*>(ads[a]) + ar * p.C + c0, ad);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] += ad[e];
                }
            }
        } else {
            float g[EPC];
            Chunk<T>::load(gg + r * p.C + c0, g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float pre = fmaf(v[e], kc[e], kd[e]);
                const float ge = pre > 0.f ? g[e] : g[e] * p.slope;
                const float xh = (v[e] - ka[e]) * kb[e];
                o[e] = kc[e] * (ge - ke[e] - xh * kf[e]);
            }
        }
        Chunk<T>::store(og + r * p.C + c0, o);
    }
}

static int bn_apply_grid() {
    return BN_APPLY_GRID;      // (swept 1024 / 2048 / 8192 in round 5: no effect, profiles/r05/README.md)
}

static int bn_reduce_grid() {
    return BN_GRID > BN_MAXPART ? BN_MAXPART : BN_GRID;      // (swept 256 / 1024 in round 5: no effect)
}

static int bn_plan(int64_t R, int C, int dtype, BnParams* p, int* grid, int target = 0) {
    const int epc = dtype == FGNN_F32 ? 4 : 8;
    if (C % epc != 0) return -1;
    const int cpr = C / epc;
    if (cpr > BN_THREADS || (cpr & (cpr - 1)) != 0) return -1;
    int sh = 0;
    while ((1 << sh) < cpr) ++sh;
    p->cshift = sh;
    p->C = C;
    p->R = R;
    int g = target > 0 ? target : bn_reduce_grid();
    int64_t rows = (R + g - 1) / g;
    const int rstep = BN_THREADS / cpr;
    if (rows < rstep) rows = rstep;
    p->rows_per_wg = (int)rows;
    *grid = (int)((R + rows - 1) / rows);
    return 0;
}

extern "C" int fgnn_bn_supported(int64_t R, int C, int dtype) {
    BnParams p;
    int grid;
    return (R > 0 && C > 0 && (dtype == FGNN_F32 || dtype == FGNN_BF16) && bn_plan(R, C, dtype, &p, &grid) == 0) ? 1 : 0;
}

extern "C" int64_t fgnn_bn_workspace_bytes(int64_t R, int C) { return (int64_t)BN_MAXPART * 2 * C * 4 + 2 * C * 4; }

// Who finalises a reduction: the producing launch's last workgroup (csrc/fgnn_gridfold.h) or a small launch of its own.  Measured on
// MI355X (gpurun_out/r05b, LDPC step, 4096 codewords): the in-kernel fold is six serialised memory-side round trips (store
// acknowledgement, ticket, loads — twice) = 8-10 us at the tail of EVERY producer (block_tail_stats 35.6 vs 27.9 us with its finaliser
// launch included, mpconv_fwd_ws 54 vs 47 us, the step 15.66 vs 15.50 ms) against ~7 us for a finaliser kernel inside the replayed
// graph: the separate launch is the default, fgnn_set_inkernel_finalisers(1) selects the fold (both run in the GPU suite).
static int g_inkernel_finalisers = -1;
int fgnn_separate_finalisers(void) {
    if (g_inkernel_finalisers < 0) g_inkernel_finalisers = 0;
    return !g_inkernel_finalisers;
}
extern "C" int fgnn_set_inkernel_finalisers(int on) {
    const int was = !fgnn_separate_finalisers();
    g_inkernel_finalisers = on ? 1 : 0;
    return was;
}

static int bn_check_final(const char* who, const fgnn_bn_final* fin) {
    if (!fin || !fin->mean || !fin->invstd || !fin->scale || !fin->shift) FGNN_FAIL(FGNN_EINVAL, "%s: null pointer in fgnn_bn_final", who);
    if (fin->count < 1 || (fin->population != 0 && fin->population < fin->count)) FGNN_FAIL(FGNN_EINVAL, "%s: bad row counts in fgnn_bn_final", who);
    if ((fin->running_mean == nullptr) != (fin->running_var == nullptr)) FGNN_FAIL(FGNN_EINVAL, "%s: running_mean and running_var go together", who);
    return FGNN_OK;
}

// Host-side helper of the other translation units: the stand-alone forward finaliser.
int fgnn_bn_finalize_launch(const float* partials, int npartials, int C, const fgnn_bn_final* fin, hipStream_t st) {
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, st, partials, npartials, C, *fin);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Forward statistics of x [R][C] (count = R rows): one reducing launch whose last workgroup finalises (fgnn_bn_final).
extern "C" int fgnn_bn_stats(const void* x, int64_t R, int C, int dtype, const fgnn_bn_final* fin, void* workspace,
                             int64_t workspace_bytes, void* fold_scratch, fgnn_stream_t stream) {
    BnParams p = {};
    int grid, rc;
    if (!x || !workspace) FGNN_FAIL(FGNN_EINVAL, "bn_stats: null pointer");
    if ((rc = bn_check_final("bn_stats", fin))) return rc;
    if (fin->count != R) FGNN_FAIL(FGNN_EINVAL, "bn_stats: fin->count must be R");
    if (bn_plan(R, C, dtype, &p, &grid)) FGNN_FAIL(FGNN_EUNSUPPORTED, "bn: C=%d not a supported channel count", C);
    if (workspace_bytes < fgnn_bn_workspace_bytes(R, C)) FGNN_FAIL(FGNN_EINVAL, "bn: workspace too small");
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    // per-channel shift K = x[0][:] (f32): row 0 of x where it lies — the reducing kernel and the finaliser read it there (until round 6 a
    // 4 us launch copied it into the workspace first: 39 launches of a synthetic-PGM training step); nothing writes x between the two
    const float* ref = dtype == FGNN_F32 ? (const float*)x : nullptr;      // bf16: accumulate against K = 0 (values are O(1) after the
                                                                           // preceding map; f32 sums, f64 finaliser)
    p.x = x; p.ws = ws; p.ref = ref;
    p.fin = *fin;
    if (ref) {
        if (fin->shift_k) FGNN_FAIL(FGNN_EINVAL, "bn_stats: shift_k is for partials produced elsewhere");
        p.fin.shift_k = ref;
    }
    const bool inkernel = fold_scratch && 2 * C <= FGNN_FOLD_MAXJ && !fgnn_separate_finalisers();
    p.fold = fgnn_fold_make(ws, inkernel ? fold_scratch : nullptr, grid, C);
    if (dtype == FGNN_F32) hipLaunchKernelGGL((bn_reduce_kernel<float, 0>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 0>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    if (!inkernel) hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, st, ws, grid, C, p.fin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "bn_stats launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// Forward statistics from per-workgroup partials somebody else produced: partials[w][0][c] = sum (y - K), partials[w][1][c] =
// sum (y - K)^2 over that workgroup's rows, K = fin->shift_k (NULL = 0).  One small launch.  (The producers of this library take the
// fgnn_bn_final themselves and need no such call; this entry point serves partials formed elsewhere.)
extern "C" int fgnn_bn_finalize(const float* partials, int npartials, int C, const fgnn_bn_final* fin, fgnn_stream_t stream) {
    int rc;
    if (!partials) FGNN_FAIL(FGNN_EINVAL, "bn_finalize: null pointer");
    if ((rc = bn_check_final("bn_finalize", fin))) return rc;
    if (npartials < 1 || npartials > BN_MAXPART || C < 1) FGNN_FAIL(FGNN_EINVAL, "bn_finalize: bad sizes");
    if (fgnn_bn_finalize_launch(partials, npartials, C, fin, (hipStream_t)stream))
        FGNN_FAIL(FGNN_ELAUNCH, "bn_finalize launch: %s", hipGetErrorString(hipGetLastError()));
    return FGNN_OK;
}

// y = act(x * scale + shift) [+ addend + addend2 + addend3], act = LeakyReLU(slope) (slope 0: ReLU, slope 1: identity).
// addend_period[a] (or NULL = all 1): addend a has one row per `period` consecutive rows of y (a per-sample vector broadcast over
// the sample's nodes: row r reads addend row r / period).
extern "C" int fgnn_bn_apply(const void* x, void* y, int64_t R, int C, int dtype, const float* scale,
                             const float* shift, float slope, const void* addend, const void* addend2,
                             const void* addend3, const int32_t* addend_period, fgnn_stream_t stream) {
    BnParams p = {};
    int grid;
    if (!x || !y || !scale || !shift) FGNN_FAIL(FGNN_EINVAL, "bn_apply: null pointer");
    if (bn_plan(R, C, dtype, &p, &grid, bn_apply_grid())) FGNN_FAIL(FGNN_EUNSUPPORTED, "bn: C=%d not a supported channel count", C);
    p.x = x; p.out = y; p.a = scale; p.b = shift; p.slope = slope; p.addend = addend; p.addend2 = addend2; p.addend3 = addend3;
    for (int a = 0; a < 3; ++a) {
        p.aperiod[a] = addend_period ? addend_period[a] : 1;
        if (p.aperiod[a] < 1) FGNN_FAIL(FGNN_EINVAL, "bn_apply: addend period < 1");
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FGNN_F32) hipLaunchKernelGGL((bn_apply_kernel<float, 0>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, 0>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "bn_apply launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// The reduction half of the backward: per-channel dbeta / dgamma -> dsum [2][C] (in the workspace, behind the partial rows) and
// ACCUMULATED into gweight / gbias (may be NULL); one launch (the last workgroup finalises).  Host-side helper of this library
// (csrc/block_tail.hip: BatchNorm1's sums in front of the fused head kernel), not C ABI.
int fgnn_bn_backward_sums(const void* x, const void* gy, int64_t R, int C, int dtype, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float slope, float* gweight, float* gbias,
                          void* workspace, void* fold_scratch, hipStream_t st, BnParams* out, const float** dsum_out) {
    BnParams p = {};
    int grid;
    if (bn_plan(R, C, dtype, &p, &grid)) return -1;
    float* ws = (float*)workspace;
    float* dsum = ws + (int64_t)BN_MAXPART * 2 * C;
    p.x = x; p.gy = gy; p.out = nullptr; p.ws = ws; p.a = mean; p.b = invstd; p.gamma = gamma; p.beta = beta;
    p.slope = slope; p.dsum_scale = 1.0f / (float)R; p.dbeta = dsum; p.dgamma = dsum + C;
    p.dsum = dsum; p.gweight = gweight; p.gbias = gbias;
    const bool inkernel = fold_scratch && 2 * C <= FGNN_FOLD_MAXJ && !fgnn_separate_finalisers();
    p.fold = fgnn_fold_make(ws, inkernel ? fold_scratch : nullptr, grid, C);
    if (dtype == FGNN_F32) hipLaunchKernelGGL((bn_reduce_kernel<float, 1>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 1>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    if (!inkernel) hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, st, ws, grid, C, dsum, gweight, gbias);
    if (out) *out = p;
    if (dsum_out) *dsum_out = dsum;
    return 0;
}

int fgnn_bn_backward_sums_bf16(const void* x, const void* gy, int64_t R, int C, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, float slope, float* gweight, float* gbias,
                               void* workspace, void* fold_scratch, hipStream_t st, const float** dsum_out) {
    return fgnn_bn_backward_sums(x, gy, R, C, FGNN_BF16, mean, invstd, gamma, beta, slope, gweight, gbias, workspace, fold_scratch, st,
                                 nullptr, dsum_out);
}

// gx, and gweight / gbias (ACCUMULATED into, may be NULL): a reducing launch (its last workgroup finalises the sums) + the
// element-wise pass
extern "C" int fgnn_bn_backward(const void* x, const void* gy, void* gx, int64_t R, int C, int dtype,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float slope, float* gweight, float* gbias, void* workspace,
                                int64_t workspace_bytes, void* fold_scratch, fgnn_stream_t stream) {
    if (!x || !gy || !gx || !mean || !invstd || !gamma || !beta || !workspace)
        FGNN_FAIL(FGNN_EINVAL, "bn_backward: null pointer");
    if (workspace_bytes < fgnn_bn_workspace_bytes(R, C)) FGNN_FAIL(FGNN_EINVAL, "bn: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BnParams p;
    if (fgnn_bn_backward_sums(x, gy, R, C, dtype, mean, invstd, gamma, beta, slope, gweight, gbias, workspace, fold_scratch, st, &p, nullptr))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "bn: C=%d not a supported channel count", C);
    int agrid;
    BnParams pa = p;
    pa.out = gx;
    pa.fold.tickets = nullptr;
    (void)bn_plan(R, C, dtype, &pa, &agrid, bn_apply_grid());   // same fields, finer row split
    if (dtype == FGNN_F32) hipLaunchKernelGGL((bn_apply_kernel<float, 1>), dim3(agrid), dim3(BN_THREADS), 0, st, pa);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, 1>), dim3(agrid), dim3(BN_THREADS), 0, st, pa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "bn_backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// backward finaliser for partials of (sum g, sum g * x) with the RAW x: dbeta = S0, dgamma = invstd (S1 - mean S0)
// (the default; with the in-kernel fold csrc/block_tail.hip's grad kernel finalises its own)
__global__ __launch_bounds__(256) void bn_bwd_final_raw_kernel(const float* ws, int nwg, int C, const float* mean,
                                                               const float* invstd, float* dsum, float* gweight, float* gbias) {
    const int c = blockIdx.x * BN_FC + (threadIdx.x & (BN_FC - 1)), pg = threadIdx.x / BN_FC;
    double s0, s1;
    bn_fold(ws, nwg, C, c, pg, c < C, s0, s1);
    if (pg != 0 || c >= C) return;
    const double dg = (double)invstd[c] * (s1 - (double)mean[c] * s0);
    dsum[c] = (float)s0;
    dsum[C + c] = (float)dg;
    if (gbias) gbias[c] += (float)s0;
    if (gweight) gweight[c] += (float)dg;
}

int fgnn_bn_bwd_final_raw_launch(const float* partials, int npartials, int C, const float* mean, const float* invstd, float* dsum,
                                 float* gweight, float* gbias, hipStream_t st) {
    hipLaunchKernelGGL(bn_bwd_final_raw_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, st, partials, npartials, C, mean,
                       invstd, dsum, gweight, gbias);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The element-wise half of the backward alone: gx = gamma invstd (g - dbeta / R - xhat dgamma / R), g = gy act'(pre), from sums
// somebody else finalised (dsum [2][C] = dbeta, dgamma: fgnn_block_tail_backward leaves BatchNorm2's).
extern "C" int fgnn_bn_backward_apply(const void* x, const void* gy, void* gx, int64_t R, int C, int dtype,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      float slope, const float* dsum, fgnn_stream_t stream) {
    BnParams p = {};
    int grid;
    if (!x || !gy || !gx || !mean || !invstd || !gamma || !beta || !dsum)
        FGNN_FAIL(FGNN_EINVAL, "bn_backward_apply: null pointer");
    if (bn_plan(R, C, dtype, &p, &grid, bn_apply_grid())) FGNN_FAIL(FGNN_EUNSUPPORTED, "bn: C=%d not a supported channel count", C);
    hipStream_t st = (hipStream_t)stream;
    p.x = x; p.gy = gy; p.out = gx; p.a = mean; p.b = invstd; p.gamma = gamma; p.beta = beta;
    p.slope = slope; p.dsum_scale = 1.0f / (float)R; p.dbeta = dsum; p.dgamma = dsum + C;
    if (dtype == FGNN_F32) hipLaunchKernelGGL((bn_apply_kernel<float, 1>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, 1>), dim3(grid), dim3(BN_THREADS), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "bn_backward_apply launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
