// ldpc_datapath.hip — the LDPC data path in front of the decoder model, on the GPU:
//
//   encode            codeword = [s | G s mod 2]        `s2t(s, 48, 48, Gfile, smn=True)`,
//                                                        /root/reference/lib/data/MNC/MNC_py.cpp:22-83
//   channel           y = 2 gcx (t - 1/2) + z1 (+ gcx sigma_b z2 where u < rho),  gcx = 10^(snr_db/20)
//                                                        `t2y`, MNC_py.cpp:86-102
//   model inputs      node / hop / edge features gathered from y along the code's incidence lists
//                                                        lib/data/ldpc_dataset.py:92-106,222-236
//
// The reference does this per codeword on the host (a pybind11 call per item, numpy takes, a DataLoader); at
// B = 4096 codewords per step per GPU that is the input bottleneck.  Here a batch is two launches of byte / gather
// work, written once, coalesced, in the storage dtype the model kernels read.  The random draws (z1, u, z2) are
// inputs (the transform is deterministic and is checked bit-for-bit (encode) / to f32 rounding (channel) against
// the oracle) or come from a counter-based generator inside the kernel (fgnn_ldpc_channel_features_rng).
#include "fgnn_common.h"
#include <stdint.h>

#define LD_THREADS 256
#define LD_CW_PER_WG 64

// gmask[r] = row r of G packed over the K (<= 64) message bits: parity bit r = popcount(gmask[r] & s) & 1
__global__ __launch_bounds__(LD_THREADS) void ldpc_encode_kernel(const uint8_t* __restrict__ s,
                                                                 const unsigned long long* __restrict__ gmask,
                                                                 int64_t B, int K, int P, uint8_t* __restrict__ cw) {
    __shared__ unsigned long long msg[LD_CW_PER_WG];
    __shared__ unsigned long long gm[64];
    const int64_t b0 = (int64_t)blockIdx.x * LD_CW_PER_WG;
    const int tid = threadIdx.x;
    if (tid < LD_CW_PER_WG) {
        unsigned long long m = 0;
        if (b0 + tid < B) {
            const uint8_t* sp = s + (b0 + tid) * K;
            for (int c = 0; c < K; ++c) m |= (unsigned long long)(sp[c] & 1) << c;
        }
        msg[tid] = m;
    } else if (tid - LD_CW_PER_WG < P) {
        gm[tid - LD_CW_PER_WG] = gmask[tid - LD_CW_PER_WG];
    }
    __syncthreads();
    const int N = K + P;
    const int64_t nb = B - b0 < LD_CW_PER_WG ? B - b0 : LD_CW_PER_WG;
    for (int o = tid; o < nb * N; o += LD_THREADS) {
        const int w = o / N, n = o - w * N;
        const unsigned long long m = msg[w];
        const unsigned bit = n < K ? (unsigned)((m >> n) & 1ull) : (unsigned)(__popcll(m & gm[n - K]) & 1);
        cw[b0 * N + o] = (uint8_t)bit;
    }
}

struct LdFeatParams {
    const uint8_t* cw;           // [B][nvar]
    const float *snr_db, *sigma_b, *z1, *u, *z2;
    const int* var_to_factors;   // [nvar][dv]   (nn_idx_f2v)
    const int* factor_to_vars;   // [nchk][dc]   (nn_idx_v2f)
    float* y;                    // [B][nvar] f32
    void *node, *hop, *ef_f2v, *ef_v2f;
    float rho;
    int nvar, nchk, dv, dc;
    unsigned long long seed, offset;     // RNG variant: Philox key and stream offset (z1 / u / z2 are not read)
};

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so the draws of
// codeword bit i of the batch depend on (seed, offset, i) only — not on the grid, the launch order or earlier calls.
// oracle/fgnn_oracle.py::philox4x32 restates it in numpy; tests compare the two streams through the channel output.
__device__ __forceinline__ void ld_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&r)[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
// standard normal from two 32-bit words (Box-Muller, cosine branch): u1 in (0, 1], u2 in [0, 1)
__device__ __forceinline__ float ld_normal(unsigned a, unsigned b) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-08f, u2 = (float)(b >> 8) * 5.9604644775390625e-08f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}


// One workgroup per codeword: y into LDS, then every output array is written by flat index (coalesced).
template <typename T, bool RNG>
__global__ __launch_bounds__(LD_THREADS) void ldpc_features_kernel(const LdFeatParams p) {
    __shared__ float ys[1024];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, nvar = p.nvar, nchk = p.nchk, dv = p.dv, dc = p.dc;
    const float snr = p.snr_db[b], sb = p.sigma_b[b];
    const float gcx = exp2f(snr * 0.16609640474436813f);          // 10^(snr/20) = 2^(snr log2(10)/20)
    for (int n = tid; n < nvar; n += LD_THREADS) {
        const int64_t i = b * nvar + n;
        float z1, u, z2;
        if (RNG) {          // the channel's draws (MNC_py.cpp:89,94-97) from the counter (i, offset): no noise tensors in HBM
            unsigned ra[4], rb[4];
            const unsigned long long ctr = (unsigned long long)i;
            ld_philox((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)p.offset, (unsigned)(p.offset >> 32), (unsigned)p.seed, (unsigned)(p.seed >> 32), ra);
            ld_philox((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)p.offset, (unsigned)(p.offset >> 32) ^ 0x80000000u, (unsigned)p.seed, (unsigned)(p.seed >> 32), rb);
            z1 = ld_normal(ra[0], ra[1]);
            u = (float)(ra[2] >> 8) * 5.9604644775390625e-08f;
            z2 = ld_normal(rb[0], rb[1]);
        } else {
            z1 = p.z1[i]; u = p.u[i]; z2 = p.z2[i];
        }
        float v = 2.f * gcx * ((float)p.cw[i] - 0.5f) + z1;
        if (sb >= 1e-20f && u < p.rho) v += gcx * sb * z2;
        ys[n] = v;
        p.y[i] = v;
    }
    __syncthreads();
    T* node = static_cast<T*>(p.node) + b * 2 * nvar;                       // [2][nvar]
    for (int o = tid; o < 2 * nvar; o += LD_THREADS) fgnn_st(node + o, o < nvar ? ys[o] : snr);
    T* hop = static_cast<T*>(p.hop) + b * dc * nchk;                        // [dc][nchk] = hop^T
    for (int o = tid; o < dc * nchk; o += LD_THREADS) {
        const int j = o / nchk, f = o - j * nchk;
        fgnn_st(hop + o, ys[p.factor_to_vars[f * dc + j]]);
    }
    T* e1 = static_cast<T*>(p.ef_f2v) + b * (int64_t)(dc + 1) * nvar * dv;  // [dc+1][nvar][dv]
    for (int o = tid; o < (dc + 1) * nvar * dv; o += LD_THREADS) {
        const int c = o / (nvar * dv), r = o - c * nvar * dv, n = r / dv, j = r - n * dv;
        const float v = c < dc ? ys[p.factor_to_vars[p.var_to_factors[n * dv + j] * dc + c]] : ys[n];
        fgnn_st(e1 + o, v);
    }
    T* e2 = static_cast<T*>(p.ef_v2f) + b * (int64_t)(dc + 1) * nchk * dc;  // [dc+1][nchk][dc]
    for (int o = tid; o < (dc + 1) * nchk * dc; o += LD_THREADS) {
        const int c = o / (nchk * dc), r = o - c * nchk * dc, f = r / dc, j = r - f * dc;
        fgnn_st(e2 + o, ys[p.factor_to_vars[f * dc + (c < dc ? c : j)]]);
    }
}

// cw [B][K+P] (bytes 0/1) = [s | G s]; s [B][K] bytes, gmask [P] 64-bit rows of G over the K <= 64 message bits.
extern "C" int fgnn_ldpc_encode(const uint8_t* s, const uint64_t* gmask, int64_t B, int K, int P, uint8_t* cw,
                                fgnn_stream_t stream) {
    if (B < 0 || K < 1 || K > 64 || P < 0 || P > 64) FGNN_FAIL(FGNN_EUNSUPPORTED, "ldpc_encode: K=%d P=%d (each <= 64)", K, P);
    if (B == 0) return FGNN_OK;
    if (!s || !gmask || !cw) FGNN_FAIL(FGNN_EINVAL, "ldpc_encode: null pointer");
    const int64_t grid = (B + LD_CW_PER_WG - 1) / LD_CW_PER_WG;
    fgnn_note_kernel("ldpc_encode_kernel");
    hipLaunchKernelGGL(ldpc_encode_kernel, dim3((unsigned)grid), dim3(LD_THREADS), 0, (hipStream_t)stream, s,
                       (const unsigned long long*)gmask, B, K, P, cw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "ldpc_encode launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// Received words and the model's inputs for B codewords of a (nvar, nchk) code with dv checks per variable and dc
// variables per check: y [B][nvar] f32; node [B][2][nvar], hop [B][dc][nchk], ef_f2v [B][dc+1][nvar][dv],
// ef_v2f [B][dc+1][nchk][dc] in `dtype` (FGNN_F32 / FGNN_BF16).
static int ld_channel_launch(bool rng, const uint8_t* cw, const float* snr_db, const float* sigma_b, float rho, const float* z1,
                             const float* u, const float* z2, uint64_t seed, uint64_t offset, const int32_t* var_to_factors,
                             const int32_t* factor_to_vars, int64_t B, int nvar, int nchk, int dv, int dc, int dtype, float* y,
                             void* node, void* hop, void* ef_f2v, void* ef_v2f, fgnn_stream_t stream) {
    if (B < 0 || nvar < 1 || nvar > 1024 || nchk < 1 || dv < 1 || dc < 1 || (dtype != FGNN_F32 && dtype != FGNN_BF16))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "ldpc_channel_features: nvar=%d (<= 1024) nchk=%d dv=%d dc=%d dtype=%d", nvar, nchk, dv,
                  dc, dtype);
    if (B == 0) return FGNN_OK;
    if (!cw || !snr_db || !sigma_b || (!rng && (!z1 || !u || !z2)) || !var_to_factors || !factor_to_vars || !y || !node || !hop ||
        !ef_f2v || !ef_v2f)
        FGNN_FAIL(FGNN_EINVAL, "ldpc_channel_features: null pointer");
    LdFeatParams p = {cw, snr_db, sigma_b, z1, u, z2, var_to_factors, factor_to_vars, y, node, hop, ef_f2v, ef_v2f,
                      rho, nvar, nchk, dv, dc, seed, offset};
    fgnn_note_kernel(rng ? "ldpc_features_kernel<rng>" : "ldpc_features_kernel");
    const dim3 grid((unsigned)B), block(LD_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FGNN_F32) {
        if (rng) hipLaunchKernelGGL((ldpc_features_kernel<float, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((ldpc_features_kernel<float, false>), grid, block, 0, st, p);
    } else {
        if (rng) hipLaunchKernelGGL((ldpc_features_kernel<bf16_t, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((ldpc_features_kernel<bf16_t, false>), grid, block, 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "ldpc_channel_features launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

extern "C" int fgnn_ldpc_channel_features(const uint8_t* cw, const float* snr_db, const float* sigma_b, float rho,
                                          const float* z1, const float* u, const float* z2,
                                          const int32_t* var_to_factors, const int32_t* factor_to_vars, int64_t B,
                                          int nvar, int nchk, int dv, int dc, int dtype, float* y, void* node,
                                          void* hop, void* ef_f2v, void* ef_v2f, fgnn_stream_t stream) {
    return ld_channel_launch(false, cw, snr_db, sigma_b, rho, z1, u, z2, 0, 0, var_to_factors, factor_to_vars, B, nvar, nchk, dv,
                             dc, dtype, y, node, hop, ef_f2v, ef_v2f, stream);
}

// The same with the channel's random draws made in the kernel: Philox4x32-10 keyed by `seed`, counter = (index of the
// codeword bit in the batch, `offset`): z1 = Box-Muller of words 0, 1, u = word 2 / 2^32 (24 bits), z2 = Box-Muller of words
// 0, 1 of the block with the counter's top bit flipped.  A training loop passes its step number as `offset`.
extern "C" int fgnn_ldpc_channel_features_rng(const uint8_t* cw, const float* snr_db, const float* sigma_b, float rho,
                                              uint64_t seed, uint64_t offset, const int32_t* var_to_factors,
                                              const int32_t* factor_to_vars, int64_t B, int nvar, int nchk, int dv, int dc,
                                              int dtype, float* y, void* node, void* hop, void* ef_f2v, void* ef_v2f,
                                              fgnn_stream_t stream) {
    return ld_channel_launch(true, cw, snr_db, sigma_b, rho, nullptr, nullptr, nullptr, seed, offset, var_to_factors,
                             factor_to_vars, B, nvar, nchk, dv, dc, dtype, y, node, hop, ef_f2v, ef_v2f, stream);
}

// ---- the training loss behind the decoder (round 5) --------------------------------------------------------------------------------
// /root/reference/train_ldpc.py:222-227:  loss = BCEWithLogits(decoded bits, message bits) + w * MSE(predicted burst amplitude,
// 10^(sigma_b / 20)), both means.  Through torch that is ~12 five-microsecond launches forward and ~13 backward (casts, log-sigmoid,
// two reductions, pow, scalar arithmetic) alone on the GPU between the model's forward and its backward; here two short launches
// forward (per-workgroup sums in double, then the partials in order: a fixed order, bit-reproducible; ONE workgroup for all
// 196 608 logits of the benched size took 193 us, gpurun_out/r05t/timeline) and one backward.
#define LL_THREADS 1024

__device__ __forceinline__ float ll_bce(float x, float y) {       // torch's stable form: max(x,0) - x y + log(1 + exp(-|x|))
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

#define LL_MAXGRID 128

// stage 1: workgroup w sums its grid-stride share of the two terms (double, fixed order) -> part[w][2]
template <typename T>
__global__ __launch_bounds__(LL_THREADS) void ldpc_loss_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ label,
                                                                   const float* __restrict__ pred, const float* __restrict__ sigma_b,
                                                                   int64_t nlogit, int64_t B, double* __restrict__ part) {
    __shared__ double red[2 * LL_THREADS];
    const int tid = threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * LL_THREADS;
    double a = 0.0, m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * LL_THREADS + tid; i < nlogit; i += step) a += (double)ll_bce(fgnn_ld(logits + i), label[i]);
    for (int64_t b = (int64_t)blockIdx.x * LL_THREADS + tid; b < B; b += step) { const float d = pred[b] - powf(10.f, sigma_b[b] * 0.05f); m += (double)(d * d); }
    red[tid] = a; red[LL_THREADS + tid] = m;
    __syncthreads();
    for (int s = LL_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) { red[tid] += red[tid + s]; red[LL_THREADS + tid] += red[LL_THREADS + tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { part[2 * blockIdx.x] = red[0]; part[2 * blockIdx.x + 1] = red[LL_THREADS]; }
}

// stage 2: the partials in workgroup order
__global__ __launch_bounds__(64) void ldpc_loss_final_kernel(const double* __restrict__ part, int n, int64_t nlogit, int64_t B, float w,
                                                             float* __restrict__ loss) {
    if (threadIdx.x != 0) return;
    double a = 0.0, m = 0.0;
    for (int i = 0; i < n; ++i) { a += part[2 * i]; m += part[2 * i + 1]; }
    loss[0] = (float)(a / (double)nlogit + (double)w * m / (double)B);
}

template <typename T>
__global__ __launch_bounds__(256) void ldpc_loss_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ label,
                                                            const float* __restrict__ pred, const float* __restrict__ sigma_b,
                                                            const float* __restrict__ gloss, int64_t nlogit, int64_t B, float w,
                                                            T* __restrict__ glogits, float* __restrict__ gpred) {
    const float g = gloss[0];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nlogit) {
        const float x = fgnn_ld(logits + i);
        const float sg = 1.f / (1.f + expf(-x));
        fgnn_st(glogits + i, g * (sg - label[i]) / (float)nlogit);
    }
    if (i < B) gpred[i] = g * w * 2.f * (pred[i] - powf(10.f, sigma_b[i] * 0.05f)) / (float)B;
}

static int ll_check(const void* logits, const float* label, const float* pred, const float* sigma_b, int64_t B, int n, int dtype) {
    if (!logits || !label || !pred || !sigma_b) FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: null pointer");
    if (B < 1 || n < 1) FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: bad sizes B=%lld n=%d", (long long)B, n);
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: unknown dtype %d", dtype);
    return FGNN_OK;
}

extern "C" int64_t fgnn_ldpc_loss_workspace_bytes(void) { return (int64_t)LL_MAXGRID * 2 * sizeof(double); }

extern "C" int fgnn_ldpc_loss_forward(const void* logits, const float* label, const float* pred, const float* sigma_b, int64_t B,
                                      int n, int dtype, float mse_weight, float* loss, void* workspace, int64_t workspace_bytes,
                                      fgnn_stream_t stream) {
    int rc = ll_check(logits, label, pred, sigma_b, B, n, dtype);
    if (rc) return rc;
    if (!loss) FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: null pointer");
    if (!workspace || workspace_bytes < fgnn_ldpc_loss_workspace_bytes() || ((uintptr_t)workspace & 7))
        FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: workspace of fgnn_ldpc_loss_workspace_bytes() bytes needed");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nl = B * n;
    int grid = (int)((nl + 4 * LL_THREADS - 1) / (4 * LL_THREADS));      // ~4 logits per thread (196 608 logits at the benched size: 48 workgroups)
    if (grid > LL_MAXGRID) grid = LL_MAXGRID;
    if (grid < 1) grid = 1;
    double* part = (double*)workspace;
    if (dtype == FGNN_F32) hipLaunchKernelGGL(ldpc_loss_fwd_kernel<float>, dim3(grid), dim3(LL_THREADS), 0, st, (const float*)logits, label, pred, sigma_b, nl, B, part);
    else hipLaunchKernelGGL(ldpc_loss_fwd_kernel<bf16_t>, dim3(grid), dim3(LL_THREADS), 0, st, (const bf16_t*)logits, label, pred, sigma_b, nl, B, part);
    hipLaunchKernelGGL(ldpc_loss_final_kernel, dim3(1), dim3(64), 0, st, part, grid, nl, B, mse_weight, loss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "ldpc_loss forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

extern "C" int fgnn_ldpc_loss_backward(const void* logits, const float* label, const float* pred, const float* sigma_b,
                                       const float* gloss, int64_t B, int n, int dtype, float mse_weight, void* glogits, float* gpred,
                                       fgnn_stream_t stream) {
    int rc = ll_check(logits, label, pred, sigma_b, B, n, dtype);
    if (rc) return rc;
    if (!gloss || !glogits || !gpred) FGNN_FAIL(FGNN_EINVAL, "ldpc_loss: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nl = B * n;
    const int grid = (int)((nl + 255) / 256);
    if (dtype == FGNN_F32) hipLaunchKernelGGL(ldpc_loss_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)logits, label, pred, sigma_b, gloss, nl, B, mse_weight, (float*)glogits, gpred);
    else hipLaunchKernelGGL(ldpc_loss_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)logits, label, pred, sigma_b, gloss, nl, B, mse_weight, (bf16_t*)glogits, gpred);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "ldpc_loss backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
