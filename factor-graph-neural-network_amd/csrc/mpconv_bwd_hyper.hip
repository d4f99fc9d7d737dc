// mpconv_bwd_hyper.hip — backward of the VF/FV message operator for the two "hyper-edge" calls of a
// factor graph with one factor that touches every variable (the LDPC hyper-factor of
// /root/reference/train_ldpc.py:60-75: V->F with M = 1 destination of degree k = 96, F->V with N = 1 source
// feeding M = 96 destinations through k = 1 edges), single edge type, max aggregator, NO_EXTENSION, and a
// CONSTANT etype (getype == NULL).  Same maths as mpconv_bwd_res.hip (autograd through
// /root/reference/lib/model/mpnn/mp_nn.py:115-175) but neither call is GEMM-shaped once the routing is known:
//
//  fan-in  (M == 1):  z[o] = max_j et[j] * P[idx[j], o].  Only ONE neighbour per output channel carries
//      gradient, so dP has nou non-zeros out of N*nou:
//          c_o = et[j*_o] * gz[o],  n*_o = idx[j*_o]
//          dx[n, :]  = sum_{o : n*_o = n} c_o * W[:, o]          (all other rows are zero)
//          dW[:, o] += c_o * x[n*_o, :]
//      = 2 * nin * nou MACs per sample instead of the dense 3 * N * nin * nou.
//  fan-out (N == 1, k == 1):  z[m, o] = et[m] * P[0, o]:
//          dP[o] = sum_m et[m] * gz[m, o],  dx = W dP,  dW += x (x) dP
//      = one weighted column sum of gz plus two matrix-vector products.
//
// Schedule: ONE WAVE PER SAMPLE (8 waves per workgroup, samples interleaved over all waves of the grid);
// lane <-> input channel for the W-shaped work, W^T resident in LDS (row o contiguous over c: conflict-free),
// routing decisions broadcast with v_readlane / ballots, dW accumulated in registers over all samples of a
// wave, folded across the 8 waves through LDS in a fixed order (bit-reproducible) and written to the
// workgroup's slab of the caller's workspace (summed by the slab reduce of mpconv_bwd_res.hip).
// Both kernels are HBM-bound: fan-in writes the (mostly zero) gx, fan-out reads gz.
#include "fgnn_common.h"
#include <stdlib.h>

#define BH_THREADS 512
#define BH_WAVES 8

struct BhParams {
    fgnn_mpconv_desc d;
    const void* x;
    const int64_t* idx;
    const void* et;
    const float* W;
    const void* gz;
    const uint8_t* argmax;
    void* gx;
    float* ws;           // per-workgroup slabs [grid][nin*nou + nou]
    int has_bias;
};

extern __shared__ __attribute__((aligned(16))) float fgnn_lds_bh[];

void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);

__device__ __forceinline__ float bh_bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Stage W [nin][nou] (row-major) transposed into LDS: Wl[o * nin + c].
template <int NIN, int NOU>
__device__ __forceinline__ void bh_stage_w(const float* __restrict__ W, float* Wl, int tid) {
    for (int f = tid; f < NIN * NOU; f += BH_THREADS) {
        const int c = f / NOU, o = f - c * NOU;
        Wl[o * NIN + c] = W[f];
    }
}

// Fold the per-wave dW / dbias accumulators across the workgroup's waves (fixed order) and write the slab.
// dW[i][o] belongs to (c = lane + 64 i, o); gb_val is this lane's dbias for channel gb_ch (or gb_ch < 0).
template <int NI, int NO>
__device__ __forceinline__ void bh_flush(const BhParams& p, float* Wl, float* bl, float (&dW)[NI][64 * NO],
                                         const float (&gbv)[4 * NO], const int (&gbc)[4 * NO], int ngb, int tid,
                                         int lane, int wave) {
    constexpr int NIN = 64 * NI, NOU = 64 * NO;
    __syncthreads();                                   // every wave is done with W^T: the buffer becomes the fold
    for (int w = 0; w < BH_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int o = 0; o < NOU; ++o) {
                    float* q = Wl + o * NIN + lane + 64 * i;
                    *q = (w == 0) ? dW[i][o] : *q + dW[i][o];
                }
#pragma unroll
            for (int t = 0; t < 4 * NO; ++t)
                if (t < ngb && gbc[t] >= 0) bl[gbc[t]] = (w == 0) ? gbv[t] : bl[gbc[t]] + gbv[t];
        }
        __syncthreads();
    }
    float* slab = p.ws + (int64_t)blockIdx.x * (NIN * NOU + NOU);
    for (int f = tid; f < NIN * NOU; f += BH_THREADS) {
        const int c = f / NOU, o = f - c * NOU;
        slab[f] = Wl[o * NIN + c];
    }
    for (int f = tid; f < NOU; f += BH_THREADS) slab[NIN * NOU + f] = bl[f];
}

// ----------------------------------------------------------------------------------------
// fan-in: M == 1, net == 1
// ----------------------------------------------------------------------------------------
template <typename T, int NI, int NO>
__global__ __launch_bounds__(BH_THREADS) void mpconv_bwd_fanin_kernel(const BhParams p) {
    constexpr int NIN = 64 * NI, NOU = 64 * NO;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Wl = fgnn_lds_bh;                           // [NOU][NIN]
    float* bl = Wl + NIN * NOU;                        // [NOU]
    bh_stage_w<NIN, NOU>(p.W, Wl, tid);
    __syncthreads();

    const T* X = reinterpret_cast<const T*>(p.x);
    const T* ET = reinterpret_cast<const T*>(p.et);
    const T* GZ = reinterpret_cast<const T*>(p.gz);
    T* GX = reinterpret_cast<T*>(p.gx);
    const int N = d.N, k = d.k;

    float dW[NI][NOU];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int o = 0; o < NOU; ++o) dW[i][o] = 0.f;
    float gbv[4 * NO];
    int gbc[4 * NO];
#pragma unroll
    for (int t = 0; t < 4 * NO; ++t) { gbv[t] = 0.f; gbc[t] = -1; }
#pragma unroll
    for (int q = 0; q < NO; ++q) gbc[q] = lane + 64 * q;

    const int nwaves = gridDim.x * BH_WAVES;
    for (int b = blockIdx.x * BH_WAVES + wave; b < d.B; b += nwaves) {
        // lane <-> output channel: winner slot, its source node and the routed coefficient
        int ns[NO];
        float cf[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            const int64_t yo = (int64_t)b * d.y_sb + (int64_t)(lane + 64 * q) * d.y_sc;
            const float g = fgnn_ld(GZ + yo);
            int js = p.argmax[yo];
            js = js < k ? js : k - 1;
            int n = (int)p.idx[(int64_t)b * d.idx_sb + (int64_t)js * d.idx_sk];
            n = n < 0 ? 0 : (n >= N ? N - 1 : n);
            ns[q] = n;
            cf[q] = g * fgnn_ld(ET + (int64_t)b * d.et_sb + (int64_t)js * d.et_sk);
            gbv[q] += g;
        }
        const T* xb = X + (int64_t)b * d.x_sb;
        // dW[:, o] += c_o * x[n*_o, :]      (lane <-> input channel; one coalesced row read per o)
#pragma unroll
        for (int q = 0; q < NO; ++q)
#pragma unroll
            for (int ol = 0; ol < 64; ++ol) {
                const int n = __builtin_amdgcn_readlane(ns[q], ol);
                const float s = bh_bcast(cf[q], ol);
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    dW[i][ol + 64 * q] =
                        fmaf(s, fgnn_ld(xb + (int64_t)n * d.x_sn + (int64_t)(lane + 64 * i) * d.x_sc), dW[i][ol + 64 * q]);
            }
        // dx rows: row n collects the channels routed to it; untouched rows are written as zeros
        T* gxb = GX + (int64_t)b * d.x_sb;
        for (int n = 0; n < N; ++n) {
            float acc[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = 0.f;
#pragma unroll
            for (int q = 0; q < NO; ++q) {
                unsigned long long mask = __ballot(ns[q] == n);
                while (mask) {
                    const int ol = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float s = bh_bcast(cf[q], ol);
                    const float* wr = Wl + (ol + 64 * q) * NIN + lane;
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i] = fmaf(s, wr[64 * i], acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
                fgnn_st(gxb + (int64_t)n * d.x_sn + (int64_t)(lane + 64 * i) * d.x_sc, acc[i]);
        }
    }
    bh_flush<NI, NO>(p, Wl, bl, dW, gbv, gbc, NO, tid, lane, wave);
}

// ----------------------------------------------------------------------------------------
// fan-out: N == 1, k == 1, net == 1; gz channel-fastest [M][nou]
// ----------------------------------------------------------------------------------------
template <typename T, int NI, int NO>
__global__ __launch_bounds__(BH_THREADS) void mpconv_bwd_fanout_kernel(const BhParams p) {
    constexpr int NIN = 64 * NI, NOU = 64 * NO, CH = 4 * NO;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Wl = fgnn_lds_bh;
    float* bl = Wl + NIN * NOU;
    bh_stage_w<NIN, NOU>(p.W, Wl, tid);
    __syncthreads();

    const T* X = reinterpret_cast<const T*>(p.x);
    const T* ET = reinterpret_cast<const T*>(p.et);
    const T* GZ = reinterpret_cast<const T*>(p.gz);
    T* GX = reinterpret_cast<T*>(p.gx);
    const int M = d.M;
    const int ch0 = (lane & 15) * CH, rc = lane >> 4;  // this lane's channels and row class of the gz sweep

    float dW[NI][NOU];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int o = 0; o < NOU; ++o) dW[i][o] = 0.f;
    float gbv[CH];
    int gbc[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) { gbv[t] = 0.f; gbc[t] = rc == 0 ? ch0 + t : -1; }

    const int nwaves = gridDim.x * BH_WAVES;
    for (int b = blockIdx.x * BH_WAVES + wave; b < d.B; b += nwaves) {
        float xv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) xv[i] = fgnn_ld(X + (int64_t)b * d.x_sb + (int64_t)(lane + 64 * i) * d.x_sc);
        // dP[o] = sum_m et[m] gz[m, o]; dbias[o] += sum_m gz[m, o].  16 lanes x CH channels per row, 4 rows per sweep
        float dp[CH], bs[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) { dp[t] = 0.f; bs[t] = 0.f; }
        const T* gzb = GZ + (int64_t)b * d.y_sb + ch0;
        const T* etb = ET + (int64_t)b * d.et_sb;
        for (int r = rc; r < M; r += 4) {
            const float e = fgnn_ld(etb + (int64_t)r * d.et_sm);
#pragma unroll
            for (int h = 0; h < NO; ++h) {
                const f32x4 v = fgnn_ld4(gzb + (int64_t)r * d.y_sm + 4 * h);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    dp[4 * h + u] = fmaf(e, v[u], dp[4 * h + u]);
                    bs[4 * h + u] += v[u];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            dp[t] += __shfl_xor(dp[t], 16);
            dp[t] += __shfl_xor(dp[t], 32);
            bs[t] += __shfl_xor(bs[t], 16);
            bs[t] += __shfl_xor(bs[t], 32);
            gbv[t] += bs[t];
        }
        // dx = W dP, dW += x (x) dP   (lane <-> input channel; dP[o] lives in lane o / CH, register o % CH)
        float acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = 0.f;
#pragma unroll
        for (int o = 0; o < NOU; ++o) {
            const float s = bh_bcast(dp[o % CH], o / CH);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i] = fmaf(s, Wl[o * NIN + lane + 64 * i], acc[i]);
                dW[i][o] = fmaf(s, xv[i], dW[i][o]);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) fgnn_st(GX + (int64_t)b * d.x_sb + (int64_t)(lane + 64 * i) * d.x_sc, acc[i]);
    }
    bh_flush<NI, NO>(p, Wl, bl, dW, gbv, gbc, CH, tid, lane, wave);
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define BH_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] hyper-edge backward rejects shape: rule %d\n", code); return 0; } while (0)

template <typename T>
static void* bh_pick(bool fanin, int NI, int NO) {
#define BH_CASE(ni, no) if (NI == ni && NO == no) return fanin ? (void*)mpconv_bwd_fanin_kernel<T, ni, no> : (void*)mpconv_bwd_fanout_kernel<T, ni, no>;
    BH_CASE(1, 1) BH_CASE(1, 2) BH_CASE(2, 1)
#undef BH_CASE
    return nullptr;
}

// Returns 1 if launched, 0 if the call is not a constant-etype hyper-edge call, <0 on error.
int fgnn_mpconv_backward_hyper(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                               const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                               float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                               fgnn_stream_t stream) {
    if (getype) BH_REJECT(1);                                    // edge-weight gradient: the resident kernel has it
    if (d->ext != FGNN_EXT_NONE || d->agg != FGNN_AGG_MAX || d->net != 1) BH_REJECT(2);
    if (d->nin % 64 || d->nou % 64 || d->nin > 128 || d->nou > 128 || d->nin * d->nou > 64 * 128) BH_REJECT(3);
    const bool fanin = d->M == 1 && d->k >= 1 && d->k <= 256 && d->N >= 1;
    const bool fanout = !fanin && d->N == 1 && d->k == 1;
    if (!fanin && !fanout) BH_REJECT(4);
    if (fanout && !(d->y_sc == 1 && d->y_sm % 4 == 0 && d->y_sb % 4 == 0 && d->y_sm >= d->nou)) BH_REJECT(5);
    const int64_t nw = (int64_t)d->nin * d->nou, slab_len = nw + d->nou;
    if (!workspace || workspace_bytes < 256 * slab_len * 4) BH_REJECT(6);
    const int NI = d->nin / 64, NO = d->nou / 64;
    void* fn = d->dtype == FGNN_F32 ? bh_pick<float>(fanin, NI, NO) : bh_pick<bf16_t>(fanin, NI, NO);
    if (!fn) BH_REJECT(7);
    const int lds = (int)(slab_len * 4);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int grid = (d->B + BH_WAVES - 1) / BH_WAVES;
    if (grid > 256) grid = 256;
    BhParams p;
    p.d = *d;
    p.x = x; p.idx = nn_idx; p.et = etype; p.W = filters; p.gz = gz; p.argmax = argmax; p.gx = gx;
    p.ws = (float*)workspace; p.has_bias = gbias != nullptr;
    fgnn_note_kernel("mpconv_bwd_%s_kernel<%s, %d, %d>", fanin ? "fanin" : "fanout", d->dtype ? "bf16_t" : "float", NI, NO);
    void* args[] = {(void*)&p};
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BH_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv hyper-edge backward launch: %s", hipGetErrorString(e));
    fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
