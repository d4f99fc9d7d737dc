// linear_wgrad.hip — weight / bias gradient of the node-wise (1x1) maps around the message operator
// (`Conv2d(cin, cout, 1)` in mp_conv_residual / iid_mapping*, reference
// /root/reference/lib/model/mpnn/mp_nn_residual.py:25-35, base_model.py:43-90):
//
//     gW[o][c] += sum_r gy[r][o] * x[r][c]        gb[o] += sum_r gy[r][o]          r over B*N rows
//
// a "tall-skinny" GEMM (K = B*N ~ 4e5, M,N <= 256).  rocBLAS runs it as one un-split tile loop
// (592 us for 64x64 at B*N = 393216, profiles/r01) although it only has to stream x and gy once
// (100 MB ~ 15 us).  Here the rows are split over ~1000 workgroups; each stages 32-row tiles of gy
// and x in LDS (row-major, as in memory) and accumulates its partial gW with exact-f32 MFMA; partials go
// to a workspace slab per workgroup and a second kernel sums them (no contended atomics).
#include "fgnn_common.h"

#define WG_THREADS 256
#define WG_WAVES 4
#define WG_ROWS 32
#define WG_TMAX 16       // 16x16 output tiles per wave

struct WgradParams {
    const void* x;       // [R][Cin]
    const void* gy;      // [R][Cout]
    float* ws;           // [grid.x * grid.y][oc * Cin_pad... see kernel
    int R, Cin, Cout, dtype;
    int rows_per_wg;
    int Cip, Cop;        // Cin / Cout-chunk padded to 16
    int oc;              // output channels handled per grid.y slice
    int XS, GS;          // LDS row strides (floats), == 16 (mod 32)
};

extern __shared__ __attribute__((aligned(16))) float wg_lds[];

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void linear_wgrad_kernel(const WgradParams p) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int o_base = blockIdx.y * p.oc;
    const int oc = min(p.oc, p.Cout - o_base);
    float* gs = wg_lds;                               // [WG_ROWS][GS]  gy tile (this slice's channels)
    float* xs = wg_lds + WG_ROWS * p.GS;              // [WG_ROWS][XS]  x tile
    const T* xg = static_cast<const T*>(p.x);
    const T* gg = static_cast<const T*>(p.gy);

    const int nct = p.Cip / 16, not_ = p.Cop / 16;    // tiles along Cin / along the channel slice
    const int ntiles = nct * not_;
    f32x4 acc[WG_TMAX];
#pragma unroll
    for (int t = 0; t < WG_TMAX; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // zero the padding columns once
    for (int f = tid; f < WG_ROWS * p.GS; f += WG_THREADS) gs[f] = 0.f;
    for (int f = tid; f < WG_ROWS * p.XS; f += WG_THREADS) xs[f] = 0.f;

    const int r_begin = blockIdx.x * p.rows_per_wg;
    const int r_end = min(p.R, r_begin + p.rows_per_wg);
    for (int r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
        __syncthreads();
        const int nr = min(WG_ROWS, r_end - r0);
        for (int f = tid; f < WG_ROWS * p.oc; f += WG_THREADS) {     // p.oc is a power of two dividing 256:
            const int r = f / p.oc, o = f - r * p.oc;                 // o == tid % p.oc for every f of a thread
            const float v = (r < nr && o < oc) ? fgnn_ld(gg + (int64_t)(r0 + r) * p.Cout + o_base + o) : 0.f;
            gs[r * p.GS + o] = v;
            bsum += v;
        }
        for (int f = tid; f < WG_ROWS * p.Cin; f += WG_THREADS) {
            const int r = f / p.Cin, c = f - r * p.Cin;
            xs[r * p.XS + c] = r < nr ? fgnn_ld(xg + (int64_t)(r0 + r) * p.Cin + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < WG_TMAX; ++t) {
            const int u = wave + WG_WAVES * t;
            if (u < ntiles) {
                const int ot = u / nct, ctile = u - ot * nct;
                const float* ap = gs + lk * p.GS + ot * 16 + li;       // A[i = o][k = row]
                const float* bp = xs + lk * p.XS + ctile * 16 + li;    // B[k = row][j = c]
                f32x4 a = acc[t];
#pragma unroll
                for (int kk = 0; kk < WG_ROWS / 4; ++kk)
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * p.GS], bp[kk * 4 * p.XS], a, 0, 0, 0);
                acc[t] = a;
            }
        }
    }
    // partial gW slab: [oc_pad16][Cip] then [oc] bias sums
    const int64_t slab_len = (int64_t)p.Cop * p.Cip + p.Cop;
    float* slab = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * slab_len;
#pragma unroll
    for (int t = 0; t < WG_TMAX; ++t) {
        const int u = wave + WG_WAVES * t;
        if (u < ntiles) {
            const int ot = u / nct, ctile = u - ot * nct;
#pragma unroll
            for (int r = 0; r < 4; ++r)              // D[i = o (4lk + r)][j = c (li)]
                slab[(int64_t)(ot * 16 + 4 * lk + r) * p.Cip + ctile * 16 + li] = acc[t][r];
        }
    }
    {
        __syncthreads();
        float* red = gs;
        if (tid < p.Cop) red[tid] = 0.f;
        __syncthreads();
        atomicAdd(&red[tid % p.oc], bsum);
        __syncthreads();
        if (tid < p.Cop) slab[(int64_t)p.Cop * p.Cip + tid] = tid < oc ? red[tid] : 0.f;
    }
}

// out[o][c] += sum_w slab[w][o][c]; gb[o] += sum_w slab[w][bias o]
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ ws, int nslab_x,
                                                                  int Cin, int Cout, int Cip, int Cop, int oc,
                                                                  float* __restrict__ gW, float* __restrict__ gb) {
    const int64_t slab_len = (int64_t)Cop * Cip + Cop;
    const int slice = blockIdx.y;                      // channel slice
    const int i = blockIdx.x * 256 + threadIdx.x;      // element of the slice's slab
    if (i >= slab_len) return;
    const float* base = ws + (int64_t)slice * nslab_x * slab_len + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 4 <= nslab_x; w += 4) {
        s0 += base[(int64_t)w * slab_len];
        s1 += base[(int64_t)(w + 1) * slab_len];
        s2 += base[(int64_t)(w + 2) * slab_len];
        s3 += base[(int64_t)(w + 3) * slab_len];
    }
    for (; w < nslab_x; ++w) s0 += base[(int64_t)w * slab_len];
    const float s = (s0 + s1) + (s2 + s3);
    if (i < (int64_t)Cop * Cip) {
        const int ol = i / Cip, c = i - ol * Cip, o = slice * oc + ol;
        if (ol < oc && o < Cout && c < Cin) gW[(int64_t)o * Cin + c] += s;
    } else if (gb) {
        const int ol = i - Cop * Cip, o = slice * oc + ol;
        if (ol < oc && o < Cout) gb[o] += s;
    }
}

static int wgrad_plan(int R, int Cin, int Cout, WgradParams* p, int* gx, int* gy_) {
    p->Cip = fgnn_round_up(Cin, 16);
    // channel slice: <= 64 output tiles per workgroup (4 waves x WG_TMAX), power-of-two channels so that
    // WG_THREADS % oc == 0 (per-thread bias column sums)
    int oc = 1;
    while (oc < Cout && oc < 256) oc *= 2;
    while (oc > 16 && (fgnn_round_up(oc, 16) / 16) * (p->Cip / 16) > WG_WAVES * WG_TMAX) oc /= 2;
    if ((fgnn_round_up(oc, 16) / 16) * (p->Cip / 16) > WG_WAVES * WG_TMAX) return -1;
    p->oc = oc;
    p->Cop = fgnn_round_up(oc, 16);
    p->XS = (p->Cip % 32 == 0) ? p->Cip + 16 : p->Cip;
    p->GS = (p->Cop % 32 == 0) ? p->Cop + 16 : p->Cop;
    *gy_ = (Cout + oc - 1) / oc;
    int g = 1024 / *gy_;
    if (g < 1) g = 1;
    int rows = (R + g - 1) / g;
    rows = fgnn_round_up(rows < WG_ROWS ? WG_ROWS : rows, WG_ROWS);
    p->rows_per_wg = rows;
    *gx = (R + rows - 1) / rows;
    return 0;
}

extern "C" int64_t fgnn_linear_wgrad_workspace_bytes(int64_t R, int Cin, int Cout) {
    WgradParams p;
    int gx, gy;
    if (R <= 0 || Cin <= 0 || Cout <= 0 || wgrad_plan((int)R, Cin, Cout, &p, &gx, &gy)) return -1;
    return (int64_t)gx * gy * ((int64_t)p.Cop * p.Cip + p.Cop) * 4;
}

// gW [Cout][Cin] f32 and gb [Cout] f32 (or NULL) are ACCUMULATED into.  x [R][Cin], gy [R][Cout] dense
// row-major, f32 or bf16.
extern "C" int fgnn_linear_wgrad(const void* x, const void* gy, int64_t R, int Cin, int Cout, int dtype,
                                 float* gW, float* gb, void* workspace, int64_t workspace_bytes,
                                 fgnn_stream_t stream) {
    if (!x || !gy || !gW || !workspace) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: null pointer");
    if (R <= 0 || R > 0x7fffffff || Cin <= 0 || Cout <= 0) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: bad sizes");
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: unknown dtype %d", dtype);
    WgradParams p;
    int gx, gyn;
    if (wgrad_plan((int)R, Cin, Cout, &p, &gx, &gyn))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_wgrad: Cin=%d too wide for the register tiling", Cin);
    const int64_t need = (int64_t)gx * gyn * ((int64_t)p.Cop * p.Cip + p.Cop) * 4;
    if (workspace_bytes < need) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: workspace %lld < %lld bytes",
                                          (long long)workspace_bytes, (long long)need);
    p.x = x; p.gy = gy; p.ws = (float*)workspace; p.R = (int)R; p.Cin = Cin; p.Cout = Cout; p.dtype = dtype;
    const int lds = WG_ROWS * (p.XS + p.GS) * 4;
    void* fn = dtype == FGNN_F32 ? (void*)linear_wgrad_kernel<float> : (void*)linear_wgrad_kernel<bf16_t>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(gx, gyn), dim3(WG_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad launch: %s", hipGetErrorString(e));
    const int64_t slab_len = (int64_t)p.Cop * p.Cip + p.Cop;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((slab_len + 255) / 256), gyn), dim3(256), 0,
                       (hipStream_t)stream, p.ws, gx, Cin, Cout, p.Cip, p.Cop, p.oc, gW, gb);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad reduce launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
