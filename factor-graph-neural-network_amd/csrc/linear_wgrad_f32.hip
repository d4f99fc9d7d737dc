// linear_wgrad_f32.hip — f32 weight / bias gradient of the node-wise (1x1) maps for the synthetic-PGM models (f32 storage,
// rows = B * nodes ~ 6e4, channel counts 64..512: /root/reference/lib/model/mpnn/factor_mpnn.py:40-70, mp_nn_residual.py:25-35):
//
//     gW[o][c] += sum_r gy[r][o] * x[r][c]        gb[o] += sum_r gy[r][o]
//
// The general kernel of linear_wgrad.hip stages 32-row tiles for 4 waves that own up to 16 output tiles each and runs the wide
// maps of these models at 0.4-0.8 TB/s (256 x 256: 315 us where the f32 matrix cores need 57 us and the bytes 25 us).  Here
// the OUTPUT is blocked (128 x 128 per workgroup, 8 waves x (2 x 4) tiles of 16 x 16 in registers), the rows are split over
// the x dimension of the grid, both operands of a 32-row tile sit in LDS row-major as in memory (row stride 144 floats: the four
// k-groups of a v_mfma_f32_16x16x4_f32 operand read four rows 16 banks apart, conflict-free), the next tile travels in
// registers.  Per-chunk partial blocks go to slabs that a second kernel folds in a fixed order (no atomics: bit-reproducible).
#include "fgnn_common.h"
#include <stdlib.h>

#define WF_THREADS 512
#define WF_TR 32             // rows per LDS tile (8 k-steps)
#define WF_BLK 128           // output block: 128 output channels x 128 input channels
#define WF_LS 144            // LDS row stride in floats (== 16 mod 64 banks)

struct WfParams {
    const float* x;          // [R][Cin]
    const float* gy;         // [R][Cout]
    float* ws;               // [nrc][Cout * Cin + Cout]
    int R, Cin, Cout;
    int rows_per_chunk, nblk_c;
    int want_bias;
};

__global__ __launch_bounds__(WF_THREADS, 2) void linear_wgrad_f32_kernel(const WfParams p) {
    __shared__ __attribute__((aligned(16))) float gs[WF_TR * WF_LS];
    __shared__ __attribute__((aligned(16))) float xs[WF_TR * WF_LS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int bo = blockIdx.y / p.nblk_c, bc = blockIdx.y - bo * p.nblk_c;
    const int o0 = bo * WF_BLK, c0 = bc * WF_BLK;
    const int Cin = p.Cin, Cout = p.Cout;
    const int wo = wave & 3, wc = wave >> 2;              // o tiles 2 wo, 2 wo + 1; c tiles 4 wc .. 4 wc + 3

    const int r_begin = blockIdx.x * p.rows_per_chunk;
    const int r_end = min(p.R, r_begin + p.rows_per_chunk);

    // staging: 2048 chunks of 4 floats per tile (1024 of gy, 1024 of x), four per thread
    uint4 pr[4];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = (tid + (i & 1) * WF_THREADS), row = q >> 5, col = (q & 31) * 4;
            const int r = r0 + row;
            pr[i] = make_uint4(0, 0, 0, 0);
            if (i < 2) { if (r < r_end && o0 + col < Cout) pr[i] = *reinterpret_cast<const uint4*>(p.gy + (int64_t)r * Cout + o0 + col); }
            else { if (r < r_end && c0 + col < Cin) pr[i] = *reinterpret_cast<const uint4*>(p.x + (int64_t)r * Cin + c0 + col); }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = (tid + (i & 1) * WF_THREADS), row = q >> 5, col = (q & 31) * 4;
            *reinterpret_cast<uint4*>((i < 2 ? gs : xs) + row * WF_LS + col) = pr[i];
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool bias_role = p.want_bias && bc == 0 && tid < WF_BLK;

    if (r_begin < r_end) prefetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += WF_TR) {
        __syncthreads();                                  // the previous tile's operands have been read
        commit();
        if (r0 + WF_TR < r_end) prefetch(r0 + WF_TR);
        __syncthreads();
        if (bias_role) {
#pragma unroll 8
            for (int rr = 0; rr < WF_TR; ++rr) bsum += gs[rr * WF_LS + tid];
        }
#pragma unroll
        for (int kk = 0; kk < WF_TR / 4; ++kk) {
            const float* ga = gs + (4 * kk + lk) * WF_LS + 32 * wo + li;
            const float* xb = xs + (4 * kk + lk) * WF_LS + 64 * wc + li;
            const float a0 = ga[0], a1 = ga[16];
            const float b0 = xb[0], b1 = xb[16], b2 = xb[32], b3 = xb[48];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2, acc[0][2], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b2, acc[1][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b3, acc[0][3], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b3, acc[1][3], 0, 0, 0);
        }
    }

    // D[i = o (4 lk + r)][j = c (li)] of tile (ot, ct) -> this chunk's slab, in gW's own layout
    float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)Cout * Cin + Cout);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = c0 + (4 * wc + b) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + (2 * wo + a) * 16 + 4 * lk + r;
                if (o < Cout && c < Cin) slab[(int64_t)o * Cin + c] = acc[a][b][r];
            }
        }
    if (bias_role && o0 + tid < Cout) slab[(int64_t)Cout * Cin + o0 + tid] = bsum;
}

// fold of the per-chunk slabs: the fixed-order, LDS-staged slab reduction shared with the operator backward kernels
// (mpconv_bwd_res.hip: 16 elements x 16 slab groups per workgroup, every group walks its slabs in order)
void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias, hipStream_t st);

static void wf_plan(int64_t R, int Cin, int Cout, int* nrc, int* rows_per, int* nblk_o, int* nblk_c) {
    *nblk_o = (Cout + WF_BLK - 1) / WF_BLK;
    *nblk_c = (Cin + WF_BLK - 1) / WF_BLK;
    const int nblk = *nblk_o * *nblk_c;
    int target = 256 / nblk;                              // one workgroup per CU: enough for the chip, and the slab traffic stays down
    if (target < 1) target = 1;
    int64_t rows = (R + target - 1) / target;
    if (rows < 2 * WF_TR) rows = 2 * WF_TR;
    rows = (rows + WF_TR - 1) / WF_TR * WF_TR;
    *rows_per = (int)rows;
    *nrc = (int)((R + rows - 1) / rows);
}

// 1 if the f32 blocked kernel takes this call
int fgnn_linear_wgrad_f32_accepts(int64_t R, int Cin, int Cout) {
    static const bool off = getenv("FGNN_NO_WGRAD_F32") != nullptr;
    // maps narrower than 64 x 64 stay with the general kernel (one block would be mostly padding)
    return !off && R >= 2048 && R <= 0x7fffffff && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 16 && Cout >= 16 && Cin <= 1024 && Cout <= 1024 &&
           (int64_t)Cin * Cout >= 4096;
}

int64_t fgnn_linear_wgrad_f32_workspace_bytes(int64_t R, int Cin, int Cout) {
    if (!fgnn_linear_wgrad_f32_accepts(R, Cin, Cout)) return 0;
    int nrc, rows, nbo, nbc;
    wf_plan(R, Cin, Cout, &nrc, &rows, &nbo, &nbc);
    return (int64_t)nrc * ((int64_t)Cout * Cin + Cout) * 4;
}

// Returns 1 if launched, 0 if outside this kernel's family, < 0 on error.
int fgnn_linear_wgrad_f32(const void* x, const void* gy, int64_t R, int Cin, int Cout, float* gW, float* gb, void* workspace,
                          int64_t workspace_bytes, fgnn_stream_t stream) {
    if (!fgnn_linear_wgrad_f32_accepts(R, Cin, Cout) || ((uintptr_t)x & 15) || ((uintptr_t)gy & 15)) return 0;
    int nrc, rows, nbo, nbc;
    wf_plan(R, Cin, Cout, &nrc, &rows, &nbo, &nbc);
    const int64_t nw = (int64_t)Cout * Cin, slab_len = nw + Cout;
    if (workspace_bytes < nrc * slab_len * 4) return 0;
    WfParams p;
    p.x = (const float*)x; p.gy = (const float*)gy; p.ws = (float*)workspace; p.R = (int)R; p.Cin = Cin; p.Cout = Cout;
    p.rows_per_chunk = rows; p.nblk_c = nbc; p.want_bias = gb != nullptr;
    hipLaunchKernelGGL(linear_wgrad_f32_kernel, dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    fgnn_launch_slab_reduce(p.ws, nrc, slab_len, nw, gW, gb, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad f32 launch: %s", hipGetErrorString(e));
    return 1;
}
