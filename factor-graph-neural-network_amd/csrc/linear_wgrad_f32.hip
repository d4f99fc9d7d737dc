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


// ---------------------------------------------------------------------------------------------------------------------------
// Piece form (round 6): the same blocked product on the bf16 matrix cores.  The kernel above is bound by the f32 matrix pipe
// (64 v_mfma_f32_16x16x4_f32 = 2 048 cycles per wave and 32-row tile; a 64 x 64 map pads its 128 x 128 block to a quarter full):
// 29 us per map at the synthetic-PGM models' 61 440 rows, 38 maps per training step = 8 % of BASELINE config 5's step.  Here a
// tile's operands are cut into three bf16 pieces when they are committed to LDS (h + m + l: 2^-24, the arithmetic of
// mpconv_fwd_extp_kernel) and both operands — gy^T and x, contracted over the tile's 32 ROWS — come out of the row-major piece
// images through ds_read_b64_tr_b16 (256-byte rows, 32-byte segments swizzled by (r & 3) | (r >> 3 & 1) << 2: conflict-free
// transpose passes): one k-step of v_mfma_f32_16x16x32_bf16 per tile, 8 fragment products x 6 piece products = 768 cycles.
// Same blocking, slabs and fold; the bias gradient is summed from the f32 registers at commit time.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __bf16 wq_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wq_s16x4 __attribute__((ext_vector_type(4)));
#define WQ_NP 3
#define WQ_ROW 256                      // bytes per image row: 128 bf16
#define WQ_PIECE (WF_TR * WQ_ROW)       // one piece image of a 32-row tile

__device__ __forceinline__ int wq_sw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ unsigned wq_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ uint2 wq_tr(unsigned lds_addr) {
    typedef __attribute__((address_space(3))) wq_s16x4 lds_v4;
    const wq_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4*>(static_cast<uintptr_t>(lds_addr)));
    return __builtin_bit_cast(uint2, v);
}

// BO x BC = the output block (64 or 128 each: a 64-channel side takes a 64-wide block instead of padding a 128-wide one);
// waves = 4 (o) x 2 (c): BO / 64 o tiles and BC / 32 c tiles per wave
template <int BO, int BC>
__global__ __launch_bounds__(WF_THREADS, 2) void linear_wgrad_f32q_kernel(const WfParams p) {
    constexpr int GROW = 2 * BO, XROW = 2 * BC;                      // image row bytes
    constexpr int GPIECE = WF_TR * GROW, XPIECE = WF_TR * XROW;
    constexpr int NO = BO / 64, NC = BC / 32;                         // tiles per wave
    constexpr int GCH = BO / 4, XCH = BC / 4;                         // 16-byte chunks (4 floats) per tile row
    constexpr int GPT = WF_TR * GCH / WF_THREADS, XPT = WF_TR * XCH / WF_THREADS;     // chunks per thread and tile (1 or 2)
    constexpr int RG = WF_THREADS / GCH;                              // row groups of the gy staging (bias partials)
    __shared__ __attribute__((aligned(16))) unsigned char gq[WQ_NP * GPIECE];      // gy pieces [piece][32 rows][BO]
    __shared__ __attribute__((aligned(16))) unsigned char xq[WQ_NP * XPIECE];      // x pieces  [piece][32 rows][BC]
    __shared__ float bpart[RG * BO];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int bo = blockIdx.y / p.nblk_c, bc = blockIdx.y - bo * p.nblk_c;
    const int o0 = bo * BO, c0 = bc * BC;
    const int Cin = p.Cin, Cout = p.Cout;
    const int wo = wave & 3, wc = wave >> 2;              // o tiles NO wo ..; c tiles NC wc ..
    const unsigned gq0 = (unsigned)(uintptr_t)gq, xq0 = (unsigned)(uintptr_t)xq;
    // 32-byte segment swizzles (conflict-free transpose passes over rows {r..r+3, r+8..r+11}): 256-byte rows / 128-byte rows
    auto swz = [](int r, int rowbytes) { return rowbytes == 256 ? ((r & 3) | (((r >> 3) & 1) << 2)) : (((r >> 1) & 1) | (((r >> 3) & 1) << 1)); };

    const int r_begin = blockIdx.x * p.rows_per_chunk;
    const int r_end = min(p.R, r_begin + p.rows_per_chunk);

    uint4 pg[GPT], px[XPT];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int q = tid + i * WF_THREADS, row = q / GCH, col = (q % GCH) * 4;
            const int r = r0 + row;
            pg[i] = (r < r_end && o0 + col < Cout) ? *reinterpret_cast<const uint4*>(p.gy + (int64_t)r * Cout + o0 + col) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int q = tid + i * WF_THREADS, row = q / XCH, col = (q % XCH) * 4;
            const int r = r0 + row;
            px[i] = (r < r_end && c0 + col < Cin) ? *reinterpret_cast<const uint4*>(p.x + (int64_t)r * Cin + c0 + col) : make_uint4(0, 0, 0, 0);
        }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};                   // bias gradient: this thread's four gy columns, over its rows
    auto put = [&](unsigned char* img, int piece, int rowbytes, int row, int col, const uint4& v) {
        float a = __uint_as_float(v.x), b = __uint_as_float(v.y), c = __uint_as_float(v.z), d = __uint_as_float(v.w);
        unsigned char* dst = img + row * rowbytes + (((col >> 4) ^ swz(row, rowbytes)) << 5) + ((col & 15) << 1);
#pragma unroll
        for (int t = 0; t < WQ_NP; ++t) {
            const unsigned w0 = wq_pack2(a, b), w1 = wq_pack2(c, d);
            *reinterpret_cast<uint2*>(dst + t * piece) = make_uint2(w0, w1);
            if (t + 1 < WQ_NP) {
                a -= __uint_as_float(w0 << 16); b -= __uint_as_float(w0 & 0xffff0000u);
                c -= __uint_as_float(w1 << 16); d -= __uint_as_float(w1 & 0xffff0000u);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int q = tid + i * WF_THREADS, row = q / GCH, col = (q % GCH) * 4;
            bs[0] += __uint_as_float(pg[i].x); bs[1] += __uint_as_float(pg[i].y); bs[2] += __uint_as_float(pg[i].z); bs[3] += __uint_as_float(pg[i].w);
            put(gq, GPIECE, GROW, row, col, pg[i]);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int q = tid + i * WF_THREADS, row = q / XCH, col = (q % XCH) * 4;
            put(xq, XPIECE, XROW, row, col, px[i]);
        }
    };

    f32x4 acc[NO][NC];
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transpose-read addresses: lane (li, lk) names rows 8 lk + (li >> 2) (+ 4), four columns from 4 (li & 3) of a 16-column segment
    const int r0t = 8 * lk + (li >> 2), r1t = r0t + 4;
    const unsigned g0 = r0t * GROW + ((li & 3) << 3), g1 = r1t * GROW + ((li & 3) << 3);
    const unsigned x0 = r0t * XROW + ((li & 3) << 3), x1 = r1t * XROW + ((li & 3) << 3);
    const int sg0 = swz(r0t, GROW), sg1 = swz(r1t, GROW), sx0 = swz(r0t, XROW), sx1 = swz(r1t, XROW);
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // smallest piece products first

    if (r_begin < r_end) prefetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += WF_TR) {
        __syncthreads();                                  // the previous tile's operands have been read
        commit();
        if (r0 + WF_TR < r_end) prefetch(r0 + WF_TR);
        __syncthreads();
        wq_bf16x8 ga[NO][WQ_NP], xb[NC][WQ_NP];
#pragma unroll
        for (int a = 0; a < NO; ++a)
#pragma unroll
            for (int t = 0; t < WQ_NP; ++t) {
                const int g = NO * wo + a;
                const uint2 lo = wq_tr(gq0 + t * GPIECE + g0 + ((g ^ sg0) << 5)), hi = wq_tr(gq0 + t * GPIECE + g1 + ((g ^ sg1) << 5));
                ga[a][t] = __builtin_bit_cast(wq_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
#pragma unroll
        for (int b = 0; b < NC; ++b)
#pragma unroll
            for (int t = 0; t < WQ_NP; ++t) {
                const int g = NC * wc + b;
                const uint2 lo = wq_tr(xq0 + t * XPIECE + x0 + ((g ^ sx0) << 5)), hi = wq_tr(xq0 + t * XPIECE + x1 + ((g ^ sx1) << 5));
                xb[b][t] = __builtin_bit_cast(wq_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
#pragma unroll
        for (int pr6 = 0; pr6 < 6; ++pr6)
#pragma unroll
            for (int a = 0; a < NO; ++a)
#pragma unroll
                for (int b = 0; b < NC; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[a][TA[pr6]], xb[b][TB[pr6]], acc[a][b], 0, 0, 0);
    }

    // D[i = o (4 lk + r)][j = c (li)] of tile (ot, ct) -> this chunk's slab, in gW's own layout
    float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)Cout * Cin + Cout);
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            const int c = c0 + (NC * wc + b) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + (NO * wo + a) * 16 + 4 * lk + r;
                if (o < Cout && c < Cin) slab[(int64_t)o * Cin + c] = acc[a][b][r];
            }
        }
    if (p.want_bias && bc == 0) {                         // the row groups' column sums, folded in a fixed order
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) bpart[(tid / GCH) * BO + 4 * (tid % GCH) + e] = bs[e];
        __syncthreads();
        if (tid < BO && o0 + tid < Cout) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) sum += bpart[g * BO + tid];
            slab[(int64_t)Cout * Cin + o0 + tid] = sum;
        }
    }
}

// fold of the per-chunk slabs: the fixed-order, LDS-staged slab reduction shared with the operator backward kernels
// (mpconv_bwd_res.hip: 16 elements x 16 slab groups per workgroup, every group walks its slabs in order)
void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias, hipStream_t st);

static void wf_plan(int64_t R, int Cin, int Cout, int* nrc, int* rows_per, int* nblk_o, int* nblk_c) {
    *nblk_o = (Cout + WF_BLK - 1) / WF_BLK;               // (a side of <= 64 channels is ONE block in both block sizes: the plan does not
    *nblk_c = (Cin + WF_BLK - 1) / WF_BLK;                //  depend on the form that runs)
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
    static const bool exact = getenv("FGNN_WGRAD_F32_EXACT") != nullptr;      // (A/B switch: the f32 matrix-core kernel)
    if (exact) hipLaunchKernelGGL(linear_wgrad_f32_kernel, dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    else if (Cout <= 64 && Cin <= 64) hipLaunchKernelGGL((linear_wgrad_f32q_kernel<64, 64>), dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    else if (Cout <= 64) hipLaunchKernelGGL((linear_wgrad_f32q_kernel<64, 128>), dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    else if (Cin <= 64) hipLaunchKernelGGL((linear_wgrad_f32q_kernel<128, 64>), dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((linear_wgrad_f32q_kernel<128, 128>), dim3(nrc, nbo * nbc), dim3(WF_THREADS), 0, (hipStream_t)stream, p);
    fgnn_launch_slab_reduce(p.ws, nrc, slab_len, nw, gW, gb, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad f32 launch: %s", hipGetErrorString(e));
    return 1;
}
