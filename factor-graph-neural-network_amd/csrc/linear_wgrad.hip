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
#include <stdlib.h>

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
    int nslab;           // slabs per channel slice (one per workgroup)
    unsigned gmagic, xmagic, omagic;   // ceil(2^32 / chunks-per-row) for gy / x rows; ceil(2^32 / oc)
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
    {   // fold the 256/oc per-thread partials of each bias column in a fixed order
        __syncthreads();
        float* red = gs;                              // WG_ROWS * GS >= WG_THREADS floats
        red[tid] = bsum;
        __syncthreads();
        if (tid < p.Cop) {
            float s = 0.f;
            if (tid < oc)
                for (int j = tid; j < WG_THREADS; j += p.oc) s += red[j];
            slab[(int64_t)p.Cop * p.Cip + tid] = s;
        }
    }
}

// Vectorised variant for 16-byte-aligned rows (Cin, Cout multiples of 8 for bf16 / 4 for f32): 64-row tiles,
// 16-byte global loads, and the NEXT tile is prefetched into registers while the current one feeds the
// matrix cores (issue-early / write-late), so HBM latency hides behind the MFMAs.
#define WV_ROWS 64
#define WV_MAXCH 10      // 16-byte chunks per thread per tile: 64 * (Cin + oc) * elem / 16 / 256 <= 10
#define WV_NSLAB 32

template <typename T, int TMAX, int MAXCH>
__global__ __launch_bounds__(WG_THREADS) void linear_wgrad_vec_kernel(const WgradParams p) {
    constexpr int EPC = 16 / sizeof(T);                // elements per 16-byte chunk (8 bf16 / 4 f32)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int o_base = blockIdx.y * p.oc;
    float* gs = wg_lds;                                // [WV_ROWS][GS]
    float* xs = wg_lds + WV_ROWS * p.GS;               // [WV_ROWS][XS]
    const T* xg = static_cast<const T*>(p.x);
    const T* gg = static_cast<const T*>(p.gy);
    const int nct = p.Cip / 16, not_ = p.Cop / 16;
    const int ntiles = nct * not_;
    f32x4 acc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int f = tid; f < WV_ROWS * p.GS; f += WG_THREADS) gs[f] = 0.f;
    for (int f = tid; f < WV_ROWS * p.XS; f += WG_THREADS) xs[f] = 0.f;

    // chunk space of a tile: [0, GCH) are gy chunks (row, o8), [GCH, GCH + XCH) are x chunks (row, c8);
    // (row, col) of a chunk by multiply-high with host-computed reciprocals (no integer division)
    const int gpr = p.oc / EPC, xpr = p.Cin / EPC;     // chunks per row
    const int GCH = WV_ROWS * gpr, XCH = WV_ROWS * xpr;
    const int r_begin = blockIdx.x * p.rows_per_wg;
    const int r_end = min(p.R, r_begin + p.rows_per_wg);
    uint4 pre[MAXCH];
    auto prefetch = [&](int r0, int t) {
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const unsigned f = t + q * WG_THREADS;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((int)f < GCH) {
                const unsigned r = gpr == 1 ? f : __umulhi(f, p.gmagic), c = f - r * gpr;
                if (r0 + (int)r < r_end) v = *reinterpret_cast<const uint4*>(gg + (int64_t)(r0 + r) * p.Cout + o_base + c * EPC);
            } else if ((int)f < GCH + XCH) {
                const unsigned g = f - GCH, r = xpr == 1 ? g : __umulhi(g, p.xmagic), c = g - r * xpr;
                if (r0 + (int)r < r_end) v = *reinterpret_cast<const uint4*>(xg + (int64_t)(r0 + r) * p.Cin + c * EPC);
            }
            pre[q] = v;
        }
    };
    auto unpack_store = [&](float* dst, const uint4& v) {
        if constexpr (sizeof(T) == 2) {
            f32x4 a = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                       __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
            f32x4 b = {__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u),
                       __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
            *reinterpret_cast<f32x4*>(dst) = a;
            *reinterpret_cast<f32x4*>(dst + 4) = b;
        } else {
            f32x4 a = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            *reinterpret_cast<f32x4*>(dst) = a;
        }
    };
    auto commit = [&](int t) {
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const unsigned f = t + q * WG_THREADS;
            if ((int)f < GCH) {
                const unsigned r = gpr == 1 ? f : __umulhi(f, p.gmagic), c = f - r * gpr;
                unpack_store(gs + r * p.GS + c * EPC, pre[q]);
            } else if ((int)f < GCH + XCH) {
                const unsigned g = f - GCH, r = xpr == 1 ? g : __umulhi(g, p.xmagic), c = g - r * xpr;
                unpack_store(xs + r * p.XS + c * EPC, pre[q]);
            }
        }
    };
    // bias column sums: thread t owns column t % oc and rows t / oc, t / oc + 256 / oc, ... of every tile
    const int bcol = tid & (p.oc - 1), brow0 = __umulhi((unsigned)tid, p.omagic), brstep = WG_THREADS / p.oc;
    if (r_begin < r_end) prefetch(r_begin, tid);
    for (int r0 = r_begin; r0 < r_end; r0 += WV_ROWS) {
        __syncthreads();                               // the previous tile's MFMAs are done with gs / xs
        int t = tid;
        asm volatile("" : "+v"(t));                    // keep the chunk address math out of long-lived registers
        commit(t);
        __syncthreads();
        if (r0 + WV_ROWS < r_end) prefetch(r0 + WV_ROWS, t);
        if (p.oc >= 1) {
            float sgy = 0.f;
            for (int r = (p.oc == 1 ? tid : brow0); r < WV_ROWS; r += brstep) sgy += gs[r * p.GS + bcol];
            bsum += sgy;
        }
        // wave's tiles u = wave + 4t: when nct == 4 they share the x column tile (ctile = wave): the B
        // fragment is read once per k-step and reused for every accumulator
        const int li_ = t & 15, lk_ = (t >> 4) & 3;
        if (nct == WG_WAVES) {
            const float* bp = xs + lk_ * p.XS + wave * 16 + li_;
            const float* ap = gs + lk_ * p.GS + li_;
#pragma unroll 4
            for (int kk = 0; kk < WV_ROWS / 4; ++kk) {
                const float bfr = bp[kk * 4 * p.XS];
#pragma unroll
                for (int tt = 0; tt < TMAX; ++tt)
                    if (tt < not_)
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * p.GS + tt * 16], bfr, acc[tt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < TMAX; ++tt) {
                const int u = wave + WG_WAVES * tt;
                if (u < ntiles) {
                    const int ot = u / nct, ctile = u - ot * nct;
                    const float* ap = gs + lk_ * p.GS + ot * 16 + li_;
                    const float* bp = xs + lk_ * p.XS + ctile * 16 + li_;
                    f32x4 a = acc[tt];
#pragma unroll
                    for (int kk = 0; kk < WV_ROWS / 4; ++kk)
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * p.GS], bp[kk * 4 * p.XS], a, 0, 0, 0);
                    acc[tt] = a;
                }
            }
        }
    }
    const int64_t slab_len = (int64_t)p.Cop * p.Cip + p.Cop;
    // one slab per workgroup, plain stores; the reduce kernel sums them 16 groups wide in a fixed order
    float* slab = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * slab_len;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int u = wave + WG_WAVES * t;
        if (u < ntiles) {
            const int ot = u / nct, ctile = u - ot * nct;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(int64_t)(ot * 16 + 4 * lk + r) * p.Cip + ctile * 16 + li] = acc[t][r];
        }
    }
    {   // fold the 256/oc per-thread partials of each bias column in a fixed order
        __syncthreads();
        float* red = gs;                              // WV_ROWS * GS >= WG_THREADS floats
        red[tid] = bsum;
        __syncthreads();
        if (tid < p.Cop) {
            float s = 0.f;
            if (tid < p.oc)
                for (int j = tid; j < WG_THREADS; j += p.oc) s += red[j];
            slab[(int64_t)p.Cop * p.Cip + tid] = s;
        }
    }
}

// out[o][c] += sum_w slab[w][o][c]; gb[o] += sum_w slab[w][bias o].  16 elements x 16 slab groups per workgroup:
// group g sums slabs g, g+16, ..., the 16 group sums fold through LDS in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ ws, int nslab_x,
                                                                  int Cin, int Cout, int Cip, int Cop, int oc,
                                                                  float* __restrict__ gW, float* __restrict__ gb) {
    __shared__ float part[16][17];
    const int64_t slab_len = (int64_t)Cop * Cip + Cop;
    const int slice = blockIdx.y;                      // channel slice
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + e;    // element of the slice's slab
    float s = 0.f;
    if (i < slab_len) {
        const float* base = ws + (int64_t)slice * nslab_x * slab_len + i;
        for (int w = g; w < nslab_x; w += 16) s += base[(int64_t)w * slab_len];
    }
    part[g][e] = s;
    __syncthreads();
    if (g != 0 || i >= slab_len) return;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][e];
    if (i < (int64_t)Cop * Cip) {
        const int ol = (int)(i / Cip), c = (int)(i - (int64_t)ol * Cip), o = slice * oc + ol;
        if (ol < oc && o < Cout && c < Cin) gW[(int64_t)o * Cin + c] += s;
    } else if (gb) {
        const int ol = (int)(i - (int64_t)Cop * Cip), o = slice * oc + ol;
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
    int g = 512 / *gy_;
    if (g < 1) g = 1;
    int rows = (R + g - 1) / g;
    rows = fgnn_round_up(rows < 64 ? 64 : rows, 64);      // multiple of both kernels' row tiles
    p->rows_per_wg = rows;
    *gx = (R + rows - 1) / rows;
    return 0;
}

// bf16, channel counts in multiples of 64: the LDS-free kernel of linear_wgrad_b16.hip
int64_t fgnn_linear_wgrad_b16_workspace_bytes(int64_t R, int Cin, int Cout);
int fgnn_linear_wgrad_b16(const void* x, const void* gy, int64_t R, int Cin, int Cout, float* gW, float* gb,
                          void* workspace, int64_t workspace_bytes, fgnn_stream_t stream);

// f32, rows >= 2048, channel counts in multiples of 4: the output-blocked kernel of linear_wgrad_f32.hip
int64_t fgnn_linear_wgrad_f32_workspace_bytes(int64_t R, int Cin, int Cout);
int fgnn_linear_wgrad_f32(const void* x, const void* gy, int64_t R, int Cin, int Cout, float* gW, float* gb, void* workspace,
                          int64_t workspace_bytes, fgnn_stream_t stream);

extern "C" int64_t fgnn_linear_wgrad_workspace_bytes(int64_t R, int Cin, int Cout) {
    WgradParams p;
    int gx, gy;
    if (R <= 0 || Cin <= 0 || Cout <= 0) return -1;
    const bool general = wgrad_plan((int)R, Cin, Cout, &p, &gx, &gy) == 0;
    if (!general && fgnn_linear_wgrad_f32_workspace_bytes(R, Cin, Cout) == 0) return -1;
    const int64_t a = general ? (int64_t)gx * gy * ((int64_t)p.Cop * p.Cip + p.Cop) * 4 : 0;
    const int64_t b = fgnn_linear_wgrad_b16_workspace_bytes(R, Cin, Cout);
    const int64_t c = fgnn_linear_wgrad_f32_workspace_bytes(R, Cin, Cout);
    const int64_t m = a > b ? a : b;
    return m > c ? m : c;
}

// gW [Cout][Cin] f32 and gb [Cout] f32 (or NULL) are ACCUMULATED into.  x [R][Cin], gy [R][Cout] dense
// row-major, f32 or bf16.
extern "C" int fgnn_linear_wgrad(const void* x, const void* gy, int64_t R, int Cin, int Cout, int dtype,
                                 float* gW, float* gb, void* workspace, int64_t workspace_bytes,
                                 fgnn_stream_t stream) {
    if (!x || !gy || !gW || !workspace) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: null pointer");
    if (R <= 0 || R > 0x7fffffff || Cin <= 0 || Cout <= 0) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: bad sizes");
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: unknown dtype %d", dtype);
    if (dtype == FGNN_BF16) {
        const int rc = fgnn_linear_wgrad_b16(x, gy, R, Cin, Cout, gW, gb, workspace, workspace_bytes, stream);
        if (rc != 0) return rc < 0 ? rc : FGNN_OK;
    }
    if (dtype == FGNN_F32) {
        const int rc = fgnn_linear_wgrad_f32(x, gy, R, Cin, Cout, gW, gb, workspace, workspace_bytes, stream);
        if (rc != 0) return rc < 0 ? rc : FGNN_OK;
    }
    WgradParams p;
    int gx, gyn;
    if (wgrad_plan((int)R, Cin, Cout, &p, &gx, &gyn))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_wgrad: Cin=%d too wide for the register tiling", Cin);
    const int64_t need = (int64_t)gx * gyn * ((int64_t)p.Cop * p.Cip + p.Cop) * 4;
    if (workspace_bytes < need) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad: workspace %lld < %lld bytes",
                                          (long long)workspace_bytes, (long long)need);
    p.x = x; p.gy = gy; p.ws = (float*)workspace; p.R = (int)R; p.Cin = Cin; p.Cout = Cout; p.dtype = dtype;
    const int epc = dtype == FGNN_F32 ? 4 : 8;
    const bool vec = Cin % epc == 0 && Cout % epc == 0 && p.oc % epc == 0 && Cout % p.oc == 0 &&
                     (WV_ROWS * (Cin + p.oc) / epc) <= WG_THREADS * WV_MAXCH &&
                     ((uintptr_t)x % 16 == 0) && ((uintptr_t)gy % 16 == 0);
    const int rows_tile = vec ? WV_ROWS : WG_ROWS;
    {
        const unsigned gpr = p.oc / epc > 0 ? p.oc / epc : 1, xpr = Cin / epc > 0 ? Cin / epc : 1;
        p.gmagic = gpr == 1 ? 0u : (unsigned)((0x100000000ULL + gpr - 1) / gpr);
        p.xmagic = xpr == 1 ? 0u : (unsigned)((0x100000000ULL + xpr - 1) / xpr);
        p.omagic = p.oc == 1 ? 0u : (unsigned)((0x100000000ULL + p.oc - 1) / p.oc);
    }
    const int lds = rows_tile * (p.XS + p.GS) * 4;
    void* fn;
    const int ntiles = (p.Cop / 16) * (p.Cip / 16);
    const int nchunks = (WV_ROWS * (Cin + p.oc) / epc + WG_THREADS - 1) / WG_THREADS;
    int nslab_x = gx;
    if (vec) {
        const int tm = (ntiles + WG_WAVES - 1) / WG_WAVES;
#define WV_PICK(T) (tm <= 4 && nchunks <= 4 ? (void*)linear_wgrad_vec_kernel<T, 4, 4> : \
                                              (void*)linear_wgrad_vec_kernel<T, 16, 10>)
        fn = dtype == FGNN_F32 ? WV_PICK(float) : WV_PICK(bf16_t);
#undef WV_PICK
        p.nslab = gx;                                  // one slab per workgroup: plain stores, no atomics
    } else {
        p.nslab = gx;
        fn = dtype == FGNN_F32 ? (void*)linear_wgrad_kernel<float> : (void*)linear_wgrad_kernel<bf16_t>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(gx, gyn), dim3(WG_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad launch: %s", hipGetErrorString(e));
    const int64_t slab_len = (int64_t)p.Cop * p.Cip + p.Cop;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((slab_len + 15) / 16), gyn), dim3(256), 0,
                       (hipStream_t)stream, p.ws, nslab_x, Cin, Cout, p.Cip, p.Cop, p.oc, gW, gb);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad reduce launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
