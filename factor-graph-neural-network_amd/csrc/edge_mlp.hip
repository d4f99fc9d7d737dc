// edge_mlp.hip — the edge-type MLP in front of the message operator,
//
//     etype[b, :, m, j] = W2 . ReLU( W1 . efeature[b, :, m, j] + b1 ) + b2        Cin (<= 8) -> 64 -> net (<= 4)
//
// (`emodel_f2v / emodel_v2f = Conv2d(7,64,1) + ReLU + Conv2d(64,4,1)`, /root/reference/train_ldpc.py:32-38,68-69).
// As three library ops the 64-channel hidden tensor (B x 288 edges x 64 = 151 MB in bf16 at B = 4096) is written,
// read, rectified, written and read again — and once more in the backward — although a row's whole MLP is
// 64 x (Cin + net) multiply-adds.  Here the hidden activations never exist in memory:
//   forward : lane <-> edge row; the row's Cin inputs sit in registers, the 64 hidden units are produced and
//             consumed one at a time (weights are wave-uniform -> scalar operands), 8 bytes written per row
//             (edge-type-fastest, the layout the operator kernels read);
//   backward: hidden units are RECOMPUTED from the inputs; a workgroup walks its rows once per block of 8 hidden
//             units with the 8 x (net + Cin + 1) gradient partials of those units in registers, folds them across
//             the workgroup and writes its slab; a second kernel sums the slabs in a fixed order.  The inputs
//             (edge features) need no gradient.
#include "fgnn_common.h"
#include <stdlib.h>

#define EM_THREADS 256
#define EM_BWD_THREADS 512
#define EM_HID 64
#define EM_UB 4          // hidden units per backward block (their 13 x 4 weights live in SGPRs)
#define EM_RPT 6         // rows per thread per forward iteration (amortises the weight broadcasts)
#define EM_MAXC 8
#define EM_MAXE 4
#define EM_REC 13        // per hidden unit: W1[u][0..7] | b1[u] | W2[0..3][u]  (zero padded)
#define EM_SLAB (EM_HID * EM_MAXC + EM_HID + EM_MAXE * EM_HID + EM_MAXE)   // dW1 | db1 | dW2 | db2
#define EM_LDS_ROWS 5120 // rows a backward workgroup stages (24 B each)

struct EmParams {
    const uint16_t* x;   // bf16, element (b, c, r) at b*x_sb + c*x_sc + r*x_sr (NCHW: Cin*E, E, 1)
    const uint16_t* gy;  // backward: gradient w.r.t. the output, element (b, e, r) at b*gy_sb + e*gy_se + r*gy_sr
    uint16_t* y;         // forward: [B][E][net] bf16 (edge-type-fastest)
    const float* W1;     // [64][Cin]
    const float* b1;     // [64]
    const float* W2;     // [net][64]
    const float* b2;     // [net]
    float* ws;           // backward: [grid][EM_SLAB]
    int64_t R;           // B * E rows
    int E, Cin, net;
    int64_t gy_sb, gy_se, gy_sr;
    int64_t x_sb, x_sc, x_sr;
    int rows_per_wg;
};

__device__ __forceinline__ float em_ld(const uint16_t* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ float em_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// lane u <- the weight record of hidden unit u
__device__ __forceinline__ void em_load_rec(const EmParams& p, int u, float (&w)[EM_REC]) {
#pragma unroll
    for (int c = 0; c < EM_MAXC; ++c) w[c] = c < p.Cin ? p.W1[u * p.Cin + c] : 0.f;
    w[8] = p.b1[u];
#pragma unroll
    for (int q = 0; q < EM_MAXE; ++q) w[9 + q] = q < p.net ? p.W2[q * EM_HID + u] : 0.f;
}

typedef float em_f2 __attribute__((ext_vector_type(2)));

// row r -> (sample, edge) with a 32-bit divide when the row count allows it (the 64-bit one is ~100 instructions)
__device__ __forceinline__ void em_split(const EmParams& p, int64_t r, int64_t& b, int& e) {
    if (p.R < (int64_t)0x7fffffff) {
        const unsigned q = (unsigned)r / (unsigned)p.E;
        b = q; e = (int)((unsigned)r - q * (unsigned)p.E);
    } else {
        b = r / p.E; e = (int)(r - b * p.E);
    }
}

template <int CIN>
__global__ __launch_bounds__(EM_THREADS) void edge_mlp_fwd_kernel(const EmParams p) {
    static_assert(EM_RPT % 2 == 0, "rows are processed as packed-f32 pairs");
    const int net = p.net;
    float wr[EM_REC];
    em_load_rec(p, threadIdx.x & 63, wr);
    float bias2[EM_MAXE];
#pragma unroll
    for (int q = 0; q < EM_MAXE; ++q) bias2[q] = q < net ? p.b2[q] : 0.f;
    const int64_t span = (int64_t)EM_THREADS * EM_RPT;
    for (int64_t base = (int64_t)blockIdx.x * span; base < p.R; base += (int64_t)gridDim.x * span) {
        em_f2 xv[EM_RPT / 2][CIN], out[EM_RPT / 2][EM_MAXE];
#pragma unroll
        for (int i = 0; i < EM_RPT; ++i) {
            const int64_t r = base + (int64_t)i * EM_THREADS + threadIdx.x;
            const bool ok = r < p.R;
            int64_t b; int e;
            em_split(p, ok ? r : 0, b, e);
            const uint16_t* xr = p.x + b * p.x_sb + (int64_t)e * p.x_sr;
#pragma unroll
            for (int c = 0; c < CIN; ++c) xv[i >> 1][c][i & 1] = (ok && c < p.Cin) ? em_ld(xr + (int64_t)c * p.x_sc) : 0.f;
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) out[i >> 1][q][i & 1] = bias2[q];
        }
#pragma unroll 2
        for (int u = 0; u < EM_HID; ++u) {
            float s[EM_REC];
#pragma unroll
            for (int j = 0; j < EM_REC; ++j) s[j] = em_bcast(wr[j], u);
#pragma unroll
            for (int i = 0; i < EM_RPT / 2; ++i) {
                em_f2 h = {s[8], s[8]};
#pragma unroll
                for (int c = 0; c < CIN; ++c) h = __builtin_elementwise_fma((em_f2){s[c], s[c]}, xv[i][c], h);
                h[0] = fmaxf(h[0], 0.f); h[1] = fmaxf(h[1], 0.f);
#pragma unroll
                for (int q = 0; q < EM_MAXE; ++q) out[i][q] = __builtin_elementwise_fma((em_f2){s[9 + q], s[9 + q]}, h, out[i][q]);
            }
        }
#pragma unroll
        for (int i = 0; i < EM_RPT; ++i) {
            const int64_t r = base + (int64_t)i * EM_THREADS + threadIdx.x;
            if (r >= p.R) continue;
            uint16_t o[EM_MAXE];
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) { const __bf16 hq = (__bf16)out[i >> 1][q][i & 1]; o[q] = __builtin_bit_cast(uint16_t, hq); }
            uint16_t* yr = p.y + r * net;
            if (net == 4) {
                uint2 v; v.x = o[0] | ((unsigned)o[1] << 16); v.y = o[2] | ((unsigned)o[3] << 16);
                *reinterpret_cast<uint2*>(yr) = v;
            } else {
#pragma unroll
                for (int q = 0; q < EM_MAXE; ++q) if (q < net) yr[q] = o[q];
            }
        }
    }
}

// sum of v over the wave, in every lane (a fixed tree: quad, row of 16 by mirrors, then the 4 rows in order)
__device__ __forceinline__ float em_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
    return (em_bcast(v, 0) + em_bcast(v, 16)) + (em_bcast(v, 32) + em_bcast(v, 48));
}

// Workgroup = 8 waves over one block of rows staged in LDS (bf16, 24 B a row).  The 16 blocks of 4 hidden units are
// dealt to the waves (2 each): a wave walks ALL the staged rows for its block, so its 52 partials fold inside the
// wave only and go straight to the slab — no cross-wave combine, one barrier in the whole kernel.
template <int CIN>
__global__ __launch_bounds__(EM_BWD_THREADS) void edge_mlp_bwd_kernel(const EmParams p) {
    extern __shared__ uint32_t em_rows[];                 // [rows][6]: x as 4 bf16 pairs, gy as 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_wg;
    const int nrows = (int)((r_begin + p.rows_per_wg < p.R ? r_begin + p.rows_per_wg : p.R) - r_begin);
    float* slab = p.ws + (int64_t)blockIdx.x * EM_SLAB;
    float wr[EM_REC];
    em_load_rec(p, lane, wr);
    constexpr int SB = 3;                                  // rows a thread has in flight while staging
    for (int i0 = tid; i0 < nrows; i0 += EM_BWD_THREADS * SB) {
        unsigned v[SB][12];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int i = i0 + k * EM_BWD_THREADS;
            const bool ok = i < nrows;
            int64_t b; int e;
            em_split(p, ok ? r_begin + i : r_begin, b, e);
            const uint16_t* xr = p.x + b * p.x_sb + (int64_t)e * p.x_sr;
            const uint16_t* gr = p.gy + b * p.gy_sb + (int64_t)e * p.gy_sr;
#pragma unroll
            for (int c = 0; c < EM_MAXC; ++c) v[k][c] = (ok && c < p.Cin) ? xr[(int64_t)c * p.x_sc] : 0u;
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) v[k][8 + q] = (ok && q < p.net) ? gr[(int64_t)q * p.gy_se] : 0u;
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int i = i0 + k * EM_BWD_THREADS;
            if (i < nrows) {
                uint32_t* dst = em_rows + i * 6;
#pragma unroll
                for (int j = 0; j < 6; ++j) dst[j] = v[k][2 * j] | (v[k][2 * j + 1] << 16);
            }
        }
    }
    __syncthreads();
    for (int u0 = wave * EM_UB; u0 < EM_HID; u0 += (EM_BWD_THREADS / 64) * EM_UB) {
        float s[EM_UB][EM_REC];
#pragma unroll
        for (int i = 0; i < EM_UB; ++i)
#pragma unroll
            for (int j = 0; j < EM_REC; ++j) s[i][j] = em_bcast(wr[j], u0 + i);
        float a2[EM_UB][EM_MAXE], a1[EM_UB][CIN], ab[EM_UB], sb2[EM_MAXE] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < EM_UB; ++i) {
            ab[i] = 0.f;
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) a2[i][q] = 0.f;
#pragma unroll
            for (int c = 0; c < CIN; ++c) a1[i][c] = 0.f;
        }
#pragma unroll 2
        for (int row = lane; row < nrows; row += 64) {
            const uint32_t* src = em_rows + row * 6;
            float xv[EM_MAXC], gv[EM_MAXE];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t w = src[j]; xv[2 * j] = __uint_as_float(w << 16); xv[2 * j + 1] = __uint_as_float(w & 0xffff0000u); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { const uint32_t w = src[4 + j]; gv[2 * j] = __uint_as_float(w << 16); gv[2 * j + 1] = __uint_as_float(w & 0xffff0000u); }
            if (u0 == 0) {
#pragma unroll
                for (int q = 0; q < EM_MAXE; ++q) sb2[q] += gv[q];
            }
#pragma unroll
            for (int i = 0; i < EM_UB; ++i) {
                float h = s[i][8];
#pragma unroll
                for (int c = 0; c < CIN; ++c) h = fmaf(s[i][c], xv[c], h);
                const bool on = h > 0.f;
                h = on ? h : 0.f;
                float gh = 0.f;
#pragma unroll
                for (int q = 0; q < EM_MAXE; ++q) { a2[i][q] = fmaf(gv[q], h, a2[i][q]); gh = fmaf(s[i][9 + q], gv[q], gh); }
                gh = on ? gh : 0.f;
                ab[i] += gh;
#pragma unroll
                for (int c = 0; c < CIN; ++c) a1[i][c] = fmaf(gh, xv[c], a1[i][c]);
            }
        }
        // fold over the lanes (a fixed DPP tree) and write this block's slice of the slab
#pragma unroll
        for (int i = 0; i < EM_UB; ++i) {
            const int u = u0 + i;
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) { const float v = em_wave_sum(a2[i][q]); if (lane == 0) slab[EM_HID * EM_MAXC + EM_HID + q * EM_HID + u] = v; }
#pragma unroll
            for (int c = 0; c < CIN; ++c) { const float v = em_wave_sum(a1[i][c]); if (lane == 0) slab[u * EM_MAXC + c] = v; }
            if (CIN < EM_MAXC && lane == 0) slab[u * EM_MAXC + EM_MAXC - 1] = 0.f;
            { const float v = em_wave_sum(ab[i]); if (lane == 0) slab[EM_HID * EM_MAXC + u] = v; }
        }
        if (u0 == 0) {
#pragma unroll
            for (int q = 0; q < EM_MAXE; ++q) { const float v = em_wave_sum(sb2[q]); if (lane == 0) slab[EM_HID * EM_MAXC + EM_HID + EM_MAXE * EM_HID + q] = v; }
        }
    }
}

// ----------------------------------------------------------------------------------------
// Round 5: the same MLP on the MATRIX cores (profiles/r04: the VALU kernels above ran at 0.06 / 0.03 of HBM — 57 / 97 us for 26 MB —
// at the serial head and tail of the training step).  A wave owns 16-row tiles; Cin (<= 8) and net (<= 4) are zero-padded to the
// K = 16 of v_mfma_f32_16x16x16_bf16.  Orientations are chosen so that every product's D fragment IS the next product's operand:
//   forward : H^T tile [hidden][row] = W1 X^T  ->  + b1, ReLU  ->  B operand (k = hidden) of  Y^T [q][row] += W2 H^T;
//   backward: H [row][hidden] = X W1^T and dH [row][hidden] = dY W2 (masked by H > 0): a lane then holds FOUR CONSECUTIVE ROWS of one
//             hidden unit = the k = row operand layout of  dW1^T [hidden][c] += dH^T X  (A) and  dW2 [q][hidden] += dY^T H  (B).
// Precision: x / gy are bf16 already; every f32 quantity that enters a product (W1, W2, H, dH) goes in as bf16 hi + bf16 lo (two
// MFMAs): 2^-17 relative, f32 accumulation — the tests' bounds (2^-8 of the output, 1e-4 of the gradients) are those of the VALU form.
// ----------------------------------------------------------------------------------------
typedef short em_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void em_split4(const float (&v)[4], em_s4& hi, em_s4& lo) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const __bf16 h = (__bf16)v[r];
        const __bf16 l = (__bf16)(v[r] - (float)h);
        hi[r] = __builtin_bit_cast(short, h);
        lo[r] = __builtin_bit_cast(short, l);
    }
}
#define EM_MMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
// Both tensors arrive PLANE-major per sample (x [B][Cin][E] as the reference's DataLoader collates it, the edge-type gradient
// [B][net][E] as the operator's backward writes it): a sample's block is one contiguous run, so a workgroup stages TWO samples
// (2 E / 16 row tiles over its four waves) with 16-byte loads, the next pair in flight in registers, and takes its MFMA operands from
// LDS: (row, four channels) = four 2-byte reads from four planes, (channel, four rows) = one 8-byte read.  Other layouts keep the
// VALU kernels (host dispatch).
#define EM_PAIR 2
#define EM_MAXCHUNK 4     // 16-byte chunks a thread stages per pair and tensor (2 samples x 8 planes x 512 rows x 2 B / 16 / 256 threads)
#define EM_MAXE_ROWS 512  // rows per sample the staging is sized for

template <bool BWD>
__device__ __forceinline__ void em_stage_load(const EmParams& p, int64_t pair, int tid, uint4 (&vx)[EM_MAXCHUNK], uint4 (&vg)[EM_MAXCHUNK]) {
    const int E = p.E;
    const int cx = p.Cin * E / 8, cg = p.net * E / 8;              // 16-byte chunks of one sample's x / gy block
#pragma unroll
    for (int k = 0; k < EM_MAXCHUNK; ++k) {
        const int i = tid + 256 * k;
        vx[k] = make_uint4(0, 0, 0, 0);
        if (i < EM_PAIR * cx) {
            const int smp = i / cx, ch = i - smp * cx;
            const int64_t b = pair * EM_PAIR + smp;
            if (b * E < p.R) vx[k] = reinterpret_cast<const uint4*>(p.x + b * p.x_sb)[ch];
        }
        if (BWD) {
            vg[k] = make_uint4(0, 0, 0, 0);
            if (i < EM_PAIR * cg) {
                const int smp = i / cg, ch = i - smp * cg;
                const int64_t b = pair * EM_PAIR + smp;
                if (b * E < p.R) vg[k] = reinterpret_cast<const uint4*>(p.gy + b * p.gy_sb)[ch];
            }
        }
    }
}
template <bool BWD>
__device__ __forceinline__ void em_stage_store(const EmParams& p, int tid, const uint4 (&vx)[EM_MAXCHUNK], const uint4 (&vg)[EM_MAXCHUNK],
                                               uint16_t* xs, uint16_t* gs) {
    const int E = p.E;
    const int cx = p.Cin * E / 8, cg = p.net * E / 8;
#pragma unroll
    for (int k = 0; k < EM_MAXCHUNK; ++k) {
        const int i = tid + 256 * k;
        if (i < EM_PAIR * cx) {
            const int smp = i / cx, ch = i - smp * cx;
            reinterpret_cast<uint4*>(xs + smp * (EM_MAXC * EM_MAXE_ROWS))[ch] = vx[k];       // planes keep their stride E inside the sample's slot
        }
        if (BWD && i < EM_PAIR * cg) {
            const int smp = i / cg, ch = i - smp * cg;
            reinterpret_cast<uint4*>(gs + smp * (EM_MAXE * EM_MAXE_ROWS))[ch] = vg[k];
        }
    }
}

__global__ __launch_bounds__(256) void edge_mlp_fwd_mfma_kernel(const EmParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t xs[EM_PAIR * EM_MAXC * EM_MAXE_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int Cin = p.Cin, net = p.net, E = p.E;
    // resident operands: W1 rows as A (i = hidden, k = channel), W2 rows as A (i = q, k = hidden); hi / lo halves
    em_s4 a1h[4], a1l[4], a2h[4], a2l[4];
    float bias1[4][4], bias2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float w1[4], w2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * kg + r;
            w1[r] = c < Cin ? p.W1[(16 * t + li) * Cin + c] : 0.f;
            w2[r] = li < net ? p.W2[li * EM_HID + 16 * t + 4 * kg + r] : 0.f;
            bias1[t][r] = p.b1[16 * t + 4 * kg + r];
        }
        em_split4(w1, a1h[t], a1l[t]);
        em_split4(w2, a2h[t], a2l[t]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias2[r] = r < net ? p.b2[r] : 0.f;
    const int64_t B = p.R / E, npair = (B + EM_PAIR - 1) / EM_PAIR;
    const int tps = E / 16;                                  // row tiles per sample
    uint4 vx[EM_MAXCHUNK], vg[EM_MAXCHUNK];
    if ((int64_t)blockIdx.x < npair) em_stage_load<false>(p, blockIdx.x, tid, vx, vg);
    for (int64_t pair = blockIdx.x; pair < npair; pair += gridDim.x) {
        __syncthreads();                                     // the previous pair's readers are done
        em_stage_store<false>(p, tid, vx, vg, xs, nullptr);
        __syncthreads();
        if (pair + gridDim.x < npair) em_stage_load<false>(p, pair + gridDim.x, tid, vx, vg);      // in flight under this pair's MFMAs
        for (int tt = wave; tt < EM_PAIR * tps; tt += 4) {
            const int smp = tt / tps, tl = tt - smp * tps;
            const int64_t b = pair * EM_PAIR + smp;
            if (b >= B) continue;                            // (wave-uniform)
            const uint16_t* xp = xs + smp * (EM_MAXC * EM_MAXE_ROWS) + tl * 16 + li;
            em_s4 bx;                                        // X^T as B: lane (row li, k-group kg) holds channels 4 kg .. 4 kg + 3
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int c = 4 * kg + r; bx[r] = c < Cin ? (short)xp[c * E] : (short)0; }
            f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 h = EM_MMA(a1h[t], bx, ((f32x4){0.f, 0.f, 0.f, 0.f}));
                h = EM_MMA(a1l[t], bx, h);                    // h[r] = hidden unit 16 t + 4 kg + r of row li
                float hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = fmaxf(h[r] + bias1[t][r], 0.f);
                em_s4 hh, hl;
                em_split4(hv, hh, hl);
                y = EM_MMA(a2h[t], hh, y);
                y = EM_MMA(a2h[t], hl, y);
                y = EM_MMA(a2l[t], hh, y);
            }
            if (kg == 0) {                                   // y[r] = output q = r of row li (k-group 0)
                uint16_t o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const __bf16 hq = (__bf16)(y[q] + bias2[q]); o[q] = __builtin_bit_cast(uint16_t, hq); }
                uint16_t* yr = p.y + (b * E + tl * 16 + li) * net;
                if (net == 4) *reinterpret_cast<uint2*>(yr) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (q < net) yr[q] = o[q];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void edge_mlp_bwd_mfma_kernel(const EmParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t xs[EM_PAIR * EM_MAXC * EM_MAXE_ROWS];
    __shared__ __attribute__((aligned(16))) uint16_t gs[EM_PAIR * EM_MAXE * EM_MAXE_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int Cin = p.Cin, net = p.net, E = p.E;
    // resident operands: W1^T as B (k = channel, j = hidden li), W2 as B (k = q, j = hidden li)
    em_s4 b1h[4], b1l[4], b2h[4], b2l[4];
    float bias1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float w1[4], w2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * kg + r;
            w1[r] = c < Cin ? p.W1[(16 * t + li) * Cin + c] : 0.f;
            w2[r] = c < net ? p.W2[c * EM_HID + 16 * t + li] : 0.f;
        }
        em_split4(w1, b1h[t], b1l[t]);
        em_split4(w2, b2h[t], b2l[t]);
        bias1[t] = p.b1[16 * t + li];
    }
    f32x4 dW1[4], dW2[4];
    float db1[4], db2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dW2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; db1[t] = 0.f; }
    const int64_t B = p.R / E, npair = (B + EM_PAIR - 1) / EM_PAIR;
    const int tps = E / 16;
    uint4 vx[EM_MAXCHUNK], vg[EM_MAXCHUNK];
    if ((int64_t)blockIdx.x < npair) em_stage_load<true>(p, blockIdx.x, tid, vx, vg);
    for (int64_t pair = blockIdx.x; pair < npair; pair += gridDim.x) {
        __syncthreads();
        em_stage_store<true>(p, tid, vx, vg, xs, gs);
        __syncthreads();
        if (pair + gridDim.x < npair) em_stage_load<true>(p, pair + gridDim.x, tid, vx, vg);
        for (int tt = wave; tt < EM_PAIR * tps; tt += 4) {
            const int smp = tt / tps, tl = tt - smp * tps;
            if (pair * EM_PAIR + smp >= B) continue;
            const uint16_t* xp = xs + smp * (EM_MAXC * EM_MAXE_ROWS) + tl * 16;
            const uint16_t* gp = gs + smp * (EM_MAXE * EM_MAXE_ROWS) + tl * 16;
            em_s4 ax, ay, bxr, ayt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                   // this lane's row li: channels 4 kg .. (A of H = X W1^T), net gradients (A of dH = dY W2)
                const int c = 4 * kg + r;
                ax[r] = c < Cin ? (short)xp[c * E + li] : (short)0;
                ay[r] = c < net ? (short)gp[c * E + li] : (short)0;
            }
            {   // rows 4 kg .. 4 kg + 3 of the tile: channel li of x (B of dW1^T = dH^T X), net gradient li (A of dW2 = dY^T H)
                const uint2 vxr = li < Cin ? *reinterpret_cast<const uint2*>(xp + li * E + 4 * kg) : make_uint2(0, 0);
                const uint2 vgr = li < net ? *reinterpret_cast<const uint2*>(gp + li * E + 4 * kg) : make_uint2(0, 0);
                bxr = __builtin_bit_cast(em_s4, vxr);
                ayt = __builtin_bit_cast(em_s4, vgr);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) db2 += __uint_as_float((unsigned)(uint16_t)ayt[r] << 16);      // (lanes li < net: q = li, rows of k-group kg)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 h = EM_MMA(ax, b1h[t], ((f32x4){0.f, 0.f, 0.f, 0.f}));
                h = EM_MMA(ax, b1l[t], h);                  // h[r] = hidden unit 16 t + li of row 4 kg + r
                f32x4 g = EM_MMA(ay, b2h[t], ((f32x4){0.f, 0.f, 0.f, 0.f}));
                g = EM_MMA(ay, b2l[t], g);                  // dH before the ReLU mask, same layout
                float hv[4], gv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = h[r] + bias1[t];
                    hv[r] = fmaxf(pre, 0.f);
                    gv[r] = pre > 0.f ? g[r] : 0.f;
                    db1[t] += gv[r];
                }
                em_s4 gh, gl, hh, hl;
                em_split4(gv, gh, gl);
                em_split4(hv, hh, hl);
                dW1[t] = EM_MMA(gh, bxr, dW1[t]);           // D[i = hidden 16 t + 4 kg + r][j = channel li]
                dW1[t] = EM_MMA(gl, bxr, dW1[t]);
                dW2[t] = EM_MMA(ayt, hh, dW2[t]);           // D[i = q 4 kg + r][j = hidden 16 t + li]
                dW2[t] = EM_MMA(ayt, hl, dW2[t]);
            }
        }
    }
    // this wave's sums -> its row of the fold (the staging buffer is free); the four waves in a fixed order -> the workgroup's slab
    __syncthreads();
    float* fold = reinterpret_cast<float*>(xs);
    float* mine = fold + wave * EM_SLAB;
    for (int f = lane; f < EM_SLAB; f += 64) mine[f] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float s1 = db1[t];
        s1 += __shfl_xor(s1, 16);
        s1 += __shfl_xor(s1, 32);
        if (kg == 0) mine[EM_HID * EM_MAXC + 16 * t + li] = s1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (li < EM_MAXC) mine[(16 * t + 4 * kg + r) * EM_MAXC + li] = dW1[t][r];
            if (kg == 0) mine[EM_HID * EM_MAXC + EM_HID + r * EM_HID + 16 * t + li] = dW2[t][r];
        }
    }
    {
        float s2 = db2;
        s2 += __shfl_xor(s2, 16);
        s2 += __shfl_xor(s2, 32);
        if (kg == 0 && li < EM_MAXE) mine[EM_HID * EM_MAXC + EM_HID + EM_MAXE * EM_HID + li] = s2;
    }
    __syncthreads();
    float* slab = p.ws + (int64_t)blockIdx.x * EM_SLAB;
    for (int f = tid; f < EM_SLAB; f += 256) slab[f] = (fold[f] + fold[EM_SLAB + f]) + (fold[2 * EM_SLAB + f] + fold[3 * EM_SLAB + f]);
}

// the layouts the MFMA kernels stage: plane-major sample blocks, 16-byte aligned, whole 16-row tiles per sample
static bool em_mfma_ok(const EmParams& p, bool bwd) {
    if (p.E % 16 || p.E > EM_MAXE_ROWS || p.R >= (int64_t)0x7fffffff) return false;
    if (p.x_sr != 1 || p.x_sc != p.E || p.x_sb % 8 || ((uintptr_t)p.x & 15) || p.x_sb < (int64_t)p.Cin * p.E) return false;
    if (EM_PAIR * p.Cin * p.E / 8 > 256 * EM_MAXCHUNK) return false;
    if (bwd && (p.gy_sr != 1 || p.gy_se != p.E || p.gy_sb % 8 || ((uintptr_t)p.gy & 15) || p.gy_sb < (int64_t)p.net * p.E)) return false;
    return true;
}

// out += sum over slabs, slab element i -> (dW1 | db1 | dW2 | db2) with the padded slab strides undone
__global__ __launch_bounds__(256) void edge_mlp_reduce_kernel(const float* __restrict__ ws, int nslab, int Cin, int net,
                                                              float* dW1, float* db1, float* dW2, float* db2) {
    __shared__ float part[16][17];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < EM_SLAB)
        for (int w = g; w < nslab; w += 16) s += ws[(int64_t)w * EM_SLAB + i];
    part[g][e] = s;
    __syncthreads();
    if (g != 0 || i >= EM_SLAB) return;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][e];
    if (i < EM_HID * EM_MAXC) {
        const int u = i / EM_MAXC, c = i - u * EM_MAXC;
        if (c < Cin && dW1) dW1[u * Cin + c] += t;
    } else if (i < EM_HID * EM_MAXC + EM_HID) {
        if (db1) db1[i - EM_HID * EM_MAXC] += t;
    } else if (i < EM_HID * EM_MAXC + EM_HID + EM_MAXE * EM_HID) {
        const int j = i - EM_HID * EM_MAXC - EM_HID, q = j / EM_HID, u = j - q * EM_HID;
        if (q < net && dW2) dW2[q * EM_HID + u] += t;
    } else {
        const int q = i - (EM_HID * EM_MAXC + EM_HID + EM_MAXE * EM_HID);
        if (q < net && db2) db2[q] += t;
    }
}

static int em_check(int64_t B, int E, int Cin, int net) {
    if (B < 0 || E < 1 || Cin < 1 || Cin > EM_MAXC || net < 1 || net > EM_MAXE)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "edge_mlp: Cin=%d (<= %d), net=%d (<= %d), hidden 64 only", Cin, EM_MAXC, net, EM_MAXE);
    return FGNN_OK;
}

// y[B][E][net] (bf16, edge-type-fastest) = W2 ReLU(W1 x + b1) + b2; x bf16 with element (b, c, r) at
// b*x_sb + c*x_sc + r*x_sr; f32 parameters, hidden 64.
extern "C" int fgnn_edge_mlp_forward(const void* x, int64_t x_sb, int64_t x_sc, int64_t x_sr, const float* W1, const float* b1, const float* W2, const float* b2,
                                     void* y, int64_t B, int E, int Cin, int net, fgnn_stream_t stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !y) FGNN_FAIL(FGNN_EINVAL, "edge_mlp_forward: null pointer");
    int rc = em_check(B, E, Cin, net);
    if (rc) return rc;
    if (B == 0) return FGNN_OK;
    EmParams p = {};
    p.x = (const uint16_t*)x; p.y = (uint16_t*)y; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2;
    p.R = B * E; p.E = E; p.Cin = Cin; p.net = net; p.x_sb = x_sb; p.x_sc = x_sc; p.x_sr = x_sr;
    if (em_mfma_ok(p, false)) {
        int64_t g = (B + EM_PAIR - 1) / EM_PAIR;
        if (g > 1024) g = 1024;
        fgnn_note_kernel("edge_mlp_fwd_mfma_kernel");
        hipLaunchKernelGGL(edge_mlp_fwd_mfma_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        int64_t g = (p.R + EM_THREADS * EM_RPT - 1) / (EM_THREADS * EM_RPT);
        if (g > 4096) g = 4096;
        fgnn_note_kernel("edge_mlp_fwd_kernel");
        if (Cin == 7) hipLaunchKernelGGL(edge_mlp_fwd_kernel<7>, dim3((unsigned)g), dim3(EM_THREADS), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(edge_mlp_fwd_kernel<8>, dim3((unsigned)g), dim3(EM_THREADS), 0, (hipStream_t)stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "edge_mlp_forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// one workgroup per CU while its rows fit the LDS staging buffer, more workgroups beyond that
static int64_t em_bwd_grid(int64_t R) {
    int64_t g = (R + EM_LDS_ROWS - 1) / EM_LDS_ROWS;
    if (g < 256) g = R < 256 * 64 ? (R + 63) / 64 : 256;
    if (g < 1) g = 1;
    return g;
}

extern "C" int64_t fgnn_edge_mlp_workspace_bytes(int64_t B, int E) {
    return em_bwd_grid(B * E) * EM_SLAB * 4;
}

// Parameter gradients (ACCUMULATED into gW1 [64][Cin], gb1 [64], gW2 [net][64], gb2 [net]; any may be NULL) from
// gy, the gradient w.r.t. the output, addressed as element (b, e, r) at b*gy_sb + e*gy_se + r*gy_sr (bf16).
extern "C" int fgnn_edge_mlp_backward(const void* x, int64_t x_sb, int64_t x_sc, int64_t x_sr, const void* gy, int64_t gy_sb, int64_t gy_se, int64_t gy_sr,
                                      const float* W1, const float* b1, const float* W2, int64_t B, int E, int Cin,
                                      int net, float* gW1, float* gb1, float* gW2, float* gb2, void* workspace,
                                      int64_t workspace_bytes, fgnn_stream_t stream) {
    if (!x || !gy || !W1 || !b1 || !W2 || !workspace) FGNN_FAIL(FGNN_EINVAL, "edge_mlp_backward: null pointer");
    int rc = em_check(B, E, Cin, net);
    if (rc) return rc;
    if (B == 0) return FGNN_OK;
    if (workspace_bytes < fgnn_edge_mlp_workspace_bytes(B, E)) FGNN_FAIL(FGNN_EINVAL, "edge_mlp_backward: workspace too small");
    EmParams p = {};
    p.x = (const uint16_t*)x; p.gy = (const uint16_t*)gy; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.ws = (float*)workspace;
    p.R = B * E; p.E = E; p.Cin = Cin; p.net = net; p.gy_sb = gy_sb; p.gy_se = gy_se; p.gy_sr = gy_sr;
    p.x_sb = x_sb; p.x_sc = x_sc; p.x_sr = x_sr;
    const int grid = (int)em_bwd_grid(p.R);
    p.rows_per_wg = (int)((p.R + grid - 1) / grid);
    const size_t lds = (size_t)p.rows_per_wg * 24;
    hipStream_t st = (hipStream_t)stream;
    if (em_mfma_ok(p, true)) {
        fgnn_note_kernel("edge_mlp_bwd_mfma_kernel");
        hipLaunchKernelGGL(edge_mlp_bwd_mfma_kernel, dim3(grid), dim3(256), 0, st, p);
    } else {
        fgnn_note_kernel("edge_mlp_bwd_kernel");
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)edge_mlp_bwd_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, EM_LDS_ROWS * 24);
            (void)hipFuncSetAttribute((const void*)edge_mlp_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, EM_LDS_ROWS * 24);
            attr_done = true;
        }
        if (Cin == 7) hipLaunchKernelGGL(edge_mlp_bwd_kernel<7>, dim3(grid), dim3(EM_BWD_THREADS), lds, st, p);
        else hipLaunchKernelGGL(edge_mlp_bwd_kernel<8>, dim3(grid), dim3(EM_BWD_THREADS), lds, st, p);
    }
    hipLaunchKernelGGL(edge_mlp_reduce_kernel, dim3((EM_SLAB + 15) / 16), dim3(256), 0, st, p.ws, grid, Cin, net, gW1, gb1,
                       gW2, gb2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "edge_mlp_backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
