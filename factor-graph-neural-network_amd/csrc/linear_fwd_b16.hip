// linear_fwd_b16.hip — forward of the node-wise (1x1) maps around the message operator for bf16 activations:
//
//     y[r][o] = sum_c x[r][c] W[o][c] + b[o]          r over R = B*N rows, Cin / Cout multiples of 64, <= 256
//
// (`Conv2d(cin, cout, 1)` in mp_conv_residual / iid_mapping*, /root/reference/lib/model/mpnn/mp_nn_residual.py:25-35,
// base_model.py:43-90), optionally with the per-channel batch statistics of y for the BatchNorm that follows it
// in the reference (sum and sum of squares per workgroup, in the partial layout csrc/bnact.hip folds), so that
// the BatchNorm needs no statistics pass of its own.
//
// A skinny GEMM (K, N <= 256, M ~ 4e5) is a pure stream: 2*R*(Cin+Cout) bytes against 2*R*Cin*Cout flops on the
// bf16 matrix cores.  W^T lives in LDS as bf16 (register-resident when it is at most 16 fragments); a wave walks
// 16-row tiles, takes the x operand straight from global memory (8 consecutive channels of a row = one 16-byte
// load per lane, each byte fetched once) and writes 4 consecutive output channels of a row per lane.  Wide
// outputs are split over the waves of a workgroup (<= 64 output channels per wave), which then share the x rows
// through L1.  No barriers in the streaming loop.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>

#define LF_THREADS 512
#define LF_WAVES 8
#define LF_MAXGRID 256   // one workgroup per CU (cold tensors, 393 k rows 64->64: 23.9 us; 512: 25.4; 768: 28.3; 1024: 25.6); <= BN_MAXPART of bnact.hip: the statistics partials reuse its workspace layout

typedef __bf16 lf_bf16x8 __attribute__((ext_vector_type(8)));

struct LfParams {
    const uint16_t* x;   // [R][Cin] bf16
    const float* W;      // [Cout][Cin] f32
    const float* bias;   // [Cout] or NULL
    uint16_t* y;         // [R][Cout] bf16
    float* part;         // [grid][2][Cout] per-workgroup (sum, sum of squares) of y, or NULL
    int R, Cin, Cout;
    int CG;              // output-channel groups per workgroup (waves = CG x 8/CG row groups)
    int wt;              // W is stored transposed, [Cin][Cout] (the grad-input product gy W of a [Cout'][Cin'] weight)
    FgnnFold fold;       // fold.tickets != NULL: the last workgroup folds the partial rows and finalises the BatchNorm statistics itself
    fgnn_bn_final fin;
};

extern __shared__ __attribute__((aligned(16))) unsigned char lf_lds[];

__device__ __forceinline__ unsigned lf_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}

// uniform 64-bit base + UNSIGNED 32-bit per-lane byte offset: the form that compiles to `global_load v, v_off, s[base]`
template <typename T> __device__ __forceinline__ const T* lf_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ T* lf_at(void* base, unsigned byte_off) {
    return reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}

// KS = Cin / 32 k-steps, OTW = 16-channel output tiles per wave (Cout / 16 / CG), WREG: W fragments in registers
template <int KS, int OTW, bool WREG>
__global__ __launch_bounds__(LF_THREADS) void linear_fwd_b16_kernel(const LfParams p) {
    constexpr int CIN = 32 * KS, WS = CIN + 8;            // LDS row stride of W (bf16 elements)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Cout = p.Cout, R = p.R;
    const int cg = wave % p.CG, rg = wave / p.CG, nrg = LF_WAVES / p.CG;
    uint16_t* Wl = reinterpret_cast<uint16_t*>(lf_lds);                    // [Cout][WS] bf16
    float* bl = reinterpret_cast<float*>(lf_lds + (size_t)Cout * WS * 2);    // [Cout] bias
    float* red = bl + Cout;                                                // [nrg][2][Cout] statistics fold
    // (four independent loads in flight per thread and pass: one value per iteration made the prologue a chain of up to 128 L2
    // round trips — tools/mbench.py found it as 90 us of a 384 -> 128 product in linear_multi_b16_kernel below)
    if (!p.wt) {
        const int total = Cout * (CIN / 2);
        for (int f0 = tid; f0 < total; f0 += 4 * LF_THREADS) {
            float2 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * LF_THREADS;
                if (f < total) { const int o = f / (CIN / 2), c2 = f - o * (CIN / 2); w[u] = *reinterpret_cast<const float2*>(p.W + (int64_t)o * CIN + 2 * c2); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * LF_THREADS;
                if (f < total) { const int o = f / (CIN / 2), c2 = f - o * (CIN / 2); *reinterpret_cast<unsigned*>(Wl + o * WS + 2 * c2) = lf_pack2(w[u].x, w[u].y); }
            }
        }
    } else {                                              // memory is [c][o]: coalesced reads, 2-byte LDS writes
        const int total = Cout * CIN;
        for (int f0 = tid; f0 < total; f0 += 4 * LF_THREADS) {
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int f = f0 + u * LF_THREADS; if (f < total) w[u] = p.W[f]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * LF_THREADS;
                if (f < total) { const int c = f / Cout, o = f - c * Cout; const __bf16 h = (__bf16)w[u]; Wl[o * WS + c] = __builtin_bit_cast(uint16_t, h); }
            }
        }
    }
    for (int f = tid; f < Cout; f += LF_THREADS) bl[f] = p.bias ? p.bias[f] : 0.f;
    __syncthreads();

    const int o_base = cg * OTW * 16;                     // this wave's first output channel
    // Operand permutations that make every lane's global traffic contiguous (OTW == 4):
    //  * output tile ot, MFMA row i  <->  channel 16 (i >> 2) + 4 ot + (i & 3): the D fragments of the four tiles then give lane
    //    (row li, lk) the 16 CONSECUTIVE channels 16 lk .. 16 lk + 15 of its row: two 16-byte stores instead of four 8-byte ones;
    //  * k-step ks, k-group lk  <->  input channels 8 KS lk + 8 ks .. + 7: a lane's KS loads are one contiguous 16 KS-byte run.
    // OTW == 8 (round 6: the 256-input maps with >= 128 outputs): TWO 64-channel slabs per wave, each with the permutation above — the
    // column-group waves of a workgroup each fetch the tile's rows through L1 as 16-byte pieces of 64 different lines per load, and at
    // four column groups that line rate, not the LDS or HBM, was the bound (2.5 TB/s at 256 -> 256); two groups halve the re-reads.
    static_assert(OTW == 4 || OTW == 8, "64 or 128 output channels per wave");
    auto orow = [&](int ot) { return 64 * (ot >> 2) + 16 * (li >> 2) + 4 * (ot & 3) + (li & 3); };
    auto ocol = [&](int ot) { return 64 * (ot >> 2) + 16 * lk + 4 * (ot & 3); };          // the lane's four channels of tile ot, from o_base
    auto kcol = [&](int ks) { return 8 * KS * lk + 8 * ks; };
    lf_bf16x8 aW[WREG ? OTW : 1][WREG ? KS : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                aW[ot][ks] = __builtin_bit_cast(lf_bf16x8, *reinterpret_cast<const uint4*>(Wl + (o_base + orow(ot)) * WS + kcol(ks)));
    }
    f32x4 bv[OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) bv[ot] = *reinterpret_cast<const f32x4*>(bl + o_base + ocol(ot));
    f32x4 s0[OTW], s1[OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) { s0[ot] = (f32x4){0.f, 0.f, 0.f, 0.f}; s1[ot] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int ntile = (R + 15) / 16;
    const int stride = gridDim.x * nrg;
    // A tile's rows: uniform 64-bit base (the tile) + a 32-bit lane offset, every load UNCONDITIONAL (a tile past the end re-reads the last
    // one, a row past R the last row: neither is stored or counted).  Round 6: per-lane 64-bit pointers + a guard per load made the
    // 256-input instance spill, and the reload in front of the stores — a memory operation — waited (vmcnt(0)) for the NEXT tile's
    // eight loads just requested: no tile was ever in flight under the products (1.1 TB/s in the training step).
    auto load_tile = [&](int tile, uint4 (&bx)[KS]) {
        const int tl = tile < ntile ? tile : ntile - 1;
        const uint16_t* xt = p.x + (int64_t)tl * 16 * CIN;
        const int rl = tl * 16 + li < R ? li : R - 1 - tl * 16;
        const unsigned off = (unsigned)((rl * CIN + kcol(0)) * 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bx[ks] = *lf_at<uint4>(xt, off + 16u * ks);
    };
    uint4 nx[KS];
    load_tile(blockIdx.x * nrg + rg, nx);
    for (int tile = blockIdx.x * nrg + rg; tile < ntile; tile += stride) {
        const int row = tile * 16 + li;
        const bool ok = row < R;
        uint4 bx[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bx[ks] = nx[ks];
        load_tile(tile + stride, nx);                     // the next tile's rows are in flight while this one is multiplied and stored
        if (!WREG && OTW == 8) asm volatile("" ::: "memory");      // the LDS operands are re-read per tile, not hoisted into (spilled) registers
        f32x4 acc[OTW];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
            acc[ot] = bv[ot];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                lf_bf16x8 a;
                if constexpr (WREG) a = aW[ot][ks];
                else a = __builtin_bit_cast(lf_bf16x8, *reinterpret_cast<const uint4*>(Wl + (o_base + orow(ot)) * WS + kcol(ks)));
                acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(lf_bf16x8, bx[ks]), acc[ot], 0, 0, 0);
            }
        }
        // D[i = 4 lk + r][j = row] of tile ot = channel o_base + 16 lk + 4 ot + r of this lane's row
        if (ok) {
            uint16_t* yt = p.y + (int64_t)tile * 16 * Cout;
            const unsigned yo = (unsigned)((li * Cout + o_base + 16 * lk) * 2);
#pragma unroll
            for (int sl = 0; sl < OTW / 4; ++sl) {
                const f32x4 (&a4)[OTW] = acc;
                *lf_at<uint4>(yt, yo + 128u * sl) = make_uint4(lf_pack2(a4[4 * sl][0], a4[4 * sl][1]), lf_pack2(a4[4 * sl][2], a4[4 * sl][3]),
                                                              lf_pack2(a4[4 * sl + 1][0], a4[4 * sl + 1][1]), lf_pack2(a4[4 * sl + 1][2], a4[4 * sl + 1][3]));
                *lf_at<uint4>(yt, yo + 128u * sl + 16u) = make_uint4(lf_pack2(a4[4 * sl + 2][0], a4[4 * sl + 2][1]), lf_pack2(a4[4 * sl + 2][2], a4[4 * sl + 2][3]),
                                                                    lf_pack2(a4[4 * sl + 3][0], a4[4 * sl + 3][1]), lf_pack2(a4[4 * sl + 3][2], a4[4 * sl + 3][3]));
            }
            if (p.part) {
#pragma unroll
                for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s0[ot][r] += acc[ot][r]; s1[ot][r] = fmaf(acc[ot][r], acc[ot][r], s1[ot][r]); }
            }
        }
    }
    if (p.part) {                                         // per-workgroup (sum, sum of squares) of every channel
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s0[ot][r], b = s1[ot][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }   // the 16 rows of a tile
                if (li == 0) {
                    const int o = o_base + ocol(ot) + r;
                    red[(rg * 2) * Cout + o] = a;
                    red[(rg * 2 + 1) * Cout + o] = b;
                }
            }
        __syncthreads();
        for (int f = tid; f < 2 * Cout; f += LF_THREADS) {
            float s = 0.f;
            for (int g = 0; g < nrg; ++g) s += red[g * 2 * Cout + f];
            fgnn_fold_store(p.part + (int64_t)blockIdx.x * 2 * Cout + f, s);
        }
        if (p.fold.tickets) {
            double* sums = reinterpret_cast<double*>(lf_lds);     // (W's image is dead; the fold starts with a barrier)
            if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) fgnn_bn_final_apply(p.fin, Cout, sums);
        }
    }
}

// ----------------------------------------------------------------------------------------
// The node-wise map FOLLOWED BY InstanceNorm (+ ReLU) — `iid_mapping_in`, /root/reference/lib/model/mpnn/base_model.py:82-90
// (Conv2d(cin, cout, 1) -> InstanceNorm2d(cout) -> ReLU; FactorNN's v2v / f2f maps, factor_mpnn_sp.py:77,140) — in ONE pass:
// the norm is per (sample, channel) over the sample's N nodes, so a wave that owns all N / 16 row tiles of a sample for its 64
// output channels has the whole population in its accumulators (NT x 16 registers): per-channel sums fold over the 16 rows
// of a tile with row-shuffles and over the tiles in registers, no LDS, no second kernel, and the pre-norm tensor z is only
// WRITTEN (when the backward needs it: zs != NULL), never read back.  Statistics are formed from the bf16-ROUNDED z — the
// population the staged path (csrc/instnorm.hip on the stored z) and the backward see — two-pass, f32.
// ----------------------------------------------------------------------------------------
struct LiParams {
    LfParams lf;         // x, W, bias, y; R = B * N rows; part unused
    uint16_t* zs;        // [R][Cout] pre-norm output (bf16) or NULL
    int B, N;            // N = 16 NT nodes per sample
    int relu;
    float eps;
};

template <int KS, bool WREG, int NT>
__global__ __launch_bounds__(LF_THREADS) void linear_instnorm_fwd_kernel(const LiParams q) {
    constexpr int OTW = 4, CIN = 32 * KS, WS = CIN + 8;
    const LfParams& p = q.lf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Cout = p.Cout;
    const int cg = wave % p.CG, rg = wave / p.CG, nrg = LF_WAVES / p.CG;
    uint16_t* Wl = reinterpret_cast<uint16_t*>(lf_lds);
    float* bl = reinterpret_cast<float*>(lf_lds + (size_t)Cout * WS * 2);
    for (int f = tid; f < Cout * (CIN / 2); f += LF_THREADS) {
        const int o = f / (CIN / 2), c2 = f - o * (CIN / 2);
        const float2 w = *reinterpret_cast<const float2*>(p.W + (int64_t)o * CIN + 2 * c2);
        *reinterpret_cast<unsigned*>(Wl + o * WS + 2 * c2) = lf_pack2(w.x, w.y);
    }
    for (int f = tid; f < Cout; f += LF_THREADS) bl[f] = p.bias ? p.bias[f] : 0.f;
    __syncthreads();

    const int o_base = cg * OTW * 16;
    auto orow = [&](int ot) { return 16 * (li >> 2) + 4 * ot + (li & 3); };     // (the channel permutation of linear_fwd_b16_kernel)
    auto kcol = [&](int ks) { return 8 * KS * lk + 8 * ks; };
    lf_bf16x8 aW[WREG ? OTW : 1][WREG ? KS : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                aW[ot][ks] = __builtin_bit_cast(lf_bf16x8, *reinterpret_cast<const uint4*>(Wl + (o_base + orow(ot)) * WS + kcol(ks)));
    }
    f32x4 bv[OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) bv[ot] = *reinterpret_cast<const f32x4*>(bl + o_base + 16 * lk + 4 * ot);

    const int N = q.N;
    const float invn = 1.0f / (float)N;
    // The K dimension is walked in chunks of KC <= 4 k-steps (KC x 16 bytes per lane and row tile): the next chunk — of this
    // tile or the first of the next — is in flight under the current one's products.  (A whole 256-channel row tile per step left
    // no registers for a second one: 207 us against 194 staged at 256 -> 256.)
    constexpr int KC = KS <= 4 ? KS : (KS % 4 == 0 ? 4 : 3), NCH = KS / KC;
    static_assert(KS % KC == 0, "k-steps in whole chunks");
    auto load_chunk = [&](int b, int idx, uint4 (&bx)[KC]) {          // idx = t * NCH + ch
        const int t = idx / NCH, ch = idx - t * NCH;
        const int64_t row = (int64_t)b * N + t * 16 + li;
#pragma unroll
        for (int k = 0; k < KC; ++k)
            bx[k] = *reinterpret_cast<const uint4*>(p.x + row * CIN + kcol(ch * KC + k));
    };
    for (int b = blockIdx.x * nrg + rg; b < q.B; b += gridDim.x * nrg) {
        f32x4 acc[NT][OTW];
        uint4 nx[KC];
        load_chunk(b, 0, nx);
#pragma unroll
        for (int idx = 0; idx < NT * NCH; ++idx) {
            constexpr int dummy = 0; (void)dummy;
            const int t = idx / NCH, ch = idx - t * NCH;
            uint4 bx[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) bx[k] = nx[k];
            if (idx + 1 < NT * NCH) load_chunk(b, idx + 1, nx);
            if (NCH > 1) __builtin_amdgcn_sched_barrier(0);          // (keep ONE chunk ahead: left alone the scheduler hoists them all and spills)
#pragma unroll
            for (int ot = 0; ot < OTW; ++ot) {
                if (ch == 0) acc[t][ot] = bv[ot];
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int ks = ch * KC + k;
                    lf_bf16x8 a;
                    if constexpr (WREG) a = aW[ot][ks];
                    else a = __builtin_bit_cast(lf_bf16x8, *reinterpret_cast<const uint4*>(Wl + (o_base + orow(ot)) * WS + kcol(ks)));
                    acc[t][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(lf_bf16x8, bx[k]), acc[t][ot], 0, 0, 0);
                }
            }
        }
        // z as stored: bf16.  acc[t][ot][r] = channel o_base + 16 lk + 4 ot + r of row 16 t + li
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4 lo, hi;
            lo = make_uint4(lf_pack2(acc[t][0][0], acc[t][0][1]), lf_pack2(acc[t][0][2], acc[t][0][3]),
                            lf_pack2(acc[t][1][0], acc[t][1][1]), lf_pack2(acc[t][1][2], acc[t][1][3]));
            hi = make_uint4(lf_pack2(acc[t][2][0], acc[t][2][1]), lf_pack2(acc[t][2][2], acc[t][2][3]),
                            lf_pack2(acc[t][3][0], acc[t][3][1]), lf_pack2(acc[t][3][2], acc[t][3][3]));
            if (q.zs) {
                uint16_t* zp = q.zs + ((int64_t)b * N + t * 16 + li) * Cout + o_base + 16 * lk;
                *reinterpret_cast<uint4*>(zp) = lo;
                *reinterpret_cast<uint4*>(zp + 8) = hi;
            }
            const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int ot = 0; ot < OTW; ++ot) {                 // the rounded values are the population
                acc[t][ot][0] = __uint_as_float(w[2 * ot] << 16);     acc[t][ot][1] = __uint_as_float(w[2 * ot] & 0xffff0000u);
                acc[t][ot][2] = __uint_as_float(w[2 * ot + 1] << 16); acc[t][ot][3] = __uint_as_float(w[2 * ot + 1] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) s += acc[t][ot][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);      // the 16 rows of a tile sit in 16 consecutive lanes
                const float mean = s * invn;
                float ss = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) { const float dv = acc[t][ot][r] - mean; acc[t][ot][r] = dv; ss = fmaf(dv, dv, ss); }
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m);
                const float rstd = rsqrtf(ss * invn + q.eps);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float v = acc[t][ot][r] * rstd;
                    acc[t][ot][r] = q.relu ? fmaxf(v, 0.f) : v;
                }
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint16_t* yp = p.y + ((int64_t)b * N + t * 16 + li) * Cout + o_base + 16 * lk;
            *reinterpret_cast<uint4*>(yp) = make_uint4(lf_pack2(acc[t][0][0], acc[t][0][1]), lf_pack2(acc[t][0][2], acc[t][0][3]),
                                                       lf_pack2(acc[t][1][0], acc[t][1][1]), lf_pack2(acc[t][1][2], acc[t][1][3]));
            *reinterpret_cast<uint4*>(yp + 8) = make_uint4(lf_pack2(acc[t][2][0], acc[t][2][1]), lf_pack2(acc[t][2][2], acc[t][2][3]),
                                                           lf_pack2(acc[t][3][0], acc[t][3][1]), lf_pack2(acc[t][3][2], acc[t][3][3]));
        }
    }
}

// 16-channel output tiles per wave of fgnn_linear_forward: 8 (two slabs) for the 256-input maps with >= 128 outputs, else 4
static int lf_otw(int Cin, int Cout) { return (Cin > 192 && Cout >= 128 && Cout % 128 == 0) ? 8 : 4; }

static int lf_plan(int64_t R, int Cin, int Cout, int* CG, int* grid, int otw = 4) {
    if (Cin % 64 || Cout % 64 || Cin > 256 || Cout > 256 || R <= 0 || R > 0x7fffffff) return -1;
    *CG = Cout / (16 * otw);                              // 64 (or 128) output channels per wave
    if (*CG == 3) return -1;
    const int nrg = LF_WAVES / *CG;
    const int64_t ntile = (R + 15) / 16;
    int64_t g = (ntile + 2 * nrg - 1) / (2 * nrg);         // >= 2 tiles per wave
    const int maxg = LF_MAXGRID;
    if (g > maxg) g = maxg;
    if (g < 1) g = 1;
    *grid = (int)g;
    return 0;
}

// Number of per-workgroup statistics partials a call with these sizes writes (0 = shape not supported).
extern "C" int fgnn_linear_forward_partials(int64_t R, int Cin, int Cout) {
    int CG, grid;
    return lf_plan(R, Cin, Cout, &CG, &grid, lf_otw(Cin, Cout)) ? 0 : grid;
}

// y = x W^T + b for bf16 x / y, f32 W / b.  stats_partials: NULL, or device scratch of
// fgnn_linear_forward_partials(..) * 2 * Cout floats receiving per-workgroup (sum y, sum y^2) per channel —
// feed it to fgnn_bn_finalize.  w_transposed: W is [Cin][Cout] in memory (y = x W; the grad-input product of a
// map whose weight is [Cin][Cout] = [cout'][cin']).  Returns FGNN_EUNSUPPORTED for other shapes (callers fall back to a library GEMM).
int fgnn_bn_finalize_launch(const float* partials, int npartials, int C, const fgnn_bn_final* fin, hipStream_t st);

extern "C" int fgnn_linear_forward(const void* x, const float* W, const float* bias, void* y, int64_t R, int Cin,
                                   int Cout, float* stats_partials, const fgnn_bn_final* fin, void* fold_scratch,
                                   int w_transposed, fgnn_stream_t stream) {
    if (!x || !W || !y) FGNN_FAIL(FGNN_EINVAL, "linear_forward: null pointer");
    int CG, grid;
    const int otw = lf_otw(Cin, Cout);
    if (lf_plan(R, Cin, Cout, &CG, &grid, otw) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)W & 7))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_forward: Cin=%d Cout=%d outside the bf16 streaming kernel's family", Cin, Cout);
    if (fin && (!stats_partials || !fin->mean || !fin->invstd || !fin->scale || !fin->shift || fin->count != R || fin->shift_k))
        FGNN_FAIL(FGNN_EINVAL, "linear_forward: fgnn_bn_final needs stats_partials, its outputs, count == R and no shift_k");
    LfParams p = {};
    p.x = (const uint16_t*)x; p.W = W; p.bias = bias; p.y = (uint16_t*)y; p.part = stats_partials;
    p.R = (int)R; p.Cin = Cin; p.Cout = Cout; p.CG = CG; p.wt = w_transposed;
    const bool inkernel = fin && fold_scratch && !fgnn_separate_finalisers();
    if (fin) p.fin = *fin;
    p.fold = fgnn_fold_make(stats_partials, inkernel ? fold_scratch : nullptr, grid, Cout);
    const int KS = Cin / 32;
    void* fn;
    switch (KS) {       // OTW == 4 always (64 output channels per wave); fragments stay in registers up to Cin 128
        case 2: fn = (void*)linear_fwd_b16_kernel<2, 4, true>; break;
        case 4: fn = (void*)linear_fwd_b16_kernel<4, 4, true>; break;
        case 6: fn = (void*)linear_fwd_b16_kernel<6, 4, false>; break;
        default: fn = otw == 8 ? (void*)linear_fwd_b16_kernel<8, 8, false> : (void*)linear_fwd_b16_kernel<8, 4, false>; break;
    }
    const int lds = Cout * (Cin + 8) * 2 + Cout * 4 + (LF_WAVES / CG) * 2 * Cout * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(LF_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_forward launch: %s", hipGetErrorString(e));
    if (fin && !inkernel && fgnn_bn_finalize_launch(stats_partials, grid, Cout, fin, (hipStream_t)stream))
        FGNN_FAIL(FGNN_ELAUNCH, "linear_forward finaliser launch: %s", hipGetErrorString(hipGetLastError()));
    return FGNN_OK;
}


// y = act(InstanceNorm(x W^T + b)) per sample over its N nodes, bf16 rows [B * N][Cin] -> [B * N][Cout]; z (or NULL): the
// pre-norm map output, stored for the backward (fgnn_instnorm_backward reads it).  N in {48, 96} (a multiple of 16 with an
// instantiated tile count), Cin / Cout multiples of 64 up to 256.  FGNN_EUNSUPPORTED otherwise (callers run the two kernels).
extern "C" int fgnn_linear_instnorm_forward(const void* x, const float* W, const float* bias, void* z, void* y, int B, int N,
                                            int Cin, int Cout, int relu, float eps, fgnn_stream_t stream) {
    if (!x || !W || !y) FGNN_FAIL(FGNN_EINVAL, "linear_instnorm_forward: null pointer");
    if (B < 0 || N < 1) FGNN_FAIL(FGNN_EINVAL, "linear_instnorm_forward: bad sizes B=%d N=%d", B, N);
    int CG, grid;
    if ((N != 48 && N != 96) || lf_plan((int64_t)(B > 0 ? B : 1) * N, Cin, Cout, &CG, &grid) || ((uintptr_t)x & 15) ||
        ((uintptr_t)y & 15) || ((uintptr_t)z & 15) || ((uintptr_t)W & 7))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_instnorm_forward: N=%d Cin=%d Cout=%d outside the fused kernel's family", N, Cin, Cout);
    if (B == 0) return FGNN_OK;
    LiParams q;
    q.lf.x = (const uint16_t*)x; q.lf.W = W; q.lf.bias = bias; q.lf.y = (uint16_t*)y; q.lf.part = nullptr;
    q.lf.R = B * N; q.lf.Cin = Cin; q.lf.Cout = Cout; q.lf.CG = CG; q.lf.wt = 0; q.lf.fold = fgnn_fold_make(nullptr, nullptr, 0, Cout);
    q.zs = (uint16_t*)z; q.B = B; q.N = N; q.relu = relu; q.eps = eps;
    const int nrg = LF_WAVES / CG;
    grid = (B + nrg - 1) / nrg;
    if (grid > LF_MAXGRID) grid = LF_MAXGRID;
    const int KS = Cin / 32;
    void* fn = nullptr;
#define LI_PICK(ks, wreg) fn = N == 96 ? (void*)linear_instnorm_fwd_kernel<ks, wreg, 6> : (void*)linear_instnorm_fwd_kernel<ks, wreg, 3>
    switch (KS) {
        case 2: LI_PICK(2, true); break;
        case 4: LI_PICK(4, true); break;
        case 6: LI_PICK(6, false); break;
        default: LI_PICK(8, false); break;
    }
#undef LI_PICK
    const int lds = Cout * (Cin + 8) * 2 + Cout * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    void* args[] = {(void*)&q};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(LF_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_instnorm_forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// ----------------------------------------------------------------------------------------
// Several maps INTO ONE tensor (round 5): y[r][o] = sum_s sum_c x_s[r][c] W_s[c][o] + addend_0[r][o] + addend_1[r][o] + addend_2[r][o].
//
// In a FactorNN layer (/root/reference/lib/model/mpnn/factor_mpnn_sp.py:136-168) the variables' state feeds the v2v map and one
// block per factor type, so its gradient is the sum of their input gradients — each one `gz_s W_s` with a different gz (64 channels
// behind a block's conv1, the next layer's width behind the node-wise map) — plus the gradients that arrive unchanged (the residual,
// a skip link).  Staged, every consumer wrote its [R][C] product and an n-input sum read them all back: 5.7 GB of sum_n traffic +
// 3 x [R][C] of writes per layer state out of the step's 52 GB (gpurun_out/r05p/traffic).  Here the narrow gz_s are the operands and
// the sum never leaves the accumulators: one K-concatenated product, the addends joined in f32 before the one rounding.
// Up to three sources (KS0 / KS1 / KS2 k-steps of 32 channels; 0 = absent), W_s [K_s][Cout] f32 row-major (a map's [cout'][cin']
// weight as it lies in memory), Cout in {64, 128, 256}; wide products are split over the grid's y dimension so that the bf16 image
// of the stacked W stays under 150 KB of LDS.
// ----------------------------------------------------------------------------------------
struct LmParams {
    const uint16_t* x[3];    // [R][K_s] bf16
    const float* W[3];       // [K_s][Cout] f32
    const uint16_t* add[3];  // [R][Cout] bf16 or NULL
    uint16_t* y;             // [R][Cout]
    int R, Cout;             // Cout: full row width of y / W / addends
    int NW;                  // output channels per workgroup (64 CG): blockIdx.y selects the slice
    int CG;
};

template <int KS0, int KS1, int KS2>
__global__ __launch_bounds__(LF_THREADS) void linear_multi_b16_kernel(const LmParams p) {
    // The W image is read once per MFMA (it does not fit the registers).  A lane reads 16 bytes of row 16 (li >> 2) + 4 ot + (li & 3):
    // rows 16 apart start a multiple of 64 dwords apart whatever the (16-byte aligned) row stride, i.e. on the SAME banks — a 4-way
    // conflict on every fragment read, which capped the wide products at ~200 TFLOP/s / 2.5 TB/s (tools/mbench.py).  So the 16-byte
    // column chunks of row o are stored XOR ((o >> 4) & 3) << 2: the four row groups of a read then sit 16 banks apart.  (The XOR
    // stays inside an aligned group of 16 chunks: rows are padded to a multiple of 128 channels.)
    constexpr int KS = KS0 + KS1 + KS2, KT = 32 * KS, WS = (KT + 127) / 128 * 128 + 8, OTW = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int R = p.R, Cout = p.Cout, NW = p.NW;
    const int n0 = blockIdx.y * NW;
    const int cg = wave % p.CG, rg = wave / p.CG, nrg = LF_WAVES / p.CG;
    uint16_t* Wl = reinterpret_cast<uint16_t*>(lf_lds);                    // [NW][WS] bf16: row o = output channel n0 + o, column = stacked k
    {
        // W_s [K_s][Cout] f32 -> the bf16 image [o][stacked k]: 16-byte reads along o, FOUR of them in flight per thread (one value per
        // iteration made the prologue a chain of ~100 L2 round trips: 90 of the 189 us of a 384 -> 128 product, tools/mbench.py)
        constexpr int K0 = 32 * KS0, K1 = 32 * KS1, K2 = 32 * KS2;
        const int q4 = NW / 4;
        auto stage = [&](const float* __restrict__ W, int K, int koff) {
            const int total = K * q4;
            for (int f0 = tid; f0 < total; f0 += 4 * LF_THREADS) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * LF_THREADS;
                    if (f < total) { const int c = f / q4, o = 4 * (f - c * q4); v[u] = *reinterpret_cast<const f32x4*>(W + (int64_t)c * Cout + n0 + o); }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * LF_THREADS;
                    if (f < total) {
                        const int c = f / q4, o = 4 * (f - c * q4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const __bf16 h = (__bf16)v[u][j];
                            const int k = koff + c, sw = (((o + j) >> 4) & 3) << 2;
                            Wl[(o + j) * WS + (((k >> 3) ^ sw) << 3) + (k & 7)] = __builtin_bit_cast(uint16_t, h);
                        }
                    }
                }
            }
        };
        if constexpr (KS0 > 0) stage(p.W[0], K0, 0);
        if constexpr (KS1 > 0) stage(p.W[1], K1, K0);
        if constexpr (KS2 > 0) stage(p.W[2], K2, K0 + K1);
    }
    __syncthreads();
    const int o_base = cg * OTW * 16;                     // this wave's first channel inside the slice
    auto orow = [&](int ot) { return 16 * (li >> 2) + 4 * ot + (li & 3); };     // (the channel permutation of linear_fwd_b16_kernel)
    // k-step ks of source s, k-group lk  <->  that source's channels 8 KS_s lk + 8 ks .. + 7: one contiguous 16 KS_s-byte run per lane and source
    auto wcol = [&](int g) {                                // stacked column of global k-step g
        return g < KS0 ? 8 * KS0 * lk + 8 * g : (g < KS0 + KS1 ? 32 * KS0 + 8 * KS1 * lk + 8 * (g - KS0) : 32 * (KS0 + KS1) + 8 * KS2 * lk + 8 * (g - KS0 - KS1));
    };
    const int ntile = (R + 15) / 16;
    const int stride = gridDim.x * nrg;
    auto load_tile = [&](int tile, uint4 (&bx)[KS]) {
        const int row = tile * 16 + li;
        const bool ok = tile < ntile && row < R;
#pragma unroll
        for (int g = 0; g < KS; ++g) {
            const uint16_t* src = g < KS0 ? p.x[0] + (int64_t)row * (32 * KS0) + 8 * KS0 * lk + 8 * g
                                : (g < KS0 + KS1 ? p.x[1] + (int64_t)row * (32 * KS1) + 8 * KS1 * lk + 8 * (g - KS0)
                                                 : p.x[2] + (int64_t)row * (32 * KS2) + 8 * KS2 * lk + 8 * (g - KS0 - KS1));
            bx[g] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        }
    };
    auto unpack8 = [](const uint4& q, float (&v)[8]) {
        v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u); v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
        v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u); v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
    };
    uint4 nx[KS];
    load_tile(blockIdx.x * nrg + rg, nx);
    for (int tile = blockIdx.x * nrg + rg; tile < ntile; tile += stride) {
        const int row = tile * 16 + li;
        const bool ok = row < R;
        uint4 bx[KS];
#pragma unroll
        for (int g = 0; g < KS; ++g) bx[g] = nx[g];
        // the addends of this lane's 16 output channels (two 16-byte pieces per addend), in flight with the next tile's rows
        uint4 ad[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint16_t* ap = p.add[a] ? p.add[a] + (int64_t)row * Cout + n0 + o_base + 16 * lk : nullptr;
            ad[a][0] = (ap && ok) ? *reinterpret_cast<const uint4*>(ap) : make_uint4(0, 0, 0, 0);
            ad[a][1] = (ap && ok) ? *reinterpret_cast<const uint4*>(ap + 8) : make_uint4(0, 0, 0, 0);
        }
        load_tile(tile + stride, nx);
        // W fragments are re-read from LDS per tile: hoisted out of the loop they are 16 KS registers (spills from 10 k-steps on).  The
        // opaque value is an OFFSET, not the pointer: behind an asm the pointer is a generic one and every read a flat_load (750 cycles
        // per MFMA: the first form of this kernel ran its wide shapes at 200 TFLOP/s)
        int woff = 0;
        if constexpr (KS > 4) asm volatile("" : "+v"(woff));
        const uint16_t* wl = Wl + woff;
        f32x4 acc[OTW];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
            acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < KS; ++g) {
                const lf_bf16x8 a = __builtin_bit_cast(lf_bf16x8, *reinterpret_cast<const uint4*>(wl + (o_base + orow(ot)) * WS + (((wcol(g) >> 3) ^ ((li >> 2) << 2)) << 3)));
                acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(lf_bf16x8, bx[g]), acc[ot], 0, 0, 0);
            }
        }
        if (ok) {       // acc[ot][r] = channel n0 + o_base + 16 lk + 4 ot + r of this lane's row
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (p.add[a]) {
                    float v0[8], v1[8];
                    unpack8(ad[a][0], v0); unpack8(ad[a][1], v1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc[0][r] += v0[r]; acc[1][r] += v0[4 + r]; acc[2][r] += v1[r]; acc[3][r] += v1[4 + r]; }
                }
            }
            uint16_t* yp = p.y + (int64_t)row * Cout + n0 + o_base + 16 * lk;
            *reinterpret_cast<uint4*>(yp) = make_uint4(lf_pack2(acc[0][0], acc[0][1]), lf_pack2(acc[0][2], acc[0][3]),
                                                       lf_pack2(acc[1][0], acc[1][1]), lf_pack2(acc[1][2], acc[1][3]));
            *reinterpret_cast<uint4*>(yp + 8) = make_uint4(lf_pack2(acc[2][0], acc[2][1]), lf_pack2(acc[2][2], acc[2][3]),
                                                           lf_pack2(acc[3][0], acc[3][1]), lf_pack2(acc[3][2], acc[3][3]));
        }
    }
}

static void* lm_pick(int k0, int k1, int k2) {
#define LM_CASE(a, b, c) if (k0 == a && k1 == b && k2 == c) return (void*)linear_multi_b16_kernel<a, b, c>;
    LM_CASE(2, 0, 0) LM_CASE(2, 0, 2)
    LM_CASE(2, 2, 0) LM_CASE(2, 4, 0) LM_CASE(2, 8, 0)
    LM_CASE(2, 2, 2) LM_CASE(2, 4, 2) LM_CASE(2, 8, 2)
    LM_CASE(0, 2, 0) LM_CASE(0, 4, 0) LM_CASE(0, 8, 0)
#undef LM_CASE
    return nullptr;
}

// 1 if the source widths (slot 0 and 2: 0 or 64 channels; slot 1: 0, 64, 128 or 256) and the output width are the kernel's.
extern "C" int fgnn_linear_multi_supported(int64_t R, const int32_t* K, int Cout) {
    if (!K || R <= 0 || R > 0x7fffffff || (Cout != 64 && Cout != 128 && Cout != 256)) return 0;
    for (int s = 0; s < 3; ++s) if (K[s] % 32) return 0;
    return lm_pick(K[0] / 32, K[1] / 32, K[2] / 32) != nullptr;
}

extern "C" int fgnn_linear_multi_forward(const void* const* x, const int32_t* K, const float* const* W, const void* const* addend,
                                         void* y, int64_t R, int Cout, fgnn_stream_t stream) {
    if (!x || !K || !W || !addend || !y) FGNN_FAIL(FGNN_EINVAL, "linear_multi_forward: null pointer");
    if (!fgnn_linear_multi_supported(R, K, Cout))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_multi_forward: K = (%d, %d, %d) -> %d outside the kernel's family", K[0], K[1], K[2], Cout);
    LmParams p = {};
    int kt = 0;
    for (int s = 0; s < 3; ++s) {
        p.x[s] = (const uint16_t*)x[s]; p.W[s] = W[s]; p.add[s] = (const uint16_t*)addend[s];
        if (K[s] && (!x[s] || !W[s] || ((uintptr_t)x[s] & 15) || ((uintptr_t)W[s] & 15))) FGNN_FAIL(FGNN_EINVAL, "linear_multi_forward: source %d missing or misaligned", s);
        if (addend[s] && ((uintptr_t)addend[s] & 15)) FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_multi_forward: addend %d is not 16-byte aligned", s);
        kt += K[s];
    }
    if ((uintptr_t)y & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_multi_forward: y is not 16-byte aligned");
    p.y = (uint16_t*)y; p.R = (int)R; p.Cout = Cout;
    int NW = Cout;
    const int ws_row = (kt + 127) / 128 * 128 + 8;       // (the kernel's padded, swizzled row)
    while ((int64_t)NW * ws_row * 2 > 150 * 1024 && NW > 64) NW /= 2;
    p.NW = NW; p.CG = NW / 64;
    const int nrg = LF_WAVES / p.CG;
    const int64_t ntile = (R + 15) / 16;
    int64_t g = (ntile + 2 * nrg - 1) / (2 * nrg);
    const int ny = Cout / NW;
    const int maxg = LF_MAXGRID / ny > 0 ? LF_MAXGRID / ny : 1;
    if (g > maxg) g = maxg;
    if (g < 1) g = 1;
    void* fn = lm_pick(K[0] / 32, K[1] / 32, K[2] / 32);
    const int lds = NW * ws_row * 2;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel("linear_multi_b16_kernel<%d, %d, %d>", K[0] / 32, K[1] / 32, K[2] / 32);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)g, (unsigned)ny), dim3(LF_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_multi_forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
