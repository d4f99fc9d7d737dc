// factor_layer_fwd.hip — inference forward of ONE WHOLE `FactorNN` layer of the LDPC model in one kernel:
//
//     var'  = ReLU(IN(Wvv var))  + f2v_0(fac0 -> var) + f2v_1(fac1 -> var) + var  (+ skip)          [96, 64]
//     fac0' = ReLU(IN(Wff fac0)) + v2f_0(var -> fac0)                       + fac0 (+ skip)          [48, 64]
//     fac1' =        0           + v2f_1(var -> fac1)                       + fac1 (+ skip)          [ 1, 64]
//
// (/root/reference/lib/model/mpnn/factor_mpnn_sp.py:136-168: per factor type an F->V and a V->F `mp_conv_residual` block
// — Conv1x1+BN+LeakyReLU -> message operator (+bias, BN, ReLU) -> Conv1x1+BN+LeakyReLU, mp_nn_residual.py:39-56 with the
// operator of mp_nn.py:115-175 — plus the node-wise `iid_mapping_in` maps v2v / f2f, base_model.py:82-90; the hyper-factor's
// own f2f map is an InstanceNorm over ONE node, i.e. exactly 0.)  In eval mode every BatchNorm is a per-channel affine, so a
// sample's layer depends on nothing but that sample: the 145 node states (18.5 KB of bf16) are staged once, every
// intermediate of the four blocks and two maps lives in LDS or registers, and only the new state goes back to HBM.  The
// staged inference path runs this layer as ~12 kernels on two streams (2 node-wise GEMMs, 2 InstanceNorms, 2 n-way sums, 2
// parity blocks, fan-in and fan-out blocks) and moves the state through HBM about nine times.
//
// Family: the 64 -> 64 layers of `LDPCModel` (layers 0, 1, 7): bf16 channel-fastest states, 96 variables, 48 parity checks of
// degree 6 (variables of degree 3) with ONE neighbour table shared by the batch, 4 edge types (edge-type-fastest etype), one
// hyper-factor listening to all variables in order, max aggregation, nmed = 64.  The wider layers keep the per-block kernels:
// their weights (0.4 - 1.2 MB of bf16) do not stay in registers, and a sample-resident workgroup would stream them from L2
// once per sample.
//
// Schedule: one 512-thread workgroup per CU, samples in a grid-stride loop, the next sample's states and edge types
// prefetched into registers while the hyper-factor's phases run.  The weight matrices are resident: the parity and hyper
// blocks' 64 x 64 maps and 64 x 256 / 64 x 64 filters as bf16 MFMA fragments in registers (88 VGPRs), the two node-wise
// maps as fragments in LDS (used once per sample), the hyper-factor's three matrix-vector maps as bf16 in LDS.  Per sample
// eleven barriers over the images {var, fac0, A (conv1 output), B (operator output), P (projection)}:
//   maps + InstanceNorm statistics | conv1 V->F | projection | gather | conv2 + conv1 F->V | projection | gather |
//   conv2 + conv1 fan-in | fan-in projection + max, fan-out operator output | conv2 fan-out, residual + skip, store
// (the hyper-factor's three matrix-vector products are split over the eight waves by input channel, partials through LDS; the
// fan-in block's closing sum is taken by wave 0 behind the next sample's first barrier).  The new state accumulates in registers — every node-wise product has the same tile -> lane
// mapping, so an element is owned by one lane throughout.  Rounding points are those of the per-block kernels (a1, P, a2
// rounded to bf16; everything summed in f32 and rounded once).
// Measured (profiles/r02/README.md): ~245 us per layer at 4 096 codewords (16 samples per workgroup, ~37 000 shader clocks per
// sample: the phases are short dependent chains with two waves per SIMD to hide them) against ~300 us for the dozen kernels
// it replaces — the inference forward goes from 4.92 to 4.78 ms, and from 0.98 to 0.83 ms at 64 codewords where launches
// dominate.  HBM would allow 21 us: the next step is a software pipeline over samples (projection of sample s + 1 under
// the gather of sample s, as csrc/mpconv_fwd_sg.hip does for the bare operator).
#include "fgnn_common.h"
#include <stdlib.h>
#include <type_traits>

#define FL_THREADS 512
#define FL_NV 96             // variables
#define FL_NF 48             // parity checks
#define FL_KF 6              // check degree   (V -> F gathers 6 variables)
#define FL_KV 3              // variable degree (F -> V gathers 3 checks)
#ifndef FL_XS
#define FL_XS 72             // row stride (bf16) of the 64-channel images
#endif
#ifndef FL_PS
#define FL_PS 264            // row stride (bf16) of the projection image: 256 columns + 8
#endif
#define FL_EPS 1e-5f

typedef __bf16 fl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fl_bf16x2 __attribute__((ext_vector_type(2)));

// packed f32 parameters (floats): two maps, then four blocks (V->F parity, F->V parity, V->F hyper, F->V hyper)
#define FL_MAP_LEN (64 * 64)
#define FL_BLK_LEN(ncol) (64 * 64 + 64 + 64 + 64 * (ncol) + 64 + 64 + 64 * 64 + 64 + 64)
#define FL_OFF_WVV 0
#define FL_OFF_WFF (FL_MAP_LEN)
#define FL_OFF_B0 (2 * FL_MAP_LEN)
#define FL_OFF_B1 (FL_OFF_B0 + FL_BLK_LEN(256))
#define FL_OFF_B2 (FL_OFF_B1 + FL_BLK_LEN(256))
#define FL_OFF_B3 (FL_OFF_B2 + FL_BLK_LEN(64))
#define FL_PARAM_LEN (FL_OFF_B3 + FL_BLK_LEN(64))
// inside a block
#define FL_W1 0
#define FL_S1 (64 * 64)
#define FL_T1 (FL_S1 + 64)
#define FL_F (FL_T1 + 64)
#define FL_S2(ncol) (FL_F + 64 * (ncol))
#define FL_T2(ncol) (FL_S2(ncol) + 64)
#define FL_W2(ncol) (FL_T2(ncol) + 64)
#define FL_S3(ncol) (FL_W2(ncol) + 64 * 64)
#define FL_T3(ncol) (FL_S3(ncol) + 64)

struct FlParams {
    const uint16_t* var;          // [B][96][64]
    const uint16_t* fac0;         // [B][48][64]
    const uint16_t* fac1;         // [B][64]
    const uint16_t* skip_var;     // same layouts, or NULL
    const uint16_t* skip_fac0;
    const uint16_t* skip_fac1;
    const int64_t* idx_v2f;       // [48][6] variables of each check
    const int64_t* idx_f2v;       // [96][3] checks of each variable
    const uint16_t* et_v2f;       // [B][48][6][4]
    const uint16_t* et_f2v;       // [B][96][3][4]
    const uint16_t* het_v2f;      // [96] weights of the hyper-factor's in-edges (shared by the batch), or NULL = 1
    const uint16_t* het_f2v;      // [96] weights of its out-edges, or NULL = 1
    const float* w;               // FL_PARAM_LEN packed parameters
    uint16_t* out_var;
    uint16_t* out_fac0;
    uint16_t* out_fac1;
    long long et_v2f_sb, et_f2v_sb;   // batch strides in elements
    int B, residual;
    float slope;
    int idx_v2f_sm, idx_v2f_sk, idx_f2v_sm, idx_f2v_sk;
    long long* prof;              // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): phase timeline of one sample
};

extern __shared__ __attribute__((aligned(16))) unsigned char fl_lds[];

#ifdef FGNN_ENABLE_PROF
#define FL_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && b == (int)(3 * gridDim.x)) p.prof[wave * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define FL_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ unsigned fl_pack2(float a, float b) {
    const fl_bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float fl_round(float a) {
    const __bf16 h = (__bf16)a;
    return __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16);
}
__device__ __forceinline__ float fl_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float fl_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ fl_bf16x8 fl_frag8(const float* p8) {            // 8 consecutive f32 -> one fragment
    const f32x4 a = *reinterpret_cast<const f32x4*>(p8), b = *reinterpret_cast<const f32x4*>(p8 + 4);
    return __builtin_bit_cast(fl_bf16x8, make_uint4(fl_pack2(a[0], a[1]), fl_pack2(a[2], a[3]), fl_pack2(b[0], b[1]), fl_pack2(b[2], b[3])));
}
// uniform 64-bit base + UNSIGNED 32-bit per-lane byte offset: the form that compiles to `global_load v, v_off, s[base]`.  Per-lane 64-bit
// pointers are hoisted out of the sample loop and SPILLED at this kernel's 256 VGPRs — and a spill reload is a memory operation that waits
// (vmcnt(0)) for every load issued before it: the next sample's prefetch then paid two or three HBM round trips back to back, ~3 000
// cycles per sample (round 6: profiles/r06/infer_layer_phase_timeline.txt, phase 7 -> 8).
template <typename T> __device__ __forceinline__ const T* fl_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ T* fl_at(void* base, unsigned byte_off) {
    return reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}
// LDS-DMA piece: 64 lanes x 16 B of global memory -> lds_dst + 16 lane, no registers in between (the idiom of mpconv_bwd_ws.hip)
__device__ __forceinline__ void fl_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// sum / max over the 16 lanes of a DPP row (the 16 nodes of a tile)
__device__ __forceinline__ float fl_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ float fl_row_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
    return v;
}

// D[i = out channel][j = node] = W (A operand, resident) x image rows (B operand), K = 64: ALL tiles of this wave at once —
// operand loads first, then the first k-step of every tile, then the second (independent MFMA chains).  WIDE: the 96-row
// images (tiles hf, hf + 2, hf + 4); otherwise the 48-row ones (tiles hf, hf + 2 where < 3; the rest stay 0).
template <bool WIDE>
__device__ __forceinline__ void fl_tiles(const fl_bf16x8 (&aW)[2], const uint16_t* img, int hf, int li, int lk, f32x4 (&acc)[3]) {
    constexpr int NT = WIDE ? 3 : 2;
    uint4 x[NT][2];
    int base = (hf * 16 + li) * FL_XS + 8 * lk;       // recomputed per call (opaque): the tiles are constant offsets from it, and
    asm volatile("" : "+v"(base));                     // a hoisted address per image and tile would cost ~40 registers
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (WIDE || hf + 2 * i < 3) {
            const uint16_t* bp = img + base + 2 * i * 16 * FL_XS;
            x[i][0] = *reinterpret_cast<const uint4*>(bp);
            x[i][1] = *reinterpret_cast<const uint4*>(bp + 32);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (WIDE || hf + 2 * i < 3)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aW[ks], __builtin_bit_cast(fl_bf16x8, x[i][ks]), acc[i], 0, 0, 0);
}

// gather + edge-type contraction + max over KC neighbours, then a2 = ReLU(s2 z + t2) as bf16 into `dst` rows (lane <-> channel);
// ND destinations in flight per wave (M = 8 waves x ND x iterations).  `off_s` holds the BYTE offset of every edge's source row in the
// projection image (the tables are shared by the batch: formed once per launch).  A wave's edge list is uniform, so its offsets and edge
// weights are BROADCAST LDS reads (every lane the same address, immediate offsets from one base register) — round 6: fetched by the
// first lanes and spread with v_readlane they cost 3 readlanes + their hazard no-ops + a scalar multiply per edge, ~11 issue slots
// for 2 dot products and a max; the gathers were 9 300 of a sample's 28 600 cycles and the kernel is VALU-issue-bound.
template <int KC, int ND>
__device__ __forceinline__ void fl_gather(const uint16_t* ps, const int* off_s, const uint2* et_s, uint16_t* dst, int M, int wave, int lane,
                                          float c2s, float c2t) {
    const unsigned char* pc = reinterpret_cast<const unsigned char*>(ps) + lane * 8;
    for (int m0 = wave; m0 < M; m0 += 8 * ND) {
        const int* ob = off_s + m0 * KC;
        const uint2* eb = et_s + m0 * KC;
        int off[ND][KC];
        uint2 ew[ND][KC];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int j = 0; j < KC; ++j) off[d][j] = ob[8 * d * KC + j];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int j = 0; j < KC; ++j) ew[d][j] = eb[8 * d * KC + j];
        uint2 pk[ND][KC];
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int d = 0; d < ND; ++d) pk[d][j] = *reinterpret_cast<const uint2*>(pc + off[d][j]);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            float best = 0.f;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                float v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fl_bf16x2, pk[d][j].x), __builtin_bit_cast(fl_bf16x2, ew[d][j].x), 0.f, false);
                v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fl_bf16x2, pk[d][j].y), __builtin_bit_cast(fl_bf16x2, ew[d][j].y), v, false);
                best = j == 0 ? v : fmaxf(best, v);
            }
            const __bf16 h = (__bf16)fmaxf(fmaf(best, c2s, c2t), 0.f);
            dst[(m0 + 8 * d) * FL_XS + lane] = __builtin_bit_cast(uint16_t, h);
        }
    }
}

__global__ __launch_bounds__(FL_THREADS) void factor_layer_fwd_kernel(const FlParams p) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int ot = wave & 3, hf = wave >> 2;          // this wave's 16-channel tile of every node-wise product; node tiles hf, hf + 2, hf + 4

    // ---- LDS map ----
    uint16_t* vs = reinterpret_cast<uint16_t*>(fl_lds);                    // [96][XS] variables
    uint16_t* fs = vs + FL_NV * FL_XS;                                     // [48][XS] parity checks
    uint16_t* as_ = fs + FL_NF * FL_XS;                                    // [96][XS] image A: conv1 output (operator input)
    uint16_t* bs = as_ + FL_NV * FL_XS;                                    // [96][XS] image B: operator output (conv2 input)
    uint16_t* ps = bs + FL_NV * FL_XS;                                     // [96][PS] projection
    int* idx_vf = reinterpret_cast<int*>(ps + FL_NV * FL_PS);              // [48 * 6]
    int* idx_fv = idx_vf + FL_NF * FL_KF;                                  // [96 * 3]
    uint2* et_vf = reinterpret_cast<uint2*>(idx_fv + FL_NV * FL_KV);       // [288] 4 x bf16
    uint2* et_fv = et_vf + FL_NF * FL_KF;                                  // [288]
    float* het_in = reinterpret_cast<float*>(et_fv + FL_NV * FL_KV);       // [96]
    float* het_out = het_in + FL_NV;                                       // [96]
    float* red = het_out + FL_NV;                                          // [2 maps][2 stats][2 halves][64]
    float* zpart = red + 512;                                              // [2 halves][64] fan-in maxima
    float* hv = zpart + 128;                                               // [64] hyper-factor state, [64] s2 * P, [64] t2, [64] spare
    float* mvp = hv + 256;                                                 // [8 waves][64] partial matrix-vector products
    float* aff = mvp + 512;                                                 // [4 blocks][s1, t1, s2, t2, s3, t3][64] folded affines
    uint4* wmap = reinterpret_cast<uint4*>(aff + 24 * 64);                 // [2 maps][2 k-steps][4 tiles][64 lanes] fragments of Wvv / Wff
    uint16_t* W2c = reinterpret_cast<uint16_t*>(wmap + 2 * 2 * 4 * 64);    // [o][c] bf16: conv2 of the fan-in block, transposed
    uint16_t* W1d = W2c + 64 * 64;                                         // [c][o] bf16: conv1 of the fan-out block, transposed
    uint16_t* Fd = W1d + 64 * 64;                                          // [o][o2] bf16: its operator filters

    const float* w = p.w;
    const float* b0 = w + FL_OFF_B0;      // V -> F parity
    const float* b1 = w + FL_OFF_B1;      // F -> V parity
    const float* b2 = w + FL_OFF_B2;      // V -> F hyper (fan-in)
    const float* b3 = w + FL_OFF_B3;      // F -> V hyper (fan-out)

    // ---- resident fragments: A[i = out channel][k = in channel] = W[out][in] ----
    fl_bf16x8 aW1a[2], aW2a[2], aW1b[2], aW2b[2], aW1c[2], aW2d[2], aFa[2][2], aFb[2][2], aFc[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int o = (ot * 16 + li) * 64 + 32 * ks + 8 * lk;
        if (hf == 0) {                                  // the maps' fragments wait in LDS, in fragment order (used once per sample)
            wmap[((0 * 2 + ks) * 4 + ot) * 64 + lane] = __builtin_bit_cast(uint4, fl_frag8(w + FL_OFF_WVV + o));
            wmap[((1 * 2 + ks) * 4 + ot) * 64 + lane] = __builtin_bit_cast(uint4, fl_frag8(w + FL_OFF_WFF + o));
        }
        aW1a[ks] = fl_frag8(b0 + FL_W1 + o);
        aW2a[ks] = fl_frag8(b0 + FL_W2(256) + o);
        aW1b[ks] = fl_frag8(b1 + FL_W1 + o);
        aW2b[ks] = fl_frag8(b1 + FL_W2(256) + o);
        aW1c[ks] = fl_frag8(b2 + FL_W1 + o);
        aW2d[ks] = fl_frag8(b3 + FL_W2(64) + o);
    }
    // operator filters: A[i = column][k = c] = F[c][column]; parity: column slabs wave and wave + 8 of 16; fan-in: slab ot of 4
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float wa[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                wa[u] = b0[FL_F + (32 * ks + 8 * lk + u) * 256 + (wave + 8 * q) * 16 + li];
                wb[u] = b1[FL_F + (32 * ks + 8 * lk + u) * 256 + (wave + 8 * q) * 16 + li];
            }
            aFa[q][ks] = __builtin_bit_cast(fl_bf16x8, make_uint4(fl_pack2(wa[0], wa[1]), fl_pack2(wa[2], wa[3]), fl_pack2(wa[4], wa[5]), fl_pack2(wa[6], wa[7])));
            aFb[q][ks] = __builtin_bit_cast(fl_bf16x8, make_uint4(fl_pack2(wb[0], wb[1]), fl_pack2(wb[2], wb[3]), fl_pack2(wb[4], wb[5]), fl_pack2(wb[6], wb[7])));
        }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float wc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wc[u] = b2[FL_F + (32 * ks + 8 * lk + u) * 64 + ot * 16 + li];
        aFc[ks] = __builtin_bit_cast(fl_bf16x8, make_uint4(fl_pack2(wc[0], wc[1]), fl_pack2(wc[2], wc[3]), fl_pack2(wc[4], wc[5]), fl_pack2(wc[6], wc[7])));
    }
    // ---- LDS-resident: the hyper-factor's matrix-vector maps, the shared tables ----
    for (int f = tid; f < 64 * 64; f += FL_THREADS) {
        const int r = f >> 6, c = f & 63;
        const __bf16 h2 = (__bf16)b2[FL_W2(64) + f];                       // W2c[out r][in c] -> [in c][out r]
        W2c[c * 64 + r] = __builtin_bit_cast(uint16_t, h2);
        const __bf16 h1 = (__bf16)b3[FL_W1 + f];                           // W1d[out r][in c] -> [in c][out r]
        W1d[c * 64 + r] = __builtin_bit_cast(uint16_t, h1);
        const __bf16 hf2 = (__bf16)b3[FL_F + f];                           // Fd[in r][out c] as in memory
        Fd[f] = __builtin_bit_cast(uint16_t, hf2);
    }
    for (int t = tid; t < FL_NF * FL_KF; t += FL_THREADS) {
        const int m = t / FL_KF, j = t - m * FL_KF;
        const long long v = p.idx_v2f[(int64_t)m * p.idx_v2f_sm + (int64_t)j * p.idx_v2f_sk];
        idx_vf[t] = (int)(v < 0 ? 0 : (v >= FL_NV ? FL_NV - 1 : v)) * (FL_PS * 2);      // byte offset of the source's projection row
    }
    for (int t = tid; t < FL_NV * FL_KV; t += FL_THREADS) {
        const int m = t / FL_KV, j = t - m * FL_KV;
        const long long v = p.idx_f2v[(int64_t)m * p.idx_f2v_sm + (int64_t)j * p.idx_f2v_sk];
        idx_fv[t] = (int)(v < 0 ? 0 : (v >= FL_NF ? FL_NF - 1 : v)) * (FL_PS * 2);
    }
    for (int f = tid; f < 24 * 64; f += FL_THREADS) {
        const int row = f >> 6, c = f & 63, blk = row / 6, which = row - blk * 6;
        const int ncol = blk < 2 ? 256 : 64;
        const float* bb = blk == 0 ? b0 : (blk == 1 ? b1 : (blk == 2 ? b2 : b3));
        const int off = which == 0 ? FL_S1 : which == 1 ? FL_T1 : which == 2 ? FL_F + 64 * ncol : which == 3 ? FL_F + 64 * ncol + 64
                        : which == 4 ? FL_F + 64 * ncol + 128 + 64 * 64 : FL_F + 64 * ncol + 192 + 64 * 64;
        aff[f] = bb[off + c];
    }
    for (int t = tid; t < FL_NV; t += FL_THREADS) {
        het_in[t] = p.het_v2f ? fl_lo(p.het_v2f[t]) : 1.f;
        het_out[t] = p.het_f2v ? fl_lo(p.het_f2v[t]) : 1.f;
    }

    // per-lane affines of the wave's conv tiles (channels ot * 16 + 4 lk + r) and of the gathers (lane <-> channel)
    const int ch4 = ot * 16 + 4 * lk;
    const float slope = p.slope;

    // ---- the next sample's inputs.  Its state (768 + 384 16-byte chunks) goes by LDS-DMA into the projection image, which is dead from
    // the end of the F -> V gather to the next sample's first projection — round 6: held in registers across the last three phases
    // (12 VGPRs of a kernel at its 256) the data or the pointers were spilled, and a spill is a memory operation that waits for every
    // load before it: two or three HBM round trips back to back, ~3 000 cycles per sample.  The 288 + 288 edge-type quads and the 64
    // hyper-factor channels stay in registers (5) ----
    uint2 e0r, e1r;
    uint16_t hr = 0, skh = 0;
    const unsigned lds_stage = (unsigned)(uintptr_t)ps;
    auto prefetch = [&](int b) {
        const unsigned char* vb = reinterpret_cast<const unsigned char*>(p.var + (int64_t)b * FL_NV * 64);      // (uniform bases: scalar registers)
        const unsigned char* fb = reinterpret_cast<const unsigned char*>(p.fac0 + (int64_t)b * FL_NF * 64);
        const uint16_t* e0b = p.et_v2f + (int64_t)b * p.et_v2f_sb;
        const uint16_t* e1b = p.et_f2v + (int64_t)b * p.et_f2v_sb;
        const uint16_t* hb = p.fac1 + (int64_t)b * 64;
        unsigned t = (unsigned)tid, l16 = (unsigned)lane * 16u;
        asm volatile("" : "+v"(t), "+v"(l16));                          // the lane offsets are formed here, not carried across the loop
        if (wave < 6) {                                                 // 12 pieces of the variables' state, 6 of the checks'
#pragma unroll
            for (int u = 0; u < 2; ++u) fl_dma16(vb + (wave * 2 + u) * 1024 + l16, lds_stage + (unsigned)((wave * 2 + u) * 1024));
        } else {
#pragma unroll
            for (int u = 0; u < 3; ++u) fl_dma16(fb + ((wave - 6) * 3 + u) * 1024 + l16, lds_stage + (unsigned)(12288 + ((wave - 6) * 3 + u) * 1024));
        }
        // UNCONDITIONAL loads (threads beyond an array re-read its first element; commit() stores under the conditions): a load in one
        // arm of a branch meets the other arm's value at the merge, and the wave waits for it there
        e0r = *fl_at<uint2>(e0b, tid < FL_NF * FL_KF ? t * 8u : 0u);
        e1r = *fl_at<uint2>(e1b, tid < FL_NV * FL_KV ? t * 8u : 0u);
        hr = *fl_at<uint16_t>(hb, (t & 63u) * 2u);
    };
    auto commit = [&]() {                              // (behind a barrier that follows `s_waitcnt vmcnt(0)` on every wave: the DMA has landed)
        const uint4* st = reinterpret_cast<const uint4*>(ps);
        int t = tid;
        asm volatile("" : "+v"(t));                    // (addresses formed here: carried across the loop they were spilled)
        const int row = (t >> 3) * FL_XS + (t & 7) * 8;
        *reinterpret_cast<uint4*>(vs + row) = st[t];
        if (tid < 256) *reinterpret_cast<uint4*>(vs + 64 * FL_XS + row) = st[512 + t];
        if (tid < 384) *reinterpret_cast<uint4*>(fs + row) = st[768 + t];
        if (tid < FL_NF * FL_KF) et_vf[tid] = e0r;
        if (tid < FL_NV * FL_KV) et_fv[tid] = e1r;
        if (tid < 64) hv[tid] = fl_lo(hr);
    };

    int b = blockIdx.x;
    if (b < p.B) prefetch(b);
    auto close_hyper = [&](int bp) {                   // wave 0: new state of sample bp's hyper-factor (hv still holds its old state)
        float acc = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) acc += mvp[w8 * 64 + lane];
        float u = fmaf(acc, aff[(2 * 6 + 4) * 64 + lane], aff[(2 * 6 + 5) * 64 + lane]);
        u = u > 0.f ? u : u * slope;
        if (p.residual) u += hv[lane];
        unsigned l2 = (unsigned)lane * 2u;
        asm volatile("" : "+v"(l2));
        if (p.skip_fac1) u += fl_lo(skh);                 // (requested with the sample's other skip terms, a phase before the sample ended)
        const __bf16 h = (__bf16)u;
        *fl_at<uint16_t>(p.out_fac1 + (int64_t)bp * 64, l2) = __builtin_bit_cast(uint16_t, h);
    };
    int b_prev = -1;
    for (; b < p.B; b += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA pieces of the sample have landed
        __syncthreads();                               // the previous sample's last readers of the images are done (and the tables are in)
        FL_STAMP(0);
        if (wave == 0 && b_prev >= 0) close_hyper(b_prev);     // before this wave's commit() overwrites hv[0..63]
        b_prev = b;
        commit();
        __syncthreads();
        FL_STAMP(1);

        f32x4 oV[3], oF[2];                            // the new state of this lane's elements: tiles (ot, hf + 2 i)

        // folded BatchNorm + LeakyReLU of a tile's four channels; rows srow (scale), srow + 1 (shift) of the affine table
        auto affine = [&](int srow, f32x4& s4, f32x4& t4) {
            int o = srow * 64 + ch4;
            asm volatile("" : "+v"(o));                                     // re-read per phase: 16 hoisted vectors would spill
            s4 = *reinterpret_cast<const f32x4*>(aff + o);
            t4 = *reinterpret_cast<const f32x4*>(aff + o + 64);
        };
        auto act4 = [&](const f32x4& acc, const f32x4& s4, const f32x4& t4, float (&v)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[r], s4[r], t4[r]); v[r] = u > 0.f ? u : u * slope; }
        };
        // conv1 of a block: image rows -> image A (bf16)
        auto conv1 = [&](auto wide, const fl_bf16x8 (&aW)[2], const uint16_t* img, int blk) {
            constexpr bool WIDE = decltype(wide)::value;
            f32x4 acc[3], s4, t4;
            fl_tiles<WIDE>(aW, img, hf, li, lk, acc);
            affine(blk * 6, s4, t4);
            int sb = (hf * 16 + li) * FL_XS + ch4;
            asm volatile("" : "+v"(sb));
#pragma unroll
            for (int i = 0; i < (WIDE ? 3 : 2); ++i) {
                if (WIDE || hf + 2 * i < 3) {
                    float v[4];
                    act4(acc[i], s4, t4, v);
                    *reinterpret_cast<uint2*>(as_ + sb + 2 * i * 16 * FL_XS) = make_uint2(fl_pack2(v[0], v[1]), fl_pack2(v[2], v[3]));
                }
            }
        };
        // projection of image A onto the 256 operator columns -> P (bf16): three node tiles x two column slabs in flight
        auto project = [&](const fl_bf16x8 (&aF)[2][2], int ntile) {
            for (int n0 = 0; n0 < ntile; n0 += 3) {
                uint4 x[3][2];
                int rb = (n0 * 16 + li) * FL_XS + 8 * lk, wb = (n0 * 16 + li) * FL_PS + wave * 16 + 4 * lk;
                asm volatile("" : "+v"(rb), "+v"(wb));
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const uint16_t* bp = as_ + rb + t * 16 * FL_XS;
                    x[t][0] = *reinterpret_cast<const uint4*>(bp);
                    x[t][1] = *reinterpret_cast<const uint4*>(bp + 32);
                }
                f32x4 acc[3][2];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[q][ks], __builtin_bit_cast(fl_bf16x8, x[t][ks]), acc[t][q], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        *reinterpret_cast<uint2*>(ps + wb + t * 16 * FL_PS + 8 * q * 16) =
                            make_uint2(fl_pack2(acc[t][q][0], acc[t][q][1]), fl_pack2(acc[t][q][2], acc[t][q][3]));
            }
        };
        const std::integral_constant<bool, true> wide_img;
        const std::integral_constant<bool, false> narrow_img;

        // ---- phase 1: the node-wise maps, InstanceNorm over the nodes, ReLU; conv1 of the V -> F parity block ----
        {
            f32x4 yv[3], yf[2];
            fl_bf16x8 aVV[2], aFF[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                aVV[ks] = __builtin_bit_cast(fl_bf16x8, wmap[((0 * 2 + ks) * 4 + ot) * 64 + lane]);
                aFF[ks] = __builtin_bit_cast(fl_bf16x8, wmap[((1 * 2 + ks) * 4 + ot) * 64 + lane]);
            }
            float sv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f}, sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
            {
                f32x4 tf[3];
                fl_tiles<true>(aVV, vs, hf, li, lk, yv);
                fl_tiles<false>(aFF, fs, hf, li, lk, tf);
                yf[0] = tf[0]; yf[1] = tf[1];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sv[r] += yv[i][r]; qv[r] = fmaf(yv[i][r], yv[i][r], qv[r]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)                    // tiles this wave does not own are exactly 0
#pragma unroll
                for (int r = 0; r < 4; ++r) { sf[r] += yf[i][r]; qf[r] = fmaf(yf[i][r], yf[i][r], qf[r]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[r] = fl_row_sum(sv[r]); qv[r] = fl_row_sum(qv[r]); sf[r] = fl_row_sum(sf[r]); qf[r] = fl_row_sum(qf[r]);
            }
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(0 * 2 + hf) * 64 + ch4 + r] = sv[r];
                    red[(2 + 0 * 2 + hf) * 64 + ch4 + r] = qv[r];
                    red[(4 + 0 * 2 + hf) * 64 + ch4 + r] = sf[r];
                    red[(6 + 0 * 2 + hf) * 64 + ch4 + r] = qf[r];
                }
            }
            {   // the fan-out block opens on the hyper-factor's own state: conv1 is a matrix-vector product, split over the
                // waves by input channel (lane <-> output channel); the partials meet after the barrier
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) part = fmaf(hv[8 * wave + i], fl_lo(W1d[(8 * wave + i) * 64 + lane]), part);
                mvp[wave * 64 + lane] = part;
            }
            __syncthreads();
            FL_STAMP(2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ch4 + r;
                const float mv = (red[c] + red[64 + c]) * (1.f / FL_NV);
                const float vv = fmaxf((red[128 + c] + red[192 + c]) * (1.f / FL_NV) - mv * mv, 0.f);
                const float rv = rsqrtf(vv + FL_EPS);
                const float mf = (red[256 + c] + red[320 + c]) * (1.f / FL_NF);
                const float vf = fmaxf((red[384 + c] + red[448 + c]) * (1.f / FL_NF) - mf * mf, 0.f);
                const float rf = rsqrtf(vf + FL_EPS);
#pragma unroll
                for (int i = 0; i < 3; ++i) oV[i][r] = fmaxf((yv[i][r] - mv) * rv, 0.f);
#pragma unroll
                for (int i = 0; i < 2; ++i) oF[i][r] = fmaxf((yf[i][r] - mf) * rf, 0.f);
            }
        }
        float fo_a1;                                   // the fan-out block's a1 (lane <-> channel), every wave its own copy
        {
            float a1 = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) a1 += mvp[w8 * 64 + lane];
            a1 = fmaf(a1, aff[(3 * 6 + 0) * 64 + lane], aff[(3 * 6 + 1) * 64 + lane]);
            fo_a1 = fl_round(a1 > 0.f ? a1 : a1 * slope);
        }

        conv1(wide_img, aW1a, vs, 0);
        __syncthreads();
        project(aFa, 6);
        {   // its projection, again split over the waves by input channel
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                part = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(fo_a1), 8 * wave + i)), fl_lo(Fd[(8 * wave + i) * 64 + lane]), part);
            mvp[wave * 64 + lane] = part;
        }
        __syncthreads();
        FL_STAMP(3);
        if (wave == 0) {
            float P = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) P += mvp[w8 * 64 + lane];
            P = fl_round(P);
            hv[64 + lane] = aff[(3 * 6 + 2) * 64 + lane] * P;     // a2[m][o] = ReLU(et[m] * (s2 P)[o] + t2[o])
            hv[128 + lane] = aff[(3 * 6 + 3) * 64 + lane];
        }
        fl_gather<FL_KF, 3>(ps, idx_vf, et_vf, bs, FL_NF, wave, lane, aff[(0 * 6 + 2) * 64 + lane], aff[(0 * 6 + 3) * 64 + lane]);
        __syncthreads();
        FL_STAMP(4);

        // ---- phase 5: its conv2 onto the checks; conv1 of the F -> V parity block ----
        {
            f32x4 acc[3], s4, t4;
            fl_tiles<false>(aW2a, bs, hf, li, lk, acc);
            affine(0 * 6 + 4, s4, t4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4];
                act4(acc[i], s4, t4, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) oF[i][r] += v[r];     // (tiles this wave does not own carry garbage nobody stores)
            }
        }
        conv1(narrow_img, aW1b, fs, 1);
        __syncthreads();
        FL_STAMP(5);
        // ---- phase 6-7: F -> V projection and gather ----
        project(aFb, 3);
        __syncthreads();
        FL_STAMP(6);
        fl_gather<FL_KV, 4>(ps, idx_fv, et_fv, bs, FL_NV, wave, lane, aff[(1 * 6 + 2) * 64 + lane], aff[(1 * 6 + 3) * 64 + lane]);
        __syncthreads();
        FL_STAMP(7);

        if (b + (int)gridDim.x < p.B) prefetch(b + gridDim.x);        // late: the registers are live for the hyper-factor phases only
        // ---- phase 8: its conv2 onto the variables; conv1 of the fan-in block ----
        {
            f32x4 acc[3], s4, t4;
            fl_tiles<true>(aW2b, bs, hf, li, lk, acc);
            affine(1 * 6 + 4, s4, t4);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float v[4];
                act4(acc[i], s4, t4, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) oV[i][r] += v[r];
            }
        }
        conv1(wide_img, aW1c, vs, 2);
        __syncthreads();
        FL_STAMP(8);

        // ---- phase 9: fan-in projection (64 columns: slab ot) and the max over the 96 variables; the fan-out block's operator
        //      output on the 96 variables -> image B ----
        {
            float zm[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            f32x4 acc[3];
            fl_tiles<true>(aFc, as_, hf, li, lk, acc);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float e = het_in[(hf + 2 * i) * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) zm[r] = fmaxf(zm[r], e * fl_round(acc[i][r]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) zm[r] = fl_row_max(zm[r]);
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zpart[hf * 64 + ch4 + r] = zm[r];
            }
        }
        for (int f = tid; f < FL_NV * 32; f += FL_THREADS) {
            const int m = f >> 5, c2 = (f & 31) * 2;
            const float e = het_out[m];
            const float v0 = fmaxf(fmaf(e, hv[64 + c2], hv[128 + c2]), 0.f), v1 = fmaxf(fmaf(e, hv[65 + c2], hv[129 + c2]), 0.f);
            *reinterpret_cast<unsigned*>(bs + m * FL_XS + c2) = fl_pack2(v0, v1);
        }
        __syncthreads();
        FL_STAMP(9);

        // the skip link's terms of this lane's elements: asked for here, consumed in the last phase
        uint2 skv[3] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)}, skf[2] = {make_uint2(0, 0), make_uint2(0, 0)};
        unsigned eo = (unsigned)(((hf * 16 + li) * 64 + ch4) * 2);     // this lane's element of tile (ot, hf) in a [nodes][64] bf16 state: + 4096 per tile step
        asm volatile("" : "+v"(eo));
        {   // UNCONDITIONAL loads (without a skip link: the layer's own input, never used): a load inside `if (skip)` meets the zero of the
            // other arm at the branch merge, i.e. the wave waits for it right there instead of behind the last block's products
            const uint16_t* sv_ = (p.skip_var ? p.skip_var : p.var) + (int64_t)b * FL_NV * 64;
            const uint16_t* sf_ = (p.skip_fac0 ? p.skip_fac0 : p.fac0) + (int64_t)b * FL_NF * 64;
#pragma unroll
            for (int i = 0; i < 3; ++i) skv[i] = *fl_at<uint2>(sv_, eo + 4096u * i);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (hf + 2 * i < 3) skf[i] = *fl_at<uint2>(sf_, eo + 4096u * i);
            unsigned l2 = (unsigned)lane * 2u;
            asm volatile("" : "+v"(l2));
            skh = *fl_at<uint16_t>((p.skip_fac1 ? p.skip_fac1 : p.fac1) + (int64_t)b * 64, l2);
        }
        // ---- phase 10: the fan-out block's conv2 onto the variables; residual, skip link, store; wave 0 first closes the
        //      fan-in block (the hyper-factor's new state: a matrix-vector product) ----
        {   // the fan-in block closes with a matrix-vector product on the hyper-factor: partials by input channel here, summed by
            // wave 0 behind the next barrier (the top of the next sample, or the one after the loop)
            float z = fmaxf(zpart[lane], zpart[64 + lane]);
            z = fl_round(fmaxf(fmaf(z, aff[(2 * 6 + 2) * 64 + lane], aff[(2 * 6 + 3) * 64 + lane]), 0.f));
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                part = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 8 * wave + i)), fl_lo(W2c[(8 * wave + i) * 64 + lane]), part);
            mvp[wave * 64 + lane] = part;
        }
        int rsb;
        {
            f32x4 acc[3], s4, t4;
            fl_tiles<true>(aW2d, bs, hf, li, lk, acc);
            affine(3 * 6 + 4, s4, t4);
            asm volatile("" : "+v"(eo));                // (the stores' lane offset stays a 32-bit offset beside a scalar base)
            rsb = (hf * 16 + li) * FL_XS + ch4;         // the residual's element of tile (ot, hf) in the LDS images, formed here (not carried)
            asm volatile("" : "+v"(rsb));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int n = (hf + 2 * i) * 16 + li;
                float v[4];
                act4(acc[i], s4, t4, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += oV[i][r];
                if (p.residual) {
                    const uint2 a = *reinterpret_cast<const uint2*>(vs + rsb + 2 * i * 16 * FL_XS);
                    v[0] += fl_lo(a.x); v[1] += fl_hi(a.x); v[2] += fl_lo(a.y); v[3] += fl_hi(a.y);
                }
                if (p.skip_var) { v[0] += fl_lo(skv[i].x); v[1] += fl_hi(skv[i].x); v[2] += fl_lo(skv[i].y); v[3] += fl_hi(skv[i].y); }
                *fl_at<uint2>(p.out_var + (int64_t)b * FL_NV * 64, eo + 4096u * i) = make_uint2(fl_pack2(v[0], v[1]), fl_pack2(v[2], v[3]));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (hf + 2 * i < 3) {
                const int n = (hf + 2 * i) * 16 + li;
                float v[4] = {oF[i][0], oF[i][1], oF[i][2], oF[i][3]};
                if (p.residual) {
                    const uint2 a = *reinterpret_cast<const uint2*>(fs + rsb + 2 * i * 16 * FL_XS);
                    v[0] += fl_lo(a.x); v[1] += fl_hi(a.x); v[2] += fl_lo(a.y); v[3] += fl_hi(a.y);
                }
                if (p.skip_fac0) { v[0] += fl_lo(skf[i].x); v[1] += fl_hi(skf[i].x); v[2] += fl_lo(skf[i].y); v[3] += fl_hi(skf[i].y); }
                *fl_at<uint2>(p.out_fac0 + (int64_t)b * FL_NF * 64, eo + 4096u * i) = make_uint2(fl_pack2(v[0], v[1]), fl_pack2(v[2], v[3]));
            }
        }
        FL_STAMP(10);
    }
    __syncthreads();
    if (wave == 0 && b_prev >= 0) close_hyper(b_prev);
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
extern "C" int64_t fgnn_factor_layer_param_count(void) { return FL_PARAM_LEN; }

// One FactorNN layer (64 -> 64) of the LDPC model, inference.  States are bf16 channel-fastest: var [B][96][64],
// fac0 [B][48][64], fac1 [B][64]; skip_* are a skip link's terms in the same layouts (NULL = none); idx_v2f [48][6] /
// idx_f2v [96][3] are the neighbour tables SHARED by the batch (strides in elements); et_* the parity edge types,
// [B][M][k][4] with the given batch strides; het_* the hyper-factor's per-edge weights ([96], shared; NULL = ones);
// `params` the fgnn_factor_layer_param_count() packed floats (maps, then per block W1 [64][64], s1, t1, F [64][ncol], s2, t2,
// W2 [64][64], s3, t3 with the biases and eval-mode BatchNorms folded into the s / t affines).
extern "C" int fgnn_factor_layer_forward(int32_t B, const void* var, const void* fac0, const void* fac1, const void* skip_var,
                                         const void* skip_fac0, const void* skip_fac1, const int64_t* idx_v2f, int32_t idx_v2f_sm,
                                         int32_t idx_v2f_sk, const int64_t* idx_f2v, int32_t idx_f2v_sm, int32_t idx_f2v_sk,
                                         const void* et_v2f, int64_t et_v2f_sb, const void* et_f2v, int64_t et_f2v_sb,
                                         const void* het_v2f, const void* het_f2v, const float* params, int32_t residual,
                                         float slope, void* out_var, void* out_fac0, void* out_fac1, fgnn_stream_t stream) {
    if (!var || !fac0 || !fac1 || !idx_v2f || !idx_f2v || !et_v2f || !et_f2v || !params || !out_var || !out_fac0 || !out_fac1)
        FGNN_FAIL(FGNN_EINVAL, "factor_layer_forward: null pointer");
    if (B < 0) FGNN_FAIL(FGNN_EINVAL, "factor_layer_forward: B = %d", B);
    if (((uintptr_t)var & 15) || ((uintptr_t)fac0 & 15) || ((uintptr_t)et_v2f & 7) || ((uintptr_t)et_f2v & 7) || (et_v2f_sb & 3) ||
        (et_f2v_sb & 3) || ((uintptr_t)out_var & 7) || ((uintptr_t)out_fac0 & 7) || ((uintptr_t)skip_var & 7) || ((uintptr_t)skip_fac0 & 7) ||
        ((uintptr_t)params & 15))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "factor_layer_forward: misaligned operand");
    if (B == 0) return FGNN_OK;
    FlParams p;
    p.var = (const uint16_t*)var; p.fac0 = (const uint16_t*)fac0; p.fac1 = (const uint16_t*)fac1;
    p.skip_var = (const uint16_t*)skip_var; p.skip_fac0 = (const uint16_t*)skip_fac0; p.skip_fac1 = (const uint16_t*)skip_fac1;
    p.idx_v2f = idx_v2f; p.idx_f2v = idx_f2v; p.et_v2f = (const uint16_t*)et_v2f; p.et_f2v = (const uint16_t*)et_f2v;
    p.het_v2f = (const uint16_t*)het_v2f; p.het_f2v = (const uint16_t*)het_f2v; p.w = params;
    p.out_var = (uint16_t*)out_var; p.out_fac0 = (uint16_t*)out_fac0; p.out_fac1 = (uint16_t*)out_fac1;
    p.et_v2f_sb = et_v2f_sb; p.et_f2v_sb = et_f2v_sb; p.B = B; p.residual = residual; p.slope = slope;
    p.idx_v2f_sm = idx_v2f_sm; p.idx_v2f_sk = idx_v2f_sk; p.idx_f2v_sm = idx_f2v_sm; p.idx_f2v_sk = idx_f2v_sk;
    const int lds = (FL_NV + FL_NF + 2 * FL_NV) * FL_XS * 2 + FL_NV * FL_PS * 2 + (FL_NF * FL_KF + FL_NV * FL_KV) * (4 + 8) +
                    2 * FL_NV * 4 + (512 + 128 + 256 + 512 + 24 * 64) * 4 + 2 * 2 * 4 * 64 * 16 + 3 * 64 * 64 * 2;
    hipError_t e = hipFuncSetAttribute((const void*)factor_layer_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    int grid = 256;
    if (grid > B) grid = B;
    fgnn_note_kernel("factor_layer_fwd_kernel");
    p.prof = nullptr;
#ifdef FGNN_ENABLE_PROF
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 128 * 8);
        (void)hipMemset(prof_buf, 0, 128 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel((const void*)factor_layer_fwd_kernel, dim3(grid), dim3(FL_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "factor_layer_forward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {                                     // tuning aid: the fourth sample of workgroup 0, shader clocks after every barrier
        long long h[128];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[fgnn prof layer] wave %d:", w);
            for (int i = 0; i < 12; ++i) fprintf(stderr, " %lld", h[w * 16 + i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return FGNN_OK;
}
