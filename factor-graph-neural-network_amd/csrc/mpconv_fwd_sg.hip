// mpconv_fwd_sg.hip — second-generation bf16 forward of the VF/FV message operator for the LDPC parity-check calls
// (NO_EXTENSION, 4 edge types, max aggregation, fixed degree 3 / 6; reference: /root/reference/lib/model/mpnn/
// mp_nn.py:115-134,160-175).  Same math and rounding points as mpconv_fwd_b16.hip (x, etype, filters and the projected
// rows P are bf16, every sum is f32, the output is rounded to bf16 once); what changed is the instruction stream.
//
// rocprofv3 SQ counters of the first-generation kernel (profiles/r01/pmc_issue_counters.json) show its four waves per
// SIMD ISSUING for ~100 % of the kernel's duration (626 VALU + 300 SALU per wave and sample): the forward is bound by
// instruction issue, not by HBM, LDS or the matrix cores.  This kernel removes every instruction that is not arithmetic
// on a message:
//   * neighbour tables: the LDS address of every P row a wave will ever read (destination x neighbour slot) is computed
//     ONCE and kept in registers for the kernel's lifetime when the graph is shared by the batch (36 VGPRs; re-derived
//     per sample otherwise) — no per-edge v_readlane / s_mul / v_add;
//   * edge-type weights: each wave fetches the contiguous block of its own destinations (<= 288 B) with ONE load per
//     sample, a sample ahead, parks it in a wave-private LDS slot and reads it back as broadcasts (every lane the same
//     address) straight into VGPR operands of v_dot2_f32_bf16 — no workgroup staging pass, no v_readlane.  (Per-
//     destination global loads, even two destinations ahead and L2-warm, left the gather waiting ~600 cycles per
//     destination: profiles/r02.)
//   * 32x32x16 MFMA tiles: half the B-operand LDS reads of the 16x16x32 form, 4 channels x 4 edge types per lane and
//     node, written as four conflict-free 8-byte stores (row stride = 2 banks mod 32);
//   * every shape parameter the LDPC calls fix (nin, nou, degree, epilogue form) is a template argument: no scalar
//     branching or register spilling in the sample loop (the first kernel carried ~100 live SGPRs).
// Per wave and sample: ~280 VALU / ~60 LDS / ~33 VMEM instructions (was 626 / 76 / 15).
//
// Layout: 512-thread workgroups (8 waves), 2 per CU, persistent over samples.  Wave w owns the 32 projection columns
// [32w, 32w+32) (8 channels x 4 edge types, W-stationary A fragments) and the contiguous destinations
// [w*DPW, (w+1)*DPW).  Per sample: x (prefetched into registers one sample ahead) -> LDS image; MFMA projection ->
// P[N][64 ch][4 et] bf16 in LDS; barrier; gather with lane = channel; barrier.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>

typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sg_f32x16 __attribute__((ext_vector_type(16)));

#define SG_THREADS 512
#define SG_WAVES 8
#ifndef SG_OPT_ASMDOT
#define SG_OPT_ASMDOT 0       // 1: seed each message with the three-address v_dot2_f32_bf16 through inline asm (measured SLOWER: 52 vs 41 us)
#endif
#ifndef SG_OPT_MAX3
#define SG_OPT_MAX3 1         // v_max3 + first-occurrence scan instead of a compare/select chain
#endif
#ifndef SG_OPT_EARLYCOMMIT
#define SG_OPT_EARLYCOMMIT 1  // stage the next sample right after the projection barrier
#endif
#define SG_ESLOT 304          // bytes of LDS per wave for its edge-type block (<= 288 used)
#define SG_PSB 520            // P row stride in bytes: 64 channels x 8 B + 8 (130 dwords = 2 banks mod 32)

enum { SG_MODE_TRAIN = 0, SG_MODE_TRAIN_STATS = 1, SG_MODE_AFFINE_RELU = 2, SG_MODE_GENERIC = 3 };

struct SgParams {
    const unsigned short* x;
    const int64_t* idx;
    const unsigned short* et;
    const float* W;
    const float* bias;
    const float* pscale;
    const float* pshift;
    unsigned short* y;
    uint8_t* argmax;
    float* stats;
    int B, N, M, Npad, DPW, relu;
    int y_ld, w_ld, st_ld;                        // row strides (elements) of y / argmax, of W, of a statistics partial row: NOU,
                                                  // 4 NOU, NOU for a whole call; the 64 -> 128 calls run as TWO 64-channel launches
                                                  // over the halves of a 128-wide output (y_ld = st_ld = 128, w_ld = 512)
    long long x_sb, et_sb, y_sb;                  // elements
    long long* prof;                              // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): phase timeline
    FgnnFold fold;                                // fold.tickets != NULL: the last workgroup finalises the BatchNorm statistics (fgnn_gridfold.h)
    fgnn_bn_final fin;
};

extern __shared__ __attribute__((aligned(16))) unsigned char sg_lds[];

// phase-timeline stamps (tuning aid): compiled in only with -DFGNN_ENABLE_PROF, read with FGNN_PROF=1
#ifdef FGNN_ENABLE_PROF
#define SG_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && b == 3 * (int)gridDim.x) p.prof[wave * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SG_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ unsigned sg_pack(float a, float b) {
    sg_bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float sg_dot2(unsigned p, unsigned e, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, p), __builtin_bit_cast(sg_bf16x2, e), acc, false);
}
// D = a.lo*b.lo + a.hi*b.hi + c as the three-address VOP3P form: the builtin always selects the two-address v_dot2c
// (accumulator tied to the destination) and pays a v_mov per message to seed it.  Operands are LDS / register values
// the compiler tracks; the instruction has no software-visible hazards of its own.
__device__ __forceinline__ float sg_dot2_seed(unsigned p, unsigned e, float c) {
#if SG_OPT_ASMDOT
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(e), "v"(c));
    return r;
#else
    return sg_dot2(p, e, c);
#endif
}

// edge-type weights of one destination: KC x 4 bf16 = KC*8 bytes from this wave's LDS slot, every lane the same address
template <int KC>
struct SgEt {
    unsigned w[2 * KC];
    __device__ __forceinline__ void load(const unsigned char* p) {
        if constexpr (KC == 6) {
            const uint4* q = reinterpret_cast<const uint4*>(p);
            const uint4 a = q[0], b = q[1], c = q[2];
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
            w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
        } else {
            static_assert(KC == 3, "degree 3 or 6");
            const uint2* q = reinterpret_cast<const uint2*>(p);
            const uint2 a = q[0], b = q[1], c = q[2];
            w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y;
        }
    }
};

// NIN / NOU in {64, 128}; KC = degree (3 or 6); MAXD = destinations per wave (ceil(M / 8)) the register arrays are sized for.
// The neighbour table is shared by the batch (idx_sb == 0: what every reference script passes, ops.shared_graph_view); a
// different graph per sample takes the first-generation kernel (mpconv_fwd_b16.hip).
template <int NIN, int NOU, int KC, int MAXD, int MODE>
__global__ __launch_bounds__(SG_THREADS, 4) void mpconv_fwd_sg_kernel(const SgParams p) {
    constexpr int KS = NIN / 16;                      // MFMA k-steps
    constexpr int NPASS = NOU / 64;                   // column passes of 64 channels (256 projection columns)
    constexpr int XSB = NIN * 2 + 16;                 // x image row stride (bytes): rows land on distinct 16-byte slots
    constexpr int C8 = NIN / 8;                       // 16-byte chunks per x row
    constexpr int XPT = (96 * C8 + SG_THREADS - 1) / SG_THREADS;     // chunks per thread (N <= 96)
    constexpr bool WANT_ARG = MODE == SG_MODE_TRAIN || MODE == SG_MODE_TRAIN_STATS || MODE == SG_MODE_GENERIC;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int l31 = lane & 31, lh = lane >> 5;
    const int N = p.N, M = p.M, Npad = p.Npad;

    unsigned char* xs = sg_lds;                                   // [Npad][XSB]
    unsigned char* ps = sg_lds + Npad * XSB;                      // [Npad][SG_PSB]

    // ---- W^T A-fragments: areg[pass][kk] = W[c = 16kk + 8lh + 0..7][col = 256 pass + 32 wave + l31] ----
    sg_bf16x8 areg[NPASS][KS];
    {
        const int ncols = p.w_ld;
#pragma unroll
        for (int ps_i = 0; ps_i < NPASS; ++ps_i) {
            const float* wc = p.W + ps_i * 256 + 32 * wave + l31;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int c = 16 * kk + 8 * lh + 2 * h;
                    w[h] = sg_pack(wc[(size_t)c * ncols], wc[(size_t)(c + 1) * ncols]);
                }
                areg[ps_i][kk] = __builtin_bit_cast(sg_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
    }
    // ---- per-lane epilogue constants (lane <-> channel within a pass) ----
    float c_bias[NPASS], c_scale[NPASS], c_shift[NPASS];
#pragma unroll
    for (int ps_i = 0; ps_i < NPASS; ++ps_i) {
        const int o = ps_i * 64 + lane;
        c_bias[ps_i] = p.bias ? p.bias[o] : 0.f;
        c_scale[ps_i] = (MODE >= SG_MODE_AFFINE_RELU && p.pscale) ? p.pscale[o] : 1.f;
        c_shift[ps_i] = (MODE >= SG_MODE_AFFINE_RELU && p.pscale) ? p.pshift[o] : 0.f;
    }
    float st0[NPASS], st1[NPASS];
#pragma unroll
    for (int ps_i = 0; ps_i < NPASS; ++ps_i) { st0[ps_i] = 0.f; st1[ps_i] = 0.f; }

    // zero the padded rows of the x image once
    for (int f = tid; f < (Npad - N) * (XSB / 4); f += SG_THREADS) reinterpret_cast<unsigned*>(xs + N * XSB)[f] = 0u;

    const int m0 = wave * p.DPW;                                  // this wave's destinations: [m0, m0 + nd)
    const int nd = __builtin_amdgcn_readfirstlane(max(0, min(p.DPW, M - m0)));
    const unsigned pbase = (unsigned)(Npad * XSB) + (unsigned)lane * 8u;      // byte offset of this lane's channel in a P row

    // LDS byte addresses of the P rows this wave gathers: addr[d][j] = pbase + nn_idx[m0 + d][j] * SG_PSB
    unsigned addr[MAXD][KC];
    {
        const int64_t* ib = p.idx + (int64_t)m0 * KC;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                long long v = d < nd ? ib[d * KC + j] : 0;
                v = v < 0 ? 0 : (v >= N ? N - 1 : v);             // never read outside the image
                addr[d][j] = pbase + (unsigned)v * (unsigned)SG_PSB;
                // keep it a per-lane register: left alone, the compiler notices that the table entry is wave-uniform, moves it
                // to an SGPR (then spills the 36 of them into VGPR lanes) and re-adds the lane's offset on every edge
                asm volatile("" : "+v"(addr[d][j]));
            }
    }

    uint4 xr[XPT];
    uint4 er = make_uint4(0, 0, 0, 0);                            // 16 bytes of this wave's edge-type block (lane < ceil(bytes / 16))
    const int xchunks = N * C8;
    const int ebytes = nd * KC * 8;                               // edge-type bytes of this wave's destinations: <= 288
    unsigned char* es = sg_lds + Npad * XSB + Npad * SG_PSB + wave * (2 * SG_ESLOT);  // wave-private, double-buffered
    unsigned char* es_next = es + SG_ESLOT;
    auto prefetch = [&](int b) {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)b * p.x_sb);
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int f = tid + q * SG_THREADS;
            xr[q] = f < xchunks ? xb[f] : make_uint4(0, 0, 0, 0);
        }
        // rows [m0, m0 + nd) of etype[b] are one contiguous block: 16 bytes per lane, one load per wave and sample
        const unsigned char* eb = reinterpret_cast<const unsigned char*>(p.et + (int64_t)b * p.et_sb + (int64_t)m0 * KC * 4);
        if (lane * 16 < ebytes) {
            if (lane * 16 + 16 <= ebytes) {
                er = *reinterpret_cast<const uint4*>(eb + lane * 16);
            } else {                                              // 8-byte tail (odd number of degree-3 rows)
                const uint2 t = *reinterpret_cast<const uint2*>(eb + lane * 16);
                er = make_uint4(t.x, t.y, 0, 0);
            }
        }
    };

    auto commit = [&]() {                                       // prefetched registers -> LDS x image
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int f = tid + q * SG_THREADS;
            if (f < xchunks) {
                const int n = f / C8, c8 = f - n * C8;
                *reinterpret_cast<uint4*>(xs + n * XSB + c8 * 16) = xr[q];
            }
        }
    };

    int b = blockIdx.x;
    if (b < p.B) {
        prefetch(b);
        commit();
        if (lane * 16 < ebytes) *reinterpret_cast<uint4*>(es + lane * 16) = er;
        if (b + (int)gridDim.x < p.B) prefetch(b + gridDim.x);
    }

    for (; b < p.B; b += gridDim.x) {
        SG_STAMP(0);
#if !SG_OPT_EARLYCOMMIT
        if (b != (int)blockIdx.x) {                               // (the first sample was staged before the loop)
            commit();
            if (lane * 16 < ebytes) *reinterpret_cast<uint4*>(es + lane * 16) = er;
            if (b + (int)gridDim.x < p.B) prefetch(b + gridDim.x);
        }
#endif
        unsigned short* yb = p.y + (int64_t)b * p.y_sb + (int64_t)m0 * p.y_ld;
        uint8_t* ab = (WANT_ARG && p.argmax) ? p.argmax + (int64_t)b * p.y_sb + (int64_t)m0 * p.y_ld : nullptr;
        SG_STAMP(1);
        __syncthreads();                                          // x image complete; every wave is done with P
        SG_STAMP(2);

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            // ---- projection: P^T[32 cols of this wave][nodes] = W^T . x^T, bf16 MFMA 32x32x16, f32 accumulate ----
            const int ntile = Npad / 32;                          // 1..3
            // B fragments move in groups of four k-steps (16 VGPRs).  nin = 64: the next tile's group is in flight under
            // this tile's MFMAs; nin = 128 (two groups per tile) has no registers to spare for that and loads in place.
            constexpr int KG = KS / 4;
            constexpr bool AHEAD = KG == 1;
            sg_bf16x8 bfr[4];
            auto load_group = [&](int tile, int grp) {
                const unsigned char* bp = xs + (tile * 32 + l31) * XSB + lh * 16 + grp * 128;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bfr[kk] = __builtin_bit_cast(sg_bf16x8, *reinterpret_cast<const uint4*>(bp + kk * 32));
            };
            if (AHEAD) load_group(0, 0);
#pragma unroll
            for (int tile = 0; tile < 3; ++tile) {
                if (tile < ntile) {
                    sg_f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int grp = 0; grp < KG; ++grp) {
                        if (!AHEAD) load_group(tile, grp);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[pass][grp * 4 + kk], bfr[kk], acc, 0, 0, 0);
                    }
                    if (AHEAD && tile + 1 < ntile) load_group(tile + 1, 0);
                    // lane holds, for node tile*32 + l31, rows 8g + 4lh + i  ->  channel 8 wave + 2g + lh, edge type i
                    unsigned char* pw = ps + (tile * 32 + l31) * SG_PSB + (8 * wave + lh) * 8;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 pk;
                        pk.x = sg_pack(acc[4 * g + 0], acc[4 * g + 1]);
                        pk.y = sg_pack(acc[4 * g + 2], acc[4 * g + 3]);
                        *reinterpret_cast<uint2*>(pw + g * 16) = pk;
                    }
                }
            }
            SG_STAMP(3);
            __syncthreads();
            SG_STAMP(4);
            if (SG_OPT_EARLYCOMMIT && pass == NPASS - 1) {
                // every wave is past the projection: the x image is free.  Stage the NEXT sample now — its loads were
                // issued a whole gather ago, and this point lies before this sample's stores in program order, so the
                // vmcnt wait does not have to drain them (at the loop top it did: ~800 cycles per sample)
                if (b + (int)gridDim.x < p.B) {
                    commit();
                    if (lane * 16 < ebytes) *reinterpret_cast<uint4*>(es_next + lane * 16) = er;
                    if (b + 2 * (int)gridDim.x < p.B) prefetch(b + 2 * gridDim.x);
                }
            }

            // ---- gather + edge-type contraction + max: one destination per wave at a time, lane = channel ----
#pragma unroll
            for (int d = 0; d < MAXD; ++d) {
                if (d < nd) {
                    SgEt<KC> et;
                    et.load(static_cast<const unsigned char*>(__builtin_assume_aligned(es + d * KC * 8, KC == 6 ? 16 : 8)));   // broadcast reads
                    uint2 pk[KC];
#pragma unroll
                    for (int j = 0; j < KC; ++j)
                        pk[j] = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(sg_lds + addr[d][j], 8));    // ds_read_b64
                    float v[KC];
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        // the bias rides in the accumulator: max_j(msg_j + bias)
                        v[j] = sg_dot2(pk[j].y, et.w[2 * j + 1], sg_dot2_seed(pk[j].x, et.w[2 * j], c_bias[pass]));
                    }
#if SG_OPT_MAX3
                    float best = v[0];
#pragma unroll
                    for (int j = 1; j + 1 < KC; j += 2) best = fmaxf(fmaxf(best, v[j]), v[j + 1]);      // v_max3_f32
                    if ((KC & 1) == 0) best = fmaxf(best, v[KC - 1]);
                    int arg = KC - 1;
                    if (WANT_ARG) {                               // first occurrence of the maximum (torch.max on CPU)
#pragma unroll
                        for (int j = KC - 2; j >= 0; --j) arg = v[j] == best ? j : arg;
                    }
#else
                    float best = v[0];
                    int arg = 0;
#pragma unroll
                    for (int j = 1; j < KC; ++j) {
                        if (WANT_ARG) { if (v[j] > best) { best = v[j]; arg = j; } }        // strict >: first occurrence
                        else best = fmaxf(best, v[j]);
                    }
#endif
                    float res = best;
                    if (MODE >= SG_MODE_AFFINE_RELU) res = fmaf(res, c_scale[pass], c_shift[pass]);
                    if (MODE == SG_MODE_AFFINE_RELU || (MODE == SG_MODE_GENERIC && p.relu)) res = fmaxf(res, 0.f);
                    const unsigned packed = sg_pack(res, 0.f);
                    const int off = d * p.y_ld + pass * 64 + lane;
                    yb[off] = (unsigned short)packed;
                    if (MODE == SG_MODE_TRAIN_STATS) {
                        const float zr = __uint_as_float(packed << 16);            // of the value as stored
                        st0[pass] += zr;
                        st1[pass] = fmaf(zr, zr, st1[pass]);
                    }
                    if (WANT_ARG && ab) ab[off] = (uint8_t)arg;
                }
            }
            if (pass + 1 < NPASS) __syncthreads();                // P is rewritten by the next pass
        }
        if (SG_OPT_EARLYCOMMIT) { unsigned char* t = es; es = es_next; es_next = t; }
        SG_STAMP(5);
    }

    if (MODE == SG_MODE_TRAIN_STATS && p.stats) {
        // BatchNorm statistics epilogue: fold the 8 waves' per-channel sums in a fixed order, one partial row per workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(sg_lds);
#pragma unroll
        for (int ps_i = 0; ps_i < NPASS; ++ps_i) {
            red[((wave * 2) * NPASS + ps_i) * 64 + lane] = st0[ps_i];
            red[((wave * 2 + 1) * NPASS + ps_i) * 64 + lane] = st1[ps_i];
        }
        __syncthreads();
        if (tid < 2 * NOU) {
            const int which = tid / NOU, c = tid - which * NOU;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SG_WAVES; ++w) sum += red[((w * 2 + which) * NPASS + (c >> 6)) * 64 + (c & 63)];
            fgnn_fold_store(p.stats + ((int64_t)blockIdx.x * 2 + which) * p.st_ld + c, sum);
        }
        if (p.fold.tickets) {
            double* sums = reinterpret_cast<double*>(sg_lds + SG_WAVES * 2 * NPASS * 64 * 4);     // (past the fold's own floats)
            if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) fgnn_bn_final_apply(p.fin, NOU, sums);
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
template <int NIN, int NOU, int KC, int MAXD>
static void* sg_pick_mode(int mode) {
#define SG_CASE(m) \
    if (mode == m) return (void*)mpconv_fwd_sg_kernel<NIN, NOU, KC, MAXD, m>;
    SG_CASE(SG_MODE_TRAIN) SG_CASE(SG_MODE_TRAIN_STATS) SG_CASE(SG_MODE_AFFINE_RELU) SG_CASE(SG_MODE_GENERIC)
#undef SG_CASE
    return nullptr;
}
template <int KC, int MAXD>
static void* sg_pick_width(int nin, int nou, int mode) {
    if (nin == 64 && nou == 64) return sg_pick_mode<64, 64, KC, MAXD>(mode);
    if (nin == 64 && nou == 128) return sg_pick_mode<64, 128, KC, MAXD>(mode);
    if (nin == 128 && nou == 64) return sg_pick_mode<128, 64, KC, MAXD>(mode);
    return nullptr;
}

int fgnn_mpconv_forward_ws(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                           const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                           void* y, uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid, int mode, int split);
void fgnn_stats_pending(const fgnn_bn_final** fin, void** scratch);
void fgnn_stats_upper_half(FgnnFold* fold, fgnn_bn_final* fin);

// Same contract as fgnn_mpconv_forward_b16 (mpconv_fwd_b16.hip): 1 = launched, 0 = shape outside this kernel's family,
// < 0 = error; stats / plan_grid as there.
int fgnn_mpconv_forward_sg(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                           const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                           void* y, uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid) {
    static const bool off = getenv("FGNN_NO_SG") != nullptr;
    if (off) return 0;
    if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->net != 4 || d->agg != FGNN_AGG_MAX) return 0;
    if (d->k != 3 && d->k != 6) return 0;
    // 64 -> 128 as ONE launch (two column passes: 32 more resident fragment registers) spills under the 128-VGPR budget of 2
    // workgroups per CU and measured slower than the first-generation kernel (200 vs 150 us; FGNN_SG_WIDE=1 selects it).  The
    // output channels are independent, so the call runs as TWO launches of the 64 -> 64 kernel over the halves of W's columns
    // and of the 128-wide y / argmax / statistics rows (x and etype are read twice: +60 MB at 4 096 codewords).
    static const bool wide = getenv("FGNN_SG_WIDE") != nullptr;
    static const bool no_split = getenv("FGNN_SG_NOSPLIT") != nullptr;
    const bool split = d->nin == 64 && d->nou == 128 && !wide && !no_split;
    if (!((d->nin == 64 && (d->nou == 64 || d->nou == 128) && !(d->nou == 128 && no_split && !wide)) || (d->nin == 128 && d->nou == 64))) return 0;
    if (d->N < 1 || d->N > 96 || d->M < 1) return 0;
    const int DPW = (d->M + SG_WAVES - 1) / SG_WAVES;
    const int MAXD = d->k == 6 ? 6 : 12;
    if (DPW > MAXD) return 0;
    // x: dense channel-fastest sample block, 16-byte aligned rows; y / argmax: channel-fastest, dense per sample
    if (!(d->x_sc == 1 && d->x_sn == d->nin) || (d->x_sb % 8) != 0) return 0;
    if (!(d->y_sc == 1 && (d->M == 1 || d->y_sm == d->nou))) return 0;
    // nn_idx: dense [M][k] per sample (or shared); etype: edge-type fastest [M][k][4], per sample, 4-byte aligned rows
    if (!(d->idx_sk == 1 && d->idx_sm == d->k)) return 0;
    if (!(d->et_se == 1 && d->et_sk == 4 && d->et_sm == 4 * d->k) || (d->et_sb % 2) != 0) return 0;
    if (x && ((((uintptr_t)x) & 15) || (((uintptr_t)etype) & 3))) {
        // The plan call (x == NULL) announced THIS kernel's grid, and the caller sized / will finalise that many statistics rows:
        // falling through to the first-generation kernel (another grid) would leave rows unwritten or overrun them, silently.
        if (stats) FGNN_FAIL(FGNN_EINVAL, "mpconv forward with statistics: x must be 16-byte and etype 4-byte aligned (the partial rows were planned for the shared-graph kernel)");
        return 0;
    }
    int mode;
    if (stats || plan_grid) mode = SG_MODE_TRAIN_STATS;
    else if (!post_scale && !d->relu) mode = SG_MODE_TRAIN;
    else if (post_scale && d->relu && !argmax && bias) mode = SG_MODE_AFFINE_RELU;
    else mode = SG_MODE_GENERIC;
    if (d->idx_sb != 0 && d->B > 1) return 0;                     // per-sample graphs: first-generation kernel
    const int knou = split ? 64 : d->nou;                         // output channels of one launch
    if (knou == 64) {                                             // third generation first (mpconv_fwd_ws.hip); 0 = not its shape
        const int r = fgnn_mpconv_forward_ws(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, argmax, stream, stats,
                                             plan_grid, mode, split ? 1 : 0);
        if (r != 0) return r;
    }
    void* fn = d->k == 6 ? sg_pick_width<6, 6>(d->nin, knou, mode) : sg_pick_width<3, 12>(d->nin, knou, mode);
    if (!fn) return 0;
    const int Npad = fgnn_round_up(d->N, 32);
    const int lds = Npad * (d->nin * 2 + 16) + Npad * SG_PSB + SG_WAVES * 2 * SG_ESLOT;
    int grid = 256 * 2;
    if (grid > d->B) grid = d->B;
    if (plan_grid) { *plan_grid = grid; return 1; }
    SgParams p = {};
    p.x = static_cast<const unsigned short*>(x); p.idx = nn_idx; p.et = static_cast<const unsigned short*>(etype);
    p.W = filters; p.bias = bias; p.pscale = post_scale; p.pshift = post_shift;
    p.y = static_cast<unsigned short*>(y); p.argmax = argmax; p.stats = stats;
    p.B = d->B; p.N = d->N; p.M = d->M; p.Npad = Npad; p.DPW = DPW; p.relu = d->relu;
    p.x_sb = d->x_sb; p.et_sb = d->et_sb; p.y_sb = d->y_sb;
    p.y_ld = d->nou; p.w_ld = d->nou * 4; p.st_ld = d->nou;
    {   // the BatchNorm behind the operator, finalised by this launch (fgnn_mpconv_forward_stats set it for this call)
        const fgnn_bn_final* fin = nullptr;
        void* scratch = nullptr;
        fgnn_stats_pending(&fin, &scratch);
        p.fold = fgnn_fold_make(stats, (stats && fin) ? scratch : nullptr, grid, knou, 2 * d->nou, d->nou);
        if (fin) p.fin = *fin;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel(split ? "mpconv_fwd_sg_kernel<%d, %d, %d, %d, %d> x2" : "mpconv_fwd_sg_kernel<%d, %d, %d, %d, %d>", d->nin, knou, d->k, MAXD, mode);
    p.prof = nullptr;
#ifdef FGNN_ENABLE_PROF
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 64 * 8);
        (void)hipMemset(prof_buf, 0, 64 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(SG_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv sg forward launch: %s", hipGetErrorString(e));
    if (split) {                                                  // the upper 64 output channels
        p.W += 256; p.y += 64;
        if (p.bias) p.bias += 64;
        if (p.pscale) { p.pscale += 64; p.pshift += 64; }
        if (p.argmax) p.argmax += 64;
        if (p.stats) p.stats += 64;
        fgnn_stats_upper_half(&p.fold, &p.fin);
        e = hipLaunchKernel(fn, dim3(grid), dim3(SG_THREADS), args, lds, (hipStream_t)stream);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv sg forward launch (upper half): %s", hipGetErrorString(e));
    }
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {                                     // tuning aid: per-wave phase timeline of one sample (shader clocks)
        long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[fgnn prof sg fwd] wave %d:", w);
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %lld", h[w * 8 + i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 1;
}
