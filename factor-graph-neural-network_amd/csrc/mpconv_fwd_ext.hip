// mpconv_fwd_ext.hip — forward of the VF/FV message operator for the synthetic-PGM calls (BASELINE configs 1 / 2 / 5):
// f32 storage and arithmetic (the 1e-4 parity path), ORIG_WITH_NEIGHBOR / ORIG_WITH_DIFF extension, 16 edge types,
// 64 -> 64 channels, N = M <= 64 nodes, degree k <= 16 (reference: /root/reference/lib/model/mpnn/mp_nn.py:136-175).
//
// The reference multiplies a [B*N*k, 128] per-edge matrix by filters [128, 1024]; split by rows of `filters`
// (W_top = rows 0..63 act on x_i, W_bot = rows 64..127 on x_j or x_i - x_j) the same sum is two NODE-level projections
//     S = x (W_top + W_bot), T = -x W_bot   (DIFF)          S = x W_top, T = x W_bot   (NEIGHBOR)
//     E[m,j,o] = sum_e etype[m,j,e] (S[m,o,e] + T[idx[m,j],o,e])
// — k times fewer FLOPs, and what remains (2 x 64 x 64 x 1024 x 2 = 16.8 MFLOP per sample) is compute-bound on the f32
// matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain; 157 TF peak; AI ~400 FLOP/B).  The shape-generic kernel
// (mpconv_fwd.hip) re-stages all of W (512 KB as f32) for every sample and runs at 10-16 % of that roof.  Here:
//   * column passes of 8 output channels (128 S + 128 T columns); a pass's filter slice lives in REGISTERS as MFMA A
//     fragments (32 VGPRs per wave) while the workgroup runs all of its samples through it (pass-major loop: W is read
//     from L2 8 times per workgroup, not 8 times per sample; workgroups start at different passes so that the chip does
//     not hammer one 64 KB slice at once);
//   * x of the next sample is prefetched into registers during the projection; P = [S | T] of the pass (64 KB) lives in
//     LDS only; edge types (shared by the batch in every reference script) are transposed into LDS once per workgroup;
//   * gather: thread = (destination, channel of the pass); per neighbour 8 LDS reads of 16 B and 16 packed FMAs.
// One 512-thread workgroup per CU (P 65 KB + edge types <= 40 KB + x 17 KB of LDS), up to 256 VGPRs per lane.
#include "fgnn_common.h"
#include <stdlib.h>

#define FX_THREADS 512
#define FX_WAVES 8
#define FX_NIN 64
#define FX_NOU 64
#define FX_NET 16
#define FX_NCOLS 1024
#define FX_PASS_CH 8
#define FX_XS 68             // x image row stride in floats (64 + 4: 16-byte aligned rows on distinct bank groups)
#define FX_PS 260            // P image row stride in floats (128 S + 128 T + 4)
#define FX_MAX_MK 640        // edges per sample the LDS edge-type image is sized for

typedef float fx_f32x2 __attribute__((ext_vector_type(2)));

struct FxParams {
    const float* x;
    const int64_t* idx;
    const float* et;
    const float* W;          // [128][1024]
    const float* bias;
    const float* pscale;
    const float* pshift;
    float* y;
    uint8_t* argmax;
    int B, N, k, ext, agg, relu;
    long long x_sb, y_sb, idx_sb, idx_sm, idx_sk, et_sb, et_se, et_sm, et_sk;     // elements
    int off_xs, off_ps, off_et, off_idx, off_par;       // byte offsets into LDS
    int dbg;                 // prof builds: FGNN_EXT_DBG bit 0 = skip the projection, bit 1 = skip the gather (timing experiments)
    long long* prof;         // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): stage timeline of workgroup 0
};

extern __shared__ __attribute__((aligned(16))) unsigned char fx_lds[];

template <int AGG>
__global__ __launch_bounds__(FX_THREADS, 2) void mpconv_fwd_ext_kernel(const FxParams p) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int N = p.N, k = p.k, mk = N * k;
    const int ntile = (N + 15) >> 4;

    float* xs = reinterpret_cast<float*>(fx_lds + p.off_xs);      // [64][XS]
    float* ps = reinterpret_cast<float*>(fx_lds + p.off_ps);      // [64][PS]: cols 0..127 = S, 128..255 = T of the pass
    float* et_s = reinterpret_cast<float*>(fx_lds + p.off_et);    // [mk][16]
    int* idx_s = reinterpret_cast<int*>(fx_lds + p.off_idx);      // [mk]

    const int chunk = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk, b_end = min(p.B, b_begin + chunk);
    if (b_begin >= b_end) return;
    const bool shared_graph = p.idx_sb == 0 && p.et_sb == 0;

    // zero the x image once (rows >= N feed the matrix cores)
    for (int f = tid; f < 64 * FX_XS; f += FX_THREADS) xs[f] = 0.f;

    auto stage_graph = [&](int b) {             // neighbour table + edge types of sample b -> LDS ([edge][16 types])
        for (int r = tid; r < mk; r += FX_THREADS) {
            const int m = r / k, j = r - m * k;
            long long v = p.idx[(int64_t)b * p.idx_sb + (int64_t)m * p.idx_sm + (int64_t)j * p.idx_sk];
            idx_s[r] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        }
        for (int f = tid; f < mk * FX_NET; f += FX_THREADS) {
            const int e = f / mk, r = f - e * mk;                 // source is edge-type slowest in the common layout: coalesced reads
            const int m = r / k, j = r - m * k;
            et_s[r * FX_NET + e] = p.et[(int64_t)b * p.et_sb + (int64_t)e * p.et_se + (int64_t)m * p.et_sm + (int64_t)j * p.et_sk];
        }
    };
    if (shared_graph) stage_graph(0);

    // x prefetch: 64 rows x 16 chunks of 16 bytes = 1024 chunks, two per thread
    uint4 xr[2];
    auto prefetch_x = [&](int b) {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)b * p.x_sb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * FX_THREADS;
            xr[q] = (f >> 4) < N ? xb[f] : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * FX_THREADS;
            if ((f >> 4) < N) *reinterpret_cast<uint4*>(xs + (f >> 4) * FX_XS + (f & 15) * 4) = xr[q];
        }
    };

    // gather role of this thread: destination m = tid >> 3, channel oc = tid & 7 of the pass
    const int gm = tid >> 3, goc = tid & 7;
    const bool gactive = gm < N;

    for (int pp = 0; pp < FX_NOU / FX_PASS_CH; ++pp) {
        const int pass = (pp + blockIdx.x) & (FX_NOU / FX_PASS_CH - 1);        // staggered start: see the header
        // ---- A fragments of the pass: slab `wave` of the S columns and slab `wave` of the T columns.
        //      lane (li, lk) holds, for k-step kk, W[c = 16 lk + kk][col]  (k index re-ordered so that a lane's B operand
        //      is 16 CONSECUTIVE input channels of a node: four 16-byte LDS reads per node tile) ----
        float aS[16], aT[16];
        {
            const int col = 128 * pass + 16 * wave + li;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = 16 * lk + kk;
                const float wt = p.W[(int64_t)c * FX_NCOLS + col], wb = p.W[(int64_t)(FX_NIN + c) * FX_NCOLS + col];
                aS[kk] = p.ext == FGNN_EXT_DIFF ? wt + wb : wt;
                aT[kk] = p.ext == FGNN_EXT_DIFF ? -wb : wb;
            }
        }
        prefetch_x(b_begin);
        for (int b = b_begin; b < b_end; ++b) {
            __syncthreads();                      // previous gather is done with P, xs, idx_s / et_s
            commit_x();
            if (!shared_graph) stage_graph(b);
            if (b + 1 < b_end) prefetch_x(b + 1);
            __syncthreads();

            // ---- projection: D[i = col][j = node] for this wave's S slab and T slab, f32 MFMA 16x16x4 ----
            for (int nt = 0; nt < ntile; ++nt) {
                f32x4 accS = {0.f, 0.f, 0.f, 0.f}, accT = {0.f, 0.f, 0.f, 0.f};
                const float* bp = xs + (nt * 16 + li) * FX_XS + 16 * lk;
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bp + 4 * q);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const float bv = bq[kk >> 2][kk & 3];
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[kk], bv, accS, 0, 0, 0);
                    accT = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[kk], bv, accT, 0, 0, 0);
                }
                float* pw = ps + (nt * 16 + li) * FX_PS + 16 * wave + 4 * lk;
                *reinterpret_cast<f32x4*>(pw) = accS;
                *reinterpret_cast<f32x4*>(pw + 128) = accT;
            }
            __syncthreads();

            // ---- gather: thread (m, oc): E[j] = sum_e et[m,j,e] (S[m][oc][e] + T[idx[m,j]][oc][e]) ----
            if (gactive) {
                const float* srow = ps + gm * FX_PS + goc * 16;
                fx_f32x2 s2[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * q);
                    s2[2 * q] = (fx_f32x2){v[0], v[1]};
                    s2[2 * q + 1] = (fx_f32x2){v[2], v[3]};
                }
                float best = 0.f, mx = -INFINITY, ssum = 0.f;
                int arg = 0;
                for (int j = 0; j < k; ++j) {
                    const int r = gm * k + j;
                    const float* trow = ps + idx_s[r] * FX_PS + 128 + goc * 16;
                    const float* erow = et_s + r * FX_NET;
                    fx_f32x2 acc = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(trow + 4 * q);
                        const f32x4 ev = *reinterpret_cast<const f32x4*>(erow + 4 * q);
                        acc = (fx_f32x2){ev[0], ev[1]} * (s2[2 * q] + (fx_f32x2){tv[0], tv[1]}) + acc;
                        acc = (fx_f32x2){ev[2], ev[3]} * (s2[2 * q + 1] + (fx_f32x2){tv[2], tv[3]}) + acc;
                    }
                    const float v = acc[0] + acc[1];
                    if (AGG == FGNN_AGG_MAX) {
                        if (j == 0 || v > best) { best = v; arg = j; }          // strict >: first occurrence (torch.max on CPU)
                    } else if (AGG == FGNN_AGG_LSE) {
                        const float v3 = 3.0f * v;
                        if (v3 > mx) { ssum = ssum * expf(mx - v3) + 1.0f; mx = v3; }
                        else ssum += expf(v3 - mx);
                    } else {
                        ssum += v;
                    }
                }
                float res;
                if (AGG == FGNN_AGG_MAX) res = best;
                else if (AGG == FGNN_AGG_LSE) res = (1.0f / 3.0f) * (mx + logf(ssum));
                else res = ssum / (float)k;
                const int o = FX_PASS_CH * pass + goc;
                if (p.bias) res += p.bias[o];
                if (p.pscale) res = res * p.pscale[o] + p.pshift[o];
                if (p.relu) res = fmaxf(res, 0.f);
                const int64_t off = (int64_t)b * p.y_sb + (int64_t)gm * FX_NOU + o;
                p.y[off] = res;
                if (AGG == FGNN_AGG_MAX && p.argmax) p.argmax[off] = (uint8_t)arg;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Pipelined form for a graph shared by the batch (idx_sb == 0 and et_sb == 0: every reference script): the projection
// (matrix cores) and the gather (LDS + VALU) of consecutive stages run CONCURRENTLY on different waves instead of taking
// turns behind barriers.  A stage = (pass of 4 output channels, sample).  Waves 0-3 ("producers", one per SIMD) project
// stage t from x image t&1 into P buffer t&1; waves 4-7 ("consumers", one per SIMD) gather stage t-1 from the other P
// buffer, write its 4 output channels and stage x of stage t+1 into the other x image; one barrier per stage.
// P row = 64 S + 64 T columns (32 quads of 16 B); the 16-column block (= one channel) of a row is XOR-swizzled with the
// row number (bits 0-1 ^ bits 2-3) so that the producers' b32 stores of a D fragment (4 rows x 16 columns per 32 lanes)
// and the consumers' b128 reads (lane (m, oc) reads chunk (q + m) & 3 of its block at step q: 16 lanes -> 16 bank quads,
// whatever rows the neighbours are) are both conflict-free.
#define FP_PCH 4
#define FP_NPASS (FX_NOU / FP_PCH)
#define FP_PROW 128
#ifdef FGNN_ENABLE_PROF
#define FP_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && t >= 8 && t < 12) p.prof[(wave * 4 + (t - 8)) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define FP_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int fp_sw(int row) { return (row ^ (row >> 2)) & 3; }

// single f32 VALU ops the compiler's SLP vectoriser cannot pair into v_pk_* instructions
__device__ __forceinline__ float fx_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fx_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---- three-term bf16 split of f32 operands (SPLIT variant) ----
// x = h + m + l exactly (three 8-bit mantissa pieces, same exponent range as f32).  A product x w is then the six bf16 MFMA
// terms  h h' + (h m' + m h') + (h l' + l h' + m m')  accumulated in f32: what is dropped (m l', l m', l l') is below 2^-24
// of |x w|, the size of one f32 rounding.  On the bf16 matrix cores six MFMAs cost 6 / 16 of one f32 MFMA, and — unlike the
// f32 MFMA — they leave the SIMD's VALU to the gather running beside them (tools/ubench/mfma_f32_partner.hip).
#define FQ_XS 72             // split x image row stride in bf16 elements (64 + 8: rows 144 bytes apart, conflict-free 16-byte reads)
typedef __bf16 fq_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned fq_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
// (a, b) -> packed bf16 pairs of the three terms
__device__ __forceinline__ float fx_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void fq_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    // single (not packed) subtractions: v_pk_add_f32 does not issue beside the producers' bf16 MFMA stream
    h = fq_pack2(a, b);
    const float ra = fx_sub(a, __uint_as_float(h << 16)), rb = fx_sub(b, __uint_as_float(h & 0xffff0000u));
    m = fq_pack2(ra, rb);
    l = fq_pack2(fx_sub(ra, __uint_as_float(m << 16)), fx_sub(rb, __uint_as_float(m & 0xffff0000u)));
}

template <int AGG, bool SPLIT>
__global__ __launch_bounds__(FX_THREADS, 2) void mpconv_fwd_extp_kernel(const FxParams p) {
    constexpr int NTHR = FX_THREADS, NCT = NTHR - 256, XQN = 1024 / NCT;      // consumer threads, x chunks per consumer thread
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int N = p.N, k = p.k, mk = N * k;
    const int ntile = (N + 15) >> 4;

    float* xs = reinterpret_cast<float*>(fx_lds + p.off_xs);      // [2][64][XS] f32 x image (exact-f32 variant)
    uint16_t* xq = reinterpret_cast<uint16_t*>(fx_lds + p.off_xs);   // SPLIT: [2][3 terms][64][FQ_XS] bf16
    constexpr int XQ_TERM = 64 * FQ_XS, XQ_BUF = 3 * XQ_TERM;
    float* ps = reinterpret_cast<float*>(fx_lds + p.off_ps);      // [2][64][128], block-swizzled rows
    float* et_s = reinterpret_cast<float*>(fx_lds + p.off_et);    // [mk][16]
    int* idx_s = reinterpret_cast<int*>(fx_lds + p.off_idx);      // [mk]: (row * 128 + 64) | swizzle of the neighbour's row

    const int chunk = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int ns = min(p.B, b_begin + chunk) - b_begin;
    if (ns <= 0) return;
    const int nstage = FP_NPASS * ns;

    float* par_s = reinterpret_cast<float*>(fx_lds + p.off_par);  // [3][64]: bias, post scale, post shift (LDS: the consumers' only
                                                                  // vector-memory traffic stays the x prefetch and the y stores)
    if (tid < FX_NOU) {
        par_s[tid] = p.bias ? p.bias[tid] : 0.f;
        par_s[64 + tid] = p.pscale ? p.pscale[tid] : 1.f;
        par_s[128 + tid] = p.pscale ? p.pshift[tid] : 0.f;
    }
    if (SPLIT) { for (int f = tid; f < 2 * XQ_BUF / 2; f += NTHR) reinterpret_cast<unsigned*>(xq)[f] = 0u; }
    else { for (int f = tid; f < 2 * 64 * FX_XS; f += NTHR) xs[f] = 0.f; }
    for (int r = tid; r < mk; r += NTHR) {
        const int m = r / k, j = r - m * k;
        long long v = p.idx[(int64_t)m * p.idx_sm + (int64_t)j * p.idx_sk];
        const int n = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        idx_s[r] = (n * FP_PROW + 64) | fp_sw(n);
    }
    for (int f = tid; f < mk * FX_NET; f += NTHR) {
        const int e = f / mk, r = f - e * mk;
        const int m = r / k, j = r - m * k;
        et_s[r * FX_NET + e] = p.et[(int64_t)e * p.et_se + (int64_t)m * p.et_sm + (int64_t)j * p.et_sk];
    }
    __syncthreads();
    // chunk f of a sample's x (row f >> 4, four channels from 4 (f & 15)) into x image `buf`
    auto put_x = [&](int buf, int f, const uint4& v) {
        if (SPLIT) {
            unsigned h0, m0, l0, h1, m1, l1;
            fq_split2(__uint_as_float(v.x), __uint_as_float(v.y), h0, m0, l0);
            fq_split2(__uint_as_float(v.z), __uint_as_float(v.w), h1, m1, l1);
            uint16_t* q = xq + buf * XQ_BUF + (f >> 4) * FQ_XS + (f & 15) * 4;
            *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(q + XQ_TERM) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(q + 2 * XQ_TERM) = make_uint2(l0, l1);
        } else {
            *reinterpret_cast<uint4*>(xs + buf * 64 * FX_XS + (f >> 4) * FX_XS + (f & 15) * 4) = v;
        }
    };
    // x of the first stage (all threads: 1024 chunks of 16 B)
    {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)b_begin * p.x_sb);
        for (int f = tid; f < N * 16; f += NTHR) put_x(0, f, xb[f]);
    }
    __syncthreads();

    if (wave < 4) {
        // =========================== producers ===========================
      if constexpr (!SPLIT) {
        // Filter rows of the NEXT pass travel as raw loads (rT = W_top, rB = W_bot columns) while this pass computes; they
        // are combined into the S / T operands only when the pass starts, so no stage waits on the L2 round trip.
        float aS[16], aT[16], rT[16], rB[16];
        auto load_w = [&](int pass) {
            const int col = 16 * (FP_PCH * pass + wave) + li;              // filter column o * 16 + e, o = 4 pass + wave, e = li
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = 16 * lk + kk;
                rT[kk] = p.W[(int64_t)c * FX_NCOLS + col];
                rB[kk] = p.W[(int64_t)(FX_NIN + c) * FX_NCOLS + col];
            }
        };
        int si = 0, pi = 0;
        const int npair = (ntile + 1) >> 1;
        load_w(blockIdx.x & (FP_NPASS - 1));
        for (int t = 0; t <= nstage; ++t) {
            FP_STAMP(0);
            if (t < nstage) {
                if (si == 0) {
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) {
                        aS[kk] = p.ext == FGNN_EXT_DIFF ? rT[kk] + rB[kk] : rT[kk];
                        aT[kk] = p.ext == FGNN_EXT_DIFF ? -rB[kk] : rB[kk];
                    }
                    if (pi + 1 < FP_NPASS) load_w((pi + 1 + blockIdx.x) & (FP_NPASS - 1));
                }
                const float* xi = xs + (t & 1) * 64 * FX_XS;
                float* pb = ps + (t & 1) * 64 * FP_PROW;
                f32x4 bq[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float* bp = xi + (h * 16 + li) * FX_XS + 16 * lk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[h][q] = *reinterpret_cast<const f32x4*>(bp + 4 * q);
                }
                for (int np = 0; np < ((p.dbg & 1) ? 0 : npair); ++np) {                       // two node tiles at once: four independent accumulators
                    f32x4 cq[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < 4; ++q) cq[h][q] = bq[h][q];
                    if (np + 1 < npair) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float* bp = xi + ((2 * np + 2 + h) * 16 + li) * FX_XS + 16 * lk;
#pragma unroll
                            for (int q = 0; q < 4; ++q) bq[h][q] = *reinterpret_cast<const f32x4*>(bp + 4 * q);
                        }
                    }
                    f32x4 accS[2], accT[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) { accS[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; accT[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float xv = cq[h][kk >> 2][kk & 3];
                            accS[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, aS[kk], accS[h], 0, 0, 0);
                            accT[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, aT[kk], accT[h], 0, 0, 0);
                        }
                    }
                    // D[i = node 4 lk + r][j = column li]
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = (2 * np + h) * 16 + 4 * lk + r;
                            float* pr = pb + row * FP_PROW + ((wave ^ fp_sw(row)) << 4) + li;
                            pr[0] = accS[h][r];
                            pr[64] = accT[h][r];
                        }
                }
                if (++si == ns) { si = 0; ++pi; }
            }
            FP_STAMP(1);
            __syncthreads();
        }
      } else {
        // ---- SPLIT: three-term bf16 operands on v_mfma_f32_16x16x32_bf16.  k-step ks, k-group lk <-> channels 32 ks + 8 lk .. + 7 ----
        float rT[16], rB[16];                                              // raw filter rows of the next pass (c = 32 (kk >> 3) + 8 lk + (kk & 7))
        fq_bf16x8 wS[3][2], wT[3][2];                                      // [term][k-step] B operands of this wave's S and T slab
        auto load_w = [&](int pass) {
            const int col = 16 * (FP_PCH * pass + wave) + li;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = 32 * (kk >> 3) + 8 * lk + (kk & 7);
                rT[kk] = p.W[(int64_t)c * FX_NCOLS + col];
                rB[kk] = p.W[(int64_t)(FX_NIN + c) * FX_NCOLS + col];
            }
        };
        auto split_w = [&]() {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned hs[4], ms[4], ls[4], ht[4], mt[4], lt[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 8 * ks + 2 * q;
                    const float s0 = p.ext == FGNN_EXT_DIFF ? rT[k0] + rB[k0] : rT[k0], s1 = p.ext == FGNN_EXT_DIFF ? rT[k0 + 1] + rB[k0 + 1] : rT[k0 + 1];
                    const float t0 = p.ext == FGNN_EXT_DIFF ? -rB[k0] : rB[k0], t1 = p.ext == FGNN_EXT_DIFF ? -rB[k0 + 1] : rB[k0 + 1];
                    fq_split2(s0, s1, hs[q], ms[q], ls[q]);
                    fq_split2(t0, t1, ht[q], mt[q], lt[q]);
                }
                wS[0][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(hs[0], hs[1], hs[2], hs[3]));
                wS[1][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(ms[0], ms[1], ms[2], ms[3]));
                wS[2][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(ls[0], ls[1], ls[2], ls[3]));
                wT[0][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(ht[0], ht[1], ht[2], ht[3]));
                wT[1][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(mt[0], mt[1], mt[2], mt[3]));
                wT[2][ks] = __builtin_bit_cast(fq_bf16x8, make_uint4(lt[0], lt[1], lt[2], lt[3]));
            }
        };
        int si = 0, pi = 0;
        load_w(blockIdx.x & (FP_NPASS - 1));
        for (int t = 0; t <= nstage; ++t) {
            FP_STAMP(0);
            if (t < nstage) {
                if (si == 0) {
                    split_w();
                    if (pi + 1 < FP_NPASS) load_w((pi + 1 + blockIdx.x) & (FP_NPASS - 1));
                }
                const uint16_t* xi = xq + (t & 1) * XQ_BUF;
                float* pb = ps + (t & 1) * 64 * FP_PROW;
                for (int nt = 0; nt < ((p.dbg & 1) ? 0 : ntile); ++nt) {
                    fq_bf16x8 xa[3][2];                                    // [term][k-step] A operand: 8 consecutive channels of node row li
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            xa[tm][ks] = __builtin_bit_cast(fq_bf16x8, *reinterpret_cast<const uint4*>(xi + tm * XQ_TERM + (nt * 16 + li) * FQ_XS + 32 * ks + 8 * lk));
                    // four independent accumulator chains (S / T x the two k-steps), smallest terms first in each:
                    // l h' + h l' + m m', then m h' + h m', then h h'
                    f32x4 aS[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, aT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            aS[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[TA[pr]][ks], wS[TB[pr]][ks], aS[ks], 0, 0, 0);
                            aT[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[TA[pr]][ks], wT[TB[pr]][ks], aT[ks], 0, 0, 0);
                        }
                    const f32x4 accS = aS[0] + aS[1], accT = aT[0] + aT[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = nt * 16 + 4 * lk + r;
                        float* pr = pb + row * FP_PROW + ((wave ^ fp_sw(row)) << 4) + li;
                        pr[0] = accS[r];
                        pr[64] = accT[r];
                    }
                }
                if (++si == ns) { si = 0; ++pi; }
            }
            FP_STAMP(1);
            __syncthreads();
        }
      }
    } else {
        // =========================== consumers ===========================
        if (!(p.dbg & 4)) __builtin_amdgcn_s_setprio(3);      // VALU issue is arbitrated by priority, then age: without it the gather starves beside the producers' MFMA stream
        const int ct = tid - 256;
        const int gm = ct >> 2, goc = ct & 3, mg = gm & 3;
        const bool gactive = gm < N;
        const int srow = gm * FP_PROW + ((goc ^ fp_sw(gm)) << 4);
        int si = 0, pi = 0;                       // of stage t - 1
        // x travels two stages ahead: loaded during stage t - 1, written to the idle x image during stage t, read by the
        // producers in stage t + 1 (the load's latency never sits between two barriers)
        uint4 xr[XQN];
        auto load_x = [&](int stage) {
            const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)(b_begin + stage % ns) * p.x_sb);
#pragma unroll
            for (int q = 0; q < XQN; ++q) {
                const int f = ct + q * NCT;
                xr[q] = (f >> 4) < N ? xb[f] : make_uint4(0, 0, 0, 0);
            }
        };
        if (1 < nstage) load_x(1);
        for (int t = 0; t <= nstage; ++t) {
            FP_STAMP(0);
            if (t + 1 < nstage) {
#pragma unroll
                for (int q = 0; q < XQN; ++q) {
                    const int f = ct + q * NCT;
                    if ((f >> 4) < N) put_x((t + 1) & 1, f, xr[q]);
                }
            }
            FP_STAMP(2);
            if (t + 2 < nstage) load_x(t + 2);
            if (t >= 1) {
                const float* pb = ps + ((t - 1) & 1) * 64 * FP_PROW;
                if (gactive && !(p.dbg & 2)) {
                    fx_f32x2 s2[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pb + srow + 4 * ((q + mg) & 3));
                        s2[2 * q] = (fx_f32x2){v[0], v[1]};
                        s2[2 * q + 1] = (fx_f32x2){v[2], v[3]};
                    }
                    float best = 0.f, mx = -INFINITY, ssum = 0.f;
                    int arg = 0;
                    auto fold = [&](float v, int j) {
                        if (AGG == FGNN_AGG_MAX) {
                            if (j == 0 || v > best) { best = v; arg = j; }
                        } else if (AGG == FGNN_AGG_LSE) {
                            const float v3 = 3.0f * v;
                            if (v3 > mx) { ssum = ssum * expf(mx - v3) + 1.0f; mx = v3; }
                            else ssum += expf(v3 - mx);
                        } else {
                            ssum += v;
                        }
                    };
                    if (SPLIT) {
                        // plain f32 VALU: packed f32 ops do not issue beside a bf16 MFMA stream (tools/ubench/mfma_f32_partner.hip).
                        // (Requesting neighbour j + 1's rows before neighbour j's products, or pairing lanes over the edge types on 8
                        // consumer waves, measured slower: the order-9 gather is bound by LDS throughput, not by its round trips.)
                        for (int j = 0; j < k; ++j) {
                            const int r = gm * k + j;
                            const int iv = idx_s[r];
                            const float* trow = pb + (iv & ~3) + ((goc ^ (iv & 3)) << 4);
                            const float* erow = et_s + r * FX_NET;
                            float a0 = 0.f, a1 = 0.f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int c4 = 4 * ((q + mg) & 3);
                                const f32x4 tv = *reinterpret_cast<const f32x4*>(trow + c4);
                                const f32x4 ev = *reinterpret_cast<const f32x4*>(erow + c4);
                                a0 = fx_fma(ev[0], fx_add(s2[2 * q][0], tv[0]), a0);
                                a1 = fx_fma(ev[1], fx_add(s2[2 * q][1], tv[1]), a1);
                                a0 = fx_fma(ev[2], fx_add(s2[2 * q + 1][0], tv[2]), a0);
                                a1 = fx_fma(ev[3], fx_add(s2[2 * q + 1][1], tv[3]), a1);
                            }
                            fold(a0 + a1, j);
                        }
                    } else {
                        for (int j = 0; j < k; ++j) {
                            const int r = gm * k + j;
                            const int iv = idx_s[r];
                            const float* trow = pb + (iv & ~3) + ((goc ^ (iv & 3)) << 4);
                            const float* erow = et_s + r * FX_NET;
                            fx_f32x2 acc = {0.f, 0.f};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int c4 = 4 * ((q + mg) & 3);
                                const f32x4 tv = *reinterpret_cast<const f32x4*>(trow + c4);
                                const f32x4 ev = *reinterpret_cast<const f32x4*>(erow + c4);
                                acc = (fx_f32x2){ev[0], ev[1]} * (s2[2 * q] + (fx_f32x2){tv[0], tv[1]}) + acc;
                                acc = (fx_f32x2){ev[2], ev[3]} * (s2[2 * q + 1] + (fx_f32x2){tv[2], tv[3]}) + acc;
                            }
                            fold(acc[0] + acc[1], j);
                        }
                    }
                    float res;
                    if (AGG == FGNN_AGG_MAX) res = best;
                    else if (AGG == FGNN_AGG_LSE) res = (1.0f / 3.0f) * (mx + logf(ssum));
                    else res = ssum / (float)k;
                    const int pass = (pi + blockIdx.x) & (FP_NPASS - 1);
                    const int o = FP_PCH * pass + goc;
                    res = (res + par_s[o]) * par_s[64 + o] + par_s[128 + o];
                    if (p.relu) res = fmaxf(res, 0.f);
                    const int64_t off = (int64_t)(b_begin + si) * p.y_sb + (int64_t)gm * FX_NOU + o;
                    p.y[off] = res;
                    if (AGG == FGNN_AGG_MAX && p.argmax) p.argmax[off] = (uint8_t)arg;
                }
                if (++si == ns) { si = 0; ++pi; }
            }
            FP_STAMP(1);
            __syncthreads();
        }
    }
}

// Returns 1 if launched, 0 if the call is outside this kernel's family, < 0 on error.
int fgnn_mpconv_forward_ext(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                            void* y, uint8_t* argmax, fgnn_stream_t stream) {
    static const bool off = getenv("FGNN_NO_EXT") != nullptr;
    if (off) return 0;
    if (d->dtype != FGNN_F32 || (d->ext != FGNN_EXT_NEIGHBOR && d->ext != FGNN_EXT_DIFF)) return 0;
    if (d->net != FX_NET || d->nin != FX_NIN || d->nou != FX_NOU) return 0;
    if (d->N != d->M || d->N < 1 || d->N > 64 || d->k < 1 || d->k > 16 || d->N * d->k > FX_MAX_MK) return 0;
    // x / y: dense channel-fastest per-sample blocks, 16-byte aligned
    if (!(d->x_sc == 1 && d->x_sn == FX_NIN && d->x_sb % 4 == 0) || ((uintptr_t)x & 15)) return 0;
    if (!(d->y_sc == 1 && d->y_sm == FX_NOU)) return 0;
    FxParams p;
    p.x = (const float*)x; p.idx = nn_idx; p.et = (const float*)etype; p.W = filters; p.bias = bias;
    p.pscale = post_scale; p.pshift = post_shift; p.y = (float*)y; p.argmax = argmax;
    p.B = d->B; p.N = d->N; p.k = d->k; p.ext = d->ext; p.agg = d->agg; p.relu = d->relu;
    p.x_sb = d->x_sb; p.y_sb = d->y_sb; p.idx_sb = d->idx_sb; p.idx_sm = d->idx_sm; p.idx_sk = d->idx_sk;
    p.et_sb = d->et_sb; p.et_se = d->et_se; p.et_sm = d->et_sm; p.et_sk = d->et_sk;
    int off_b = 0;
    auto take = [&](int bytes) { const int o = off_b; off_b = fgnn_round_up(off_b + bytes, 16); return o; };
    const bool piped = (d->idx_sb == 0 && d->et_sb == 0) || d->B == 1;
    static const bool exact = getenv("FGNN_EXT_F32MFMA") != nullptr;      // projections on the f32 matrix cores (bitwise an fmaf chain)
    // default: three-term bf16 split on the bf16 matrix cores, when its larger x images still fit next to the edge types
    const bool split = piped && !exact && (2 * 3 * 64 * FQ_XS * 2 + 2 * 64 * FP_PROW * 4 + d->N * d->k * (FX_NET * 4 + 4) + 3 * 64 * 4 + 64 <= 160 * 1024);
    p.off_xs = take(split ? 2 * 3 * 64 * FQ_XS * 2 : (piped ? 2 : 1) * 64 * FX_XS * 4);
    p.off_ps = take(piped ? 2 * 64 * FP_PROW * 4 : 64 * FX_PS * 4);
    p.off_et = take(d->N * d->k * FX_NET * 4);
    p.off_idx = take(d->N * d->k * 4);
    p.off_par = take(3 * 64 * 4);
    const int lds = off_b;
    if (lds > 160 * 1024) return 0;
    void* fn = d->agg == FGNN_AGG_MAX ? (void*)mpconv_fwd_ext_kernel<FGNN_AGG_MAX>
             : d->agg == FGNN_AGG_LSE ? (void*)mpconv_fwd_ext_kernel<FGNN_AGG_LSE> : (void*)mpconv_fwd_ext_kernel<FGNN_AGG_MEAN>;
    if (piped && split)
        fn = d->agg == FGNN_AGG_MAX ? (void*)mpconv_fwd_extp_kernel<FGNN_AGG_MAX, true>
           : d->agg == FGNN_AGG_LSE ? (void*)mpconv_fwd_extp_kernel<FGNN_AGG_LSE, true> : (void*)mpconv_fwd_extp_kernel<FGNN_AGG_MEAN, true>;
    else if (piped)
        fn = d->agg == FGNN_AGG_MAX ? (void*)mpconv_fwd_extp_kernel<FGNN_AGG_MAX, false>
           : d->agg == FGNN_AGG_LSE ? (void*)mpconv_fwd_extp_kernel<FGNN_AGG_LSE, false> : (void*)mpconv_fwd_extp_kernel<FGNN_AGG_MEAN, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    int grid = 256;
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;
    fgnn_note_kernel(piped ? (split ? "mpconv_fwd_extp_kernel<%d, true>" : "mpconv_fwd_extp_kernel<%d, false>") : "mpconv_fwd_ext_kernel<%d>", d->agg);
    p.prof = nullptr;
    p.dbg = 0;
#ifdef FGNN_ENABLE_PROF
    if (getenv("FGNN_EXT_DBG")) p.dbg = atoi(getenv("FGNN_EXT_DBG"));
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 128 * 8);
        (void)hipMemset(prof_buf, 0, 128 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(FX_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext forward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof && piped) {                            // tuning aid: stages 8..11 of workgroup 0 (shader clocks): start, x written, end
        long long h[128];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[fgnn prof ext fwd] wave %d:", w);
            for (int t = 0; t < 4; ++t)
                fprintf(stderr, "  | %lld %lld %lld", h[(w * 4 + t) * 4] - h[0], h[(w * 4 + t) * 4 + 2] - h[0], h[(w * 4 + t) * 4 + 1] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 1;
}
