// mpconv_block_fwd.hip — inference forward of a whole `mp_conv_residual` block in ONE kernel:
//
//     a1  = LeakyReLU( BN1( Conv1x1(x) ) )                         on the SOURCE nodes      [N, 64]
//     z   = max_j sum_e etype[e,m,j] * (a1 W)[idx[m,j], o, e] + bias                         [M, 64]
//     a2  = ReLU( BN2(z) )
//     out = LeakyReLU( BN3( Conv1x1(a2) ) ) (+ addend)             on the DESTINATION nodes [M, 64]
//
// (/root/reference/lib/model/mpnn/mp_nn_residual.py:39-56 with the message operator of mp_nn.py:115-175).  In
// eval mode every BatchNorm is a per-channel affine, so the three stages of a sample depend on nothing but that
// sample: a persistent workgroup keeps x -> a1 -> P -> a2 in LDS and only `out` goes back to HBM.  The unfused
// inference path runs 5 kernels over these tensors (two streaming GEMMs, two affine+activation passes, the
// operator) and moves each intermediate through HBM twice.
//
// Family: bf16 channel-fastest tensors, 4 edge types (edge-type-fastest etype), max aggregator, NO_EXTENSION,
// nmed = 64, nin and nout in {64, 128, 256}, fixed degree k in {3, 6} (the parity-check calls of the LDPC model).  Schedule: as
// mpconv_fwd_b16.hip (512 threads, next sample prefetched into registers, W fragments of all three products
// resident in registers, two destinations in flight per wave in the gather) plus one MFMA phase in front and one
// behind; the a2 image re-uses the x image's LDS, so the 64-wide block stays under 80 KB (2 workgroups per CU).
#include "fgnn_common.h"
#include <stdlib.h>

#define KB_THREADS 512
#define KB_WAVES 8
#ifndef KB_XSB
#define KB_XSB 72        // row stride (bf16 elements) of the a1 / a2 images: 64 channels + 8 (x image: nin + 8)
#endif
#ifndef KB_PSB
#define KB_PSB 264       // row stride of the P image: 256 columns + 8
#endif

typedef __bf16 kb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kb_bf16x2 __attribute__((ext_vector_type(2)));

struct KbParams {
    fgnn_mpconv_desc d;  // of the message operator inside the block (nin = nou = 64, net = 4)
    const uint16_t* x;   // [B][N][nin]
    const int64_t* idx;
    const uint16_t* et;  // [B][M][k][4]
    const float* W1;     // [64][nin]  conv1 weight [out][in]
    const float* s1;     // [64] folded scale / shift of conv1 bias + BN1
    const float* t1;
    const float* F;      // [64][256] operator filters
    const float* s2;     // [64] folded (bias, BN2)
    const float* t2;
    const float* W2;     // [nout][64] conv2 weight [out][in]
    const float* s3;
    const float* t3;
    const uint16_t* addend;   // [B][M][nout] or NULL
    const uint16_t* addend1;  // two more of the same layout (or NULL): the kernel adds them in f32, one rounding at the output
    const uint16_t* addend2;
    uint16_t* y;         // [B][M][nout]
    float slope;         // LeakyReLU slope of conv1 / conv2
    int Npad, Mpad;
    int off_a1, off_ps, off_idx, off_et;   // byte offsets (x / a2 image at 0)
    int ad_bcast;        // bit a: addend a is ONE row per sample [B][nout], added to every destination (the hyper-factor's message to
                         // the variables: M identical rows the producer no longer writes — round 6)
};

extern __shared__ __attribute__((aligned(16))) unsigned char fgnn_lds_kb[];

__device__ __forceinline__ unsigned kb_pack2(float a, float b) {
    const kb_bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ kb_bf16x8 kb_frag8(const float* p8) {          // 8 consecutive f32 -> one fragment
    const f32x4 a = *reinterpret_cast<const f32x4*>(p8), b = *reinterpret_cast<const f32x4*>(p8 + 4);
    return __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(a[0], a[1]), kb_pack2(a[2], a[3]),
                                                    kb_pack2(b[0], b[1]), kb_pack2(b[2], b[3])));
}

// KC = degree, NI = nin / 64, NO = nout / 64
template <int KC, int NI, int NO>
__global__ __launch_bounds__(KB_THREADS) void mpconv_block_fwd_kernel(const KbParams p) {
    constexpr int NIN = 64 * NI, NOUT = 64 * NO, XS1 = NIN + 8, KS1 = 2 * NI, XQ = (96 * NIN / 8 + KB_THREADS - 1) / KB_THREADS;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = d.N, M = d.M;
    constexpr int k = KC;
    const int mk = M * k;
    uint16_t* xs = reinterpret_cast<uint16_t*>(fgnn_lds_kb);                 // [Npad][XS1] x, later [Mpad][XSB] a2
    uint16_t* a1s = reinterpret_cast<uint16_t*>(fgnn_lds_kb + p.off_a1);     // [Npad][XSB] a1
    uint16_t* ps = reinterpret_cast<uint16_t*>(fgnn_lds_kb + p.off_ps);      // [Npad][PSB] P
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_kb + p.off_idx);
    uint2* et_s = reinterpret_cast<uint2*>(fgnn_lds_kb + p.off_et);          // [mk] 4 x bf16

    // ---- resident fragments ----
    const int li0 = lane & 15, lk0 = lane >> 4;
    const int ot = wave & 3;                           // this wave's 16-channel output tile of conv1 / conv2
    kb_bf16x8 aW1[KS1], aW2[NO][2], aF[2][2];
    f32x4 c1s, c1t, c3s[NO], c3t[NO];                  // per-lane affine of the wave's conv tiles: channels tile*16 + 4lk + r
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) aW1[ks] = kb_frag8(p.W1 + (ot * 16 + li0) * NIN + 32 * ks + 8 * lk0);   // A[i = o][k = c] = W1[o][c]
    c1s = *reinterpret_cast<const f32x4*>(p.s1 + ot * 16 + 4 * lk0);
    c1t = *reinterpret_cast<const f32x4*>(p.t1 + ot * 16 + 4 * lk0);
#pragma unroll
    // conv2: the wave's NO output tiles are tiles u = ot * NO + q of the 4 * NO, with PERMUTED rows: tile (g = u >> 2, qt = u & 3), row i
    // <-> channel 64 g + 16 (i >> 2) + 4 qt + (i & 3).  Lane (row li, lk) then holds the 4 * NO CONSECUTIVE channels
    // 64 g + 16 lk + 4 qt0 .. of its node: addend and output move as 16-byte pieces, not 8-byte ones
    for (int q = 0; q < NO; ++q) {
        const int u = ot * NO + q, g = u >> 2, qt = u & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aW2[q][ks] = kb_frag8(p.W2 + (64 * g + 16 * (li0 >> 2) + 4 * qt + (li0 & 3)) * 64 + 32 * ks + 8 * lk0);
        c3s[q] = *reinterpret_cast<const f32x4*>(p.s3 + 64 * g + 16 * lk0 + 4 * qt);
        c3t[q] = *reinterpret_cast<const f32x4*>(p.t3 + 64 * g + 16 * lk0 + 4 * qt);
    }
    const int cbase = 64 * ((ot * NO) >> 2) + 16 * lk0 + 4 * ((ot * NO) & 3);     // first of this lane's 4 * NO output channels
    // operator filters: A[i = col][k = c] = F[c][col]; slabs wave and wave + 8 of the 16 column slabs
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = p.F[(32 * ks + 8 * lk0 + u) * 256 + (wave + 8 * q) * 16 + li0];
            aF[q][ks] = __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(w8[0], w8[1]), kb_pack2(w8[2], w8[3]),
                                                                 kb_pack2(w8[4], w8[5]), kb_pack2(w8[6], w8[7])));
        }
    const float c2s = p.s2[lane], c2t = p.t2[lane];   // gather epilogue: lane <-> channel

    // ---- prefetch registers ----
    uint4 xr[XQ];
    uint2 er;
    long long ir = 0;
    const int xchunks = N * (NIN / 8);
    auto prefetch = [&](int b, int t) {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)b * d.x_sb);
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = t + q * KB_THREADS;
            xr[q] = f < xchunks ? xb[f] : make_uint4(0, 0, 0, 0);
        }
        er = make_uint2(0, 0);
        if (t < mk) {
            er = *reinterpret_cast<const uint2*>(p.et + (int64_t)b * d.et_sb + (int64_t)t * 4);
            const int m = t / k, j = t - m * k;
            ir = (p.idx + (int64_t)b * d.idx_sb)[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
        }
    };
    auto commit = [&](int t) {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = t + q * KB_THREADS;
            if (f < xchunks) *reinterpret_cast<uint4*>(xs + (f / (NIN / 8)) * XS1 + (f % (NIN / 8)) * 8) = xr[q];
        }
        if (t < mk) {
            et_s[t] = er;
            const long long v = ir;
            idx_s[t] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v)) * (KB_PSB * 2);      // BYTE offset of the source's projection row
        }
    };
    // rows N..Npad of the a1 image are read by the projection and never written by conv1's stores (n < N only)
    for (int f = tid; f < p.Npad * KB_XSB / 2; f += KB_THREADS) reinterpret_cast<unsigned*>(a1s)[f] = 0u;

    int b = blockIdx.x;
    if (b < d.B) prefetch(b, tid);
    const int ntile = p.Npad / 16, mtile = p.Mpad / 16;
    for (; b < d.B; b += gridDim.x) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int li = t & 15, lk = (t >> 4) & 3;
        __syncthreads();                               // previous sample's conv2 is done with the a2 (= x) image
        commit(t);
        __syncthreads();
        if (b + (int)gridDim.x < d.B) prefetch(b + gridDim.x, t);

        // ---- conv1 + BN1 + LeakyReLU on the sources: D[i = o][j = n], tile (ot, nt), nt = wave/4 + 2 i ----
        for (int nt = wave >> 2; nt < ntile; nt += 2) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const uint16_t* bp = xs + (nt * 16 + li) * XS1 + 8 * lk;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aW1[ks], __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32 * ks)), acc, 0, 0, 0);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[r], c1s[r], c1t[r]); v[r] = u > 0.f ? u : u * p.slope; }
            if (nt * 16 + li < N)
                *reinterpret_cast<uint2*>(a1s + (nt * 16 + li) * KB_XSB + ot * 16 + 4 * lk) = make_uint2(kb_pack2(v[0], v[1]), kb_pack2(v[2], v[3]));
        }
        __syncthreads();

        // ---- projection P^T = F^T a1: wave's slabs w and w + 8, all node tiles ----
        for (int nt = 0; nt < ntile; ++nt) {
            const uint16_t* bp = a1s + (nt * 16 + li) * KB_XSB + 8 * lk;
            const kb_bf16x8 b0 = __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(bp));
            const kb_bf16x8 b1 = __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[q][0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[q][1], b1, acc, 0, 0, 0);
                *reinterpret_cast<uint2*>(ps + (nt * 16 + li) * KB_PSB + (wave + 8 * q) * 16 + 4 * lk) =
                    make_uint2(kb_pack2(acc[0], acc[1]), kb_pack2(acc[2], acc[3]));
            }
        }
        __syncthreads();                               // P complete; the x image is dead: it becomes a2

        // the addend of this wave's first conv2 node tile: asked for here, it lands under the gather
        // (up to three addends: their pieces are summed in f32 as they arrive — 4 registers per channel tile instead of 2)
        const int rows0 = (p.ad_bcast & 1) ? 1 : M, rows1 = (p.ad_bcast & 2) ? 1 : M, rows2 = (p.ad_bcast & 4) ? 1 : M;
        const uint16_t* adb = p.addend ? p.addend + (int64_t)b * rows0 * NOUT : nullptr;
        const uint16_t* adb1 = p.addend1 ? p.addend1 + (int64_t)b * rows1 * NOUT : nullptr;
        const uint16_t* adb2 = p.addend2 ? p.addend2 + (int64_t)b * rows2 * NOUT : nullptr;
        f32x4 adr[NO];
        auto ad_fetch = [&](int m0) {
#pragma unroll
            for (int q = 0; q < NO; ++q) adr[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const uint16_t* src[3] = {adb, adb1, adb2};
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (src[a]) {
                    const int64_t off = ((p.ad_bcast >> a) & 1) ? (int64_t)cbase : (int64_t)m0 * NOUT + cbase;
                    const uint2* ap = reinterpret_cast<const uint2*>(src[a] + off);
#pragma unroll
                    for (int q = 0; q < NO; ++q) {
                        const uint2 w = ap[q];
                        adr[q][0] += __uint_as_float(w.x << 16); adr[q][1] += __uint_as_float(w.x & 0xffff0000u);
                        adr[q][2] += __uint_as_float(w.y << 16); adr[q][3] += __uint_as_float(w.y & 0xffff0000u);
                    }
                }
        };
#pragma unroll
        for (int q = 0; q < NO; ++q) adr[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (adb && (wave >> 2) * 16 + li < M) ad_fetch((wave >> 2) * 16 + li);
        // ---- gather + edge-type contraction + max, two destinations in flight per wave; a2 = ReLU(BN2(z)) -> LDS ----
        // A wave's edge list is uniform: the source rows' byte offsets and the edge weights are BROADCAST LDS reads (every lane the same
        // address).  Round 6: fetched by the first lanes and spread with v_readlane they cost 3 readlanes + hazard no-ops + a scalar
        // multiply per edge — ~11 issue slots for two dot products and a max (factor_layer_fwd.hip: fl_gather).
        {
            const unsigned char* pc = reinterpret_cast<const unsigned char*>(ps) + lane * 8;
            for (int m0 = wave; m0 < M; m0 += 2 * KB_WAVES) {
                const bool two = m0 + KB_WAVES < M;
                const int m1 = two ? m0 + KB_WAVES : m0;
                const int* o0 = idx_s + m0 * KC;
                const int* o1 = idx_s + m1 * KC;
                const uint2* w0 = et_s + m0 * KC;
                const uint2* w1 = et_s + m1 * KC;
                int of0[KC], of1[KC];
                uint2 ew0[KC], ew1[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) { of0[j] = o0[j]; of1[j] = o1[j]; }
#pragma unroll
                for (int j = 0; j < KC; ++j) { ew0[j] = w0[j]; ew1[j] = w1[j]; }
                uint2 pk0[KC], pk1[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    pk0[j] = *reinterpret_cast<const uint2*>(pc + of0[j]);
                    pk1[j] = *reinterpret_cast<const uint2*>(pc + of1[j]);
                }
                float b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    float v0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(kb_bf16x2, pk0[j].x), __builtin_bit_cast(kb_bf16x2, ew0[j].x), 0.f, false);
                    v0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(kb_bf16x2, pk0[j].y), __builtin_bit_cast(kb_bf16x2, ew0[j].y), v0, false);
                    float v1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(kb_bf16x2, pk1[j].x), __builtin_bit_cast(kb_bf16x2, ew1[j].x), 0.f, false);
                    v1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(kb_bf16x2, pk1[j].y), __builtin_bit_cast(kb_bf16x2, ew1[j].y), v1, false);
                    b0 = j == 0 ? v0 : fmaxf(b0, v0);
                    b1 = j == 0 ? v1 : fmaxf(b1, v1);
                }
                const __bf16 h0 = (__bf16)fmaxf(fmaf(b0, c2s, c2t), 0.f);
                xs[m0 * KB_XSB + lane] = __builtin_bit_cast(uint16_t, h0);
                if (two) {
                    const __bf16 h1 = (__bf16)fmaxf(fmaf(b1, c2s, c2t), 0.f);
                    xs[m1 * KB_XSB + lane] = __builtin_bit_cast(uint16_t, h1);
                }
            }
        }
        __syncthreads();

        // ---- conv2 + BN3 + LeakyReLU (+ addend) on the destinations: node tiles mt = wave/4 + 2 i, the wave's NO channel tiles ----
        {
            uint16_t* yb = p.y + (int64_t)b * M * NOUT;
            int mt = wave >> 2;
            for (; mt < mtile; mt += 2) {
                const uint16_t* bp = xs + (mt * 16 + li) * KB_XSB + 8 * lk;
                const kb_bf16x8 b0 = __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(bp));
                const kb_bf16x8 b1 = __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32));
                const int m = mt * 16 + li;
                f32x4 acur[NO];
#pragma unroll
                for (int q = 0; q < NO; ++q) acur[q] = adr[q];
                if (adb && mt + 2 < mtile && m + 32 < M) ad_fetch(m + 32);        // the next node tile's addends travel under this one's products
                f32x4 acc[NO];
#pragma unroll
                for (int q = 0; q < NO; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aW2[q][0], b0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NO; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aW2[q][1], b1, acc[q], 0, 0, 0);
                if (m < M) {
                    unsigned ow[2 * NO];
#pragma unroll
                    for (int q = 0; q < NO; ++q) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[q][r], c3s[q][r], c3t[q][r]); v[r] = u > 0.f ? u : u * p.slope; }
                        if (adb) { v[0] += acur[q][0]; v[1] += acur[q][1]; v[2] += acur[q][2]; v[3] += acur[q][3]; }
                        ow[2 * q] = kb_pack2(v[0], v[1]);
                        ow[2 * q + 1] = kb_pack2(v[2], v[3]);
                    }
                    uint16_t* yp = yb + (int64_t)m * NOUT + cbase;
                    if (NO == 1) *reinterpret_cast<uint2*>(yp) = make_uint2(ow[0], ow[1]);
                    else {
#pragma unroll
                        for (int h = 0; h < NO / 2; ++h) *reinterpret_cast<uint4*>(yp + 8 * h) = make_uint4(ow[4 * h], ow[4 * h + 1], ow[4 * h + 2], ow[4 * h + 3]);
                    }
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
// Inference forward of conv1+BN1+LeakyReLU -> message operator (+bias, BN2, ReLU) -> conv2+BN3+LeakyReLU (+addend).
// s1/t1, s2/t2: per-channel float32 [64] affines, s3/t3: [nout], with the conv / operator biases folded in;
// W1 is [64][nin], W2 is [nout][64].  Returns FGNN_EUNSUPPORTED
// outside the family described at the top of this file (callers then run the stages separately).
extern "C" int fgnn_mpconv_block_forward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                         const void* etype, const float* W1, const float* s1, const float* t1,
                                         const float* filters, const float* s2, const float* t2, const float* W2,
                                         const float* s3, const float* t3, float slope, int nin, int nout,
                                         const void* addend, const void* addend1, const void* addend2, void* y, fgnn_stream_t stream) {
    return fgnn_mpconv_block_forward_rows(d, x, nn_idx, etype, W1, s1, t1, filters, s2, t2, W2, s3, t3, slope, nin, nout, addend, addend1,
                                          addend2, 0, y, stream);
}

// The same with per-sample ROW addends: bit a of `addend_row_mask` marks addend a as [B][nout] — one row per sample, added to every
// destination (fgnn_mpconv_block_forward_fanout run with M = 1 produces it: the hyper-factor sends every variable the same message).
extern "C" int fgnn_mpconv_block_forward_rows(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                              const void* etype, const float* W1, const float* s1, const float* t1,
                                              const float* filters, const float* s2, const float* t2, const float* W2,
                                              const float* s3, const float* t3, float slope, int nin, int nout,
                                              const void* addend, const void* addend1, const void* addend2, int32_t addend_row_mask,
                                              void* y, fgnn_stream_t stream) {
    if (!d || !x || !nn_idx || !etype || !W1 || !s1 || !t1 || !filters || !s2 || !t2 || !W2 || !s3 || !t3 || !y)
        FGNN_FAIL(FGNN_EINVAL, "mpconv_block_forward: null pointer");
    const bool ok = d->dtype == FGNN_BF16 && d->ext == FGNN_EXT_NONE && d->agg == FGNN_AGG_MAX && d->net == 4 &&
                    d->nin == 64 && d->nou == 64 && (d->k == 3 || d->k == 6) && d->N >= 1 && d->N <= 96 &&
                    d->M >= 1 && d->M <= 96 && d->M * d->k <= KB_THREADS &&
                    (nin == 64 || nin == 128 || nin == 256) && (nout == 64 || nout == 128 || nout == 256) &&
                    d->x_sc == 1 && d->x_sn == nin && d->x_sb % 8 == 0 &&
                    d->y_sc == 1 && d->y_sm == nout && d->y_sb == (int64_t)d->M * nout &&
                    d->et_se == 1 && d->et_sk == 4 && d->et_sm == 4 * d->k && d->et_sb % 4 == 0 &&
                    !((uintptr_t)x & 15) && !((uintptr_t)etype & 7) && !((uintptr_t)y & 15) &&
                    !(((uintptr_t)addend | (uintptr_t)addend1 | (uintptr_t)addend2) & 7) && (addend || (!addend1 && !addend2));
    if (!ok) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_block_forward: outside the fused block's family");
    if (d->B == 0) return FGNN_OK;
    KbParams p;
    p.d = *d;
    p.x = (const uint16_t*)x; p.idx = nn_idx; p.et = (const uint16_t*)etype;
    p.W1 = W1; p.s1 = s1; p.t1 = t1; p.F = filters; p.s2 = s2; p.t2 = t2; p.W2 = W2; p.s3 = s3; p.t3 = t3;
    p.addend = (const uint16_t*)addend; p.addend1 = (const uint16_t*)addend1; p.addend2 = (const uint16_t*)addend2;
    p.y = (uint16_t*)y; p.slope = slope;
    p.ad_bcast = addend_row_mask & 7;
    p.Npad = fgnn_round_up(d->N, 16);
    p.Mpad = fgnn_round_up(d->M, 16);
    const int img0 = p.Npad * (nin + 8) > p.Mpad * KB_XSB ? p.Npad * (nin + 8) : p.Mpad * KB_XSB;   // x, later a2
    int off = fgnn_round_up(img0 * 2, 16);
    p.off_a1 = off; off += fgnn_round_up(p.Npad * KB_XSB * 2, 16);
    p.off_ps = off; off += fgnn_round_up(p.Npad * KB_PSB * 2, 16);
    p.off_idx = off; off += fgnn_round_up(d->M * d->k * 4, 16);
    p.off_et = off; off += fgnn_round_up(d->M * d->k * 8, 16);
    const int lds = off;
    void* fn = nullptr;
#define KB_CASE(kc, ni, no) if (d->k == kc && nin == 64 * ni && nout == 64 * no) fn = (void*)mpconv_block_fwd_kernel<kc, ni, no>;
    KB_CASE(3, 1, 1) KB_CASE(6, 1, 1) KB_CASE(3, 2, 4) KB_CASE(6, 2, 4) KB_CASE(3, 4, 4) KB_CASE(6, 4, 4) KB_CASE(3, 4, 2) KB_CASE(6, 4, 2)
    KB_CASE(3, 1, 2) KB_CASE(6, 1, 2) KB_CASE(3, 2, 1) KB_CASE(6, 2, 1)
#undef KB_CASE
    if (!fn) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_block_forward: nin=%d nout=%d not instantiated", nin, nout);
    if (lds > 160 * 1024) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_block_forward: %d B of LDS", lds);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int wg_per_cu = (160 * 1024) / lds;
    if (wg_per_cu > 2) wg_per_cu = 2;
    if (wg_per_cu < 1) wg_per_cu = 1;
    int grid = 256 * wg_per_cu;
    if (grid > d->B) grid = d->B;
    fgnn_note_kernel("mpconv_block_fwd_kernel<%d, %d, %d>", d->k, nin / 64, nout / 64);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(KB_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv_block_forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// ========================================================================================
// The same block around the hyper-factor FAN-OUT call (one source node feeding M destinations through k = 1
// edges, single edge type: /root/reference/train_ldpc.py:60-75): per sample the sources are ONE nin-vector, so
// conv1 and the projection are two matrix-vector products (lane <-> channel, weights in LDS), the operator is
// z[m,o] = etype[m] * P[o], and only conv2 is GEMM-shaped: its B fragments (8 consecutive channels of a2 for one
// destination) are built in registers from P, so a2 never exists in memory.  One wave per sample, no barriers.
// ========================================================================================
struct KfParams {
    fgnn_mpconv_desc d;
    const uint16_t* x;   // [B][nin]
    const uint16_t* et;  // [B][M] (strides from d)
    const float* W1;     // [64][nin]
    const float* s1; const float* t1;
    const float* F;      // [64][64]
    const float* s2; const float* t2;
    const float* W2;     // [nout][64]
    const float* s3; const float* t3;
    const uint16_t* addend;
    const uint16_t* addend1;
    const uint16_t* addend2;
    uint16_t* y;         // [B][M][nout]
    float slope;
    int nin, nout, Mpad;
};

template <int NI>
__global__ __launch_bounds__(512) void mpconv_block_fanout_kernel(const KfParams p) {
    constexpr int NIN = 64 * NI;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int nout = p.nout, M = d.M;
    float* W1t = reinterpret_cast<float*>(fgnn_lds_kb);                    // [NIN][64]: W1t[c][o] = W1[o][c]
    float* Fl = W1t + NIN * 64;                                            // [64][64] as in memory: F[o][o2]
    uint16_t* W2l = reinterpret_cast<uint16_t*>(Fl + 64 * 64);             // [nout][XSB] bf16
    float* pw = reinterpret_cast<float*>(W2l + nout * KB_XSB) + wave * 128; // per wave: s2*P [64], t2 [64]
    for (int f = tid; f < NIN * 64; f += 512) { const int o = f / NIN, c = f - o * NIN; W1t[c * 64 + o] = p.W1[f]; }
    for (int f = tid; f < 64 * 64; f += 512) Fl[f] = p.F[f];
    for (int f = tid; f < nout * 32; f += 512) {
        const int q = f >> 5, c2 = f & 31;
        const float2 w = *reinterpret_cast<const float2*>(p.W2 + q * 64 + 2 * c2);
        *reinterpret_cast<unsigned*>(W2l + q * KB_XSB + 2 * c2) = kb_pack2(w.x, w.y);
    }
    __syncthreads();
    const float c1s = p.s1[lane], c1t = p.t1[lane], c2s = p.s2[lane], c2t = p.t2[lane];
    const int mtile = p.Mpad / 16, qtile = nout / 16;
    const int nwaves = gridDim.x * 8;
    for (int b = blockIdx.x * 8 + wave; b < d.B; b += nwaves) {
        float xv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) xv[i] = __uint_as_float((unsigned)p.x[(int64_t)b * d.x_sb + lane + 64 * i] << 16);
        // conv1 + BN1 + LeakyReLU (lane <-> channel o), rounded to bf16 like the staged path's a1
        float a1 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll 8
            for (int c = 0; c < 64; ++c)
                a1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[i]), c)), W1t[(64 * i + c) * 64 + lane], a1);
        a1 = fmaf(a1, c1s, c1t);
        a1 = a1 > 0.f ? a1 : a1 * p.slope;
        { const __bf16 h = (__bf16)a1; a1 = __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16); }
        // projection (lane <-> o2), bf16-rounded like the P of mpconv_fwd_hyper.hip
        float P = 0.f;
#pragma unroll 8
        for (int o = 0; o < 64; ++o) P = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), o)), Fl[o * 64 + lane], P);
        { const __bf16 h = (__bf16)P; P = __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16); }
        pw[lane] = c2s * P;                             // a2[m][o2] = ReLU(et[m] * (s2 P)[o2] + t2[o2])
        pw[64 + lane] = c2t;
        const uint16_t* eb = p.et + (int64_t)b * d.et_sb;
        uint16_t* yb = p.y + (int64_t)b * M * nout;
        const uint16_t* adbs[3] = {p.addend ? p.addend + (int64_t)b * M * nout : nullptr, p.addend1 ? p.addend1 + (int64_t)b * M * nout : nullptr,
                                   p.addend2 ? p.addend2 + (int64_t)b * M * nout : nullptr};
        for (int mt = 0; mt < mtile; ++mt) {
            const int m = mt * 16 + li;
            const float e = m < M ? __uint_as_float((unsigned)eb[(int64_t)m * d.et_sm] << 16) : 0.f;
            kb_bf16x8 bf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(pw + 32 * ks + 8 * lk), p1 = *reinterpret_cast<const f32x4*>(pw + 32 * ks + 8 * lk + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(pw + 64 + 32 * ks + 8 * lk), t1 = *reinterpret_cast<const f32x4*>(pw + 64 + 32 * ks + 8 * lk + 4);
                float v[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = fmaxf(fmaf(e, p0[u], t0[u]), 0.f); v[4 + u] = fmaxf(fmaf(e, p1[u], t1[u]), 0.f); }
                bf[ks] = __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(v[0], v[1]), kb_pack2(v[2], v[3]), kb_pack2(v[4], v[5]), kb_pack2(v[6], v[7])));
            }
            // Output channels in groups of 64 = four MFMA tiles whose rows are permuted (tile qt, row i <-> channel 16 (i >> 2) + 4 qt
            // + (i & 3)): lane (row li, lk) then holds the 16 CONSECUTIVE channels 16 lk .. + 15 of its row and moves them (and the
            // addend) as two 16-byte accesses instead of four 8-byte ones.
            for (int g = 0; g < qtile / 4; ++g) {
                f32x4 acc[4];
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    const uint16_t* wr = W2l + (64 * g + 16 * (li >> 2) + 4 * qt + (li & 3)) * KB_XSB + 8 * lk;
                    acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(wr)), bf[0], acc[qt], 0, 0, 0);
                    acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(wr + 32)), bf[1], acc[qt], 0, 0, 0);
                }
                if (m < M) {
                    const int ch = 64 * g + 16 * lk;                       // D[i = 4 lk + r] of tile qt = channel ch + 4 qt + r
                    float v[16];
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt) {
                        const f32x4 s3 = *reinterpret_cast<const f32x4*>(p.s3 + ch + 4 * qt), t3 = *reinterpret_cast<const f32x4*>(p.t3 + ch + 4 * qt);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[qt][r], s3[r], t3[r]); v[4 * qt + r] = u > 0.f ? u : u * p.slope; }
                    }
                    const int64_t off = (int64_t)m * nout + ch;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (adbs[a]) {
                            const uint4 a0 = *reinterpret_cast<const uint4*>(adbs[a] + off), a1 = *reinterpret_cast<const uint4*>(adbs[a] + off + 8);
                            const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                            for (int u = 0; u < 8; ++u) { v[2 * u] += __uint_as_float(aw[u] << 16); v[2 * u + 1] += __uint_as_float(aw[u] & 0xffff0000u); }
                        }
                    *reinterpret_cast<uint4*>(yb + off) = make_uint4(kb_pack2(v[0], v[1]), kb_pack2(v[2], v[3]), kb_pack2(v[4], v[5]), kb_pack2(v[6], v[7]));
                    *reinterpret_cast<uint4*>(yb + off + 8) = make_uint4(kb_pack2(v[8], v[9]), kb_pack2(v[10], v[11]), kb_pack2(v[12], v[13]), kb_pack2(v[14], v[15]));
                }
            }
        }
    }
}


// ----------------------------------------------------------------------------------------
// The fan-out block on ONE row per sample (round 6; M = 1: the caller hands the result on as a per-sample broadcast,
// blocks.py `_fused_eval`).  With one destination the block is three small GEMMs over the batch — [B x nin] W1^T, [B x 64] F,
// [B x 64] W2^T — and the wave-per-sample kernel above spends its time in two readlane mat-vec chains per sample behind a 36 us
// weight-staging prologue per workgroup (140 us per launch at B = 4096, on the inference forward's critical path:
// profiles/r06/README.md).  Here a wave takes 16 SAMPLES as the N dimension of v_mfma_f32_16x16x32_bf16 and computes the
// TRANSPOSED products a1^T = W1 x^T, P^T = F^T a1^T, y^T = W2 a2^T: the accumulator tiles of one product (lane: sample li, rows
// 16 t + 4 lk + r) ARE the B fragments of the next (k-slot (ks, i) <-> row 16 (2 ks + (i >> 2)) + 4 lk + (i & 3); the weight
// fragments are gathered in that order), so nothing goes through LDS and there is no prologue.  The f32 weights of conv1 and of
// the operator enter as two bf16 pieces (two MFMAs: the f32 products of the kernel above up to the summation order), conv2's as
// bf16, like there; the same bf16 rounding points (a1, P, a2, y).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ void kr_split8(const float* w, kb_bf16x8& hi, kb_bf16x8& lo) {      // 8 f32 -> bf16 hi + lo fragments
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = w[2 * q], b = w[2 * q + 1];
        h[q] = kb_pack2(a, b);
        l[q] = kb_pack2(a - __uint_as_float(h[q] << 16), b - __uint_as_float(h[q] & 0xffff0000u));
    }
    hi = __builtin_bit_cast(kb_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(kb_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

template <int NI>
__global__ __launch_bounds__(64) void mpconv_block_rows1_kernel(const KfParams p) {
    constexpr int NIN = 64 * NI, KS1 = NIN / 32;
    const fgnn_mpconv_desc& d = p.d;
    const int lane = threadIdx.x;
    const int li = lane & 15, lk = lane >> 4;
    const int nout = p.nout;
    const int b = blockIdx.x * 16 + li;
    const bool live = b < d.B;
    const int64_t bs = live ? b : d.B - 1;                 // (rows beyond the batch compute on the last sample and store nothing)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // ---- a1^T [64 o][16 samples] = W1 [o][c] x^T [c][sample] ----
    f32x4 acc1[4] = {zero, zero, zero, zero};
    const uint16_t* xrow = p.x + bs * d.x_sb + 8 * lk;
#pragma unroll 2
    for (int ks = 0; ks < KS1; ++ks) {
        const kb_bf16x8 xb = __builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(xrow + 32 * ks));
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const float* wr = p.W1 + (int64_t)(16 * ot + li) * NIN + 32 * ks + 8 * lk;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
            const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            kb_bf16x8 hi, lo;
            kr_split8(w, hi, lo);
            acc1[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, xb, acc1[ot], 0, 0, 0);
            acc1[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, xb, acc1[ot], 0, 0, 0);
        }
    }
    // BatchNorm1 + LeakyReLU, rounded to bf16: the B fragments of the next product
    kb_bf16x8 b2[2];
    {
        unsigned wq[8];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const int o = 16 * ot + 4 * lk;
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.s1 + o), t1 = *reinterpret_cast<const f32x4*>(p.t1 + o);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float u = fmaf(acc1[ot][r], s1[r], t1[r]); v[r] = u > 0.f ? u : u * p.slope; }
            wq[2 * ot] = kb_pack2(v[0], v[1]);
            wq[2 * ot + 1] = kb_pack2(v[2], v[3]);
        }
        b2[0] = __builtin_bit_cast(kb_bf16x8, make_uint4(wq[0], wq[1], wq[2], wq[3]));
        b2[1] = __builtin_bit_cast(kb_bf16x8, make_uint4(wq[4], wq[5], wq[6], wq[7]));
    }
    // ---- P^T [64 o2][16 samples] = F^T a1^T: A[o2 = 16 t + li][k-slot (ks, i)] = F[o = 16 (2 ks + (i >> 2)) + 4 lk + (i & 3)][o2] ----
    f32x4 acc2[4] = {zero, zero, zero, zero};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = p.F[(16 * (2 * ks + (i >> 2)) + 4 * lk + (i & 3)) * 64 + 16 * t + li];
            kb_bf16x8 hi, lo;
            kr_split8(w, hi, lo);
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b2[ks], acc2[t], 0, 0, 0);
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b2[ks], acc2[t], 0, 0, 0);
        }
    // P rounded to bf16, a2 = ReLU(e (s2 P) + t2) rounded to bf16: the B fragments of conv2
    const float e = __uint_as_float((unsigned)p.et[bs * d.et_sb] << 16);
    kb_bf16x8 b3[2];
    {
        unsigned wq[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o2 = 16 * t + 4 * lk;
            const f32x4 s2 = *reinterpret_cast<const f32x4*>(p.s2 + o2), t2 = *reinterpret_cast<const f32x4*>(p.t2 + o2);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const __bf16 h = (__bf16)acc2[t][r];
                const float P = __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16);
                v[r] = fmaxf(fmaf(e, s2[r] * P, t2[r]), 0.f);
            }
            wq[2 * t] = kb_pack2(v[0], v[1]);
            wq[2 * t + 1] = kb_pack2(v[2], v[3]);
        }
        b3[0] = __builtin_bit_cast(kb_bf16x8, make_uint4(wq[0], wq[1], wq[2], wq[3]));
        b3[1] = __builtin_bit_cast(kb_bf16x8, make_uint4(wq[4], wq[5], wq[6], wq[7]));
    }
    // ---- y^T [nout q][16 samples] = W2 a2^T; BatchNorm3 + LeakyReLU + addends; 8 bytes (channels 16 qt + 4 lk ..) per lane and tile ----
    uint16_t* yb = p.y + bs * nout;
    const uint16_t* adbs[3] = {p.addend ? p.addend + bs * nout : nullptr, p.addend1 ? p.addend1 + bs * nout : nullptr,
                               p.addend2 ? p.addend2 + bs * nout : nullptr};
#pragma unroll 2
    for (int qt = 0; qt < nout / 16; ++qt) {
        f32x4 acc = zero;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wr = p.W2 + (int64_t)(16 * qt + li) * 64 + 4 * lk;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 16 * (2 * ks)), w1 = *reinterpret_cast<const f32x4*>(wr + 16 * (2 * ks + 1));
            const kb_bf16x8 a = __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(w0[0], w0[1]), kb_pack2(w0[2], w0[3]), kb_pack2(w1[0], w1[1]), kb_pack2(w1[2], w1[3])));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b3[ks], acc, 0, 0, 0);
        }
        const int q = 16 * qt + 4 * lk;
        const f32x4 s3 = *reinterpret_cast<const f32x4*>(p.s3 + q), t3 = *reinterpret_cast<const f32x4*>(p.t3 + q);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[r], s3[r], t3[r]); v[r] = u > 0.f ? u : u * p.slope; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (adbs[a]) {
                const uint2 aw = *reinterpret_cast<const uint2*>(adbs[a] + q);
                v[0] += __uint_as_float(aw.x << 16); v[1] += __uint_as_float(aw.x & 0xffff0000u);
                v[2] += __uint_as_float(aw.y << 16); v[3] += __uint_as_float(aw.y & 0xffff0000u);
            }
        if (live) *reinterpret_cast<uint2*>(yb + q) = make_uint2(kb_pack2(v[0], v[1]), kb_pack2(v[2], v[3]));
    }
}

// ----------------------------------------------------------------------------------------
// Fan-in block (variables -> the hyper-factor): M = 1 destination listening to all N nodes in order (idx[j] = j,
// k = N), one edge type with per-neighbour weights et[j].  One wave per sample, nothing but the two weight images
// in LDS:  a1^T = W1 x^T by MFMA with x straight from global memory (B operand) and W1 fragments from LDS;
// the accumulator tiles of two 16-channel groups ARE the next B fragment (a1 never leaves registers: the k-slot
// permutation this implies is baked into the resident F fragments);  P^T = F^T a1^T;  the max over nodes runs on the
// accumulators (DPP over the 16 nodes of a tile, registers across tiles);  the 64 -> nout map on the single
// destination is a mat-vec with lane <-> output channel and z broadcast by v_readlane.
// ----------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(512) void mpconv_block_fanin_kernel(const KfParams p) {
    constexpr int NIN = 64 * NI, KS1 = NIN / 32, XW = NIN + 8;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int nout = p.nout, N = d.N, NQ = nout / 64;
    uint16_t* W1l = reinterpret_cast<uint16_t*>(fgnn_lds_kb);               // [64][XW] bf16: W1[c1][c]
    uint16_t* W2t = W1l + 64 * XW;                                           // [64][nout] bf16: W2t[o][oo] = W2[oo][o]
    float* zl = reinterpret_cast<float*>(W2t + 64 * nout) + wave * 64;       // per wave: z[64]
    for (int f = tid; f < 64 * (NIN / 2); f += 512) {
        const int c1 = f / (NIN / 2), c2 = f - c1 * (NIN / 2);
        const float2 w = *reinterpret_cast<const float2*>(p.W1 + (int64_t)c1 * NIN + 2 * c2);
        *reinterpret_cast<unsigned*>(W1l + c1 * XW + 2 * c2) = kb_pack2(w.x, w.y);
    }
    for (int f = tid; f < nout * 64; f += 512) {
        const int oo = f >> 6, o = f & 63;
        const __bf16 h = (__bf16)p.W2[f];
        W2t[o * nout + oo] = __builtin_bit_cast(uint16_t, h);
    }
    // resident A fragments of P^T = F^T a1^T: A[i = o][k-slot]; slot u of lane group lk in step ks is channel
    // c1 = 16 (2 ks + (u >> 2)) + 4 lk + (u & 3) — the order the a1 accumulators come in
    kb_bf16x8 aF[4][2];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = p.F[(int64_t)(16 * (2 * ks + (u >> 2)) + 4 * lk + (u & 3)) * 64 + ot * 16 + li];
            aF[ot][ks] = __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(w8[0], w8[1]), kb_pack2(w8[2], w8[3]),
                                                                  kb_pack2(w8[4], w8[5]), kb_pack2(w8[6], w8[7])));
        }
    float c1s[4][4], c1t[4][4];                                              // channel 16 ot + 4 lk + r
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * ot + 4 * lk + r;
            c1s[ot][r] = p.s1[c]; c1t[ot][r] = p.t1[c];
        }
    __syncthreads();
    const int ntile = (N + 15) / 16;
    const int nwaves = gridDim.x * 8;
    for (int b = blockIdx.x * 8 + wave; b < d.B; b += nwaves) {
        const uint16_t* xb = p.x + (int64_t)b * d.x_sb;
        const uint16_t* eb = p.et + (int64_t)b * d.et_sb;
        float zm[4][4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) zm[ot][r] = -3.0e38f;
        for (int nt = 0; nt < ntile; ++nt) {
            const int n = nt * 16 + li;
            const bool valid = n < N;
            uint4 bx[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                bx[ks] = valid ? *reinterpret_cast<const uint4*>(xb + (int64_t)n * NIN + 32 * ks + 8 * lk) : make_uint4(0, 0, 0, 0);
            const float e = valid ? __uint_as_float((unsigned)eb[(int64_t)n * d.et_sk] << 16) : 0.f;
            float a1v[4][4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                int woff = (16 * ot + li) * XW + 8 * lk;
                asm volatile("" : "+v"(woff));                    // per tile: keeps the W1 fragments in LDS, not in 128 hoisted registers
                const uint16_t* wr = W1l + woff;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kb_bf16x8, *reinterpret_cast<const uint4*>(wr + 32 * ks)),
                                                                  __builtin_bit_cast(kb_bf16x8, bx[ks]), acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float u = fmaf(acc[r], c1s[ot][r], c1t[ot][r]); a1v[ot][r] = u > 0.f ? u : u * p.slope; }
            }
            kb_bf16x8 bf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bf[ks] = __builtin_bit_cast(kb_bf16x8, make_uint4(kb_pack2(a1v[2 * ks][0], a1v[2 * ks][1]), kb_pack2(a1v[2 * ks][2], a1v[2 * ks][3]),
                                                                  kb_pack2(a1v[2 * ks + 1][0], a1v[2 * ks + 1][1]), kb_pack2(a1v[2 * ks + 1][2], a1v[2 * ks + 1][3])));
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[ot][0], bf[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[ot][1], bf[1], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {                              // P rounded to bf16 like the staged path's image
                    const __bf16 h = (__bf16)acc[r];
                    const float pe = e * __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16);
                    if (valid) zm[ot][r] = fmaxf(zm[ot][r], pe);
                }
            }
        }
        // max over the 16 nodes of a row of lanes, then operator bias + BN2 + ReLU (bf16 like the staged z) -> zl[o]
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = zm[ot][r];
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
                const int o = 16 * ot + 4 * lk + r;
                v = fmaxf(fmaf(v, p.s2[o], p.t2[o]), 0.f);
                const __bf16 h = (__bf16)v;
                if (li == 0) zl[o] = __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16);
            }
        // conv2 on the single destination: lane <-> output channel oo = 64 q + lane, z broadcast from the wave's LDS row
        // (same wave wrote it: in-order LDS, no barrier)
        float acc3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int o = 0; o < 64; ++o) {
            const float zb = zl[o];
            const uint16_t* wr = W2t + o * nout + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < NQ) acc3[q] = fmaf(zb, __uint_as_float((unsigned)wr[64 * q] << 16), acc3[q]);
        }
        uint16_t* yb = p.y + (int64_t)b * nout;
        const uint16_t* adbs[3] = {p.addend ? p.addend + (int64_t)b * nout : nullptr, p.addend1 ? p.addend1 + (int64_t)b * nout : nullptr,
                                   p.addend2 ? p.addend2 + (int64_t)b * nout : nullptr};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < NQ) {
                const int oo = 64 * q + lane;
                float v = fmaf(acc3[q], p.s3[oo], p.t3[oo]);
                v = v > 0.f ? v : v * p.slope;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (adbs[a]) v += __uint_as_float((unsigned)adbs[a][oo] << 16);
                const __bf16 h = (__bf16)v;
                yb[oo] = __builtin_bit_cast(uint16_t, h);
            }
    }
}

// Fan-out form of fgnn_mpconv_block_forward: d describes the inner operator with N = 1, k = 1, net = 1, nin = nou = 64
// (x strides: the block's input [B, nin]; y: [B, M, nout]); F is [64][64].
static const bool ROWS1 = getenv("FGNN_NO_BLOCK_ROWS1") == nullptr;      // (A/B switch of the one-row form, tools/gpu_ab.sh)
extern "C" int fgnn_mpconv_block_forward_fanout(const fgnn_mpconv_desc* d, const void* x, const void* etype,
                                                const float* W1, const float* s1, const float* t1, const float* filters,
                                                const float* s2, const float* t2, const float* W2, const float* s3,
                                                const float* t3, float slope, int nin, int nout, const void* addend,
                                                const void* addend1, const void* addend2, void* y, fgnn_stream_t stream) {
    if (!d || !x || !etype || !W1 || !s1 || !t1 || !filters || !s2 || !t2 || !W2 || !s3 || !t3 || !y)
        FGNN_FAIL(FGNN_EINVAL, "mpconv_block_forward_fanout: null pointer");
    const bool ok = d->dtype == FGNN_BF16 && d->ext == FGNN_EXT_NONE && d->net == 1 && d->nin == 64 && d->nou == 64 &&
                    d->N == 1 && d->k == 1 && d->M >= 1 && d->M <= 256 &&
                    (nin == 64 || nin == 128 || nin == 256) && (nout == 64 || nout == 128 || nout == 256) &&
                    d->x_sc == 1 && d->y_sc == 1 && d->y_sm == nout && d->y_sb == (int64_t)d->M * nout &&
                    !((uintptr_t)y & 15) && !(((uintptr_t)addend | (uintptr_t)addend1 | (uintptr_t)addend2) & 15);
    if (!ok) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_block_forward_fanout: outside the fused block's family");
    if (d->B == 0) return FGNN_OK;
    KfParams p;
    p.d = *d;
    p.x = (const uint16_t*)x; p.et = (const uint16_t*)etype; p.W1 = W1; p.s1 = s1; p.t1 = t1; p.F = filters;
    p.s2 = s2; p.t2 = t2; p.W2 = W2; p.s3 = s3; p.t3 = t3; p.addend = (const uint16_t*)addend; p.addend1 = (const uint16_t*)addend1; p.addend2 = (const uint16_t*)addend2; p.y = (uint16_t*)y;
    p.slope = slope; p.nin = nin; p.nout = nout; p.Mpad = fgnn_round_up(d->M, 16);
    if (d->M == 1 && ROWS1 && d->x_sb % 8 == 0 && !((uintptr_t)x & 15) && !(((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)s1 | (uintptr_t)t1 |
        (uintptr_t)s2 | (uintptr_t)t2 | (uintptr_t)s3 | (uintptr_t)t3) & 15)) {
        // one row per sample: 16 samples per wave on the matrix cores, no weight staging
        void* fr = nin == 64 ? (void*)mpconv_block_rows1_kernel<1> : nin == 128 ? (void*)mpconv_block_rows1_kernel<2> : (void*)mpconv_block_rows1_kernel<4>;
        fgnn_note_kernel("mpconv_block_rows1_kernel<%d>", nin / 64);
        void* rargs[] = {(void*)&p};
        hipError_t er = hipLaunchKernel(fr, dim3((d->B + 15) / 16), dim3(64), rargs, 0, (hipStream_t)stream);
        if (er != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv_block_forward_fanout (one row) launch: %s", hipGetErrorString(er));
        return FGNN_OK;
    }
    const int lds = nin * 64 * 4 + 64 * 64 * 4 + nout * KB_XSB * 2 + 8 * 128 * 4;
    void* fn = nin == 64 ? (void*)mpconv_block_fanout_kernel<1> : nin == 128 ? (void*)mpconv_block_fanout_kernel<2>
                                                                             : (void*)mpconv_block_fanout_kernel<4>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int grid = (d->B + 7) / 8;
    if (grid > 512) grid = 512;
    fgnn_note_kernel("mpconv_block_fanout_kernel<%d>", nin / 64);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(512), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv_block_forward_fanout launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// Fan-in form: d describes the inner operator with M = 1, k = N, net = 1, nin = nou = 64, and the neighbour list must
// be the identity (idx[j] = j — the caller checks; the kernel does not read nn_idx).  x strides: the block's input
// [B][N][nin] (channel-fastest); y / addend: [B][nout].
extern "C" int fgnn_mpconv_block_forward_fanin(const fgnn_mpconv_desc* d, const void* x, const void* etype,
                                               const float* W1, const float* s1, const float* t1, const float* filters,
                                               const float* s2, const float* t2, const float* W2, const float* s3,
                                               const float* t3, float slope, int nin, int nout, const void* addend,
                                               const void* addend1, const void* addend2, void* y, fgnn_stream_t stream) {
    if (!d || !x || !etype || !W1 || !s1 || !t1 || !filters || !s2 || !t2 || !W2 || !s3 || !t3 || !y)
        FGNN_FAIL(FGNN_EINVAL, "mpconv_block_forward_fanin: null pointer");
    const bool ok = d->dtype == FGNN_BF16 && d->ext == FGNN_EXT_NONE && d->agg == FGNN_AGG_MAX && d->net == 1 &&
                    d->nin == 64 && d->nou == 64 && d->M == 1 && d->k == d->N && d->N >= 1 && d->N <= 4096 &&
                    (nin == 64 || nin == 128 || nin == 256) && (nout == 64 || nout == 128 || nout == 256) &&
                    d->x_sc == 1 && d->x_sn == nin && d->x_sb % 8 == 0 && !((uintptr_t)x & 15);
    if (!ok) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_block_forward_fanin: outside the fused block's family");
    if (d->B == 0) return FGNN_OK;
    KfParams p;
    p.d = *d;
    p.x = (const uint16_t*)x; p.et = (const uint16_t*)etype; p.W1 = W1; p.s1 = s1; p.t1 = t1; p.F = filters;
    p.s2 = s2; p.t2 = t2; p.W2 = W2; p.s3 = s3; p.t3 = t3; p.addend = (const uint16_t*)addend; p.addend1 = (const uint16_t*)addend1; p.addend2 = (const uint16_t*)addend2; p.y = (uint16_t*)y;
    p.slope = slope; p.nin = nin; p.nout = nout; p.Mpad = 16;
    const int lds = 64 * (nin + 8) * 2 + 64 * nout * 2 + 8 * 64 * 4;
    void* fn = nin == 64 ? (void*)mpconv_block_fanin_kernel<1> : nin == 128 ? (void*)mpconv_block_fanin_kernel<2>
                                                                            : (void*)mpconv_block_fanin_kernel<4>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int grid = (d->B + 7) / 8;
    if (grid > 512) grid = 512;
    fgnn_note_kernel("mpconv_block_fanin_kernel<%d>", nin / 64);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(512), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv_block_forward_fanin launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
