// mpconv_bwd_b16.hip — backward of the VF/FV message operator on the bf16 matrix cores, for the parity-check
// calls of the LDPC model: bf16 channel-fastest x / gz / etype, 4 edge types, max aggregator, NO_EXTENSION,
// nin in {64,128}, nou in {64,128}.  Same maths and the same atomic-free, bit-reproducible structure as
// mpconv_bwd_res.hip (autograd through /root/reference/lib/model/mpnn/mp_nn.py:115-175):
//
//     P[n,col]   = sum_c x[c,n] W[c,col]                              (recomputed, col = o*4+e)
//     detype[e,m,j] = sum_o gz[o,m] [j == argmax[o,m]] P[idx[m,j], o*4+e]
//     dP[n,col]  = sum_{(m,j): idx[m,j]=n} gz[o,m] [j == argmax[o,m]] etype[e,m,j]   (sorted CSR transpose)
//     dx[c,n]    = sum_col W[c,col] dP[n,col]
//     dW[c,col] += sum_n x[c,n] dP[n,col]          dbias[o] += sum_m gz[o,m]
//
// What changes is the arithmetic and the LDS image: x, P and dP live in LDS as bf16 rows, the three GEMM-shaped
// phases run on v_mfma_f32_16x16x32_bf16 (f32 accumulate; 16x the f32-MFMA rate, so they stop being the
// bottleneck) and the gather phases read / write bf16.  P and dx take their B operand with one 16-byte LDS
// read per lane (8 consecutive k of a row).  dW contracts over NODES, i.e. down the LDS rows: each lane reads
// 4 columns of 8 consecutive rows (8 x ds_read_b64) and transposes the 8x4 block in registers with v_perm_b32,
// which yields four fragments whose 16 lanes cover columns {4i + p} — the same trick as linear_wgrad_b16.hip;
// the column permutation is undone when the accumulators are flushed.
// Rounding: x and gz arrive as bf16; P and dP are rounded to bf16 (2^-9 relative) before they feed the matrix
// cores, which is what a bf16 autocast backward does everywhere else in the model.
#include "fgnn_common.h"
#include <stdlib.h>

#define BB_THREADS 512
#define BB_WAVES 8
#define BB_MAXN 128
#define BB_GS 40         // gz_s / am_s row stride (32 channels of a pass + 8: rows stay 8-byte aligned)
#ifndef BB_PSB
#define BB_PSB 136       // P / dP row stride in bf16 elements (128 columns of a pass + 8)
#endif
#ifndef BB_XPAD
#define BB_XPAD 8
#endif

typedef __bf16 bb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bb_bf16x4 __attribute__((ext_vector_type(4)));

struct Bb16Params {
    fgnn_mpconv_desc d;
    const uint16_t* x;
    const int64_t* idx;
    const uint16_t* et;
    const float* W;      // [nin][ncols]
    const float* Wt;     // [ncols][nin] (workspace copy)
    const uint16_t* gz;
    const uint8_t* argmax;
    uint16_t* gx;
    uint16_t* get;       // or NULL
    float* ws;           // per-workgroup slabs [grid][nin*ncols + nou]
    int Npad16, Npad32;
    int off_xb, off_pb, off_db, off_gz, off_am, off_et, off_det, off_idx, off_cs, off_cl, off_ce;   // byte offsets
    unsigned kmagic;     // ceil(2^32 / k)
    long long* prof;     // FGNN_PROF: wave 0 of workgroup 0 stamps s_memtime at the phase boundaries of its 3rd sample
    int dbg;             // FGNN_DBG ablation mask (tuning only): 1 P, 2 dP, 4 det, 8 dx, 16 dW, 32 commit, 64 prefetch, 128 aT
};

extern __shared__ __attribute__((aligned(16))) unsigned char fgnn_lds_bb[];

void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);
void fgnn_launch_w_transpose(const float* W, float* Wt, int nin, int ncols, hipStream_t st);

__device__ __forceinline__ float bb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned bb_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ bb_bf16x8 bb_frag_f32(const float* p8) {      // 8 consecutive f32 -> one fragment
    const f32x4 a = *reinterpret_cast<const f32x4*>(p8), b = *reinterpret_cast<const f32x4*>(p8 + 4);
    return __builtin_bit_cast(bb_bf16x8, make_uint4(bb_pack2(a[0], a[1]), bb_pack2(a[2], a[3]),
                                                    bb_pack2(b[0], b[1]), bb_pack2(b[2], b[3])));
}
// rows r0..r7 each hold columns (c0 c1 | c2 c3) as two dwords: gather column P's eight values
template <int P>
__device__ __forceinline__ bb_bf16x8 bb_tr(const uint2 (&r)[8]) {
    constexpr unsigned sel = (P & 1) ? 0x07060302u : 0x05040100u;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned a = P < 2 ? r[2 * q].x : r[2 * q].y, b = P < 2 ? r[2 * q + 1].x : r[2 * q + 1].y;
        w[q] = __builtin_amdgcn_perm(b, a, sel);
    }
    return __builtin_bit_cast(bb_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}
__device__ __forceinline__ bb_bf16x8 bb_tr_dyn(const uint2 (&r)[8], int P) {
    switch (P) {
        case 0: return bb_tr<0>(r);
        case 1: return bb_tr<1>(r);
        case 2: return bb_tr<2>(r);
        default: return bb_tr<3>(r);
    }
}

// phase-timeline stamps (tuning aid): compiled in only with -DFGNN_ENABLE_PROF, read with FGNN_PROF=1
#ifdef FGNN_ENABLE_PROF
#define BB_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && tid == 0 && b == b_begin + 2) p.prof[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define BB_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ float bb_quad_sum(float v) {       // sum over the 4 lanes of a quad (all lanes get it)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

// KS2 = nin / 32 (k-steps of the P projection), NPASS = ncols / 128 (column passes, 32 output channels each)
template <int KS2, int NPASS>
__global__ __launch_bounds__(BB_THREADS) void mpconv_bwd_b16_kernel(const Bb16Params p) {
    constexpr int NIN = 32 * KS2;
    constexpr int NCOLS = 128 * NPASS;
    constexpr int NOU = 32 * NPASS;
    constexpr int XSB = NIN + BB_XPAD;                // x row stride (bf16 elements)
    constexpr int NCT = NIN / 16;                     // 16-channel tiles of dx: 4 or 8
    constexpr int DXT = (NCT == 4) ? 3 : 6;           // dx tiles per wave (Npad16 <= 96)
    constexpr int HX = NIN / 64;                      // 64-channel groups of x for the dW transposes
    constexpr int XQ = (96 * NIN / 8 + BB_THREADS - 1) / BB_THREADS;   // 16-byte x chunks per thread
    constexpr bool AT_RES = NPASS == 2;               // dx-projection fragments of both passes stay in registers
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = d.N, M = d.M, k = d.k;
    const int mk = M * k;

    uint16_t* xb = reinterpret_cast<uint16_t*>(fgnn_lds_bb + p.off_xb);     // [Npad32][XSB]   bf16 x
    uint16_t* pb = reinterpret_cast<uint16_t*>(fgnn_lds_bb + p.off_pb);     // [Npad32][PSB]   bf16 P of the pass
    uint16_t* db = reinterpret_cast<uint16_t*>(fgnn_lds_bb + p.off_db);     // [Npad32][PSB]   bf16 dP of the pass
    float* gz_s = reinterpret_cast<float*>(fgnn_lds_bb + p.off_gz);         // [M][GS]         f32
    uint8_t* am_s = fgnn_lds_bb + p.off_am;                                 // [M][GS]         argmax
    float* et_s = reinterpret_cast<float*>(fgnn_lds_bb + p.off_et);         // [mk][4]
    float* det_s = reinterpret_cast<float*>(fgnn_lds_bb + p.off_det);       // [mk][4]         detype of the sample
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_idx);
    int* cs_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_cs);             // CSR: start[N+1], then cnt[N]
    int* cl_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_cl);             // CSR: in-edge list (packed)
    int* ce_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_ce);             // CSR: et_s offset of each in-edge

    const int ct = wave % NCT;                        // this wave's channel tile of dx
    // ---- resident W fragments ----
    // aP: A of P^T = W^T x   : A[i = col][k = c] = W[c][col], 8 consecutive c (stride NCOLS)
    // aT: A of dx^T = W dP^T : A[i = c][k = col] = W[c][col],  8 consecutive cols
    bb_bf16x8 aP[NPASS][KS2];
    bb_bf16x8 aT[AT_RES ? NPASS : 1][4];
    {
        const int li0 = lane & 15, lk0 = lane >> 4;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                {   // W^T read in place (8 strided loads, once per kernel): no separate transpose launch
                    const float* wp = p.W + (int64_t)(32 * ks + 8 * lk0) * NCOLS + ps * 128 + wave * 16 + li0;
                    alignas(16) float w8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w8[u] = wp[(int64_t)u * NCOLS];
                    aP[ps][ks] = bb_frag_f32(w8);
                }
            if constexpr (AT_RES) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    aT[ps][ks] = bb_frag_f32(p.W + (int64_t)(ct * 16 + li0) * NCOLS + ps * 128 + 32 * ks + 8 * lk0);
            }
        }
    }

    f32x4 gw[NPASS][4 * HX];                          // dW accumulators: (pass, x group h, column slot p)
#pragma unroll
    for (int a = 0; a < NPASS; ++a)
#pragma unroll
        for (int t = 0; t < 4 * HX; ++t) gw[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gbacc[NPASS];                               // dbias of channel pass*32 + (tid & 31), one partial per 16 rows
#pragma unroll
    for (int a = 0; a < NPASS; ++a) gbacc[a] = 0.f;

    // ---- prefetch registers (raw bf16 chunks; nothing is decoded before the commit) ----
    uint4 px[XQ], pg;
    unsigned pa[2];
    uint2 pe;
    long long ir = 0;
    const int xchunks = N * (NIN / 8);
    auto prefetch_x = [&](int b, int t) {
        const uint4* xg = reinterpret_cast<const uint4*>(p.x + (int64_t)b * d.x_sb);
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = t + q * BB_THREADS;
            px[q] = f < xchunks ? xg[f] : make_uint4(0, 0, 0, 0);
        }
        pe = make_uint2(0, 0);
        if (t < mk) {
            pe = *reinterpret_cast<const uint2*>(p.et + (int64_t)b * d.et_sb + (int64_t)t * 4);
            const unsigned m = __umulhi((unsigned)t, p.kmagic), j = t - m * k;
            ir = (p.idx + (int64_t)b * d.idx_sb)[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
        }
    };
    // gz / argmax slice of (sample b, pass): 32 channels [pass*32, pass*32+32) of every destination m
    auto prefetch_g = [&](int b, int pass, int t) {
        const int64_t base = (int64_t)b * d.y_sb + pass * 32;
        pg = make_uint4(0, 0, 0, 0);
        if (t < 4 * M) pg = *reinterpret_cast<const uint4*>(p.gz + base + (int64_t)(t >> 2) * NOU + (t & 3) * 8);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int w = t + q * BB_THREADS;
            pa[q] = w < 8 * M ? *reinterpret_cast<const unsigned*>(p.argmax + base + (int64_t)(w >> 3) * NOU + (w & 7) * 4) : 0u;
        }
    };
    auto commit_x = [&](int t) {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = t + q * BB_THREADS;
            if (f < xchunks) {
                const int n = f / (NIN / 8), c8 = f % (NIN / 8);
                *reinterpret_cast<uint4*>(xb + n * XSB + c8 * 8) = px[q];
            }
        }
        if (t < mk) {
            *reinterpret_cast<f32x4*>(et_s + t * 4) = (f32x4){bb_lo(pe.x), bb_hi(pe.x), bb_lo(pe.y), bb_hi(pe.y)};
            *reinterpret_cast<f32x4*>(det_s + t * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            const long long v = ir;
            idx_s[t] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        }
    };
    auto commit_g = [&](int t) {
        if (t < 4 * M) {
            float* dst = gz_s + (t >> 2) * BB_GS + (t & 3) * 8;
            *reinterpret_cast<f32x4*>(dst) = (f32x4){bb_lo(pg.x), bb_hi(pg.x), bb_lo(pg.y), bb_hi(pg.y)};
            *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){bb_lo(pg.z), bb_hi(pg.z), bb_lo(pg.w), bb_hi(pg.w)};
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int w = t + q * BB_THREADS;
            if (w < 8 * M) *reinterpret_cast<unsigned*>(am_s + (w >> 3) * BB_GS + (w & 7) * 4) = pa[q];
        }
    };
    // Transposed incidence as a CSR over source nodes (sorted in-edge lists: fixed summation order)
    auto build_csr = [&]() {
        int* cnt = cs_s + BB_MAXN + 1;
        if (tid <= N) cs_s[tid] = 0;
        if (tid < N) cnt[tid] = 0;
        __syncthreads();
        int pos = 0, n = 0;
        if (tid < mk) { n = idx_s[tid]; pos = atomicAdd(&cnt[n], 1); }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int q = 0; q < N; ++q) { cs_s[q] = run; run += cnt[q]; }
            cs_s[N] = run;
        }
        __syncthreads();
        if (tid < mk) cl_s[cs_s[n] + pos] = tid;
        __syncthreads();
        if (tid < N) {
            const int lo = cs_s[tid], hi = cs_s[tid + 1];
            for (int a = lo + 1; a < hi; ++a) {
                const int v = cl_s[a];
                int c = a - 1;
                while (c >= lo && cl_s[c] > v) { cl_s[c + 1] = cl_s[c]; --c; }
                cl_s[c + 1] = v;
            }
            for (int a = lo; a < hi; ++a) {
                const int r = cl_s[a], m = r / k;
                cl_s[a] = ((m * BB_GS) << 8) | (r - m * k);
                ce_s[a] = r * 4;
            }
        }
        __syncthreads();
    };

    // zero the LDS images once: padding rows / columns are read by the matrix cores and never written again
    for (int f = tid; f < p.Npad32 * XSB / 2; f += BB_THREADS) reinterpret_cast<unsigned*>(xb)[f] = 0u;
    for (int f = tid; f < p.Npad32 * BB_PSB / 2; f += BB_THREADS) {
        reinterpret_cast<unsigned*>(pb)[f] = 0u;
        reinterpret_cast<unsigned*>(db)[f] = 0u;
    }

    const int ntile = p.Npad16 / 16;
    const int nkst = p.Npad32 / 32;
    const int chunk = (d.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int b_end = min(d.B, b_begin + chunk);
    if (b_begin < b_end) { prefetch_x(b_begin, tid); prefetch_g(b_begin, 0, tid); }
    const bool shared_graph = d.idx_sb == 0;
    const int ndet = p.get ? 4 * mk : 0;              // detype work items: (edge, 8-channel octet)
    const int ngath = ndet + p.Npad16 * 8;            // + dP work items: (source node, 4-channel group)

    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();                              // previous sample's MFMAs are done with xb / db, its writers with det_s
        BB_STAMP(0);
        int t = tid;
        asm volatile("" : "+v"(t));                   // keep per-lane offsets out of long-lived registers
        const int li = t & 15, lk = (t >> 4) & 3;
        if (!(p.dbg & 32)) { commit_x(t); commit_g(t); }
        BB_STAMP(1);
        __syncthreads();
        BB_STAMP(2);
        if (!(p.dbg & 64)) {
            if (b + 1 < b_end) prefetch_x(b + 1, t);
            BB_STAMP(20);
            prefetch_g(b, 1, t);                      // NPASS >= 2
            BB_STAMP(21);
        }
        if (!shared_graph || b == b_begin) build_csr();
        BB_STAMP(3);

        f32x4 dxacc[DXT];
#pragma unroll
        for (int i = 0; i < DXT; ++i) dxacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass > 0) {
                // gz_s / am_s were last read by the previous pass's gather, which ended at a barrier
                if (!(p.dbg & 32)) commit_g(t);
                if (!(p.dbg & 64)) {
                    if (pass + 1 < NPASS) prefetch_g(b, pass + 1, t);
                    else if (b + 1 < b_end) prefetch_g(b + 1, 0, t);
                }
            }
            const int APS = AT_RES ? pass : 0;
            if constexpr (!AT_RES) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    aT[0][ks] = bb_frag_f32(p.W + (int64_t)(ct * 16 + li) * NCOLS + pass * 128 + 32 * ks + 8 * lk);
            }
            BB_STAMP(5 + 8 * pass);
            // ---- P^T slab (wave = 16-column slab of the pass): D[i = col][j = n]; only detype needs P ----
            for (int nt = 0; nt < ((p.get && !(p.dbg & 1)) ? ntile : 0); ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const uint16_t* bp = xb + (nt * 16 + li) * XSB + 8 * lk;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        aP[pass][ks], __builtin_bit_cast(bb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32 * ks)), acc, 0, 0, 0);
                *reinterpret_cast<uint2*>(pb + (nt * 16 + li) * BB_PSB + wave * 16 + 4 * lk) =
                    make_uint2(bb_pack2(acc[0], acc[1]), bb_pack2(acc[2], acc[3]));
            }
            // dbias: thread (row group t >> 5, channel t & 31) sums rows t >> 5, +16, ... of this pass's gz slice
            // (gz_s of this pass was committed before the previous barrier or, for pass 0, before the sample's)
            __syncthreads();
            BB_STAMP(6 + 8 * pass);
            {
                float sgz = 0.f;
                for (int m = t >> 5; m < M; m += BB_THREADS / 32) sgz += gz_s[m * BB_GS + (t & 31)];
                gbacc[pass] += sgz;
            }

            // ---- gather phase: one work list, no barrier inside.  Items [0, ndet) are detype partials of
            //      (edge r, channel octet q) — the 4 octets of an edge sit in one quad and fold by DPP;
            //      items [ndet, ngath) are dP owners (source node n, 4-channel group og) gathering over n's
            //      in-edges.  All LDS reads of an item are issued before its arithmetic (masked, branch-free).
            for (int it = t; it < ngath; it += BB_THREADS) {
                if (it < ndet) {
                    if (p.dbg & 4) continue;
                    const int r = it >> 2, q = it & 3;
                    const unsigned m = __umulhi((unsigned)r, p.kmagic), j = r - m * k;
                    const uint2 a8 = *reinterpret_cast<const uint2*>(am_s + m * BB_GS + 8 * q);
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gz_s + m * BB_GS + 8 * q);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gz_s + m * BB_GS + 8 * q + 4);
                    const uint4* pp = reinterpret_cast<const uint4*>(pb + idx_s[r] * BB_PSB + 32 * q);
                    const uint4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];    // channels 8q..8q+7, 4 edge types each
                    const unsigned jj = j * 0x01010101u;
                    const unsigned xa = a8.x ^ jj, xb4 = a8.y ^ jj;                // zero byte <=> routed through edge j
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    auto one = [&](unsigned x4, int u, float g, unsigned lo2, unsigned hi2) {
                        const float gg = ((x4 >> (8 * u)) & 0xffu) == 0u ? g : 0.f;
                        acc[0] = fmaf(gg, bb_lo(lo2), acc[0]);
                        acc[1] = fmaf(gg, bb_hi(lo2), acc[1]);
                        acc[2] = fmaf(gg, bb_lo(hi2), acc[2]);
                        acc[3] = fmaf(gg, bb_hi(hi2), acc[3]);
                    };
                    one(xa, 0, g0[0], p0.x, p0.y); one(xa, 1, g0[1], p0.z, p0.w);
                    one(xa, 2, g0[2], p1.x, p1.y); one(xa, 3, g0[3], p1.z, p1.w);
                    one(xb4, 0, g1[0], p2.x, p2.y); one(xb4, 1, g1[1], p2.z, p2.w);
                    one(xb4, 2, g1[2], p3.x, p3.y); one(xb4, 3, g1[3], p3.z, p3.w);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = bb_quad_sum(acc[e]);
                    if (q == 0) {
                        f32x4* dsl = reinterpret_cast<f32x4*>(det_s + r * 4);
                        const f32x4 old = *dsl;
                        *dsl = (f32x4){old[0] + acc[0], old[1] + acc[1], old[2] + acc[2], old[3] + acc[3]};
                    }
                } else {
                    if (p.dbg & 2) continue;
                    const int o = it - ndet;
                    const int n = o >> 3, og = o & 7;
                    float acc[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
                    if (n < N) {
                        const int q1 = cs_s[n + 1];
                        for (int q0 = cs_s[n]; q0 < q1; q0 += 3) {
                            int ent[3], ceo[3];
#pragma unroll
                            for (int u = 0; u < 3; ++u) {
                                const bool ok = q0 + u < q1;
                                ent[u] = ok ? cl_s[q0 + u] : 0xff;          // slot 255 never matches an argmax
                                ceo[u] = ok ? ce_s[q0 + u] : 0;
                            }
                            unsigned a4[3];
                            f32x4 g4[3], e4[3];
#pragma unroll
                            for (int u = 0; u < 3; ++u) {
                                const int mrow = ent[u] >> 8;
                                a4[u] = *reinterpret_cast<const unsigned*>(am_s + mrow + og * 4);
                                g4[u] = *reinterpret_cast<const f32x4*>(gz_s + mrow + og * 4);
                                e4[u] = *reinterpret_cast<const f32x4*>(et_s + ceo[u]);
                            }
#pragma unroll
                            for (int u = 0; u < 3; ++u) {
                                const unsigned x4 = a4[u] ^ ((unsigned)(ent[u] & 0xff) * 0x01010101u);
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float g = ((x4 >> (8 * c)) & 0xffu) == 0u ? g4[u][c] : 0.f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) acc[c * 4 + e] = fmaf(g, e4[u][e], acc[c * 4 + e]);
                                }
                            }
                        }
                    }
                    uint4* dst = reinterpret_cast<uint4*>(db + n * BB_PSB + og * 16);
                    dst[0] = make_uint4(bb_pack2(acc[0], acc[1]), bb_pack2(acc[2], acc[3]),
                                        bb_pack2(acc[4], acc[5]), bb_pack2(acc[6], acc[7]));
                    dst[1] = make_uint4(bb_pack2(acc[8], acc[9]), bb_pack2(acc[10], acc[11]),
                                        bb_pack2(acc[12], acc[13]), bb_pack2(acc[14], acc[15]));
                }
            }
            __syncthreads();
            BB_STAMP(8 + 8 * pass);

            // ---- dx^T tiles (ct fixed per wave): D[i = c][j = n] += W[c][pass cols] . dP^T ----
#pragma unroll
            for (int i = 0; i < DXT; ++i) {
                const int nt = (NCT == 4) ? (wave / 4 + 2 * i) : i;
                if (nt < ntile && !(p.dbg & 8)) {
                    const uint16_t* bp = db + (nt * 16 + li) * BB_PSB + 8 * lk;
                    f32x4 acc = dxacc[i];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            aT[APS][ks], __builtin_bit_cast(bb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32 * ks)), acc, 0, 0, 0);
                    dxacc[i] = acc;
                }
            }
            BB_STAMP(9 + 8 * pass);
            // ---- dW: contraction over nodes.  A = x^T (4 column slots per 64-channel group), B = dP^T slot
            //      (h', p') = (wave / 4, wave % 4) of this pass's 128 columns ----
            for (int kst = 0; kst < ((p.dbg & 16) ? 0 : nkst); ++kst) {
                const int row0 = 32 * kst + 8 * lk;
                uint2 rd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    rd[j] = *reinterpret_cast<const uint2*>(db + (row0 + j) * BB_PSB + 64 * (wave >> 2) + 4 * li);
                const bb_bf16x8 bfr = bb_tr_dyn(rd, wave & 3);
#pragma unroll
                for (int h = 0; h < HX; ++h) {
                    uint2 rx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        rx[j] = *reinterpret_cast<const uint2*>(xb + (row0 + j) * XSB + 64 * h + 4 * li);
                    gw[pass][4 * h + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb_tr<0>(rx), bfr, gw[pass][4 * h + 0], 0, 0, 0);
                    gw[pass][4 * h + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb_tr<1>(rx), bfr, gw[pass][4 * h + 1], 0, 0, 0);
                    gw[pass][4 * h + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb_tr<2>(rx), bfr, gw[pass][4 * h + 2], 0, 0, 0);
                    gw[pass][4 * h + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb_tr<3>(rx), bfr, gw[pass][4 * h + 3], 0, 0, 0);
                }
            }
            BB_STAMP(10 + 8 * pass);
        }   // passes

        // ---- write gx (bf16, channel-fastest) and getype for this sample ----
        {
            uint16_t* gxb = p.gx + (int64_t)b * d.x_sb;
#pragma unroll
            for (int i = 0; i < DXT; ++i) {
                const int nt = (NCT == 4) ? (wave / 4 + 2 * i) : i;
                const int n = nt * 16 + li;
                if (nt < ntile && n < N)
                    *reinterpret_cast<uint2*>(gxb + (int64_t)n * NIN + ct * 16 + 4 * lk) =
                        make_uint2(bb_pack2(dxacc[i][0], dxacc[i][1]), bb_pack2(dxacc[i][2], dxacc[i][3]));
            }
            if (p.get && t < mk) {                     // det_s is complete since the last pass's gather barrier
                const f32x4 dv = *reinterpret_cast<const f32x4*>(det_s + t * 4);
                uint16_t* gb = p.get + (int64_t)b * 4 * mk;          // [4][M][k] contiguous
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 h = (__bf16)dv[e];
                    gb[e * mk + t] = __builtin_bit_cast(uint16_t, h);
                }
            }
        }
        BB_STAMP(40);
    }   // samples

    // ---- flush dW tiles and dbias into this workgroup's slab (summed by the slab reduce) ----
    if (b_begin < b_end) {
        const int li = lane & 15, lk = lane >> 4;
        float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)NIN * NCOLS + NOU);
        float* red = gz_s;                             // 16 row-group partials per dbias channel, folded in order
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int col = pass * 128 + 64 * (wave >> 2) + 4 * li + (wave & 3);
#pragma unroll
            for (int h = 0; h < HX; ++h)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(int64_t)(64 * h + 4 * (4 * lk + r) + pp) * NCOLS + col] = gw[pass][4 * h + pp][r];
            red[tid] = gbacc[pass];
            __syncthreads();
            if (tid < 32) {
                float sgz = 0.f;
                for (int g = 0; g < BB_THREADS / 32; ++g) sgz += red[g * 32 + tid];
                slab[(int64_t)NIN * NCOLS + pass * 32 + tid] = sgz;
            }
            __syncthreads();
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define BB_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] bf16-MFMA backward rejects shape: rule %d\n", code); return 0; } while (0)

static void* bb_pick(int KS2, int NPASS) {
#define BB_CASE(ks, np) if (KS2 == ks && NPASS == np) return (void*)mpconv_bwd_b16_kernel<ks, np>;
    BB_CASE(2, 2) BB_CASE(2, 4) BB_CASE(4, 2)
#undef BB_CASE
    return nullptr;
}

// Returns 1 if launched, 0 if the call is outside this kernel's family, <0 on error.
int fgnn_mpconv_backward_b16(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                             const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                             float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                             fgnn_stream_t stream) {
    static const bool off = getenv("FGNN_NO_BWD_B16") != nullptr;
    if (off) return 0;
    if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->agg != FGNN_AGG_MAX || d->net != 4) BB_REJECT(1);
    if ((d->nin != 64 && d->nin != 128) || (d->nou != 64 && d->nou != 128)) BB_REJECT(2);
    const int KS2 = d->nin / 32, NPASS = d->nou / 32;
    void* fn = bb_pick(KS2, NPASS);
    if (!fn) BB_REJECT(3);
    if (!(d->x_sc == 1 && d->x_sn == d->nin && d->x_sb % 8 == 0)) BB_REJECT(4);
    if (!(d->y_sc == 1 && d->y_sm == d->nou && d->y_sb % 8 == 0)) BB_REJECT(5);
    const int mk = d->M * d->k;
    if (!(d->et_se == 1 && d->et_sk == 4 && (d->et_sm == 4 * d->k || d->M == 1) && d->et_sb % 4 == 0)) BB_REJECT(6);
    if (d->N > 96 || d->N < 9 || mk > BB_THREADS || d->k > 255 || d->M > 128) BB_REJECT(7);
    if (((uintptr_t)x & 15) || ((uintptr_t)gz & 15) || ((uintptr_t)etype & 7) || ((uintptr_t)argmax & 3) ||
        ((uintptr_t)gx & 7)) BB_REJECT(8);
    const int64_t nw = (int64_t)d->nin * d->nou * 4, slab_len = nw + d->nou;
    if (!workspace || workspace_bytes < (256 * slab_len + nw) * 4) BB_REJECT(9);

    Bb16Params p;
    p.d = *d;
    p.x = (const uint16_t*)x; p.idx = nn_idx; p.et = (const uint16_t*)etype; p.W = filters;
    p.gz = (const uint16_t*)gz; p.argmax = argmax; p.gx = (uint16_t*)gx; p.get = (uint16_t*)getype;
    p.ws = (float*)workspace;
    p.Wt = p.ws + 256 * slab_len;
    p.Npad16 = fgnn_round_up(d->N, 16);
    p.Npad32 = fgnn_round_up(d->N, 32);
    { const char* e = getenv("FGNN_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.prof = nullptr;
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) { (void)hipMalloc(&prof_buf, 64 * 8); }
        (void)hipMemset(prof_buf, 0, 64 * 8);
        p.prof = prof_buf;
    }
    p.kmagic = d->k == 1 ? 0u : (unsigned)((0x100000000ULL + d->k - 1) / d->k);
    if (d->k == 1) BB_REJECT(10);
    int off_b = 0;
    auto take = [&](int bytes) { const int o = off_b; off_b = fgnn_round_up(off_b + bytes, 16); return o; };
    p.off_xb = take(p.Npad32 * (d->nin + BB_XPAD) * 2);
    p.off_pb = take(p.Npad32 * BB_PSB * 2);
    p.off_db = take(p.Npad32 * BB_PSB * 2);
    p.off_gz = take((d->M * BB_GS > BB_THREADS ? d->M * BB_GS : BB_THREADS) * 4);
    p.off_am = take(d->M * BB_GS);
    p.off_et = take(mk * 16);
    p.off_det = take(mk * 16);
    p.off_idx = take(mk * 4);
    p.off_cs = take((2 * BB_MAXN + 4) * 4);
    p.off_cl = take(mk * 4);
    p.off_ce = take(mk * 4);
    const int lds = off_b;
    if (lds > 160 * 1024) BB_REJECT(11);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int grid = 256;
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    fgnn_note_kernel("mpconv_bwd_b16_kernel<%d, %d>", KS2, NPASS);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BB_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv bf16 backward launch: %s", hipGetErrorString(e));
    fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    if (p.prof) {                                     // tuning aid: phase timeline of one sample (cycles at 100 MHz s_memtime)
        long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[fgnn prof]");
        for (int i = 0; i < 41; ++i) if (h[i]) fprintf(stderr, " %d:%lld", i, h[i] - h[0]);
        fprintf(stderr, "\n");
    }
    return 1;
}
