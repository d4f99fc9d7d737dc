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
#define BB_GS 36         // gz_s / am_s row stride (32 channels of a pass + 4)
#define BB_PSB 136       // P / dP row stride in bf16 elements (128 columns of a pass + 8)

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
    int off_xb, off_pb, off_gz, off_am, off_et, off_idx, off_cs, off_cl, off_ce;   // byte offsets
    unsigned kmagic;     // ceil(2^32 / k)
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

// KS2 = nin / 32 (k-steps of the P projection), NPASS = ncols / 128 (column passes, 32 output channels each)
template <int KS2, int NPASS>
__global__ __launch_bounds__(BB_THREADS) void mpconv_bwd_b16_kernel(const Bb16Params p) {
    constexpr int NIN = 32 * KS2;
    constexpr int NCOLS = 128 * NPASS;
    constexpr int NOU = 32 * NPASS;
    constexpr int XSB = NIN + 8;                      // x row stride (bf16 elements)
    constexpr int NCT = NIN / 16;                     // 16-channel tiles of dx: 4 or 8
    constexpr int DXT = (NCT == 4) ? 3 : 6;           // dx tiles per wave (Npad16 <= 96)
    constexpr int HX = NIN / 64;                      // 64-channel groups of x for the dW transposes
    constexpr int XQ = (96 * NIN / 8 + BB_THREADS - 1) / BB_THREADS;   // 16-byte x chunks per thread
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = d.N, M = d.M, k = d.k;
    const int mk = M * k;

    uint16_t* xb = reinterpret_cast<uint16_t*>(fgnn_lds_bb + p.off_xb);     // [Npad32][XSB]   bf16
    uint16_t* pb = reinterpret_cast<uint16_t*>(fgnn_lds_bb + p.off_pb);     // [Npad32][PSB]   bf16: P then dP
    float* gz_s = reinterpret_cast<float*>(fgnn_lds_bb + p.off_gz);         // [M][GS]         f32
    uint8_t* am_s = fgnn_lds_bb + p.off_am;                                 // [M][GS]         argmax
    float* et_s = reinterpret_cast<float*>(fgnn_lds_bb + p.off_et);         // [mk][4]
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_idx);
    int* cs_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_cs);             // CSR: start[N+1], then cnt[N]
    int* cl_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_cl);             // CSR: in-edge list (packed)
    int* ce_s = reinterpret_cast<int*>(fgnn_lds_bb + p.off_ce);             // CSR: et_s offset of each in-edge

    // ---- resident W fragments of the P projection: A[i = col][k = c] = Wt[col][c], 8 consecutive c ----
    bb_bf16x8 aP[NPASS][KS2];
    {
        const int li0 = lane & 15, lk0 = lane >> 4;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                aP[ps][ks] = bb_frag_f32(p.Wt + (int64_t)(ps * 128 + wave * 16 + li0) * NIN + 32 * ks + 8 * lk0);
    }
    const int ct = wave % NCT;                        // this wave's channel tile of dx

    f32x4 gw[NPASS][4 * HX];                          // dW accumulators: (pass, x group h, column slot p)
#pragma unroll
    for (int a = 0; a < NPASS; ++a)
#pragma unroll
        for (int t = 0; t < 4 * HX; ++t) gw[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gbacc[NPASS];                               // dbias of channel pass*32 + (tid - 448), wave 7 only
#pragma unroll
    for (int a = 0; a < NPASS; ++a) gbacc[a] = 0.f;

    // ---- prefetch registers (raw bf16 chunks) ----
    uint4 px[XQ], pg;
    unsigned pa[2];
    uint2 pe;
    int ir = 0;
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
            long long v = (p.idx + (int64_t)b * d.idx_sb)[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
            v = v < 0 ? 0 : (v >= N ? N - 1 : v);
            ir = (int)v;
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
            idx_s[t] = ir;
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
    for (int f = tid; f < p.Npad32 * BB_PSB / 2; f += BB_THREADS) reinterpret_cast<unsigned*>(pb)[f] = 0u;

    const int ntile = p.Npad16 / 16;
    const int nkst = p.Npad32 / 32;
    const int chunk = (d.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int b_end = min(d.B, b_begin + chunk);
    if (b_begin < b_end) { prefetch_x(b_begin, tid); prefetch_g(b_begin, 0, tid); }
    const bool shared_graph = d.idx_sb == 0;

    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();                              // previous sample's MFMAs are done with xb / pb
        int t = tid;
        asm volatile("" : "+v"(t));                   // keep per-lane offsets out of long-lived registers
        const int li = t & 15, lk = (t >> 4) & 3;
        commit_x(t);
        commit_g(t);
        __syncthreads();
        if (b + 1 < b_end) prefetch_x(b + 1, t);
        prefetch_g(b, 1, t);                          // NPASS >= 2
        if (!shared_graph || b == b_begin) build_csr();

        f32x4 dxacc[DXT];
#pragma unroll
        for (int i = 0; i < DXT; ++i) dxacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float dacc[4] = {0.f, 0.f, 0.f, 0.f};         // detype of edge r = tid (< mk)

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass > 0) {
                __syncthreads();                      // previous pass's dx / dW MFMAs are done with pb
                commit_g(t);
                if (pass + 1 < NPASS) prefetch_g(b, pass + 1, t);
                else if (b + 1 < b_end) prefetch_g(b + 1, 0, t);
            }
            // W fragments of the dx projection: A[i = c][k = col] = W[c][col], 8 consecutive cols (L2-resident)
            bb_bf16x8 aT[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                aT[ks] = bb_frag_f32(p.W + (int64_t)(ct * 16 + li) * NCOLS + pass * 128 + 32 * ks + 8 * lk);

            // ---- P^T slab (wave = 16-column slab of the pass): D[i = col][j = n]; only detype needs P ----
            for (int nt = 0; nt < (p.get ? ntile : 0); ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const uint16_t* bp = xb + (nt * 16 + li) * XSB + 8 * lk;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        aP[pass][ks], __builtin_bit_cast(bb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32 * ks)), acc, 0, 0, 0);
                *reinterpret_cast<uint2*>(pb + (nt * 16 + li) * BB_PSB + wave * 16 + 4 * lk) =
                    make_uint2(bb_pack2(acc[0], acc[1]), bb_pack2(acc[2], acc[3]));
            }
            __syncthreads();

            // ---- dbias (one idle half-wave) and detype owners: edge r = (m, j) sums over this pass's channels ----
            if (t >= 448 && t < 480) {
                float sgz = 0.f;
                for (int m = 0; m < M; ++m) sgz += gz_s[m * BB_GS + (t - 448)];
                gbacc[pass] += sgz;
            }
            if (p.get) {
                if (t < mk) {
                    const unsigned m = __umulhi((unsigned)t, p.kmagic), j = t - m * k;
                    const uint16_t* pn = pb + idx_s[t] * BB_PSB;
                    const float* gm = gz_s + m * BB_GS;
                    const uint8_t* am = am_s + m * BB_GS;
#pragma unroll 2
                    for (int ol = 0; ol < 32; ol += 4) {
                        const unsigned a4 = *reinterpret_cast<const unsigned*>(am + ol);
                        const unsigned x4 = a4 ^ (j * 0x01010101u);
                        if (((x4 - 0x01010101u) & ~x4 & 0x80808080u) != 0u) {
                            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gm + ol);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (((x4 >> (8 * u)) & 0xffu) == 0u) {
                                    const uint2 p4 = *reinterpret_cast<const uint2*>(pn + (ol + u) * 4);
                                    const float g = g4[u];
                                    dacc[0] = fmaf(g, bb_lo(p4.x), dacc[0]);
                                    dacc[1] = fmaf(g, bb_hi(p4.x), dacc[1]);
                                    dacc[2] = fmaf(g, bb_lo(p4.y), dacc[2]);
                                    dacc[3] = fmaf(g, bb_hi(p4.y), dacc[3]);
                                }
                            }
                        }
                    }
                }
                __syncthreads();                      // P is dead: the buffer becomes dP
            }

            // ---- dP owners: (source node n, group of 4 channels = 16 columns) gathers over n's in-edges ----
            for (int it = t; it < p.Npad16 * 8; it += BB_THREADS) {
                const int n = it >> 3, og = it & 7;
                float acc[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u] = 0.f;
                if (n < N) {
                    for (int q = cs_s[n]; q < cs_s[n + 1]; ++q) {
                        const int ent = cl_s[q];
                        const int mrow = ent >> 8, j = ent & 0xff;
                        const unsigned a4 = *reinterpret_cast<const unsigned*>(am_s + mrow + og * 4);
                        const unsigned x4 = a4 ^ ((unsigned)j * 0x01010101u);   // zero byte <=> routed through edge j
                        if (((x4 - 0x01010101u) & ~x4 & 0x80808080u) != 0u) {
                            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gz_s + mrow + og * 4);
                            const f32x4 e4 = *reinterpret_cast<const f32x4*>(et_s + ce_s[q]);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float g = ((x4 >> (8 * u)) & 0xffu) == 0u ? g4[u] : 0.f;
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[u * 4 + e] = fmaf(g, e4[e], acc[u * 4 + e]);
                            }
                        }
                    }
                }
                uint4* dst = reinterpret_cast<uint4*>(pb + n * BB_PSB + og * 16);
                dst[0] = make_uint4(bb_pack2(acc[0], acc[1]), bb_pack2(acc[2], acc[3]),
                                    bb_pack2(acc[4], acc[5]), bb_pack2(acc[6], acc[7]));
                dst[1] = make_uint4(bb_pack2(acc[8], acc[9]), bb_pack2(acc[10], acc[11]),
                                    bb_pack2(acc[12], acc[13]), bb_pack2(acc[14], acc[15]));
            }
            __syncthreads();

            // ---- dx^T tiles (ct fixed per wave): D[i = c][j = n] += W[c][pass cols] . dP^T ----
#pragma unroll
            for (int i = 0; i < DXT; ++i) {
                const int nt = (NCT == 4) ? (wave / 4 + 2 * i) : i;
                if (nt < ntile) {
                    const uint16_t* bp = pb + (nt * 16 + li) * BB_PSB + 8 * lk;
                    f32x4 acc = dxacc[i];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            aT[ks], __builtin_bit_cast(bb_bf16x8, *reinterpret_cast<const uint4*>(bp + 32 * ks)), acc, 0, 0, 0);
                    dxacc[i] = acc;
                }
            }
            // ---- dW: contraction over nodes.  A = x^T (4 column slots per 64-channel group), B = dP^T slot
            //      (h', p') = (wave / 4, wave % 4) of this pass's 128 columns ----
            for (int kst = 0; kst < nkst; ++kst) {
                const int row0 = 32 * kst + 8 * lk;
                uint2 rd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    rd[j] = *reinterpret_cast<const uint2*>(pb + (row0 + j) * BB_PSB + 64 * (wave >> 2) + 4 * li);
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
            if (p.get && t < mk) {
                uint16_t* gb = p.get + (int64_t)b * 4 * mk;          // [4][M][k] contiguous
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 h = (__bf16)dacc[e];
                    gb[e * mk + t] = __builtin_bit_cast(uint16_t, h);
                }
            }
        }
    }   // samples

    // ---- flush dW tiles and dbias into this workgroup's slab (summed by the slab reduce) ----
    if (b_begin < b_end) {
        const int li = lane & 15, lk = lane >> 4;
        float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)NIN * NCOLS + NOU);
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
            if (tid >= 448 && tid < 480) slab[(int64_t)NIN * NCOLS + pass * 32 + (tid - 448)] = gbacc[pass];
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
    p.kmagic = d->k == 1 ? 0u : (unsigned)((0x100000000ULL + d->k - 1) / d->k);
    if (d->k == 1) BB_REJECT(10);
    int off_b = 0;
    auto take = [&](int bytes) { const int o = off_b; off_b = fgnn_round_up(off_b + bytes, 16); return o; };
    p.off_xb = take(p.Npad32 * (d->nin + 8) * 2);
    p.off_pb = take(p.Npad32 * BB_PSB * 2);
    p.off_gz = take(d->M * BB_GS * 4);
    p.off_am = take(d->M * BB_GS);
    p.off_et = take(mk * 16);
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
    fgnn_launch_w_transpose(filters, (float*)p.Wt, d->nin, d->nou * 4, st);
    fgnn_note_kernel("mpconv_bwd_b16_kernel<%d, %d>", KS2, NPASS);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BB_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv bf16 backward launch: %s", hipGetErrorString(e));
    fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
