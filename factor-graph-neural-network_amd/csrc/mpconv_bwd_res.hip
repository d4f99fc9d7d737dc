// mpconv_bwd_res.hip — "resident" backward of the VF/FV message operator for the LDPC shape family
// (NO_EXTENSION, max aggregator, nin in {64,128}, nou*net in {64,...,512}).
//
// Math (reference autograd through /root/reference/lib/model/mpnn/mp_nn.py:115-175), per sample:
//     P[n,col]   = sum_c x[c,n] W[c,col]                              (recomputed, col = o*net+e)
//     dE[m,j,o]  = gz[o,m] * [j == argmax[o,m]]
//     detype[e,m,j] = sum_o dE[m,j,o] * P[idx[m,j], o*net+e]          (one owner thread per edge)
//     dP[n,col]  = sum_{(m,j): idx[m,j]=n} dE[m,j,o] * etype[e,m,j]   (one owner thread per (n,o): the
//                  transposed incidence is built in LDS as a sorted CSR, so the "scatter" is a gather:
//                  no atomics, no zero-fill, bit-reproducible)
//     dx[c,n]    = sum_col W[c,col] dP[n,col]                         (MFMA; accumulated over the column
//                                                                       passes in registers, written once)
//     dW[c,col] += sum_n x[c,n] dP[n,col]                             (MFMA; f32 accumulators live in
//                                                                       registers across ALL samples of a WG)
//     dbias[o]  += sum_m gz[o,m]
//
// Schedule: persistent 512-thread workgroups (1 per CU) loop over a contiguous chunk of samples; the next
// sample's x / etype / nn_idx / gz / argmax are prefetched into registers during the current one; columns
// are processed in passes of 128, P and dP of a pass share ONE [N x 128] f32 LDS buffer (P is dead once
// detype has been taken).  dW / dbias partials go to a per-workgroup slab of the caller's workspace and
// are summed by a second tiny kernel (global atomics on 16 K addresses from 256 workgroups cost more
// than the whole kernel: profiles/r01 notes).  A transposed copy of W in the workspace makes the
// dx-projection fragments coalesced loads.
#include "fgnn_common.h"
#include <stdlib.h>

#define BR_THREADS 512
#define BR_WAVES 8
#define BR_EPT 4        // etype prefetch registers per thread  (M*k*net <= 512*4)
#define BR_APT 3        // argmax prefetch words per thread per pass (M*otcP/4 <= 512*3)
#define BR_PASS_COLS 128
#define BR_MAXN 128     // CSR arrays are sized for N <= 128 source nodes

struct BresParams {
    fgnn_mpconv_desc d;
    const void* x;
    const int64_t* idx;
    const void* et;
    const float* W;
    const float* Wt;     // [ncols][nin] transposed copy of W (workspace)
    const void* gz;
    const uint8_t* argmax;
    void* gx;            // dtype T, x's element strides
    void* get;           // dtype T, or NULL (edge-weight gradient not wanted)
    float* ws;           // per-workgroup partial [grid][nin*ncols + nou]
    int has_bias;
    int Npad, Kpad, XS, PS, GS;     // GS: row stride of the per-pass gz / argmax images
    int cl_in, cl_y, et_mode;
    unsigned xdiv, xmagic, mkmagic, ymagic, ydiv;   // ydiv: slice width otcP (channel-fastest gz) or M
    unsigned y4magic;                               // ceil(2^32 / (otcP/4)): argmax words of a channel-fastest slice
    int otcP;                                       // channels per pass slice = min(128/net, nou)
    int XQ;              // prefetch slots [0,XQ) carry x, the rest gz
    int dbg;             // FGNN_DBG ablation mask (tuning only)
    int off_xs, off_pb, off_idx, off_et, off_gz, off_am, off_cs, off_cl, off_ce, off_red;
    int JS;              // in-edge list split of the dP owners (1 = off)
    int fast_bias;       // dbias straight from the prefetch registers (gz channel-fastest, 512 % nou == 0)
};

extern __shared__ __attribute__((aligned(16))) float fgnn_lds_br[];

// W [nin][ncols] -> Wt [ncols][nin]
__global__ __launch_bounds__(256) void bres_transpose_kernel(const float* __restrict__ W, float* __restrict__ Wt,
                                                             int nin, int ncols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (c0 + r < nin && k0 + tx < ncols) ? W[(int64_t)(c0 + r) * ncols + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < ncols && c0 + tx < nin) Wt[(int64_t)(k0 + r) * nin + c0 + tx] = tile[tx][r];
}

// KS = nin/4 (k-steps of the P projection), NPASS = column passes of 128, AP_RES = keep the P-projection
// fragments of all passes in registers (else re-read them, coalesced, at the top of each pass)
template <typename T, int NET, int KS, int NPASS, bool AP_RES>
__global__ __launch_bounds__(BR_THREADS) void mpconv_bwd_res_kernel(const BresParams p) {
    constexpr int NCT = KS / 4;                       // channel tiles of dx / dW: 4 (nin 64) or 8 (nin 128)
    constexpr int PSLAB = BR_PASS_COLS / 16;          // 8 column slabs per pass = one per wave
    constexpr int PKS = BR_PASS_COLS / 4;             // 32 k-steps of the dx projection per pass
    constexpr int DXT = (NCT == 4) ? 3 : 6;           // dx tiles per wave (Npad <= 96)
    constexpr int DWT = NCT * PSLAB / BR_WAVES;       // dW tiles per wave per pass (4 or 8)
    constexpr int PT = (KS == 32) ? 30 : 18;          // prefetch registers shared by x and gz
    constexpr int APN = AP_RES ? NPASS : 1;
    constexpr int net = NET;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nin = d.nin, nou = d.nou, N = d.N, M = d.M, k = d.k;
    const int ncols = nou * net;
    const int mk = M * k;
    const int XS = p.XS, PS = p.PS, GS = p.GS;

    float* xs = fgnn_lds_br + p.off_xs;               // [Npad][XS]  x, node-major
    float* pb = fgnn_lds_br + p.off_pb;               // [Npad][PS]  P, then dP, of the current pass
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_br + p.off_idx);
    float* et_s = fgnn_lds_br + p.off_et;             // [mk][net]
    float* gz_s = fgnn_lds_br + p.off_gz;             // [M][GS]     gz, channel-fastest
    uint8_t* am_s = reinterpret_cast<uint8_t*>(fgnn_lds_br + p.off_am);   // [M][GS] argmax
    int* cs_s = reinterpret_cast<int*>(fgnn_lds_br + p.off_cs);           // CSR: start[N+1], then cnt[N]
    int* cl_s = reinterpret_cast<int*>(fgnn_lds_br + p.off_cl);           // CSR: in-edge list (packed)
    int* ce_s = reinterpret_cast<int*>(fgnn_lds_br + p.off_ce);           // CSR: et_s offset of each in-edge

    const T* xg = static_cast<const T*>(p.x);
    const T* etg = static_cast<const T*>(p.et);
    const T* gzg = static_cast<const T*>(p.gz);

    // ---- W fragments ----
    // aP[kk] : A of P^T = W^T x   : W[c = 4kk+lk][col = pass*128 + wave*16 + li]
    // aT[ks] : A of dx^T = W dP^T : W[c = ct*16+li][col = pass*128 + 4ks + lk] = Wt[col][c], re-read per pass
    const int ct = wave % NCT;
    float aP[APN][KS];
    auto load_aP = [&](int slot, int ps_i) {
        const int colP = ps_i * BR_PASS_COLS + wave * 16 + li;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = 4 * kk + lk;
            aP[slot][kk] = colP < ncols ? p.W[(int64_t)c * ncols + colP] : 0.f;
        }
    };
    if constexpr (AP_RES) {
#pragma unroll
        for (int ps_i = 0; ps_i < NPASS; ++ps_i) load_aP(ps_i, ps_i);
    }
    float aT[PKS];

    // dW accumulators: tile u = wave + 8t of the pass's NCT x 8 tile grid, (ctile, colt) = (u % NCT, u / NCT)
    f32x4 gw[NPASS][DWT];
#pragma unroll
    for (int a = 0; a < NPASS; ++a)
#pragma unroll
        for (int t = 0; t < DWT; ++t) gw[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gbacc[NPASS];                               // dbias partials, channel pass*otp + tid % otcP
#pragma unroll
    for (int a = 0; a < NPASS; ++a) gbacc[a] = 0.f;

    // ---- prefetch registers ----
    float pr[PT], er[BR_EPT];
    unsigned ar[BR_APT];
    int ir = 0;
    const int xtot = nin * N, ytot = M * p.otcP;      // x block of a sample; gz / argmax slice of a pass
    const int xpad = XS - nin;                        // xs is always [n][c]
    auto prefetch_x = [&](int b, int t) {
        const T* xb = xg + (int64_t)b * d.x_sb;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            if (q < p.XQ) {
                const int f = t + q * BR_THREADS;
                pr[q] = f < xtot ? fgnn_ld(xb + f) : 0.f;
            }
        }
        const T* eb = etg + (int64_t)b * d.et_sb;
#pragma unroll
        for (int q = 0; q < BR_EPT; ++q) {
            const int f = t + q * BR_THREADS;
            er[q] = f < mk * net ? fgnn_ld(eb + f) : 0.f;
        }
        if (t < mk) {
            const int m = t / k, j = t - m * k;
            long long v = (p.idx + (int64_t)b * d.idx_sb)[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
            v = v < 0 ? 0 : (v >= N ? N - 1 : v);
            ir = (int)v;
        }
    };
    // gz / argmax slice of (sample b, pass): channels [o0, o0 + otcP).  Channel-fastest tensors: element f
    // of the slice is (m, ol) = (f / otcP, f % otcP) at m*nou + o0 + ol; node-fastest: one dense block.
    auto prefetch_g = [&](int b, int pass, int t) {
        const int o0 = pass * p.otcP;
        const T* gb = gzg + (int64_t)b * d.y_sb + (p.cl_y ? o0 : o0 * M);
        const uint8_t* ab = p.argmax + (int64_t)b * d.y_sb + (p.cl_y ? o0 : o0 * M);
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            if (q >= p.XQ) {
                const unsigned f = t + (q - p.XQ) * BR_THREADS;
                float v = 0.f;
                if ((int)f < ytot) {
                    if (p.cl_y) { const unsigned m = __umulhi(f, p.ymagic); v = fgnn_ld(gb + m * nou + (f - m * p.ydiv)); }
                    else v = fgnn_ld(gb + f);
                }
                pr[q] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < BR_APT; ++q) {
            const unsigned w = t + q * BR_THREADS;
            unsigned v = 0u;
            if ((int)(w * 4) < ytot) {
                if (p.cl_y) { const unsigned m = __umulhi(w, p.y4magic); v = *reinterpret_cast<const unsigned*>(ab + m * nou + (w * 4 - m * p.ydiv)); }
                else v = reinterpret_cast<const unsigned*>(ab)[w];
            }
            ar[q] = v;
        }
    };
    auto commit_x = [&](int t) {
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            if (q < p.XQ) {
                // x: dense element f is (n,c) = (f/nin, f%nin) channel-fastest, or (c,n) = (f/N, f%N)
                const unsigned f = t + q * BR_THREADS;
                if ((int)f < xtot) {
                    const unsigned hi = __umulhi(f, p.xmagic);          // f / xdiv
                    if (p.cl_in) xs[f + hi * xpad] = pr[q];
                    else xs[(f - hi * p.xdiv) * XS + hi] = pr[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BR_EPT; ++q) {
            const unsigned f = t + q * BR_THREADS;
            if ((int)f < mk * net) {
                if (p.et_mode == 1 || net == 1) {
                    et_s[f] = er[q];
                } else {
                    const unsigned e = __umulhi(f, p.mkmagic), r = f - e * mk;
                    et_s[r * net + e] = er[q];
                }
            }
        }
        if (t < mk) idx_s[t] = ir;
    };
    auto commit_g = [&](int t, float& bias_acc) {
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            if (q >= p.XQ) {
                const unsigned f = t + (q - p.XQ) * BR_THREADS;
                if ((int)f < ytot) {
                    const unsigned hi = __umulhi(f, p.ymagic);          // f / ydiv
                    if (p.cl_y) gz_s[hi * GS + (f - hi * p.ydiv)] = pr[q];     // (m, ol)
                    else gz_s[(f - hi * p.ydiv) * GS + hi] = pr[q];            // (ol, m) -> [m][ol]
                    if (p.fast_bias) bias_acc += pr[q];                        // ol == t % otcP for every slot
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BR_APT; ++q) {
            const unsigned f0 = (t + q * BR_THREADS) * 4;
            if ((int)f0 < ytot) {
                if (p.cl_y) {                                           // 4 consecutive channels of one m
                    const unsigned hi = __umulhi(f0, p.ymagic);
                    *reinterpret_cast<unsigned*>(am_s + hi * GS + (f0 - hi * p.ydiv)) = ar[q];
                } else {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const unsigned f = f0 + bb, hi = __umulhi(f, p.ymagic);
                        am_s[(f - hi * p.ydiv) * GS + hi] = (uint8_t)(ar[q] >> (8 * bb));
                    }
                }
            }
        }
    };
    // Transposed incidence as a CSR over source nodes: cs_s[n..n+1) delimits node n's in-edges in cl_s;
    // an entry packs (gz/argmax row offset m*GS) << 8 | j.  Lists are sorted by edge id so that the
    // summation order — hence the result — does not depend on thread scheduling.
    auto build_csr = [&]() {
        int* cnt = cs_s + BR_MAXN + 1;
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
        if (tid < N) {                                 // insertion sort of a short list, then pack
            const int lo = cs_s[tid], hi = cs_s[tid + 1];
            for (int a = lo + 1; a < hi; ++a) {
                const int v = cl_s[a];
                int c = a - 1;
                while (c >= lo && cl_s[c] > v) { cl_s[c + 1] = cl_s[c]; --c; }
                cl_s[c + 1] = v;
            }
            for (int a = lo; a < hi; ++a) {
                const int r = cl_s[a], m = r / k;
                cl_s[a] = ((m * GS) << 8) | (r - m * k);
                ce_s[a] = r * net;
            }
        }
        __syncthreads();
    };

    // zero the padded part of xs once
    for (int f = tid; f < p.Npad * p.Kpad; f += BR_THREADS) {
        const int n = f / p.Kpad, c = f - n * p.Kpad;
        if (n >= N || c >= nin) xs[n * XS + c] = 0.f;
    }

    const int ntile = p.Npad / 16;
    const int chunk = (d.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int b_end = min(d.B, b_begin + chunk);
    if (b_begin < b_end) { prefetch_x(b_begin, tid); prefetch_g(b_begin, 0, tid); }
    const int otp = BR_PASS_COLS / net;               // channels per pass (32 or 128)
    const bool shared_graph = d.idx_sb == 0;

    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();                              // previous sample's MFMAs are done with xs / pb
        int t = tid;
        asm volatile("" : "+v"(t));
        // lane coordinates re-derived from the opaque `t`: keeps the dozens of loop-invariant LDS
        // offsets below from being hoisted out of the sample loop into (spilled) registers
        const int li = t & 15, lk = (t >> 4) & 3;
        if (!(p.dbg & 128)) { commit_x(t); commit_g(t, gbacc[0]); }
        __syncthreads();
        if (!(p.dbg & 512)) {
            if (b + 1 < b_end) prefetch_x(b + 1, t);
            if (NPASS > 1) prefetch_g(b, 1, t);
            else if (b + 1 < b_end) prefetch_g(b + 1, 0, t);
        }
        if (!shared_graph || b == b_begin) build_csr();

        f32x4 dxacc[DXT];
#pragma unroll
        for (int i = 0; i < DXT; ++i) dxacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float dacc[NET];                              // detype of edge r = tid (< mk)
#pragma unroll
        for (int e = 0; e < NET; ++e) dacc[e] = 0.f;

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int o0 = pass * otp;
            const int otc = min(otp, nou - o0);
            if (pass > 0) {
                __syncthreads();                      // previous pass's dx/dW MFMAs are done with pb; gz_s is free
                commit_g(t, gbacc[pass]);
                if (pass + 1 < NPASS) prefetch_g(b, pass + 1, t);
                else if (b + 1 < b_end) prefetch_g(b + 1, 0, t);
            }
            const int APS = AP_RES ? pass : 0;        // folds after full unrolling of the pass loop
            if constexpr (!AP_RES) load_aP(0, pass);
            {
                const float* wt = p.Wt + (int64_t)(pass * BR_PASS_COLS + lk) * nin + ct * 16 + li;
#pragma unroll
                for (int ks = 0; ks < PKS; ++ks)
                    aT[ks] = (pass * BR_PASS_COLS + 4 * ks + lk < ncols && !(p.dbg & 64)) ? wt[(int64_t)ks * 4 * nin] : 0.f;
            }

            // ---- P^T slab (wave = column slab of the pass) ----
            for (int tp = 0; tp < ((p.dbg & 1) ? 0 : (ntile + 1) / 2); ++tp) {
                const int t0 = tp * 2;
                const bool two = (t0 + 1) < ntile;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* bp0 = xs + (t0 * 16 + li) * XS + lk;
                const float* bp1 = two ? bp0 + 16 * XS : bp0;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aP[APS][kk], bp0[kk * 4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aP[APS][kk], bp1[kk * 4], acc1, 0, 0, 0);
                }
                float* dst = pb + (t0 * 16 + li) * PS + wave * 16 + 4 * lk;
                *reinterpret_cast<f32x4*>(dst) = acc0;
                if (two) *reinterpret_cast<f32x4*>(dst + 16 * PS) = acc1;
            }
            __syncthreads();

            if (!p.fast_bias && tid < otc) {           // dbias without the register shortcut: walk the LDS column
                float sgz = 0.f;
                for (int m = 0; m < M; ++m) sgz += gz_s[m * GS + tid];
                gbacc[pass] += sgz;
            }
            // ---- detype owners: edge r = (m, j) sums over this pass's channels ----
            if (tid < mk && !(p.dbg & 4)) {
                const int m = tid / k, j = tid - m * k;
                const float* pn = pb + idx_s[tid] * PS;
                const float* gm = gz_s + m * GS;
                const uint8_t* am = am_s + m * GS;
                for (int ol = 0; ol < otc; ol += 4) {
                    const unsigned a4 = *reinterpret_cast<const unsigned*>(am + ol);
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gm + ol);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if ((int)((a4 >> (8 * u)) & 0xff) == j && ol + u < otc) {
                            const float g = g4[u];
                            if constexpr (NET == 4) {
                                const f32x4 p4 = *reinterpret_cast<const f32x4*>(pn + (ol + u) * 4);
                                dacc[0] = fmaf(g, p4[0], dacc[0]);
                                dacc[1] = fmaf(g, p4[1], dacc[1]);
                                dacc[2] = fmaf(g, p4[2], dacc[2]);
                                dacc[3] = fmaf(g, p4[3], dacc[3]);
                            } else {
                                dacc[0] = fmaf(g, pn[ol + u], dacc[0]);
                            }
                        }
                    }
                }
            }
            __syncthreads();                          // P is dead: the buffer becomes dP

            // ---- dP owners: (source node n, group of 4 channels) gathers over n's in-edges ----
            // One 32-bit read gives the argmax of 4 channels of a destination; weights and gz are only
            // fetched when one of them routes through this edge.  Few owners with long in-edge lists (one
            // source node feeding every destination: the LDPC hyper-factor F->V call) split each list over
            // JS thread groups and fold the partials, in a fixed order, through LDS.
            {
                constexpr int NA = 4 * NET;                    // accumulators per owner: 4 channels x net
                const int ogp = otp / 4;                       // channel groups per pass row
                const int ogv = (otc + 3) / 4;                 // ... that hold valid channels
                auto edge_into = [&](int q, int og, float (&acc)[NA]) {
                    const int ent = cl_s[q];
                    const int mrow = ent >> 8, j = ent & 0xff;
                    const unsigned a4 = *reinterpret_cast<const unsigned*>(am_s + mrow + og * 4);
                    const unsigned jj = (unsigned)j * 0x01010101u;
                    const unsigned x4 = a4 ^ jj;               // a zero byte <=> that channel routes through edge j
                    if (((x4 - 0x01010101u) & ~x4 & 0x80808080u) != 0u) {
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gz_s + mrow + og * 4);
                        const float* etp = et_s + ce_s[q];
                        float e[NET];
                        if constexpr (NET == 4) {
                            const f32x4 e4 = *reinterpret_cast<const f32x4*>(etp);
                            e[0] = e4[0]; e[1] = e4[1]; e[2] = e4[2]; e[3] = e4[3];
                        } else {
                            e[0] = etp[0];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float g = ((x4 >> (8 * u)) & 0xffu) == 0u ? g4[u] : 0.f;
#pragma unroll
                            for (int t2 = 0; t2 < NET; ++t2) acc[u * NET + t2] = fmaf(g, e[t2], acc[u * NET + t2]);
                        }
                    }
                };
                auto store_owner = [&](int n, int og, const float (&acc)[NA]) {
                    float* dst = pb + n * PS + og * 4 * NET;
#pragma unroll
                    for (int u = 0; u < NA; u += 4)
                        *reinterpret_cast<f32x4*>(dst + u) = (f32x4){acc[u], acc[u + 1], acc[u + 2], acc[u + 3]};
                };
                if (p.JS == 1) {
                    for (int it = tid; it < ((p.dbg & 2) ? 0 : p.Npad * ogp); it += BR_THREADS) {
                        const int n = it / ogp, og = it - n * ogp;
                        float acc[NA];
#pragma unroll
                        for (int u = 0; u < NA; ++u) acc[u] = 0.f;
                        if (n < N && og < ogv)
                            for (int q = cs_s[n]; q < cs_s[n + 1]; ++q) edge_into(q, og, acc);
                        store_owner(n, og, acc);
                    }
                } else {
                    for (int f = tid; f < p.Npad * (PS / 4); f += BR_THREADS)
                        reinterpret_cast<f32x4*>(pb)[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int owners = N * ogv;                // <= 64 by construction of JS
                    const int part = tid / owners, own = tid - part * owners;
                    float* red = fgnn_lds_br + p.off_red;
                    float acc[NA];
#pragma unroll
                    for (int u = 0; u < NA; ++u) acc[u] = 0.f;
                    if (part < p.JS && !(p.dbg & 2)) {
                        const int n = own / ogv, og = own - n * ogv;
                        for (int q = cs_s[n] + part; q < cs_s[n + 1]; q += p.JS) edge_into(q, og, acc);
#pragma unroll
                        for (int u = 0; u < NA; ++u) red[(part * owners + own) * NA + u] = acc[u];
                    }
                    __syncthreads();
                    if (part == 0) {
                        for (int q = 1; q < p.JS; ++q)
#pragma unroll
                            for (int u = 0; u < NA; ++u) acc[u] += red[(q * owners + own) * NA + u];
                        store_owner(own / ogv, own - (own / ogv) * ogv, acc);
                    }
                }
            }
            __syncthreads();

            // ---- dx^T tiles (ct fixed per wave): += W[ct rows][pass cols] . dP^T ----
#pragma unroll
            for (int i = 0; i < DXT; ++i) {
                const int nt = (NCT == 4) ? (wave / 4 + 2 * i) : i;
                if (nt < ntile && !(p.dbg & 8)) {
                    const float* bp = pb + (nt * 16 + li) * PS + lk;
                    f32x4 acc = dxacc[i];
#pragma unroll
                    for (int ks = 0; ks < PKS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[ks], bp[ks * 4], acc, 0, 0, 0);
                    dxacc[i] = acc;
                }
            }
            // ---- dW tiles: += x^T . dP over the nodes ----
            {
                const int ksteps = (p.dbg & 16) ? 0 : p.Npad / 4;
#pragma unroll
                for (int tt = 0; tt < DWT; ++tt) {
                    const int u = wave + BR_WAVES * tt;
                    const int ctile = u % NCT, colt = u / NCT;
                    const float* ap = xs + lk * XS + ctile * 16 + li;
                    const float* bp = pb + lk * PS + colt * 16 + li;
                    f32x4 acc = gw[pass][tt];
                    for (int kk = 0; kk < ksteps; ++kk)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * XS], bp[kk * 4 * PS], acc, 0, 0, 0);
                    gw[pass][tt] = acc;
                }
            }
        }   // passes

        // ---- write gx (x's dtype and element strides) and getype for this sample ----
        if (!(p.dbg & 256)) {
            T* gxb = reinterpret_cast<T*>(p.gx) + (int64_t)b * d.x_sb;
#pragma unroll
            for (int i = 0; i < DXT; ++i) {
                const int nt = (NCT == 4) ? (wave / 4 + 2 * i) : i;
                const int n = nt * 16 + li;
                if (nt < ntile && n < N) {
                    const int c0 = ct * 16 + 4 * lk;
                    if (p.cl_in) {
                        fgnn_st4(gxb + (int64_t)n * d.x_sn + c0, dxacc[i]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            fgnn_st(gxb + (int64_t)(c0 + r) * d.x_sc + (int64_t)n * d.x_sn, dxacc[i][r]);
                    }
                }
            }
            if (tid < mk && p.get) {
                T* gb = reinterpret_cast<T*>(p.get) + (int64_t)b * net * mk;      // [net][M][k] contiguous
#pragma unroll
                for (int e = 0; e < NET; ++e) fgnn_st(gb + e * mk + tid, dacc[e]);
            }
        }
    }   // samples

    // ---- flush dW tiles and dbias into this workgroup's slab (summed by bres_reduce_kernel) ----
    if (b_begin < b_end) {
        float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)nin * ncols + nou);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int tt = 0; tt < DWT; ++tt) {
                const int u = wave + BR_WAVES * tt;
                const int ctile = u % NCT, colt = u / NCT;
                const int col = pass * BR_PASS_COLS + colt * 16 + li;
                if (col < ncols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(int64_t)(ctile * 16 + 4 * lk + r) * ncols + col] = gw[pass][tt][r];
                }
            }
        }
        __syncthreads();
        float* red = pb;                              // P / dP are dead now
        float* parts = pb + nou;                      // [BR_THREADS] per-thread partials of one pass
        for (int f = tid; f < nou; f += BR_THREADS) red[f] = 0.f;
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (p.fast_bias) {                        // 512/otcP threads hold partials of channel tid % otcP: folded by the
                parts[tid] = gbacc[pass];             // channel's first thread in thread order (no float atomics: same bits every run)
                __syncthreads();
                const int o = pass * p.otcP + tid;
                if (tid < p.otcP && o < nou) {
                    float sgb = 0.f;
                    for (int q = tid; q < BR_THREADS; q += p.otcP) sgb += parts[q];
                    red[o] = sgb;
                }
                __syncthreads();
            } else if (tid < p.otcP && pass * p.otcP + tid < nou) {
                red[pass * p.otcP + tid] = gbacc[pass];
            }
        }
        __syncthreads();
        for (int f = tid; f < nou; f += BR_THREADS) slab[(int64_t)nin * ncols + f] = red[f];
    }
}

// Sums the per-workgroup slabs into gW / gbias (accumulating): out[i] += sum_w ws[w][i].
// 1024 threads = 64 consecutive elements (a 256-byte line per slab) x 16 slab groups: wave g walks slabs g, g+16, ... (up to 16
// independent line loads in flight), the 16 partials of an element are folded in a fixed order through LDS (bit-reproducible).
__global__ __launch_bounds__(1024) void bres_reduce_kernel(const float* __restrict__ ws, int nslab, int64_t slab_len,
                                                           int64_t nw, float* __restrict__ gW,
                                                           float* __restrict__ gbias, int overwrite, int ncols, int ld) {
    __shared__ float part[16][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < slab_len) {
        const float* base = ws + i;
        int w = g;
        for (; w + 48 < nslab; w += 64) {
            s0 += base[(int64_t)w * slab_len];
            s1 += base[(int64_t)(w + 16) * slab_len];
            s2 += base[(int64_t)(w + 32) * slab_len];
            s3 += base[(int64_t)(w + 48) * slab_len];
        }
        for (; w < nslab; w += 16) s0 += base[(int64_t)w * slab_len];
    }
    part[g][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < slab_len) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += part[q][e];
        // the slab's weight part is [rows][ncols]; its rows land ld apart in gW (ld == ncols: dense, the usual case)
        const int64_t o = ncols == ld ? i : (i / ncols) * ld + i % ncols;
        if (i < nw) gW[o] = overwrite ? s : gW[o] + s;
        else if (gbias) gbias[i - nw] = overwrite ? s : gbias[i - nw] + s;
    }
}

// Shared with mpconv_bwd_hyper.hip: fold `nslab` slabs of [nw + nou] floats into gW / gbias.
bool fgnn_fold_push(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gb, int kind, int a, int b, int c, int d);   // fold_batch.hip

void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st) {
    if (fgnn_fold_push(ws, nslab, slab_len, nw, gW, gbias, 0, 1, 1, 0, 0)) return;       // recorded: one launch folds every slab set of the pass
    hipLaunchKernelGGL(bres_reduce_kernel, dim3((unsigned)((slab_len + 63) / 64)), dim3(1024), 0, st, ws, nslab,
                       slab_len, nw, gW, gbias, 0, 1, 1);
}

// The same with the slab's [nw / ncols][ncols] weight block landing in rows `ld` apart (a column block of a wider gfilters).
void fgnn_launch_slab_reduce_ld(const float* ws, int nslab, int64_t slab_len, int64_t nw, int ncols, int ld, float* gW,
                                float* gbias, hipStream_t st) {
    if (fgnn_fold_push(ws, nslab, slab_len, nw, gW, gbias, 0, ncols, ld, 0, 0)) return;
    hipLaunchKernelGGL(bres_reduce_kernel, dim3((unsigned)((slab_len + 63) / 64)), dim3(1024), 0, st, ws, nslab,
                       slab_len, nw, gW, gbias, 0, ncols, ld);
}

// The same fold, STORED (out[i] = sum_w ws[w][i]): for outputs that are not accumulators (the batch-summed edge-type gradient).
void fgnn_launch_slab_store(const float* ws, int nslab, int64_t slab_len, float* out, hipStream_t st) {
    hipLaunchKernelGGL(bres_reduce_kernel, dim3((unsigned)((slab_len + 63) / 64)), dim3(1024), 0, st, ws, nslab,
                       slab_len, slab_len, out, (float*)nullptr, 1, 1, 1);
}

void fgnn_launch_w_transpose(const float* W, float* Wt, int nin, int ncols, hipStream_t st) {
    hipLaunchKernelGGL(bres_transpose_kernel, dim3((ncols + 31) / 32, (nin + 31) / 32), dim3(256), 0, st, W, Wt, nin,
                       ncols);
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
template <typename T, int NET>
static void* bres_pick(int KS, int NPASS) {
#define BR_CASE(ks, np, res) if (KS == ks && NPASS == np) return (void*)mpconv_bwd_res_kernel<T, NET, ks, np, res>;
    BR_CASE(16, 1, true) BR_CASE(16, 2, true) BR_CASE(16, 4, false)
    BR_CASE(32, 1, false) BR_CASE(32, 2, false)
#undef BR_CASE
    return nullptr;
}

int64_t fgnn_mpconv_backward_ext_extra_bytes(const fgnn_mpconv_desc* d);
extern "C" int64_t fgnn_mpconv_backward_workspace_bytes(const fgnn_mpconv_desc* d) {
    if (!d) return 0;
    const int64_t R = d->ext == FGNN_EXT_NONE ? d->nin : 2 * d->nin;
    const int64_t nw = R * d->nou * d->net;
    return (256 * (nw + d->nou) + nw) * 4 + fgnn_mpconv_backward_ext_extra_bytes(d);   // 256 slabs + the transposed filter copy (+ edge-type slabs)
}

#define BR_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] resident backward rejects shape: rule %d (line %d)\n", code, __LINE__); return 0; } while (0)

// Returns 1 if launched, 0 if the shape is outside this kernel's family, <0 on error.
int fgnn_mpconv_backward_resident(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                  const void* etype, const float* filters, const void* gz,
                                  const uint8_t* argmax, void* gx, void* getype, float* gfilters,
                                  float* gbias, void* workspace, int64_t workspace_bytes,
                                  fgnn_stream_t stream) {
    if (d->ext != FGNN_EXT_NONE || d->agg != FGNN_AGG_MAX) BR_REJECT(1);
    if (d->net != 1 && d->net != 4) BR_REJECT(2);
    const int ncols = d->nou * d->net;
    if (ncols % 16 != 0 || ncols > 4 * BR_PASS_COLS) BR_REJECT(3);
    if (d->nin != 64 && d->nin != 128) BR_REJECT(4);
    if (!workspace || workspace_bytes < fgnn_mpconv_backward_workspace_bytes(d)) BR_REJECT(5);
    const int Kpad = d->nin;
    const int Npad = fgnn_round_up(d->N, 16);
    if (Npad > 96 || d->N > BR_MAXN) BR_REJECT(6);
    if (d->nou % 4 != 0 || d->nou > BR_THREADS) BR_REJECT(7);
    const bool nchw = (d->x_sn == 1 && d->x_sc == d->N);
    const bool cl = d->x_sc == 1 && (d->x_sn == d->nin || d->N == 1);   // N == 1: the node stride is moot
    if (!nchw && !cl) BR_REJECT(8);
    const int cl_in = cl ? 1 : 0;
    if (!cl_in && d->N == 1) BR_REJECT(9);
    // gz / argmax: dense per sample, channel-fastest [M][nou] or node-fastest [nou][M]
    const bool y_cl = d->y_sc == 1 && (d->y_sm == d->nou || d->M == 1);
    const bool y_nchw = (d->y_sm == 1 || d->M == 1) && d->y_sc == d->M;
    if (!y_cl && !y_nchw) BR_REJECT(10);
    const int cl_y = y_cl ? 1 : 0;
    if (!cl_y && d->M == 1) BR_REJECT(11);
    const int mk = d->M * d->k;
    const int NPASS = (ncols + BR_PASS_COLS - 1) / BR_PASS_COLS;
    const int KS = Kpad / 4;
    const int PT = KS == 32 ? 30 : 18;
    const int otp = BR_PASS_COLS / d->net;
    const int otcP = otp < d->nou ? otp : d->nou;                      // channels per pass slice
    if (NPASS > 1 && d->nou % otp != 0) BR_REJECT(21);
    const int XQ = (d->nin * d->N + BR_THREADS - 1) / BR_THREADS;
    const int GQ = (d->M * otcP + BR_THREADS - 1) / BR_THREADS;
    if (XQ + GQ > PT) BR_REJECT(12);
    if (mk * d->net > BR_THREADS * BR_EPT || mk > BR_THREADS) BR_REJECT(13);
    if (d->M * otcP > BR_THREADS * BR_APT * 4 || otcP % 4 != 0 || d->nou % 4 != 0) BR_REJECT(14);
    int et_mode;
    if (d->net == 1) {
        if (!((d->et_sk == 1 || d->k == 1) && (d->et_sm == d->k || d->M == 1))) BR_REJECT(15);
        et_mode = 1;
    } else if (mk == 1) {
        if (d->et_se != 1) BR_REJECT(16);
        et_mode = 1;
    } else if ((d->et_sk == 1 || d->k == 1) && (d->et_sm == d->k || d->M == 1) && d->et_se == mk) {
        et_mode = 0;
    } else if (d->et_se == 1 && d->et_sk == d->net && (d->et_sm == d->k * d->net || d->M == 1)) {
        et_mode = 1;
    } else {
        BR_REJECT(99);
    }

    BresParams p;
    p.d = *d;
    p.x = x; p.idx = nn_idx; p.et = etype; p.W = filters; p.gz = gz; p.argmax = argmax;
    p.gx = gx; p.get = getype; p.has_bias = gbias != nullptr;
    p.Npad = Npad; p.Kpad = Kpad;
    p.XS = (Kpad + 29) / 32 * 32 + 2;
    p.PS = BR_PASS_COLS + 4;
    p.GS = otcP + 4;
    p.otcP = otcP;
    if ((int64_t)d->M * p.GS >= (1 << 23)) BR_REJECT(17);
    p.cl_in = cl_in; p.cl_y = cl_y; p.et_mode = et_mode; p.XQ = XQ;
    p.fast_bias = (cl_y && BR_THREADS % otcP == 0) ? 1 : 0;
    { const char* e = getenv("FGNN_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.xdiv = cl_in ? d->nin : d->N;
    p.ydiv = cl_y ? otcP : d->M;
    if (p.xdiv == 1 || p.ydiv == 1) BR_REJECT(18);
    p.xmagic = (unsigned)((0x100000000ULL + p.xdiv - 1) / p.xdiv);
    p.mkmagic = mk == 1 ? 0u : (unsigned)((0x100000000ULL + mk - 1) / mk);
    p.ymagic = (unsigned)((0x100000000ULL + p.ydiv - 1) / p.ydiv);
    p.y4magic = otcP / 4 == 1 ? 0u : (unsigned)((0x100000000ULL + otcP / 4 - 1) / (otcP / 4));
    if (cl_y && otcP / 4 == 1) BR_REJECT(22);
    int off = 0;
    p.off_xs = off;  off += Npad * p.XS;                    off = fgnn_round_up(off, 4);
    p.off_pb = off;  off += Npad * p.PS;                    off = fgnn_round_up(off, 4);
    p.off_idx = off; off += fgnn_round_up(mk, 4);
    p.off_et = off;  off += fgnn_round_up(mk * d->net, 4);
    p.off_gz = off;  off += d->M * p.GS;                    off = fgnn_round_up(off, 4);
    p.off_am = off;  off += (d->M * p.GS + 3) / 4;          off = fgnn_round_up(off, 4);
    p.off_cs = off;  off += 2 * BR_MAXN + 4;
    p.off_cl = off;  off += fgnn_round_up(mk, 4);
    p.off_ce = off;  off += fgnn_round_up(mk, 4);
    {   // few owners (N * channel groups <= 64): split their in-edge lists over up to 32 thread groups
        const int ogv = (otcP + 3) / 4, owners = d->N * ogv;
        p.JS = 1;
        if (owners <= 64 && mk >= 4 * owners) { p.JS = BR_THREADS / owners; if (p.JS > 32) p.JS = 32; }
        p.off_red = off;
        if (p.JS > 1) off += p.JS * owners * 4 * d->net;
    }
    const int lds = off * 4;
    if (lds > 160 * 1024) BR_REJECT(19);
    void* fn = nullptr;
    if (d->dtype == FGNN_F32) fn = d->net == 1 ? bres_pick<float, 1>(KS, NPASS) : bres_pick<float, 4>(KS, NPASS);
    else fn = d->net == 1 ? bres_pick<bf16_t, 1>(KS, NPASS) : bres_pick<bf16_t, 4>(KS, NPASS);
    if (!fn) BR_REJECT(20);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    int grid = 256;
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;                // every workgroup owns >= 1 sample (writes its slab)
    const int64_t nw = (int64_t)d->nin * ncols;
    const int64_t slab_len = nw + d->nou;
    p.ws = (float*)workspace;
    float* Wt = p.ws + 256 * slab_len;
    p.Wt = Wt;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bres_transpose_kernel, dim3((ncols + 31) / 32, (d->nin + 31) / 32), dim3(256), 0, st,
                       filters, Wt, d->nin, ncols);
    fgnn_note_kernel("mpconv_bwd_res_kernel<%s, %d, %d, %d, %s>", d->dtype ? "bf16_t" : "float", d->net, KS, NPASS,
                     (KS == 16 && NPASS <= 2) ? "true" : "false");
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BR_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv resident backward launch: %s", hipGetErrorString(e));
    fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
