// mpconv_bwd_ext.hip — backward of the VF/FV message operator for the synthetic-PGM calls (BASELINE configs 1 / 2 / 5):
// f32, ORIG_WITH_NEIGHBOR / ORIG_WITH_DIFF, 16 edge types, 64 -> 64 channels, max aggregation, N = M <= 64, graph and
// edge types shared by the batch.  Counterpart of mpconv_fwd_ext.hip (same node-level split S = x Ws, T = x Wt); the
// reference trains this operator through autograd (/root/reference/lib/model/mpnn/mp_nn.py:136-175).
//
// With z[m,o] = max_j E[m,j,o] + bias, E[m,j,o] = sum_e et[m,j,e] (S[m,o,e] + T[idx[m,j],o,e]) and j* the forward's argmax:
//     dS[m,o,e]   = gz[m,o] et[m,j*,e]
//     dT[n,o,e]   = sum over edges (m,j) into n with j == j*(m,o):  gz[m,o] et[m,j,e]
//     get[m,j*,e] += gz[m,o] (S[m,o,e] + T[n,o,e])                       (summed over the batch: the edge types are shared)
//     gx  = dS Ws^T + dT Wt^T          gWs = x^T dS        gWt = x^T dT        gbias = sum gz
// The shape-generic kernel (mpconv_bwd.hip) scatters dT, gW and gx with float atomics (order-dependent bits) and spends
// 6.6 ms per call at B = 1024; this one has NO atomics anywhere — dT is a gather over a CSR of the shared graph built once
// per workgroup, gW / gbias / get go through per-workgroup slabs folded in a fixed order — and all three GEMMs
// (P = x W recomputed, gx, gW: 50 MFLOP per sample) run on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32).
//
// Loop order is pass-major like the forward: a pass = 4 output channels = 64 S + 64 T columns of the projection; its
// filter slice (as B operand of P = x W and of gx = dP W^T) and the gW accumulators of the pass stay in registers while
// the workgroup runs all of its samples through it; gx of a sample is accumulated across the 16 passes by the same lanes
// with plain global read-modify-write (L2-resident, deterministic).  f32 MFMAs keep the SIMD's VALU busy (measured:
// tools/ubench/mfma_f32_partner.hip), so the phases of a stage simply follow each other on all 8 waves:
//     A  P = x [Ws | Wt]_pass                    (64 MFMAs per wave)     -> LDS P
//     B1 get += gz (S + T) at the argmax         (VALU, 256 threads)
//     B2 dS (waves 0-3), dT (waves 4-7)          (VALU)                  -> LDS dP (over P)
//     C  gx += dP W^T (global RMW), gW += x^T dP (64 + 64 MFMAs per wave)
// LDS rows: x image stride 68 floats, P / dP stride 132: every access pattern above is bank-conflict-free (the k index of
// the gW product runs over nodes in the order 4 lk + (kk & 3) + 16 (kk >> 2) for that reason).
#include "fgnn_common.h"
#include <stdlib.h>

#define BX_THREADS 512
#define BX_NIN 64
#define BX_NOU 64
#define BX_NET 16
#define BX_NCOLS 1024
#define BX_PCH 4
#define BX_NPASS (BX_NOU / BX_PCH)
#define BX_XS 68
#define BX_PS 132
#define BX_MAX_MK 640

struct BxParams {
    const float* x;
    const int64_t* idx;
    const float* et;
    const float* W;
    const float* gz;
    const uint8_t* am;
    float* gx;
    float* ws;               // [grid][nw + nou] gW / gbias slabs
    float* get_ws;           // [grid][16][N][k] edge-type gradient slabs, or null
    int B, N, k, ext, nou, ncols, npass, wvec;      // wvec: filter rows are 16-byte aligned (float4 loads)
    long long x_sb, y_sb, idx_sm, idx_sk, et_se, et_sm, et_sk;
    int off_xs, off_ps, off_dp, off_et, off_get, off_idx, off_csr, off_e16, off_gz, off_am, off_w;      // off_dp == off_ps: dP overwrites P
    long long slab_len, get_len;
    long long* prof;         // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): phase timeline of one stage
    int dbg;                 // prof builds: FGNN_EXT_DBG bits switch parts of phase B off (timing experiments)
};

extern __shared__ __attribute__((aligned(16))) unsigned char bx_lds[];

#ifdef FGNN_ENABLE_PROF
#define BX_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && pp == 1 && s == (ns > 1 ? 1 : 0)) p.prof[wave * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define BX_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int bx_krow(int kk, int lk) { return 4 * lk + (kk & 3) + 16 * (kk >> 2); }

// NARROW: fewer than 64 output channels (runtime column counts, guarded loads); WVEC: filter rows 16-byte aligned.
template <int AGG, bool NARROW, bool WVEC>
__global__ __launch_bounds__(BX_THREADS, 2) void mpconv_bwd_ext_kernel(const BxParams p) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int N = p.N, k = p.k, mk = N * k;
    const int ntile = (N + 15) >> 4;
    const bool want_get = p.get_ws != nullptr;
    const bool diff = p.ext == FGNN_EXT_DIFF;

    float* xs = reinterpret_cast<float*>(bx_lds + p.off_xs);          // [64][XS]
    float* ps = reinterpret_cast<float*>(bx_lds + p.off_ps);          // [64][PS]: P
    float* dps = reinterpret_cast<float*>(bx_lds + p.off_dp);         // [64][PS]: dP (its own image when LDS allows: B1 / dS / dT then run side by side)
    const bool sep = p.off_dp != p.off_ps;
    unsigned short* e16 = reinterpret_cast<unsigned short*>(bx_lds + p.off_e16);         // [64][16] first 16 in-edges (m << 4 | j) of a node, 0xffff = none
    float* et_s = reinterpret_cast<float*>(bx_lds + p.off_et);        // [mk][16]
    float* get_s = reinterpret_cast<float*>(bx_lds + p.off_get);      // [mk][16] batch-summed edge-type gradient
    int* idx_s = reinterpret_cast<int*>(bx_lds + p.off_idx);          // [mk]
    unsigned short* csr_ent = reinterpret_cast<unsigned short*>(bx_lds + p.off_csr);     // [mk] edge ids by source node
    int* csr_off = reinterpret_cast<int*>(bx_lds + p.off_csr + ((mk * 2 + 15) & ~15));  // [65]
    float* gz_s = reinterpret_cast<float*>(bx_lds + p.off_gz);        // [64][4]
    unsigned* am_s = reinterpret_cast<unsigned*>(bx_lds + p.off_am);  // [64] four argmax bytes
    float* w_s = reinterpret_cast<float*>(bx_lds + p.off_w);          // LSE: [mk][4] softmax weight of edge (m, j) per channel of the pass
    const int nou = NARROW ? p.nou : BX_NOU, ncols = NARROW ? p.ncols : BX_NCOLS, npass = NARROW ? p.npass : BX_NPASS;

    const int chunk = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int ns = min(p.B, b_begin + chunk) - b_begin;
    float* slab = p.ws + (int64_t)blockIdx.x * p.slab_len;
    if (ns <= 0) {                                                    // the fold reads every slab
        for (int64_t f = tid; f < p.slab_len; f += BX_THREADS) slab[f] = 0.f;
        if (want_get) for (int64_t f = tid; f < p.get_len; f += BX_THREADS) p.get_ws[(int64_t)blockIdx.x * p.get_len + f] = 0.f;
        return;
    }

    // ---------------- one-time setup: graph tables, edge types, CSR by source node ----------------
    for (int f = tid; f < 64 * BX_XS; f += BX_THREADS) xs[f] = 0.f;
    for (int r = tid; r < mk; r += BX_THREADS) {
        const int m = r / k, j = r - m * k;
        long long v = p.idx[(int64_t)m * p.idx_sm + (int64_t)j * p.idx_sk];
        idx_s[r] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
    }
    for (int f = tid; f < mk * BX_NET; f += BX_THREADS) {
        const int e = f / mk, r = f - e * mk;
        const int m = r / k, j = r - m * k;
        et_s[r * BX_NET + e] = p.et[(int64_t)e * p.et_se + (int64_t)m * p.et_sm + (int64_t)j * p.et_sk];
        get_s[r * BX_NET + e] = 0.f;
    }
    __syncthreads();
    if (tid < 64) {                                                   // in-degree of node tid
        int c = 0;
        if (tid < N) for (int r = 0; r < mk; ++r) c += idx_s[r] == tid;
        gz_s[tid] = __int_as_float(c);
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int n = 0; n < 64; ++n) { csr_off[n] = a; a += __float_as_int(gz_s[n]); }
        csr_off[64] = a;
    }
    __syncthreads();
    if (tid < N) {                                                    // edges into node tid, ascending edge id: a fixed order
        int o = csr_off[tid];
        for (int r = 0; r < mk; ++r) if (idx_s[r] == tid) csr_ent[o++] = (unsigned short)(((r / k) << 4) | (r % k));      // (m, j)
    }
    __syncthreads();
    for (int f = tid; f < 64 * 16; f += BX_THREADS) {                 // padded copy: one 32-byte read gives a thread all of a node's in-edges
        const int n = f >> 4, i = csr_off[n] + (f & 15);
        e16[f] = (n < N && i < csr_off[n + 1]) ? csr_ent[i] : (unsigned short)0xffff;
    }

    // ---------------- per-thread roles ----------------
    // phase A: wave = (half, sl): slab of 16 columns = channel sl of the pass, S (half 0) or T (half 1)
    const int a_half = wave >> 2, a_sl = wave & 3;
    // phase B1: 256 threads = (destination m, quad of edge types eq); lanes of a 16-group take m 4 apart (banks)
    const int b1_eq = lane & 3, b1_m = 16 * wave + 4 * ((lane >> 2) & 3) + (lane >> 4);
    // phase B2: waves 0-3 thread (m, oc) -> dS; waves 4-7 thread (n, oc) -> dT
    const int b2_t = tid & 255, b2_row = b2_t >> 2, b2_oc = b2_t & 3;
    // phase C: gx: wave = (c tile ct, node-tile pair np); gW: wave = (column tile wt of S and of T, c-tile pair cp)
    const int c_ct = wave & 3, c_np = wave >> 2;
    const int w_t = wave & 3, w_cp = wave >> 2;

    // prefetch registers: x of the next stage (2 x 16 B per thread), gz / argmax slices (threads < 64)
    uint4 xr[2];
    uint4 gzr = make_uint4(0, 0, 0, 0);
    unsigned amr = 0;
    auto prefetch = [&](int b, int pass) {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x + (int64_t)b * p.x_sb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * BX_THREADS;
            xr[q] = (f >> 4) < N ? xb[f] : make_uint4(0, 0, 0, 0);
        }
        if (tid < 64) {
            gzr = make_uint4(0, 0, 0, 0);
            amr = 0;
            if (tid < N) {
                const int64_t o = (int64_t)b * p.y_sb + (int64_t)tid * nou + BX_PCH * pass;
                if (!NARROW) {
                    gzr = *reinterpret_cast<const uint4*>(p.gz + o);
                    if (AGG == FGNN_AGG_MAX) amr = *reinterpret_cast<const unsigned*>(p.am + o);
                } else {                                              // narrow outputs (64 -> 2 closes factor_mpnn): guarded scalars
                    float gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (BX_PCH * pass + i < nou) {
                            gq[i] = p.gz[o + i];
                            if (AGG == FGNN_AGG_MAX) amr |= (unsigned)p.am[o + i] << (8 * i);
                        }
                    gzr = make_uint4(__float_as_uint(gq[0]), __float_as_uint(gq[1]), __float_as_uint(gq[2]), __float_as_uint(gq[3]));
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * BX_THREADS;
            if ((f >> 4) < N) *reinterpret_cast<uint4*>(xs + (f >> 4) * BX_XS + (f & 15) * 4) = xr[q];
        }
        if (tid < 64) {
            *reinterpret_cast<uint4*>(gz_s + tid * 4) = gzr;
            am_s[tid] = amr;
        }
    };

    __syncthreads();
    prefetch(b_begin, blockIdx.x % npass);

    for (int pp = 0; pp < npass; ++pp) {
        const int pass = (pp + blockIdx.x) % npass;                   // staggered: the chip does not read one filter slice at once
        const int pass_next = (pp + 1 + blockIdx.x) % npass;
        // ---- filter slice of the pass ----
        float aW[16];                                                 // phase A B-operand: W[c = 16 lk + kk][column of this wave's slab]
        {
            const int col = 16 * (BX_PCH * pass + a_sl) + li;
            const bool live = !NARROW || BX_PCH * pass + a_sl < nou;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = 16 * lk + kk;
                const float wt = live ? p.W[(int64_t)c * ncols + col] : 0.f, wb = live ? p.W[(int64_t)(BX_NIN + c) * ncols + col] : 0.f;
                aW[kk] = a_half == 0 ? (diff ? wt + wb : wt) : (diff ? -wb : wb);
            }
        }
        float gB[32];                                                 // gx B-operand: Wc[dP column 32 lk + kk][c = 16 ct + li]
        {
            const int c = 16 * c_ct + li;
            const float* wtop = p.W + (int64_t)c * ncols + 64 * pass + 32 * (lk & 1);
            const float* wbot = wtop + (int64_t)BX_NIN * ncols;
            const bool live = !NARROW || 64 * pass + 32 * (lk & 1) < ncols;     // ncols is a multiple of 32 (nou even)
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 t = zero, b = zero;
                if (live && WVEC) {
                    t = *reinterpret_cast<const f32x4*>(wtop + 4 * q);
                    b = *reinterpret_cast<const f32x4*>(wbot + 4 * q);
                } else if (live) {                                    // parameters living at odd offsets of a flat buffer
#pragma unroll
                    for (int i = 0; i < 4; ++i) { t[i] = wtop[4 * q + i]; b[i] = wbot[4 * q + i]; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    gB[4 * q + i] = lk < 2 ? (diff ? t[i] + b[i] : t[i]) : (diff ? -b[i] : b[i]);
            }
        }
        f32x4 accW[2][2];                                             // [c tile of the pair][S, T]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h) accW[a][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float gb_acc = 0.f;

        for (int s = 0; s < ns; ++s) {
            const int b = b_begin + s;
            __syncthreads();                                          // previous stage is done with xs, P, gz_s
            commit();
            if (s + 1 < ns) prefetch(b + 1, pass);
            else if (pp + 1 < npass) prefetch(b_begin, pass_next);
            __syncthreads();
            BX_STAMP(0);

            // ================= phase A: P = x [Ws | Wt] =================
            {
                const int colbase = 64 * a_half + 16 * a_sl + li;
                for (int np = 0; np < (ntile + 1) >> 1; ++np) {
                    f32x4 bq[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float* bp = xs + ((2 * np + h) * 16 + li) * BX_XS + 16 * lk;
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[h][q] = *reinterpret_cast<const f32x4*>(bp + 4 * q);
                    }
                    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[h][kk >> 2][kk & 3], aW[kk], acc[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ps[((2 * np + h) * 16 + 4 * lk + r) * BX_PS + colbase] = acc[h][r];
                }
            }
            BX_STAMP(1);
            __syncthreads();
            BX_STAMP(2);

            if (AGG == FGNN_AGG_LSE) {
                // ================= phase B0 (softmax aggregator): w[m][j][oc] = exp(3 E_j - 3 agg) =================
                if (tid < 256 && b2_row < N) {
                    const float* srow = ps + b2_row * BX_PS + 16 * b2_oc;
                    f32x4 sv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const f32x4*>(srow + 4 * q);
                    float mx = -INFINITY, ssum = 0.f;
                    for (int j = 0; j < k; ++j) {
                        const int r = b2_row * k + j;
                        const float* trow = ps + idx_s[r] * BX_PS + 64 + 16 * b2_oc;
                        const float* er = et_s + r * BX_NET;
                        float v = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 tv = *reinterpret_cast<const f32x4*>(trow + 4 * q), ev = *reinterpret_cast<const f32x4*>(er + 4 * q);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v = fmaf(ev[i], sv[q][i] + tv[i], v);
                        }
                        v *= 3.0f;
                        w_s[r * 4 + b2_oc] = v;
                        if (v > mx) { ssum = ssum * expf(mx - v) + 1.0f; mx = v; }
                        else ssum += expf(v - mx);
                    }
                    const float zagg = mx + logf(ssum);
                    for (int j = 0; j < k; ++j) {
                        const int r = b2_row * k + j;
                        w_s[r * 4 + b2_oc] = expf(w_s[r * 4 + b2_oc] - zagg);
                    }
                }
                __syncthreads();
            }

            // ================= phase B: B1 get += gz w (S + T) and dS on waves 0-3, dT on waves 4-7 =================
            auto phase_b1 = [&]() {                                   // get[m][j][e] += gz w_j (S + T)   (max: w = 1 at the argmax only)
                if (want_get && b1_m < N) {
                    const unsigned am4 = am_s[b1_m];
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gz_s + b1_m * 4);
#pragma unroll
                    for (int oc = 0; oc < BX_PCH; ++oc) {
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(ps + b1_m * BX_PS + 16 * oc + 4 * b1_eq);
                        if (AGG == FGNN_AGG_MAX) {
                            const int j = min((int)((am4 >> (8 * oc)) & 255u), k - 1);
                            const int r = b1_m * k + j;
                            const f32x4 tv = *reinterpret_cast<const f32x4*>(ps + idx_s[r] * BX_PS + 64 + 16 * oc + 4 * b1_eq);
                            f32x4* gp = reinterpret_cast<f32x4*>(get_s + r * BX_NET + 4 * b1_eq);
                            *gp = *gp + g4[oc] * (sv + tv);
                        } else {
                            for (int j = 0; j < k; ++j) {
                                const int r = b1_m * k + j;
                                const f32x4 tv = *reinterpret_cast<const f32x4*>(ps + idx_s[r] * BX_PS + 64 + 16 * oc + 4 * b1_eq);
                                f32x4* gp = reinterpret_cast<f32x4*>(get_s + r * BX_NET + 4 * b1_eq);
                                *gp = *gp + (g4[oc] * w_s[r * 4 + oc]) * (sv + tv);
                            }
                        }
                    }
                }
            };
            auto phase_ds = [&]() {                                   // thread (m, oc): dS[m][oc][:] = gz sum_j w_j et[m][j][:]
                f32x4 row[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) row[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (b2_row < N) {
                    const float g = gz_s[b2_row * 4 + b2_oc];
                    if (AGG == FGNN_AGG_MAX) {
                        const int j = min((int)((am_s[b2_row] >> (8 * b2_oc)) & 255u), k - 1);
                        const float* er = et_s + (b2_row * k + j) * BX_NET;
#pragma unroll
                        for (int q = 0; q < 4; ++q) row[q] = g * *reinterpret_cast<const f32x4*>(er + 4 * q);
                    } else {
                        for (int j = 0; j < k; ++j) {
                            const int r = b2_row * k + j;
                            const float gw = g * w_s[r * 4 + b2_oc];
                            const float* er = et_s + r * BX_NET;
#pragma unroll
                            for (int q = 0; q < 4; ++q) row[q] = gw * *reinterpret_cast<const f32x4*>(er + 4 * q) + row[q];
                        }
                    }
                    gb_acc += g;
                }
                float* dr = dps + b2_row * BX_PS + 16 * b2_oc;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dr + 4 * q) = row[q];
            };
            auto phase_dt = [&]() {                                   // thread (n, oc): dT[n][oc][:] = sum over in-edges gz w et
                f32x4 row[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) row[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (b2_row < N) {
                    // all in-edge ids in two 16-byte reads, then every edge's loads at once: no branch on the argmax test
                    // (a losing edge gets weight zero), no load waits for an earlier edge
                    const uint4 ea = *reinterpret_cast<const uint4*>(e16 + b2_row * 16), eb = *reinterpret_cast<const uint4*>(e16 + b2_row * 16 + 8);
                    const unsigned ew[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ent = (ew[i >> 1] >> (16 * (i & 1))) & 0xffff;
                        if (ent != 0xffff) {
                            const int m = ent >> 4, j = ent & 15;
                            const int r = m * k + j;
                            float g = gz_s[m * 4 + b2_oc];
                            if (AGG == FGNN_AGG_MAX) g = (int)((am_s[m] >> (8 * b2_oc)) & 255u) == j ? g : 0.f;
                            else g *= w_s[r * 4 + b2_oc];
                            const float* er = et_s + r * BX_NET;
#pragma unroll
                            for (int q = 0; q < 4; ++q) row[q] = g * *reinterpret_cast<const f32x4*>(er + 4 * q) + row[q];
                        }
                    }
                    const int e1 = csr_off[b2_row + 1];
                    for (int i = csr_off[b2_row] + 16; i < e1; ++i) {  // in-degree above 16: the tail of the list
                        const int ent = csr_ent[i];
                        const int m = ent >> 4, j = ent & 15;
                        const int r = m * k + j;
                        float g = gz_s[m * 4 + b2_oc];
                        if (AGG == FGNN_AGG_MAX) g = (int)((am_s[m] >> (8 * b2_oc)) & 255u) == j ? g : 0.f;
                        else g *= w_s[r * 4 + b2_oc];
                        const float* er = et_s + r * BX_NET;
#pragma unroll
                        for (int q = 0; q < 4; ++q) row[q] = g * *reinterpret_cast<const f32x4*>(er + 4 * q) + row[q];
                    }
                }
                float* dr = dps + b2_row * BX_PS + 64 + 16 * b2_oc;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dr + 4 * q) = row[q];
            };
            if (sep) {
                if (tid < 256) { phase_b1(); phase_ds(); }
                else phase_dt();
            } else {                                                  // dP overwrites P: get must have read P first
                if (tid < 256) phase_b1();
                BX_STAMP(3);
                __syncthreads();
                if (tid < 256) phase_ds();
                else phase_dt();
            }
            BX_STAMP(4);
            __syncthreads();
            BX_STAMP(5);

            // ================= phase C: gx += dP Wc (global RMW), gW += x^T dP =================
            {
                float* gxb = p.gx + (int64_t)b * p.x_sb + 16 * c_ct + li;
                f32x4 old[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    old[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (pp > 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int node = (2 * c_np + h) * 16 + 4 * lk + r;
                            if (node < N) old[h][r] = gxb[(int64_t)node * BX_NIN];
                        }
                    }
                }
                // gW first (the gx read-modify-write loads fly meanwhile)
                float xa[2][16];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) xa[a][kk] = xs[bx_krow(kk, lk) * BX_XS + 16 * (2 * w_cp + a) + li];
                float db[2][16];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) db[h][kk] = dps[bx_krow(kk, lk) * BX_PS + 64 * h + 16 * w_t + li];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            accW[a][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[a][kk], db[h][kk], accW[a][h], 0, 0, 0);
                // gx tiles: nodes (2 np, 2 np + 1) x c tile ct, K = 128 dP columns in the order 32 lk + kk
                f32x4 accx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int half = 0; half < 2; ++half) {               // 16 k-steps at a time: 2 x 4 b128 of operands in flight
                    f32x4 dq[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float* dp = dps + ((2 * c_np + h) * 16 + li) * BX_PS + 32 * lk + 16 * half;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dq[h][q] = *reinterpret_cast<const f32x4*>(dp + 4 * q);
                    }
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            accx[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[h][kk >> 2][kk & 3], gB[16 * half + kk], accx[h], 0, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int node = (2 * c_np + h) * 16 + 4 * lk + r;
                        if (node < N) gxb[(int64_t)node * BX_NIN] = old[h][r] + accx[h][r];
                    }
                BX_STAMP(6);
            }
        }

        // ---- end of pass: filter-gradient tiles and the bias gradient of the pass -> this workgroup's slab ----
        {
            // D[i = c = 16 (2 cp + a) + 4 lk + r][j = column 16 t + li]; filters row c (top) and 64 + c (bottom)
            const int col = 64 * pass + 16 * w_t + li;
            if (!NARROW || col < ncols) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * (2 * w_cp + a) + 4 * lk + r;
                        const float gs = accW[a][0][r], gt = accW[a][1][r];
                        slab[(int64_t)c * ncols + col] = gs;
                        slab[(int64_t)(BX_NIN + c) * ncols + col] = diff ? gs - gt : gt;
                    }
            }
            __syncthreads();                                          // phase C is done with P: scratch for the bias reduction
            if (tid < 256) ps[tid] = gb_acc;                          // [m][oc]
            __syncthreads();
            if (tid < BX_PCH && (!NARROW || BX_PCH * pass + tid < nou)) {
                float sum = 0.f;
                for (int m = 0; m < 64; ++m) sum += ps[m * 4 + tid];
                slab[(int64_t)2 * BX_NIN * ncols + BX_PCH * pass + tid] = sum;
            }
        }
    }
    if (want_get) {
        __syncthreads();
        float* go = p.get_ws + (int64_t)blockIdx.x * p.get_len;       // [16][N][k]
        for (int f = tid; f < mk * BX_NET; f += BX_THREADS) {
            const int e = f / mk, r = f - e * mk;
            go[f] = get_s[r * BX_NET + e];
        }
    }
}

void fgnn_launch_slab_store(const float* ws, int nslab, int64_t slab_len, float* out, hipStream_t st);
void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);

// bytes of workspace this kernel wants on top of the gW / gbias slabs: the edge-type gradient slabs
int64_t fgnn_mpconv_backward_ext_extra_bytes(const fgnn_mpconv_desc* d) {
    if (d->ext == FGNN_EXT_NONE || d->dtype != FGNN_F32 || d->net != BX_NET) return 0;
    return (int64_t)256 * BX_NET * d->M * d->k * 4;
}

// 1 when this descriptor's backward sums the edge-type gradient over the batch itself (getype = [net, M, k]); the caller
// asks for that form with FGNN_DESC_GETYPE_REDUCED in d->reserved.
int fgnn_mpconv_backward_ext_accepts(const fgnn_mpconv_desc* d) {
    static const bool off = getenv("FGNN_NO_EXT") != nullptr;
    if (off) return 0;
    if (d->dtype != FGNN_F32 || (d->ext != FGNN_EXT_NEIGHBOR && d->ext != FGNN_EXT_DIFF)) return 0;
    if (d->agg != FGNN_AGG_MAX && d->agg != FGNN_AGG_LSE) return 0;
    if (d->net != BX_NET || d->nin != BX_NIN || d->nou < 2 || d->nou > BX_NOU || (d->nou & 1)) return 0;
    if (d->N != d->M || d->N < 1 || d->N > 64 || d->k < 1 || d->k > 16 || d->N * d->k > BX_MAX_MK) return 0;
    if ((d->idx_sb != 0 || d->et_sb != 0) && d->B != 1) return 0;
    if (!(d->x_sc == 1 && d->x_sn == BX_NIN && d->x_sb % 4 == 0)) return 0;
    if (!(d->y_sc == 1 && d->y_sm == d->nou)) return 0;
    if ((d->nou & 3) == 0 && d->y_sb % 4 != 0) return 0;
    return 1;
}

// Returns 1 if launched, 0 if the call is outside this kernel's family, < 0 on error.
int fgnn_mpconv_backward_ext(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                             const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                             float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                             fgnn_stream_t stream) {
    const bool reduced = (d->reserved & FGNN_DESC_GETYPE_REDUCED) != 0;
    if (!fgnn_mpconv_backward_ext_accepts(d)) {
        if (reduced) FGNN_FAIL(FGNN_EUNSUPPORTED, "batch-reduced edge-type gradient asked of a shape without that kernel");
        return 0;
    }
    if (getype && !reduced && d->B != 1) return 0;                   // per-sample edge-type gradient: the generic kernel
    if (((uintptr_t)x & 15) || ((d->nou & 3) == 0 && (((uintptr_t)gz & 15) || ((uintptr_t)argmax & 3)))) {
        if (reduced) FGNN_FAIL(FGNN_EINVAL, "mpconv ext backward needs 16-byte aligned x / gz");
        return 0;
    }
    const int mk = d->N * d->k;
    const int ncols = d->nou * BX_NET;
    const int64_t nw = (int64_t)2 * BX_NIN * ncols, slab_len = nw + d->nou, get_len = (int64_t)BX_NET * mk;
    int grid = 256;
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;
    const int64_t need = (grid * slab_len + (getype ? grid * get_len : 0)) * 4;
    if (!workspace || workspace_bytes < need)
        FGNN_FAIL(FGNN_EINVAL, "mpconv ext backward needs %lld bytes of workspace (fgnn_mpconv_backward_workspace_bytes)", (long long)need);
    BxParams p;
    p.x = (const float*)x; p.idx = nn_idx; p.et = (const float*)etype; p.W = filters; p.gz = (const float*)gz;
    p.am = argmax; p.gx = (float*)gx; p.ws = (float*)workspace;
    p.get_ws = getype ? (float*)workspace + grid * slab_len : nullptr;
    p.B = d->B; p.N = d->N; p.k = d->k; p.ext = d->ext; p.nou = d->nou; p.ncols = ncols; p.wvec = ((uintptr_t)filters & 15) == 0; p.npass = (d->nou + BX_PCH - 1) / BX_PCH;
    p.x_sb = d->x_sb; p.y_sb = d->y_sb; p.idx_sm = d->idx_sm; p.idx_sk = d->idx_sk;
    p.et_se = d->et_se; p.et_sm = d->et_sm; p.et_sk = d->et_sk;
    p.slab_len = slab_len; p.get_len = get_len;
    int off_b = 0;
    auto take = [&](int bytes) { const int o = off_b; off_b = fgnn_round_up(off_b + bytes, 16); return o; };
    p.off_xs = take(64 * BX_XS * 4);
    p.off_ps = take(64 * BX_PS * 4);
    p.off_dp = p.off_ps;
    p.off_et = take(mk * BX_NET * 4);
    p.off_get = take(mk * BX_NET * 4);
    p.off_idx = take(mk * 4);
    p.off_csr = take(((mk * 2 + 15) & ~15) + 65 * 4);
    p.off_e16 = take(64 * 16 * 2);
    p.off_gz = take(64 * 4 * 4);
    p.off_am = take(64 * 4);
    p.off_w = take(d->agg == FGNN_AGG_LSE ? mk * 4 * 4 : 16);
    if (off_b + 64 * BX_PS * 4 <= 160 * 1024) p.off_dp = take(64 * BX_PS * 4);      // room for dP's own image
    const int lds = off_b;
    if (lds > 160 * 1024) {
        if (reduced) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv ext backward: %d bytes of LDS", lds);
        return 0;
    }
    const bool narrow = d->nou != BX_NOU, wvec = p.wvec != 0;
    void* fn;
#define BX_PICK(A) (narrow ? (wvec ? (void*)mpconv_bwd_ext_kernel<A, true, true> : (void*)mpconv_bwd_ext_kernel<A, true, false>) \
                           : (wvec ? (void*)mpconv_bwd_ext_kernel<A, false, true> : (void*)mpconv_bwd_ext_kernel<A, false, false>))
    fn = d->agg == FGNN_AGG_MAX ? BX_PICK(FGNN_AGG_MAX) : BX_PICK(FGNN_AGG_LSE);
#undef BX_PICK
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    fgnn_note_kernel("mpconv_bwd_ext_kernel<%d, %s, %s>", d->agg, narrow ? "true" : "false", wvec ? "true" : "false");
    p.prof = nullptr;
    p.dbg = 0;
#ifdef FGNN_ENABLE_PROF
    if (getenv("FGNN_EXT_DBG")) p.dbg = atoi(getenv("FGNN_EXT_DBG"));
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 64 * 8);
        (void)hipMemset(prof_buf, 0, 64 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(BX_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext backward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {              // tuning aid: one stage of workgroup 0 (shader clocks): A start, A end, B1 start, B1 end, B2 end, C start, C end
        long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[fgnn prof ext bwd] wave %d:", w);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", h[w * 8 + i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, (hipStream_t)stream);
    if (getype) {
        fgnn_launch_slab_store(p.get_ws, grid, get_len, (float*)getype, (hipStream_t)stream);
    }
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
