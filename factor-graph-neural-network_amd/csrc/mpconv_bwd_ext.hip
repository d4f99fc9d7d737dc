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
    // split form (mpconv_bwd_extq_kernel): the filters as bf16 pieces, written by bq_prep_kernel into the workspace
    const uint16_t* wq1;     // [piece][S | T][column 0..1023][c 0..63]   (B operand of P = x W: 8 consecutive input channels per lane)
    const uint16_t* wq2;     // [piece][S | T][c 0..63][column 0..1023]   (B operand of gx = dP Wc: 8 consecutive columns per lane)
    int off_xq, off_dq;      // LDS: x pieces [piece][64][64] bf16, dP pieces [piece][64][128] bf16 (off_dq == off_ps: dP overwrites P)
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


// =====================================================================================================================
// Split form (round 6): the three GEMMs of a stage on the bf16 matrix cores with every f32 operand cut into NP bf16
// pieces (x = h + m + l: three 8-bit mantissa pieces, exact to 2^-24 — the arithmetic mpconv_fwd_extp_kernel uses for the
// forward; NP = 2 keeps h + l: 2^-17).  The exact-f32 kernel above is bound by the f32 matrix pipe (192 v_mfma_f32_16x16x4_f32
// = 6 144 cycles per wave and stage: 0.43 of the 157 TF f32 roof at B = 1024, and 42 % of the kernel time of BASELINE config 5's
// training step); the same products as bf16 pieces are 24 fragment products x 6 (NP = 3) or x 3 (NP = 2) v_mfma_f32_16x16x32_bf16
// of 16 cycles: 2 304 / 1 152 cycles.  What changes:
//   * the filters arrive as bf16 pieces in both operand layouts (bq_prep_kernel, once per call): a pass's B operands are 18
//     16-byte loads per lane instead of 96 scalar loads + their combination;
//   * x and dP live in LDS as bf16 PIECE images (x: 128-byte rows, dP: 256-byte rows; 32-byte segments XOR-swizzled by the row so
//     that the 16-byte operand reads AND the transpose reads are conflict-free); P stays f32 (the edge-type gradient reads it);
//   * gW's node-contracted operands (x^T, dP with K = nodes) come out of the row-major piece images through ds_read_b64_tr_b16.
// Max aggregation, 64 output channels; everything else (tables, phases B1 / dS / dT, slabs, fixed summation orders) is the
// exact kernel's.  Same bits run to run.
typedef __bf16 bq_bf16x8 __attribute__((ext_vector_type(8)));
typedef short bq_s16x4 __attribute__((ext_vector_type(4)));
#define BQ_XROW 128           // bytes per row of an x piece image (64 bf16)
#define BQ_DROW 256           // bytes per row of a dP piece image (128 bf16)
#define BQ_XPIECE (64 * BQ_XROW)
#define BQ_DPIECE (64 * BQ_DROW)
#define BQ_WPIECE (2 * 1024 * 64)   // elements per piece of wq1 / wq2

// segment (32 bytes) swizzles: a transpose read's 32-lane pass takes rows {r..r+3, r+8..r+11} of ONE logical segment, a 16-byte
// operand read's service group rows {0-3, 12-15} (first half of the segment) + {4-11} (second half)
__device__ __forceinline__ int bq_swx(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }      // 128-byte rows: bank bit 5 is the row's parity
__device__ __forceinline__ int bq_swd(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }             // 256-byte rows
__device__ __forceinline__ unsigned bq_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
template <int NP> __device__ __forceinline__ void bq_split2(float a, float b, unsigned (&o)[NP]) {
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        o[t] = bq_pack2(a, b);
        if (t + 1 < NP) { a -= __uint_as_float(o[t] << 16); b -= __uint_as_float(o[t] & 0xffff0000u); }
    }
}
__device__ __forceinline__ uint2 bq_tr(unsigned lds_addr) {
    typedef __attribute__((address_space(3))) bq_s16x4 lds_v4;
    const bq_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4*>(static_cast<uintptr_t>(lds_addr)));
    return __builtin_bit_cast(uint2, v);
}
// the piece products of a fragment pair, smallest first: (l h' + h l' + m m') + (m h' + h m') + h h'   |   NP = 2: l h' + h l' + h h'
template <int NP> struct BqTerms;
template <> struct BqTerms<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}; static constexpr int B[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct BqTerms<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {0, 1, 0}; };

// filters [128][1024] f32 -> the S / T combinations as bf16 pieces in both operand layouts
template <int NP>
__global__ __launch_bounds__(256) void bq_prep_kernel(const float* W, uint16_t* wq1, uint16_t* wq2, int diff) {
    const int col = blockIdx.x * 256 + threadIdx.x;       // 0..1023
    const int c = blockIdx.y;                             // 0..63
    const float wt = W[(int64_t)c * BX_NCOLS + col], wb = W[(int64_t)(BX_NIN + c) * BX_NCOLS + col];
    const float sv = diff ? wt + wb : wt, tv = diff ? -wb : wb;
    unsigned o[NP];
    bq_split2<NP>(sv, tv, o);
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const uint16_t hs = (uint16_t)(o[t] & 0xffffu), ht = (uint16_t)(o[t] >> 16);
        wq1[(int64_t)t * BQ_WPIECE + ((int64_t)0 * 1024 + col) * 64 + c] = hs;
        wq1[(int64_t)t * BQ_WPIECE + ((int64_t)1 * 1024 + col) * 64 + c] = ht;
        wq2[(int64_t)t * BQ_WPIECE + ((int64_t)0 * 64 + c) * 1024 + col] = hs;
        wq2[(int64_t)t * BQ_WPIECE + ((int64_t)1 * 64 + c) * 1024 + col] = ht;
    }
}

template <int NP, bool SEP>
__global__ __launch_bounds__(BX_THREADS, 1) void mpconv_bwd_extq_kernel(const BxParams p) {
    typedef BqTerms<NP> TM;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int N = p.N, k = p.k, mk = N * k;
    const int ntile = (N + 15) >> 4;
    const bool want_get = p.get_ws != nullptr;
    const bool diff = p.ext == FGNN_EXT_DIFF;
    constexpr int nou = BX_NOU, ncols = BX_NCOLS, npass = BX_NPASS;

    const unsigned lds0 = (unsigned)(uintptr_t)bx_lds;
    unsigned char* xq = bx_lds + p.off_xq;                            // [NP][64][128 B]
    unsigned char* dq = bx_lds + p.off_dq;                            // [NP][64][256 B]
    float* ps = reinterpret_cast<float*>(bx_lds + p.off_ps);          // [64][PS]: P (f32)
    unsigned short* e16 = reinterpret_cast<unsigned short*>(bx_lds + p.off_e16);
    float* et_s = reinterpret_cast<float*>(bx_lds + p.off_et);
    float* get_s = reinterpret_cast<float*>(bx_lds + p.off_get);
    int* idx_s = reinterpret_cast<int*>(bx_lds + p.off_idx);
    unsigned short* csr_ent = reinterpret_cast<unsigned short*>(bx_lds + p.off_csr);
    int* csr_off = reinterpret_cast<int*>(bx_lds + p.off_csr + ((mk * 2 + 15) & ~15));
    float* gz_s = reinterpret_cast<float*>(bx_lds + p.off_gz);
    unsigned* am_s = reinterpret_cast<unsigned*>(bx_lds + p.off_am);
    float* sel_s = reinterpret_cast<float*>(bx_lds + p.off_w);        // [mk + 1][4]: gz[m][oc] where edge (m, j) is the argmax of (m, oc), else 0; row mk = 0

    const int chunk = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int ns = min(p.B, b_begin + chunk) - b_begin;
    float* slab = p.ws + (int64_t)blockIdx.x * p.slab_len;
    if (ns <= 0) {
        for (int64_t f = tid; f < p.slab_len; f += BX_THREADS) slab[f] = 0.f;
        if (want_get) for (int64_t f = tid; f < p.get_len; f += BX_THREADS) p.get_ws[(int64_t)blockIdx.x * p.get_len + f] = 0.f;
        return;
    }

    // ---------------- one-time setup (as the exact kernel): graph tables, edge types, CSR by source node ----------------
    for (int f = tid; f < NP * BQ_XPIECE / 4; f += BX_THREADS) reinterpret_cast<unsigned*>(xq)[f] = 0u;
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
    if (tid < BX_NET) et_s[mk * BX_NET + tid] = 0.f;                  // the row the empty in-edge slots name (weight 0 x a finite value)
    if (tid < 4) sel_s[mk * 4 + tid] = 0.f;
    __syncthreads();
    if (tid < 64) {
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
    if (tid < N) {
        int o = csr_off[tid];
        for (int r = 0; r < mk; ++r) if (idx_s[r] == tid) csr_ent[o++] = (unsigned short)(((r / k) << 4) | (r % k));
    }
    __syncthreads();
    for (int f = tid; f < 64 * 16; f += BX_THREADS) {
        const int n = f >> 4, i = csr_off[n] + (f & 15);
        unsigned short v = (unsigned short)mk;                         // first 16 in-edges of node n as EDGE INDICES m k + j; empty slot: row mk
        if (n < N && i < csr_off[n + 1]) { const int ent = csr_ent[i]; v = (unsigned short)((ent >> 4) * k + (ent & 15)); }
        e16[f] = v;
    }

    // ---------------- per-thread roles (the exact kernel's) ----------------
    const int a_half = wave >> 2, a_sl = wave & 3;
    const int b1_eq = lane & 3, b1_m = 16 * wave + 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int b2_t = tid & 255, b2_row = b2_t >> 2, b2_oc = b2_t & 3;
    const int c_ct = wave & 3, c_np = wave >> 2;
    const int w_t = wave & 3, w_cp = wave >> 2;

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
                gzr = *reinterpret_cast<const uint4*>(p.gz + o);
                amr = *reinterpret_cast<const unsigned*>(p.am + o);
            }
        }
    };
    auto commit = [&]() {                                             // x chunk (row f >> 4, channels 4 (f & 15)..) -> the NP piece images
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + q * BX_THREADS;
            const int row = f >> 4, ch = f & 15;
            if (row < N) {
                unsigned a[NP], b[NP];
                bq_split2<NP>(__uint_as_float(xr[q].x), __uint_as_float(xr[q].y), a);
                bq_split2<NP>(__uint_as_float(xr[q].z), __uint_as_float(xr[q].w), b);
                unsigned char* dst = xq + row * BQ_XROW + (((ch >> 2) ^ bq_swx(row)) << 5) + ((ch & 3) << 3);
#pragma unroll
                for (int t = 0; t < NP; ++t) *reinterpret_cast<uint2*>(dst + t * BQ_XPIECE) = make_uint2(a[t], b[t]);
            }
        }
        if (tid < 64) {
            *reinterpret_cast<uint4*>(gz_s + tid * 4) = gzr;
            am_s[tid] = amr;
            if (tid < N) {                                            // the routed gradient of this node's out-edges, per channel of the pass
                const int j0 = min((int)(amr & 255u), k - 1), j1 = min((int)((amr >> 8) & 255u), k - 1);
                const int j2 = min((int)((amr >> 16) & 255u), k - 1), j3 = min((int)(amr >> 24), k - 1);
                for (int j = 0; j < k; ++j)
                    *reinterpret_cast<uint4*>(sel_s + (tid * k + j) * 4) = make_uint4(j == j0 ? gzr.x : 0u, j == j1 ? gzr.y : 0u, j == j2 ? gzr.z : 0u, j == j3 ? gzr.w : 0u);
            }
        }
    };
    // in-degree of this thread's dT row (b2_row), capped at the 16 slots; the wave's loop bound is the largest of its 16 rows
    int dt_deg = 0, dt_max = 0;
    {
        const int n = (tid & 255) >> 2;
        dt_deg = n < N ? min(csr_off[n + 1] - csr_off[n], 16) : 0;
        int mx = dt_deg;
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        dt_max = __builtin_amdgcn_readfirstlane(mx);
    }

    __syncthreads();
    prefetch(b_begin, blockIdx.x % npass);

    // ---- filter pieces of a pass: 6 NP 16-byte loads per lane; the next pass's are asked for behind the last stage of a pass, under
    //      the slab stores and the bias reduction ----
    bq_bf16x8 wA[NP][2];                                              // P = x W: B operand [k = c 32 ks + 8 lk ..][n = this wave's column li]
    bq_bf16x8 wG[NP][4];                                              // gx = dP Wc: B operand [k = dP column 32 ks + 8 lk ..][n = c 16 ct + li]
    auto load_w = [&](int pass) {
        const int col = 16 * (BX_PCH * pass + a_sl) + li;
        const uint16_t* base = p.wq1 + ((int64_t)a_half * 1024 + col) * 64 + 8 * lk;
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wA[t][ks] = __builtin_bit_cast(bq_bf16x8, *reinterpret_cast<const uint4*>(base + (int64_t)t * BQ_WPIECE + 32 * ks));
        const int c = 16 * c_ct + li;
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wG[t][ks] = __builtin_bit_cast(bq_bf16x8, *reinterpret_cast<const uint4*>(
                    p.wq2 + (int64_t)t * BQ_WPIECE + ((int64_t)(ks >> 1) * 64 + c) * 1024 + 64 * pass + 32 * (ks & 1) + 8 * lk));
    };

    for (int pp = 0; pp < npass; ++pp) {
        const int pass = (pp + blockIdx.x) % npass;
        const int pass_next = (pp + 1 + blockIdx.x) % npass;
        load_w(pass);             // (asked for behind the previous pass's last stage instead: measured, slower — two more registers spill)
        f32x4 accW[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h) accW[a][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float gb_acc = 0.f;

        for (int s = 0; s < ns; ++s) {
            const int b = b_begin + s;
            __syncthreads();                                          // previous stage is done with the x pieces, P / dP, gz_s
            commit();
            if (s + 1 < ns) prefetch(b + 1, pass);
            else if (pp + 1 < npass) prefetch(b_begin, pass_next);
            // gx of this sample as the earlier passes left it (read-modify-write through L2): asked for now, used at the end of phase C
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
            __syncthreads();
            BX_STAMP(0);
            // the lane's tile coordinates as values the compiler cannot prove loop-invariant: every operand address of the phases
            // below is then formed where it is used (hoisted out of the stage loop they hold ~60 registers for the whole pass
            // and the MFMA fragments spill)
            int oz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
            const int li_ = li + oz, lk_ = lk + oz;

            // ================= phase A: P = x [Ws | Wt] =================
            {
                const int li = li_, lk = lk_;
                const int colbase = 64 * a_half + 16 * a_sl + li;
                for (int nt = 0; nt < ntile; ++nt) {
                    const int row = 16 * nt + li;
                    const unsigned char* xrow = xq + row * BQ_XROW + ((lk & 1) << 4);
                    const int sw = bq_swx(row);
                    bq_bf16x8 xa[NP][2];
#pragma unroll
                    for (int t = 0; t < NP; ++t)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            xa[t][ks] = __builtin_bit_cast(bq_bf16x8, *reinterpret_cast<const uint4*>(xrow + t * BQ_XPIECE + (((2 * ks + (lk >> 1)) ^ sw) << 5)));
                    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int pr = 0; pr < TM::N; ++pr)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            acc[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[TM::A[pr]][ks], wA[TM::B[pr]][ks], acc[ks], 0, 0, 0);
                    const f32x4 out = acc[0] + acc[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ps[(16 * nt + 4 * lk + r) * BX_PS + colbase] = out[r];
                }
            }
            BX_STAMP(1);
            __syncthreads();
            BX_STAMP(2);

            // ================= phase B: get += gz (S + T) at the argmax; dS on waves 0-3, dT on waves 4-7 -> dP pieces =================
            auto phase_b1 = [&]() {                                   // (one thread owns all four channels of (m, eq): two channels whose
                if (want_get && b1_m < N) {                           // argmax is the same edge add into the same get_s words)
                    const unsigned am4 = am_s[b1_m];
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gz_s + b1_m * 4);
#pragma unroll
                    for (int oc = 0; oc < BX_PCH; ++oc) {
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(ps + b1_m * BX_PS + 16 * oc + 4 * b1_eq);
                        const int j = min((int)((am4 >> (8 * oc)) & 255u), k - 1);
                        const int r = b1_m * k + j;
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(ps + idx_s[r] * BX_PS + 64 + 16 * oc + 4 * b1_eq);
                        f32x4* gp = reinterpret_cast<f32x4*>(get_s + r * BX_NET + 4 * b1_eq);
                        *gp = *gp + g4[oc] * (sv + tv);
                    }
                }
            };
            auto store_row = [&](int h, const f32x4 (&row)[4]) {      // 16 columns of dP row b2_row (segment 4 h + oc) -> the piece images
                unsigned w[NP][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned a[NP], b[NP];
                    bq_split2<NP>(row[q][0], row[q][1], a);
                    bq_split2<NP>(row[q][2], row[q][3], b);
#pragma unroll
                    for (int t = 0; t < NP; ++t) { w[t][2 * q] = a[t]; w[t][2 * q + 1] = b[t]; }
                }
                unsigned char* dst = dq + b2_row * BQ_DROW + (((4 * h + b2_oc) ^ bq_swd(b2_row)) << 5);
#pragma unroll
                for (int t = 0; t < NP; ++t) {
                    *reinterpret_cast<uint4*>(dst + t * BQ_DPIECE) = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
                    *reinterpret_cast<uint4*>(dst + t * BQ_DPIECE + 16) = make_uint4(w[t][4], w[t][5], w[t][6], w[t][7]);
                }
            };
            auto phase_ds = [&]() {
                f32x4 row[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) row[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (b2_row < N) {
                    const float g = gz_s[b2_row * 4 + b2_oc];
                    const int j = min((int)((am_s[b2_row] >> (8 * b2_oc)) & 255u), k - 1);
                    const float* er = et_s + (b2_row * k + j) * BX_NET;
#pragma unroll
                    for (int q = 0; q < 4; ++q) row[q] = g * *reinterpret_cast<const f32x4*>(er + 4 * q);
                    gb_acc += g;
                }
                store_row(0, row);
            };
            f32x4 acc[4];                                             // dT of thread (n, eq): [oc][4 eq ..]
            auto dt_compute = [&]() {                                 // dT[n][oc][4 eq ..] = sum over in-edges r of sel[r][oc] et[r][4 eq ..]
                const int eq = b2_oc;
#pragma unroll
                for (int oc = 0; oc < 4; ++oc) acc[oc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int i0 = 0; i0 < dt_max; i0 += 4) {              // four slots at a time: eight 16-byte reads in flight, then their 64 FMAs
                    const uint2 e4 = *reinterpret_cast<const uint2*>(e16 + b2_row * 16 + i0);       // (a row with fewer in-edges names the zero row)
                    const int r0 = e4.x & 0xffff, r1 = e4.x >> 16, r2 = e4.y & 0xffff, r3 = e4.y >> 16;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sel_s + r0 * 4), v0 = *reinterpret_cast<const f32x4*>(et_s + r0 * BX_NET + 4 * eq);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sel_s + r1 * 4), v1 = *reinterpret_cast<const f32x4*>(et_s + r1 * BX_NET + 4 * eq);
                    const f32x4 s2 = *reinterpret_cast<const f32x4*>(sel_s + r2 * 4), v2 = *reinterpret_cast<const f32x4*>(et_s + r2 * BX_NET + 4 * eq);
                    const f32x4 s3 = *reinterpret_cast<const f32x4*>(sel_s + r3 * 4), v3 = *reinterpret_cast<const f32x4*>(et_s + r3 * BX_NET + 4 * eq);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) acc[oc] = s0[oc] * v0 + acc[oc];
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) acc[oc] = s1[oc] * v1 + acc[oc];
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) acc[oc] = s2[oc] * v2 + acc[oc];
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) acc[oc] = s3[oc] * v3 + acc[oc];
                }
                if (b2_row < N) {
                    const int e1 = csr_off[b2_row + 1];
                    for (int i = csr_off[b2_row] + 16; i < e1; ++i) {  // in-degree above 16: the tail of the list
                        const int ent = csr_ent[i];
                        const int r = (ent >> 4) * k + (ent & 15);
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(sel_s + r * 4);
                        const f32x4 ev = *reinterpret_cast<const f32x4*>(et_s + r * BX_NET + 4 * eq);
#pragma unroll
                        for (int oc = 0; oc < 4; ++oc) acc[oc] = sv[oc] * ev + acc[oc];
                    }
                }
                BX_STAMP(7);
            };
            auto dt_store = [&]() {
                const int eq = b2_oc;
                unsigned char* drow = dq + b2_row * BQ_DROW + (eq << 3);
                const int sd = bq_swd(b2_row);
#pragma unroll
                for (int oc = 0; oc < 4; ++oc) {
                    unsigned a[NP], b[NP];
                    bq_split2<NP>(acc[oc][0], acc[oc][1], a);
                    bq_split2<NP>(acc[oc][2], acc[oc][3], b);
#pragma unroll
                    for (int t = 0; t < NP; ++t) *reinterpret_cast<uint2*>(drow + t * BQ_DPIECE + (((4 + oc) ^ sd) << 5)) = make_uint2(a[t], b[t]);
                }
            };
            if (SEP) {
                if (tid < 256) { phase_b1(); phase_ds(); }
                else { dt_compute(); dt_store(); }
            } else {                                                  // the dP pieces overwrite P: the edge-type gradient reads P first
                if (tid < 256) phase_b1();                            // (dT formed in registers beside it: measured, no gain — 4 more registers spill)
                BX_STAMP(3);
                __syncthreads();
                if (tid < 256) phase_ds();
                else { dt_compute(); dt_store(); }
            }
            BX_STAMP(4);
            __syncthreads();
            BX_STAMP(5);

            // ================= phase C: gW += x^T dP (transpose reads), gx += dP Wc (global RMW) =================
            {
                const int li = li_, lk = lk_;
                // gW: tiles (c tile 2 cp + a) x (S / T column tile t); K = 64 nodes in two k-steps; both operands by transpose reads
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r0 = 32 * ks + 8 * lk + (li >> 2), r1 = r0 + 4;
                    const unsigned xo0 = lds0 + p.off_xq + r0 * BQ_XROW + ((li & 3) << 3), xo1 = lds0 + p.off_xq + r1 * BQ_XROW + ((li & 3) << 3);
                    const unsigned do0 = lds0 + p.off_dq + r0 * BQ_DROW + ((li & 3) << 3), do1 = lds0 + p.off_dq + r1 * BQ_DROW + ((li & 3) << 3);
                    const int sx0 = bq_swx(r0), sx1 = bq_swx(r1), sd0 = bq_swd(r0), sd1 = bq_swd(r1);
                    bq_bf16x8 xa[2][NP], db[2][NP];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int t = 0; t < NP; ++t) {
                            const int ct = 2 * w_cp + a;
                            const uint2 lo = bq_tr(xo0 + t * BQ_XPIECE + ((ct ^ sx0) << 5)), hi = bq_tr(xo1 + t * BQ_XPIECE + ((ct ^ sx1) << 5));
                            xa[a][t] = __builtin_bit_cast(bq_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < NP; ++t) {
                            const int g = 4 * h + w_t;
                            const uint2 lo = bq_tr(do0 + t * BQ_DPIECE + ((g ^ sd0) << 5)), hi = bq_tr(do1 + t * BQ_DPIECE + ((g ^ sd1) << 5));
                            db[h][t] = __builtin_bit_cast(bq_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                        }
#pragma unroll
                    for (int pr = 0; pr < TM::N; ++pr)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                accW[a][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a][TM::A[pr]], db[h][TM::B[pr]], accW[a][h], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                // (one k-step's fragments at a time: hoisted together they spill)
                }
                // gx tiles: nodes (2 np, 2 np + 1) x c tile ct, K = 128 dP columns in four k-steps
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = (2 * c_np + h) * 16 + li;
                    const unsigned char* drow = dq + row * BQ_DROW + ((lk & 1) << 4);
                    const int sd = bq_swd(row);
                    f32x4 accx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // two chains (even / odd k-steps)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) {
                        bq_bf16x8 da[2][NP];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int t = 0; t < NP; ++t)
                                da[q][t] = __builtin_bit_cast(bq_bf16x8, *reinterpret_cast<const uint4*>(drow + t * BQ_DPIECE + (((2 * (2 * kp + q) + (lk >> 1)) ^ sd) << 5)));
#pragma unroll
                        for (int pr = 0; pr < TM::N; ++pr)
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                accx[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[q][TM::A[pr]], wG[TM::B[pr]][2 * kp + q], accx[q], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int node = (2 * c_np + h) * 16 + 4 * lk + r;
                        if (node < N) gxb[(int64_t)node * BX_NIN] = old[h][r] + (accx[0][r] + accx[1][r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                BX_STAMP(6);
            }
        }

        // ---- end of pass: filter-gradient tiles and the bias gradient of the pass -> this workgroup's slab ----
        {
            const int col = 64 * pass + 16 * w_t + li;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (2 * w_cp + a) + 4 * lk + r;
                    const float gs = accW[a][0][r], gt = accW[a][1][r];
                    slab[(int64_t)c * ncols + col] = gs;
                    slab[(int64_t)(BX_NIN + c) * ncols + col] = diff ? gs - gt : gt;
                }
            // bias gradient of the pass: the (m, oc) threads' sums folded over m in a fixed tree (16 rows by lane exchanges, 4 waves in LDS)
            float gsum = gb_acc;
#pragma unroll
            for (int o = 4; o <= 32; o <<= 1) gsum += __shfl_xor(gsum, o);
            __syncthreads();                                          // phase C is done with the images: P's storage is scratch
            if (tid < 256 && lane < 4) ps[wave * 4 + lane] = gsum;
            __syncthreads();
            if (tid < BX_PCH) slab[(int64_t)2 * BX_NIN * ncols + BX_PCH * pass + tid] = (ps[tid] + ps[4 + tid]) + (ps[8 + tid] + ps[12 + tid]);
        }
    }
    if (want_get) {
        __syncthreads();
        float* go = p.get_ws + (int64_t)blockIdx.x * p.get_len;
        for (int f = tid; f < mk * BX_NET; f += BX_THREADS) {
            const int e = f / mk, r = f - e * mk;
            go[f] = get_s[r * BX_NET + e];
        }
    }
}

void fgnn_launch_slab_store(const float* ws, int nslab, int64_t slab_len, float* out, hipStream_t st);
void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);

// which arithmetic the backward of this family runs in (process-wide; FGNN_EXT_BWD_PIECES gives the initial value)
static int g_bq_pieces = -1;
static int bq_pieces() {
    if (g_bq_pieces < 0) g_bq_pieces = getenv("FGNN_EXT_BWD_PIECES") ? atoi(getenv("FGNN_EXT_BWD_PIECES")) : 2;
    return g_bq_pieces;
}
extern "C" int fgnn_set_ext_backward_pieces(int pieces) {
    const int before = bq_pieces();
    if (pieces == 0 || pieces == 2 || pieces == 3) g_bq_pieces = pieces;
    return before;
}

// bytes of workspace this kernel wants on top of the gW / gbias slabs: the edge-type gradient slabs
int64_t fgnn_mpconv_backward_ext_extra_bytes(const fgnn_mpconv_desc* d) {
    if (d->ext == FGNN_EXT_NONE || d->dtype != FGNN_F32 || d->net != BX_NET) return 0;
    return (int64_t)256 * BX_NET * d->M * d->k * 4 + (int64_t)2 * 3 * BQ_WPIECE * 2;      // edge-type slabs + the filters' bf16 pieces (split form)
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
    const int64_t pieces_bytes = (int64_t)2 * 3 * BQ_WPIECE * 2;
    const int64_t need = (grid * slab_len + (getype ? grid * get_len : 0)) * 4 + pieces_bytes;
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
    // ---- split form (bf16 pieces on the bf16 matrix cores): max aggregation, 64 output channels, when its images fit ----
    // FGNN_EXT_BWD_PIECES: 2 (default: h + l pieces, three bf16 MFMAs per fragment product; gradients within 5e-6 of the exact kernel's,
    // tools/xbench.py), 3 (h + m + l, six MFMAs: within 4e-7, 1.35 x the time), 0 (the exact-f32 kernel above: 2 x the time)
    const int want_np = bq_pieces();
    if (want_np >= 2 && d->agg == FGNN_AGG_MAX && !narrow && (d->y_sb % 4) == 0) {
        int np = 0, sep = 0, lds_q = 0;
        for (int cand = want_np >= 3 ? 3 : 2; cand >= 2 && !np; --cand)
            for (int sp = 1; sp >= 0 && !np; --sp) {
                int ob = 0;
                auto tk = [&](int bytes) { const int o = ob; ob = fgnn_round_up(ob + bytes, 16); return o; };
                const int oxq = tk(cand * BQ_XPIECE);
                const int ops_ = tk(sp ? 64 * BX_PS * 4 : (cand * BQ_DPIECE > 64 * BX_PS * 4 ? cand * BQ_DPIECE : 64 * BX_PS * 4));
                const int odq = sp ? tk(cand * BQ_DPIECE) : ops_;
                const int oet = tk((mk + 1) * BX_NET * 4), oget = tk(mk * BX_NET * 4), oidx = tk(mk * 4), osel = tk((mk + 1) * 16);
                const int ocsr = tk(((mk * 2 + 15) & ~15) + 65 * 4), oe16 = tk(64 * 16 * 2), ogz = tk(64 * 4 * 4), oam = tk(64 * 4);
                if (ob <= 160 * 1024) {
                    np = cand; sep = sp; lds_q = ob;
                    p.off_xq = oxq; p.off_ps = ops_; p.off_dq = odq; p.off_dp = odq; p.off_et = oet; p.off_get = oget; p.off_idx = oidx;
                    p.off_csr = ocsr; p.off_e16 = oe16; p.off_gz = ogz; p.off_am = oam; p.off_w = osel; p.off_xs = 0;
                }
            }
        if (np) {
            uint16_t* wq = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + (grid * slab_len + (getype ? grid * get_len : 0)) * 4);
            p.wq1 = wq;
            p.wq2 = wq + (int64_t)3 * BQ_WPIECE;
            void* prep = np == 3 ? (void*)bq_prep_kernel<3> : (void*)bq_prep_kernel<2>;
            const float* Wf = filters;
            uint16_t* a1 = wq;
            uint16_t* a2 = wq + (int64_t)3 * BQ_WPIECE;
            int is_diff = d->ext == FGNN_EXT_DIFF;
            void* pargs[] = {(void*)&Wf, (void*)&a1, (void*)&a2, (void*)&is_diff};
            hipError_t e = hipLaunchKernel(prep, dim3(BX_NCOLS / 256, BX_NIN), dim3(256), pargs, 0, (hipStream_t)stream);
            if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext backward (filter pieces) launch: %s", hipGetErrorString(e));
            fn = np == 3 ? (sep ? (void*)mpconv_bwd_extq_kernel<3, true> : (void*)mpconv_bwd_extq_kernel<3, false>)
                         : (sep ? (void*)mpconv_bwd_extq_kernel<2, true> : (void*)mpconv_bwd_extq_kernel<2, false>);
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
            if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds_q, hipGetErrorString(e));
            fgnn_note_kernel("mpconv_bwd_extq_kernel<%d, %s>", np, sep ? "true" : "false");
            p.prof = nullptr;
            p.dbg = 0;
#ifdef FGNN_ENABLE_PROF
            static long long* prof_q = nullptr;
            if (getenv("FGNN_PROF")) {
                if (!prof_q) (void)hipMalloc(&prof_q, 64 * 8);
                (void)hipMemset(prof_q, 0, 64 * 8);
                p.prof = prof_q;
            }
#endif
            void* qargs[] = {(void*)&p};
            e = hipLaunchKernel(fn, dim3(grid), dim3(BX_THREADS), qargs, lds_q, (hipStream_t)stream);
            if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext backward (split form) launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
            if (p.prof) {
                long long h[64];
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
                for (int w = 0; w < 8; ++w) {
                    fprintf(stderr, "[fgnn prof extq bwd] wave %d:", w);
                    for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", h[w * 8 + i] - h[0]);
                    fprintf(stderr, "\n");
                }
            }
#endif
            fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, (hipStream_t)stream);
            if (getype) fgnn_launch_slab_store(p.get_ws, grid, get_len, (float*)getype, (hipStream_t)stream);
            e = hipGetLastError();
            if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ext backward helper launch: %s", hipGetErrorString(e));
            return 1;
        }
    }
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
