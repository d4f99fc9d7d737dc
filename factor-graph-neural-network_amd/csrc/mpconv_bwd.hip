// mpconv_bwd.hip — backward of the fused VF/FV message operator, gfx950 (MI355X).
//
// The reference trains mp_conv_v2 (/root/reference/lib/model/mpnn/mp_nn.py:115-175) through
// autograd over ~11 ATen ops; this is the hand-written counterpart of that whole chain for
//     z[b,o,m] = agg_j sum_e etype[b,e,m,j] * ( P[b, idx[b,m,j], o,e] (+ Q[b,m,o,e]) ) + bias[o]
// with P = x W_nb, Q = x W_self recomputed on the fly (never stored by the forward).
//
// Work decomposition: one workgroup owns a CHUNK of consecutive samples.
//   for each tile of OT output channels:            (W tile staged once per chunk)
//     for each sample of the chunk:
//       stage x, nn_idx, etype, gz tile            -> LDS
//       P tile = MFMA(W^T, x)                       -> LDS               (forward recompute)
//       routing weights w[m,o,j] = gz[m,o] * dagg/dE_j  (argmax one-hot | softmax weights | 1/k), then — NO atomics, every
//       output element has one owner thread and a fixed summation order (round 3; was LDS / global float atomicAdd):
//            detype[:, m, j]  = sum_o w[m,o,j] * (P[idx[m,j], o, :] + Q[m, o, :])        owner: the edge (m, j)
//            dP[n, o, :]      = sum over the in-edges (m, j) of n, in (m, j) order, of w[m,o,j] * etype[:, m, j]
//                               owner: (n, o); the in-edge lists = a CSR transpose of nn_idx[b] built in LDS per sample
//            dQ[m, o, :]      = sum_j w[m,o,j] * etype[:, m, j]                            owner: (m, o)
//       dx^T  += MFMA(W_tile, dP^T)     -> read-modify-write of this sample's gx (owner = this WG)
//       dW    += MFMA(x^T, dP)          -> f32 accumulators in registers across the chunk
//     dW tile / dbias of the chunk -> this workgroup's SLAB (filters' own [R][nou*net] layout); fgnn_launch_slab_reduce sums
//     the slabs in a fixed order into gfilters / gbias
// so no gradient tensor the size of the edge set is ever written to HBM, and two runs give identical bits.
//
// SCOPE (round 2): this file is the fallback for shapes no specialised kernel takes — odd channel counts, per-sample graphs
// with an extension, per-sample edge-type gradients, the `mean` aggregator.  Every call of the reference's models goes
// elsewhere: the LDPC family to mpconv_bwd_sg / _b16 / _res / _hyper.hip, the synthetic-PGM family (16 edge types,
// DIFF / NEIGHBOR, max and softmax, 2..64 output channels) to mpconv_bwd_ext.hip.  Like those, this kernel is free of
// atomics (tests/test_mpconv_gpu.py::test_generic_backward_is_bitwise_reproducible).
#include "fgnn_common.h"
#include <stdlib.h>

#define BWD_MAXT 8   // 16x16 dW tiles a wave can own within one channel tile

struct BwdParams {
    fgnn_mpconv_desc d;
    const void* x;
    const int64_t* idx;
    const void* et;
    const float* W;
    const void* gz;
    const void* z;
    const uint8_t* argmax;
    void* gx;            // dtype T, x's element strides
    void* get;           // dtype T, [net][M][k] contiguous per sample, or NULL
    float* slab;         // per-workgroup slabs [grid][R * nou * net + nou] (filters' layout + dbias), folded by fgnn_launch_slab_reduce
    int has_gbias;
    const float* bias;   // unused (z - bias is recovered from gz-side data only for LSE via zagg)
    int chunk;           // samples per workgroup
    int OT, CT, nproj, Npad, Kpad, XS, WS, PS;
    int off_xs, off_ws, off_ps, off_dps, off_idx, off_et, off_det, off_gz, off_aux, off_gb, off_cs, off_cl;
};

extern __shared__ __attribute__((aligned(16))) float fgnn_lds_b[];

template <int NET>
__device__ __forceinline__ float bwd_edge_dot(const float* etp, const float* pn, const float* pself,
                                              int net) {
    float v = 0.f;
    if constexpr (NET > 0) {
#pragma unroll
        for (int e = 0; e < NET; ++e) v = fmaf(etp[e], pn[e] + (pself ? pself[e] : 0.f), v);
    } else {
        for (int e = 0; e < net; ++e) v = fmaf(etp[e], pn[e] + (pself ? pself[e] : 0.f), v);
    }
    return v;
}

template <typename T, int NET, int AGG>
__global__ __launch_bounds__(FGNN_THREADS) void mpconv_bwd_kernel(const BwdParams p) {
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nin = d.nin, nou = d.nou, net = d.net, N = d.N, M = d.M, k = d.k;
    const int ncols = nou * net;
    const int CT = p.CT, CTT = p.CT * p.nproj;
    const bool self = p.nproj == 2;
    const int mk = M * k;

    float* xs = fgnn_lds_b + p.off_xs;
    float* ws = fgnn_lds_b + p.off_ws;
    float* ps = fgnn_lds_b + p.off_ps;
    float* dps = fgnn_lds_b + p.off_dps;
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_b + p.off_idx);
    float* et_s = fgnn_lds_b + p.off_et;
    float* det_s = fgnn_lds_b + p.off_det;
    float* gz_s = fgnn_lds_b + p.off_gz;     // [OT][M]
    float* aux_s = fgnn_lds_b + p.off_aux;   // [OT][M]: argmax (as int)
    float* gb_s = fgnn_lds_b + p.off_gb;     // [OT]
    int* cs_start = reinterpret_cast<int*>(fgnn_lds_b + p.off_cs);     // [N + 1] CSR transpose of nn_idx[b]: in-edges of node n are
    int* cs_list = reinterpret_cast<int*>(fgnn_lds_b + p.off_cl);      // [M k]   cs_list[cs_start[n] .. cs_start[n + 1]) = r = m k + j, ascending
    const int64_t R_rows = d.ext == FGNN_EXT_NONE ? nin : 2 * nin;
    float* slab = p.slab + (int64_t)blockIdx.x * (R_rows * ncols + nou);

    const T* xg = static_cast<const T*>(p.x);
    const T* etg = static_cast<const T*>(p.et);
    const T* gzg = static_cast<const T*>(p.gz);

    const int b_begin = blockIdx.x * p.chunk;
    const int b_end = min(d.B, b_begin + p.chunk);
    if (b_begin >= b_end) return;

    const int nct = p.Kpad / 16;              // 16-row tiles over input channels
    const int ncolt = CTT / 16;               // 16-col tiles over the W tile's columns
    const int ntile = p.Npad / 16;
    const int tpt = nct * ncolt;              // dW tiles per channel tile (<= 4*BWD_MAXT)

    for (int o0 = 0; o0 < nou; o0 += p.OT) {
        const int otc = min(p.OT, nou - o0);
        const int vcols = otc * net;
        // ---- stage the W tile (effective weights), once per chunk ----
        __syncthreads();
        for (int f = tid; f < p.Kpad * CTT; f += FGNN_THREADS) {
            const int c = f / CTT, tc = f - c * CTT;
            const int proj = tc / CT, q = tc - proj * CT;
            float v = 0.f;
            if (c < nin && q < vcols) {
                const int g = o0 * net + q;
                if (d.ext == FGNN_EXT_NONE) {
                    v = p.W[(int64_t)c * ncols + g];
                } else {
                    const float top = p.W[(int64_t)c * ncols + g];
                    const float bot = p.W[(int64_t)(nin + c) * ncols + g];
                    if (d.ext == FGNN_EXT_NEIGHBOR) v = proj == 0 ? bot : top;
                    else v = proj == 0 ? -bot : top + bot;
                }
            }
            ws[c * p.WS + tc] = v;
        }
        if (tid < p.OT) gb_s[tid] = 0.f;

        f32x4 gw[BWD_MAXT];
#pragma unroll
        for (int t = 0; t < BWD_MAXT; ++t) gw[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int b = b_begin; b < b_end; ++b) {
            __syncthreads();    // previous sample's MFMAs / flushes are done with xs, dps, det_s
            // ---- stage x[b], nn_idx[b], etype[b], gz / argmax / z tile; zero dP and detype ----
            {
                const T* xb = xg + (int64_t)b * d.x_sb;
                const int tot = p.Npad * p.Kpad;
                if (d.x_sn == 1 || d.x_sc != 1) {
                    for (int f = tid; f < tot; f += FGNN_THREADS) {
                        const int c = f / p.Npad, n = f - c * p.Npad;
                        float v = 0.f;
                        if (c < nin && n < N) v = fgnn_ld(xb + (int64_t)c * d.x_sc + (int64_t)n * d.x_sn);
                        xs[n * p.XS + c] = v;
                    }
                } else {
                    for (int f = tid; f < tot; f += FGNN_THREADS) {
                        const int n = f / p.Kpad, c = f - n * p.Kpad;
                        float v = 0.f;
                        if (c < nin && n < N) v = fgnn_ld(xb + (int64_t)c * d.x_sc + (int64_t)n * d.x_sn);
                        xs[n * p.XS + c] = v;
                    }
                }
                const int64_t* ib = p.idx + (int64_t)b * d.idx_sb;
                for (int f = tid; f < mk; f += FGNN_THREADS) {
                    const int m = f / k, j = f - m * k;
                    long long v = ib[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
                    v = v < 0 ? 0 : (v >= N ? N - 1 : v);
                    idx_s[f] = (int)v;
                }
                const T* eb = etg + (int64_t)b * d.et_sb;
                for (int f = tid; f < mk * net; f += FGNN_THREADS) {
                    int e, r;
                    if (d.et_se == 1) { r = f / net; e = f - r * net; }     // edge-type fastest in memory
                    else { e = f / mk; r = f - e * mk; }
                    const int m = r / k, j = r - m * k;
                    et_s[r * net + e] =
                        fgnn_ld(eb + (int64_t)e * d.et_se + (int64_t)m * d.et_sm + (int64_t)j * d.et_sk);
                    det_s[r * net + e] = 0.f;
                }
                const T* gb = gzg + (int64_t)b * d.y_sb;
                for (int f = tid; f < otc * M; f += FGNN_THREADS) {
                    int ol, m;
                    if (d.y_sc == 1 && d.y_sm != 1) { m = f / otc; ol = f - m * otc; }
                    else { ol = f / M; m = f - ol * M; }
                    gz_s[ol * M + m] = fgnn_ld(gb + (int64_t)(o0 + ol) * d.y_sc + (int64_t)m * d.y_sm);
                    if (AGG == FGNN_AGG_MAX)
                        reinterpret_cast<int*>(aux_s)[ol * M + m] =
                            p.argmax[(int64_t)b * d.y_sb + (int64_t)(o0 + ol) * d.y_sc + (int64_t)m * d.y_sm];
                }
                for (int f = tid; f < p.Npad * p.PS; f += FGNN_THREADS) dps[f] = 0.f;
            }
            __syncthreads();
            // ---- CSR transpose of this sample's neighbour table: who reads node n?  Thread n counts, one thread scans, thread n
            //      fills its list in ascending r = m k + j: a fixed order, no atomics ----
            for (int n = tid; n < N; n += FGNN_THREADS) {
                int c = 0;
                for (int r = 0; r < mk; ++r) c += idx_s[r] == n ? 1 : 0;
                cs_start[n + 1] = c;
            }
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int n = 0; n < N; ++n) { const int c = cs_start[n + 1]; cs_start[n] = run; run += c; }
                cs_start[N] = run;
            }
            __syncthreads();
            for (int n = tid; n < N; n += FGNN_THREADS) {
                int pos = cs_start[n];
                for (int r = 0; r < mk; ++r)
                    if (idx_s[r] == n) cs_list[pos++] = r;
            }
            // (the barrier behind the projection below orders these lists before their readers)

            // ---- forward recompute: P^T tile = W^T . x ----
            {
                const int nslab = CTT / 16, ntp = (ntile + 1) / 2;
                const int ksteps = p.Kpad / 4;
                for (int u = wave; u < nslab * ntp; u += FGNN_WAVES) {
                    const int slab = u % nslab, tp = u / nslab;
                    const int t0 = tp * 2;
                    const bool two = (t0 + 1) < ntile;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* ap = ws + lk * p.WS + slab * 16 + li;
                    const float* bp0 = xs + (t0 * 16 + li) * p.XS + lk;
                    const float* bp1 = bp0 + 16 * p.XS;
                    for (int kk = 0; kk < ksteps; ++kk) {
                        const float a = ap[kk * 4 * p.WS];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp0[kk * 4], acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp1[kk * 4], acc1, 0, 0, 0);
                    }
                    float* dst = ps + (t0 * 16 + li) * p.PS + slab * 16 + 4 * lk;
                    *reinterpret_cast<f32x4*>(dst) = acc0;
                    if (two) *reinterpret_cast<f32x4*>(dst + 16 * p.PS) = acc1;
                }
            }
            __syncthreads();

            // ---- routing weight of edge r = (m, j) for output channel ol: dz/dE_j ----
            auto edge_w = [&](int m, int j, int ol, int r, int n) -> float {
                const float g = gz_s[ol * M + m];
                if (AGG == FGNN_AGG_MAX) return reinterpret_cast<const int*>(aux_s)[ol * M + m] == j ? g : 0.f;
                if (AGG == FGNN_AGG_MEAN) return g / (float)k;
                const float* pself = self ? ps + m * p.PS + CT + ol * net : nullptr;
                return g * expf(3.0f * bwd_edge_dot<NET>(et_s + r * net, ps + n * p.PS + ol * net, pself, net) - aux_s[ol * M + m]);
            };
            // ---- (m, o) owners: softmax normaliser (3 * agg, by the forward's online log-sum-exp: the saved z may have been
            //      clobbered by an in-place ReLU) and the self-projection gradient dQ[m, o, :] ----
            if (AGG == FGNN_AGG_LSE) {
                for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                    const int m = it / otc, ol = it - m * otc;
                    const float* pself = self ? ps + m * p.PS + CT + ol * net : nullptr;
                    const int* ip = idx_s + m * k;
                    float mx = -INFINITY, sm = 0.f;
                    for (int j = 0; j < k; ++j) {
                        const float v = 3.0f * bwd_edge_dot<NET>(et_s + (m * k + j) * net, ps + ip[j] * p.PS + ol * net, pself, net);
                        if (v > mx) { sm = sm * expf(mx - v) + 1.0f; mx = v; }
                        else sm += expf(v - mx);
                    }
                    aux_s[ol * M + m] = mx + logf(sm);
                }
                __syncthreads();
            }
            if (tid < otc) {                                  // dbias of this tile: one owner per channel, m ascending
                float sgz = gb_s[tid];
                for (int m = 0; m < M; ++m) sgz += gz_s[tid * M + m];
                gb_s[tid] = sgz;
            }
            if (self) {
                for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                    const int m = it / otc, ol = it - m * otc;
                    float* dself = dps + m * p.PS + CT + ol * net;
                    for (int e = 0; e < net; ++e) dself[e] = 0.f;
                    int j_lo = 0, j_hi = k;
                    if (AGG == FGNN_AGG_MAX) { j_lo = reinterpret_cast<const int*>(aux_s)[ol * M + m]; j_hi = j_lo + 1; }
                    for (int j = j_lo; j < j_hi; ++j) {
                        const int r = m * k + j;
                        const float w = edge_w(m, j, ol, r, idx_s[r]);
                        for (int e = 0; e < net; ++e) dself[e] = fmaf(w, et_s[r * net + e], dself[e]);
                    }
                }
            }
            // ---- edge owners: detype[r, :] = sum over ol of w * (P[n, ol, :] + Q[m, ol, :]) ----
            for (int r = tid; r < mk; r += FGNN_THREADS) {
                const int m = r / k, j = r - m * k, n = idx_s[r];
                for (int e0 = 0; e0 < net; e0 += 4) {         // four edge types at a time (weights recomputed per group)
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int ol = 0; ol < otc; ++ol) {
                        const float w = edge_w(m, j, ol, r, n);
                        if (AGG == FGNN_AGG_MAX && w == 0.f) continue;
                        const float* pn = ps + n * p.PS + ol * net + e0;
                        const float* pq = ps + m * p.PS + CT + ol * net + e0;
                        a0 = fmaf(w, pn[0] + (self ? pq[0] : 0.f), a0);
                        if (e0 + 1 < net) a1 = fmaf(w, pn[1] + (self ? pq[1] : 0.f), a1);
                        if (e0 + 2 < net) a2 = fmaf(w, pn[2] + (self ? pq[2] : 0.f), a2);
                        if (e0 + 3 < net) a3 = fmaf(w, pn[3] + (self ? pq[3] : 0.f), a3);
                    }
                    det_s[r * net + e0] = a0;
                    if (e0 + 1 < net) det_s[r * net + e0 + 1] = a1;
                    if (e0 + 2 < net) det_s[r * net + e0 + 2] = a2;
                    if (e0 + 3 < net) det_s[r * net + e0 + 3] = a3;
                }
            }
            // ---- (n, ol) owners: dP[n, ol, :] = sum over the in-edges of n (ascending r) of w * etype[r, :] ----
            for (int it = tid; it < N * otc; it += FGNN_THREADS) {
                const int n = it / otc, ol = it - n * otc;
                float* dpn = dps + n * p.PS + ol * net;
                const int q0 = cs_start[n], q1 = cs_start[n + 1];
                for (int e0 = 0; e0 < net; e0 += 4) {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int q = q0; q < q1; ++q) {
                        const int r = cs_list[q];
                        const int m = r / k, j = r - m * k;
                        const float w = edge_w(m, j, ol, r, n);
                        if (AGG == FGNN_AGG_MAX && w == 0.f) continue;
                        const float* etp = et_s + r * net + e0;
                        a0 = fmaf(w, etp[0], a0);
                        if (e0 + 1 < net) a1 = fmaf(w, etp[1], a1);
                        if (e0 + 2 < net) a2 = fmaf(w, etp[2], a2);
                        if (e0 + 3 < net) a3 = fmaf(w, etp[3], a3);
                    }
                    dpn[e0] = a0;
                    if (e0 + 1 < net) dpn[e0 + 1] = a1;
                    if (e0 + 2 < net) dpn[e0 + 2] = a2;
                    if (e0 + 3 < net) dpn[e0 + 3] = a3;
                }
            }
            __syncthreads();

            // ---- detype: owners add this channel tile's contribution (first tile writes) ----
            if (p.get) {
                T* gb = reinterpret_cast<T*>(p.get) + (int64_t)b * net * mk;
                for (int f = tid; f < mk * net; f += FGNN_THREADS) {
                    const int e = f / mk, r = f - e * mk;
                    const float v = det_s[r * net + e];
                    fgnn_st(gb + f, (o0 == 0) ? v : fgnn_ld(gb + f) + v);
                }
            }

            // ---- dx^T[c][n] (+)= sum_col W[c][col] dP[n][col]; this WG owns sample b's gx ----
            {
                T* gxb = reinterpret_cast<T*>(p.gx) + (int64_t)b * d.x_sb;
                const int ksteps = CTT / 4;
                for (int u = wave; u < nct * ntile; u += FGNN_WAVES) {
                    const int ctile = u % nct, nt = u / nct;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const float* ap = ws + (ctile * 16 + li) * p.WS + lk;
                    const float* bp = dps + (nt * 16 + li) * p.PS + lk;
                    for (int kk = 0; kk < ksteps; ++kk)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4], bp[kk * 4], acc, 0, 0, 0);
                    const int n = nt * 16 + li;
                    if (n < N) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = ctile * 16 + 4 * lk + r;
                            if (c < nin) {
                                T* q = gxb + (int64_t)c * d.x_sc + (int64_t)n * d.x_sn;
                                fgnn_st(q, (o0 == 0) ? acc[r] : fgnn_ld(q) + acc[r]);
                            }
                        }
                    }
                }
            }

            // ---- dW tile[c][col] += sum_n x[c][n] dP[n][col]; registers, across the chunk ----
            {
                const int ksteps = p.Npad / 4;
#pragma unroll
                for (int t = 0; t < BWD_MAXT; ++t) {
                    const int u = wave + FGNN_WAVES * t;
                    if (u < tpt) {
                        const int ctile = u % nct, colt = u / nct;
                        const float* ap = xs + lk * p.XS + ctile * 16 + li;
                        const float* bp = dps + lk * p.PS + colt * 16 + li;
                        f32x4 acc = gw[t];
                        for (int kk = 0; kk < ksteps; ++kk)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * p.XS], bp[kk * 4 * p.PS],
                                                                      acc, 0, 0, 0);
                        gw[t] = acc;
                    }
                }
            }
        }   // samples

        // ---- dW tile and dbias of this chunk -> this workgroup's slab, in the filters' own layout.  ORIG_WITH_DIFF sends the
        //      neighbour projection's gradient to -W_bot and the self projection's to W_top AND W_bot: the self tiles store first,
        //      the neighbour tiles subtract after a barrier (every element has one writer per step) ----
        for (int step = 0; step < (d.ext == FGNN_EXT_DIFF ? 2 : 1); ++step) {
#pragma unroll
            for (int t = 0; t < BWD_MAXT; ++t) {
                const int u = wave + FGNN_WAVES * t;
                if (u < tpt) {
                    const int ctile = u % nct, colt = u / nct;
                    const int tc = colt * 16 + li;            // column inside the staged tile
                    const int proj = tc / CT, q = tc - proj * CT;
                    if (q < vcols) {
                        const int g = o0 * net + q;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = ctile * 16 + 4 * lk + r;
                            if (c >= nin) continue;
                            const float v = gw[t][r];
                            if (d.ext == FGNN_EXT_NONE) {
                                slab[(int64_t)c * ncols + g] = v;
                            } else if (d.ext == FGNN_EXT_NEIGHBOR) {          // W_nb = W_bot, W_self = W_top
                                slab[(int64_t)(proj == 0 ? nin + c : c) * ncols + g] = v;
                            } else if (step == 0) {                           // W_nb = -W_bot, W_self = W_top + W_bot
                                if (proj == 1) { slab[(int64_t)c * ncols + g] = v; slab[(int64_t)(nin + c) * ncols + g] = v; }
                            } else if (proj == 0) {
                                slab[(int64_t)(nin + c) * ncols + g] -= v;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (tid < otc) slab[R_rows * ncols + o0 + tid] = gb_s[tid];
    }   // channel tiles
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
int fgnn_check_desc(const fgnn_mpconv_desc* d);
void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias, hipStream_t st);
int fgnn_mpconv_backward_resident(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                  const void* etype, const float* filters, const void* gz,
                                  const uint8_t* argmax, void* gx, void* getype, float* gfilters,
                                  float* gbias, void* workspace, int64_t workspace_bytes,
                                  fgnn_stream_t stream);

int fgnn_mpconv_backward_hyper(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                               const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                               float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                               fgnn_stream_t stream);

int fgnn_mpconv_backward_b16(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                             const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                             float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                             fgnn_stream_t stream);
int fgnn_mpconv_backward_sg(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                            float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                            fgnn_stream_t stream);

static int plan_backward(const fgnn_mpconv_desc* d, BwdParams* p) {
    const int nproj = d->ext == FGNN_EXT_NONE ? 1 : 2;
    p->nproj = nproj;
    p->Npad = fgnn_round_up(d->N, 16);
    p->Kpad = fgnn_round_up(d->nin, 16);
    p->XS = (p->Kpad + 29) / 32 * 32 + 2;
    const int mk = d->M * d->k;
    const int nct = p->Kpad / 16;
    int best = 0;
    for (int pass = 0; pass < 2 && !best; ++pass) {
        const int64_t budget = pass == 0 ? 78 * 1024 : 160 * 1024;
        for (int ot = d->nou < 64 ? d->nou : 64; ot >= 1; ot = (ot == 1) ? 0 : (ot + 1) / 2) {
            const int CT = fgnn_round_up(ot * d->net, 16);
            const int CTT = CT * nproj;
            if (nct * (CTT / 16) > FGNN_WAVES * BWD_MAXT) continue;     // dW tiles must fit in registers
            const int WS = (CTT + 13) / 32 * 32 + 18;                    // == 18 (mod 32), >= CTT
            const int PS = CTT + 4;
            int64_t fl = (int64_t)p->Npad * p->XS + (int64_t)p->Kpad * WS + 2 * (int64_t)p->Npad * PS +
                         fgnn_round_up(mk, 4) + 2 * (int64_t)fgnn_round_up(mk * d->net, 4) +
                         2 * (int64_t)fgnn_round_up(ot * d->M, 4) + fgnn_round_up(ot, 4) +
                         fgnn_round_up(d->N + 1, 4) + fgnn_round_up(mk, 4);               // + the CSR transpose
            if (fl * 4 <= budget) { best = ot; break; }
        }
    }
    if (!best) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv backward: shape does not fit LDS/register tiling");
    p->OT = best;
    p->CT = fgnn_round_up(best * d->net, 16);
    const int CTT = p->CT * nproj;
    p->WS = (CTT + 13) / 32 * 32 + 18;
    p->PS = CTT + 4;
    int off = 0;
    p->off_xs = off;  off += p->Npad * p->XS;  off = fgnn_round_up(off, 4);
    p->off_ws = off;  off += p->Kpad * p->WS;  off = fgnn_round_up(off, 4);
    p->off_ps = off;  off += p->Npad * p->PS;  off = fgnn_round_up(off, 4);
    p->off_dps = off; off += p->Npad * p->PS;  off = fgnn_round_up(off, 4);
    p->off_idx = off; off += fgnn_round_up(mk, 4);
    p->off_et = off;  off += fgnn_round_up(mk * d->net, 4);
    p->off_det = off; off += fgnn_round_up(mk * d->net, 4);
    p->off_gz = off;  off += fgnn_round_up(best * d->M, 4);
    p->off_aux = off; off += fgnn_round_up(best * d->M, 4);
    p->off_gb = off;  off += fgnn_round_up(best, 4);
    p->off_cs = off;  off += fgnn_round_up(d->N + 1, 4);
    p->off_cl = off;  off += fgnn_round_up(mk, 4);
    return off * 4;
}

template <typename T, int NET>
static void* pick_agg_b(int agg) {
    switch (agg) {
        case FGNN_AGG_MAX: return (void*)mpconv_bwd_kernel<T, NET, FGNN_AGG_MAX>;
        case FGNN_AGG_LSE: return (void*)mpconv_bwd_kernel<T, NET, FGNN_AGG_LSE>;
        default: return (void*)mpconv_bwd_kernel<T, NET, FGNN_AGG_MEAN>;
    }
}
template <typename T>
static void* pick_net_b(int net, int agg) {
    switch (net) {
        case 1: return pick_agg_b<T, 1>(agg);
        case 4: return pick_agg_b<T, 4>(agg);
        case 16: return pick_agg_b<T, 16>(agg);
        default: return pick_agg_b<T, 0>(agg);
    }
}

int fgnn_mpconv_backward_ext(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                             const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                             float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                             fgnn_stream_t stream);
int fgnn_mpconv_backward_ext_accepts(const fgnn_mpconv_desc* d);

extern "C" int fgnn_mpconv_backward_reduces_getype(const fgnn_mpconv_desc* d) {
    if (!d || fgnn_check_desc(d)) return 0;
    return fgnn_mpconv_backward_ext_accepts(d);
}

void fgnn_bw_set_pending_tables(const void* t);

// fgnn_mpconv_backward with the per-graph tables of fgnn_mpconv_backward_tables (NULL = none): the table-driven kernel then
// copies them instead of rebuilding the transposed incidence in every workgroup of every launch.
extern "C" int fgnn_mpconv_backward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                    const void* etype, const float* filters, const void* gz,
                                    const void* z, const uint8_t* argmax, void* gx, void* getype,
                                    float* gfilters, float* gbias, void* workspace,
                                    int64_t workspace_bytes, fgnn_stream_t stream);
extern "C" int fgnn_mpconv_backward_with_tables(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                                const void* etype, const float* filters, const void* gz,
                                                const void* z, const uint8_t* argmax, void* gx, void* getype,
                                                float* gfilters, float* gbias, void* workspace,
                                                int64_t workspace_bytes, const void* tables, fgnn_stream_t stream) {
    fgnn_bw_set_pending_tables(tables);
    const int rc = fgnn_mpconv_backward(d, x, nn_idx, etype, filters, gz, z, argmax, gx, getype, gfilters, gbias, workspace,
                                        workspace_bytes, stream);
    fgnn_bw_set_pending_tables(nullptr);
    return rc;
}

extern "C" int fgnn_mpconv_backward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                    const void* etype, const float* filters, const void* gz,
                                    const void* z, const uint8_t* argmax, void* gx, void* getype,
                                    float* gfilters, float* gbias, void* workspace,
                                    int64_t workspace_bytes, fgnn_stream_t stream) {
    int rc = fgnn_check_desc(d);
    if (rc) return rc;
    if (!x || !nn_idx || !etype || !filters || !gz || !gx || !gfilters)
        FGNN_FAIL(FGNN_EINVAL, "null tensor pointer");
    if (d->agg == FGNN_AGG_MAX && !argmax) FGNN_FAIL(FGNN_EINVAL, "max aggregator needs the forward's argmax");
    if (d->B == 0) return FGNN_OK;
    {   // LDPC shape family: W-stationary persistent kernel (mpconv_bwd_res.hip)
        static const bool force_generic = getenv("FGNN_FORCE_GENERIC") != nullptr;
        if (!force_generic) {
            rc = fgnn_mpconv_backward_ext(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias,
                                          workspace, workspace_bytes, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_backward_hyper(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias,
                                            workspace, workspace_bytes, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_backward_sg(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias,
                                         workspace, workspace_bytes, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_backward_b16(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias,
                                          workspace, workspace_bytes, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_backward_resident(d, x, nn_idx, etype, filters, gz, argmax, gx, getype,
                                               gfilters, gbias, workspace, workspace_bytes, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
        }
    }
    if (d->reserved & FGNN_DESC_GETYPE_REDUCED)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "batch-reduced edge-type gradient asked of a shape without that kernel");
    BwdParams p;
    p.d = *d;
    p.x = x; p.idx = nn_idx; p.et = etype; p.W = filters; p.gz = gz; p.z = z; p.argmax = argmax;
    p.gx = gx; p.get = getype; p.bias = nullptr; p.has_gbias = gbias != nullptr;
    const int lds = plan_backward(d, &p);
    if (lds < 0) return lds;
    // one slab per workgroup (<= 256: the workspace every backward kernel of this library sizes), folded in a fixed order
    const int64_t R_rows = d->ext == FGNN_EXT_NONE ? d->nin : 2 * d->nin;
    const int64_t nw = R_rows * d->nou * d->net, slab_len = nw + d->nou;
    if (!workspace || workspace_bytes < 256 * slab_len * 4)
        FGNN_FAIL(FGNN_EINVAL, "mpconv backward: workspace of fgnn_mpconv_backward_workspace_bytes(d) bytes needed");
    p.slab = (float*)workspace;
    int chunk = (d->B + 255) / 256;
    if (chunk < 1) chunk = 1;
    p.chunk = chunk;
    const int grid = (d->B + chunk - 1) / chunk;
    void* fn = d->dtype == FGNN_F32 ? pick_net_b<float>(d->net, d->agg) : pick_net_b<bf16_t>(d->net, d->agg);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel("mpconv_bwd_kernel<%s, %d, %d>", d->dtype ? "bf16_t" : "float",
                     (d->net == 1 || d->net == 4 || d->net == 16) ? d->net : 0, d->agg);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(FGNN_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward launch: %s", hipGetErrorString(e));
    fgnn_launch_slab_reduce(p.slab, grid, slab_len, nw, gfilters, gbias, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward fold launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
