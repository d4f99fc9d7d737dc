// mpconv_fwd.hip — fused VF/FV message operator, forward, gfx950 (MI355X).
//
// Replaces the ATen op chain of mp_conv_v2.forward (/root/reference/lib/model/mpnn/mp_nn.py:115-175,
// SURVEY §2 steps a-k) with ONE kernel per call:
//
//   per sample b (one workgroup, 4 waves), per tile of OT output channels:
//     1. stage x[b] (N x nin) and the W tile (nin x OT*net [x2 with a self term]) in LDS
//     2. project with f32-input MFMA (v_mfma_f32_16x16x4_f32, exact f32):
//            P[n, o*net+e] = sum_c x[c,n] * W[c, o*net+e]          -> LDS, never to HBM
//     3. per (destination m, channel o): gather the k neighbour rows of P from LDS,
//        contract with the per-edge type weights etype[:,m,j], aggregate over j
//        (max with first-occurrence argmax / online log-sum-exp gamma=3 / mean)
//     4. + bias, optional folded eval-BatchNorm affine, optional ReLU; transpose through LDS,
//        coalesced store.
//
// ORIG_WITH_DIFF / ORIG_WITH_NEIGHBOR use the split form:
//     [x_i, z] @ filters = x_i @ W_self + x_j @ W_nb   with
//       NEIGHBOR: W_self = W_top,          W_nb =  W_bot
//       DIFF    : W_self = W_top + W_bot,  W_nb = -W_bot
// so the per-edge [2nin] GEMM of the reference becomes two node-level projections.
//
// HBM traffic per call = x + etype + nn_idx read once, y written once, filters from L2:
// the algorithmic bytes of SURVEY §8d.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>

struct FwdParams {
    fgnn_mpconv_desc d;
    const void* x;
    const int64_t* idx;
    const void* et;
    const float* W;
    const float* bias;
    const float* pscale;
    const float* pshift;
    void* y;
    uint8_t* argmax;
    int OT, CT, nproj, Npad, Kpad, XS, WS, PS, YS;
    int off_xs, off_ws, off_ps, off_idx, off_et, off_ys, off_ya;   // float offsets into LDS
};

extern __shared__ __attribute__((aligned(16))) float fgnn_lds[];

template <int NET>
__device__ __forceinline__ float edge_dot(const float* __restrict__ etp, const float* __restrict__ pn,
                                          const float* __restrict__ pself, int net) {
    float v = 0.f;
    if constexpr (NET == 1) {
        float p = pn[0];
        if (pself) p += pself[0];
        v = etp[0] * p;
    } else if constexpr (NET == 4 || NET == 16) {
#pragma unroll
        for (int q = 0; q < NET / 4; ++q) {
            f32x4 e4 = *reinterpret_cast<const f32x4*>(etp + 4 * q);
            f32x4 p4 = *reinterpret_cast<const f32x4*>(pn + 4 * q);
            if (pself) p4 += *reinterpret_cast<const f32x4*>(pself + 4 * q);
            v = fmaf(e4[0], p4[0], v);
            v = fmaf(e4[1], p4[1], v);
            v = fmaf(e4[2], p4[2], v);
            v = fmaf(e4[3], p4[3], v);
        }
    } else {
        for (int e = 0; e < net; ++e) {
            float p = pn[e];
            if (pself) p += pself[e];
            v = fmaf(etp[e], p, v);
        }
    }
    return v;
}

template <typename T, int NET, int AGG>
__global__ __launch_bounds__(FGNN_THREADS) void mpconv_fwd_kernel(const FwdParams p) {
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nin = d.nin, nou = d.nou, net = d.net, N = d.N, M = d.M, k = d.k;
    const int ncols = nou * net;
    const int CT = p.CT, CTT = p.CT * p.nproj;
    const bool self = p.nproj == 2;

    float* xs = fgnn_lds + p.off_xs;
    float* ws = fgnn_lds + p.off_ws;
    float* ps = fgnn_lds + p.off_ps;
    int* idx_s = reinterpret_cast<int*>(fgnn_lds + p.off_idx);
    float* et_s = fgnn_lds + p.off_et;
    float* ys = fgnn_lds + p.off_ys;
    int* ya = reinterpret_cast<int*>(fgnn_lds + p.off_ya);

    const T* xg = static_cast<const T*>(p.x);
    const T* etg = static_cast<const T*>(p.et);
    T* yg = static_cast<T*>(p.y);

    for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
        // ---- stage x[b] -> xs[n][c] (zero padded), nn_idx[b], etype[b] -> et_s[m*k+j][e] ----
        {
            const T* xb = xg + (int64_t)b * d.x_sb;
            const int tot = p.Npad * p.Kpad;
            if (d.x_sn == 1 || d.x_sc != 1) {          // node-fastest global order (NCHW)
                for (int f = tid; f < tot; f += FGNN_THREADS) {
                    const int c = f / p.Npad, n = f - c * p.Npad;
                    float v = 0.f;
                    if (c < nin && n < N) v = fgnn_ld(xb + (int64_t)c * d.x_sc + (int64_t)n * d.x_sn);
                    xs[n * p.XS + c] = v;
                }
            } else {                                    // channel-fastest (channels-last)
                for (int f = tid; f < tot; f += FGNN_THREADS) {
                    const int n = f / p.Kpad, c = f - n * p.Kpad;
                    float v = 0.f;
                    if (c < nin && n < N) v = fgnn_ld(xb + (int64_t)c * d.x_sc + (int64_t)n * d.x_sn);
                    xs[n * p.XS + c] = v;
                }
            }
            const int64_t* ib = p.idx + (int64_t)b * d.idx_sb;
            for (int f = tid; f < M * k; f += FGNN_THREADS) {
                const int m = f / k, j = f - m * k;
                long long v = ib[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
                v = v < 0 ? 0 : (v >= N ? N - 1 : v);   // clamp: never fault on a bad index
                idx_s[f] = (int)v;
            }
            const T* eb = etg + (int64_t)b * d.et_sb;
            const int mk = M * k;
            for (int f = tid; f < mk * net; f += FGNN_THREADS) {
                int e, r;
                if (d.et_se == 1) { r = f / net; e = f - r * net; }         // edge-type fastest in memory
                else { e = f / mk; r = f - e * mk; }
                const int m = r / k, j = r - m * k;
                et_s[r * net + e] =
                    fgnn_ld(eb + (int64_t)e * d.et_se + (int64_t)m * d.et_sm + (int64_t)j * d.et_sk);
            }
        }

        for (int o0 = 0; o0 < nou; o0 += p.OT) {
            const int otc = min(p.OT, nou - o0);          // channels in this tile
            const int vcols = otc * net;                  // valid columns per projection
            // ---- stage the W tile: ws[c][proj*CT + q], zero padded ----
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
            __syncthreads();

            // ---- projection: P^T tile = W^T (cols x nin) . x (nin x nodes), f32 MFMA ----
            {
                const int nslab = CTT / 16, ntile = p.Npad / 16, ntp = (ntile + 1) / 2;
                const int ksteps = p.Kpad / 4;
                for (int u = wave; u < nslab * ntp; u += FGNN_WAVES) {
                    const int slab = u % nslab, tp = u / nslab;
                    const int t0 = tp * 2;
                    const bool two = (t0 + 1) < ntile;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* ap = ws + lk * p.WS + slab * 16 + li;
                    const float* bp0 = xs + (t0 * 16 + li) * p.XS + lk;
                    const float* bp1 = bp0 + 16 * p.XS;
                    if (two) {
#pragma unroll 4
                        for (int kk = 0; kk < ksteps; ++kk) {
                            const float a = ap[kk * 4 * p.WS];
                            const float b0 = bp0[kk * 4];
                            const float b1 = bp1[kk * 4];
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
                        }
                    } else {
#pragma unroll 4
                        for (int kk = 0; kk < ksteps; ++kk) {
                            const float a = ap[kk * 4 * p.WS];
                            const float b0 = bp0[kk * 4];
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
                        }
                    }
                    float* dst = ps + (t0 * 16 + li) * p.PS + slab * 16 + 4 * lk;
                    *reinterpret_cast<f32x4*>(dst) = acc0;
                    if (two) *reinterpret_cast<f32x4*>(dst + 16 * p.PS) = acc1;
                }
            }
            __syncthreads();

            // ---- gather + edge-type contraction + aggregation over the k neighbours ----
            for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                const int m = it / otc, ol = it - m * otc;
                const float* pself = self ? ps + m * p.PS + CT + ol * net : nullptr;
                const int* ip = idx_s + m * k;
                const float* ep = et_s + (m * k) * net;
                float res;
                int arg = 0;
                if constexpr (AGG == FGNN_AGG_MAX) {
                    float best = 0.f;
                    for (int j = 0; j < k; ++j) {
                        const float v = edge_dot<NET>(ep + j * net, ps + ip[j] * p.PS + ol * net, pself, net);
                        if (j == 0 || v > best) { best = v; arg = j; }   // strict >: first occurrence wins ties
                    }
                    res = best;
                } else if constexpr (AGG == FGNN_AGG_LSE) {
                    float mx = -INFINITY, s = 0.f;
                    for (int j = 0; j < k; ++j) {
                        const float v = 3.0f * edge_dot<NET>(ep + j * net, ps + ip[j] * p.PS + ol * net, pself, net);
                        if (v > mx) { s = s * expf(mx - v) + 1.0f; mx = v; }
                        else s += expf(v - mx);
                    }
                    res = (1.0f / 3.0f) * (mx + logf(s));
                } else {
                    float s = 0.f;
                    for (int j = 0; j < k; ++j)
                        s += edge_dot<NET>(ep + j * net, ps + ip[j] * p.PS + ol * net, pself, net);
                    res = s / (float)k;
                }
                const int o = o0 + ol;
                if (p.bias) res += p.bias[o];
                if (p.pscale) res = res * p.pscale[o] + p.pshift[o];
                if (d.relu) res = fmaxf(res, 0.f);
                ys[ol * p.YS + m] = res;
                if (AGG == FGNN_AGG_MAX) ya[ol * p.YS + m] = arg;
            }
            __syncthreads();

            // ---- store the [otc x M] tile ----
            {
                T* yb = yg + (int64_t)b * d.y_sb;
                if (d.y_sc == 1 && d.y_sm != 1) {           // channels-last: o fastest
                    for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                        const int m = it / otc, ol = it - m * otc;
                        fgnn_st(yb + (int64_t)(o0 + ol) * d.y_sc + (int64_t)m * d.y_sm, ys[ol * p.YS + m]);
                    }
                } else {                                     // NCHW: m fastest
                    for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                        const int ol = it / M, m = it - ol * M;
                        fgnn_st(yb + (int64_t)(o0 + ol) * d.y_sc + (int64_t)m * d.y_sm, ys[ol * p.YS + m]);
                    }
                }
                if (AGG == FGNN_AGG_MAX && p.argmax) {     // argmax shares y's element strides
                    uint8_t* ab = p.argmax + (int64_t)b * d.y_sb;
                    for (int it = tid; it < M * otc; it += FGNN_THREADS) {
                        const int ol = it / M, m = it - ol * M;
                        ab[(int64_t)(o0 + ol) * d.y_sc + (int64_t)m * d.y_sm] = (uint8_t)ya[ol * p.YS + m];
                    }
                }
            }
            // next tile's W staging touches only ws (free since the MFMA barrier); the barrier
            // after it orders these ys reads before the next gather's ys writes.
        }
        __syncthreads();   // xs / idx_s / et_s are rewritten by the next sample
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
static int plan_forward(const fgnn_mpconv_desc* d, FwdParams* p) {
    const int nproj = d->ext == FGNN_EXT_NONE ? 1 : 2;
    p->nproj = nproj;
    p->Npad = fgnn_round_up(d->N, 16);
    p->Kpad = fgnn_round_up(d->nin, 4);
    // XS == 2 (mod 32): conflict-free ds_read_b32 of the B operand (16 rows x 2 k per half-wave)
    p->XS = (p->Kpad + 29) / 32 * 32 + 2;
    p->YS = d->M + 1;
    const int mk = d->M * d->k;
    int best_ot = 0;
    int64_t best_bytes = 0;
    for (int pass = 0; pass < 2 && !best_ot; ++pass) {
        const int64_t budget = pass == 0 ? 72 * 1024 : 160 * 1024;
        for (int ot = d->nou < 64 ? d->nou : 64; ot >= 1; ot = (ot == 1) ? 0 : (ot + 1) / 2) {
            const int CT = fgnn_round_up(ot * d->net, 16);
            const int CTT = CT * nproj;
            if (pass == 0 && CTT > 128 && ot > 1) continue;      // keep the P tile small: 2 WG/CU
            const int WS = (CTT % 32 == 0) ? CTT + 16 : CTT;     // == 16 (mod 32)
            const int PS = CTT + 4;
            int64_t fl = (int64_t)p->Npad * p->XS + (int64_t)p->Kpad * WS + (int64_t)p->Npad * PS +
                         fgnn_round_up(mk, 4) + fgnn_round_up(mk * d->net, 4) +
                         2 * (int64_t)fgnn_round_up(ot * p->YS, 4);
            if (fl * 4 <= budget) { best_ot = ot; best_bytes = fl * 4; break; }
        }
    }
    if (!best_ot) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv forward: shape needs more than 160 KiB of LDS");
    p->OT = best_ot;
    p->CT = fgnn_round_up(best_ot * d->net, 16);
    const int CTT = p->CT * nproj;
    p->WS = (CTT % 32 == 0) ? CTT + 16 : CTT;
    p->PS = CTT + 4;
    int off = 0;
    p->off_xs = off;  off += p->Npad * p->XS;  off = fgnn_round_up(off, 4);
    p->off_ws = off;  off += p->Kpad * p->WS;  off = fgnn_round_up(off, 4);
    p->off_ps = off;  off += p->Npad * p->PS;  off = fgnn_round_up(off, 4);
    p->off_idx = off; off += fgnn_round_up(mk, 4);
    p->off_et = off;  off += fgnn_round_up(mk * d->net, 4);
    p->off_ys = off;  off += fgnn_round_up(best_ot * p->YS, 4);
    p->off_ya = off;  off += fgnn_round_up(best_ot * p->YS, 4);
    (void)best_bytes;
    return off * 4;
}

int fgnn_mpconv_forward_resident(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                 const void* etype, const float* filters, const float* bias,
                                 const float* post_scale, const float* post_shift, void* y,
                                 uint8_t* argmax, fgnn_stream_t stream);

int fgnn_mpconv_forward_hyper(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                              const float* filters, const float* bias, const float* post_scale,
                              const float* post_shift, void* y, uint8_t* argmax, fgnn_stream_t stream);
int fgnn_mpconv_forward_b16(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                            const void* etype, const float* filters, const float* bias,
                            const float* post_scale, const float* post_shift, void* y,
                            uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid);

int fgnn_mpconv_forward_ext(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                            void* y, uint8_t* argmax, fgnn_stream_t stream);

int fgnn_check_desc(const fgnn_mpconv_desc* d) {
    if (!d) FGNN_FAIL(FGNN_EINVAL, "null descriptor");
    if (d->B < 0 || d->nin < 1 || d->nou < 1 || d->net < 1 || d->N < 1 || d->M < 1 || d->k < 1)
        FGNN_FAIL(FGNN_EINVAL, "bad sizes B=%d nin=%d nou=%d net=%d N=%d M=%d k=%d", d->B, d->nin,
                  d->nou, d->net, d->N, d->M, d->k);
    if (d->k > 255) FGNN_FAIL(FGNN_EUNSUPPORTED, "k=%d > 255 neighbours per destination", d->k);
    if (d->ext < 0 || d->ext > 2) FGNN_FAIL(FGNN_EINVAL, "extension must one of mp_conv_type");
    if (d->agg < 0 || d->agg > 2) FGNN_FAIL(FGNN_EINVAL, "unknown aggregator %d", d->agg);
    if (d->dtype != FGNN_F32 && d->dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "unknown dtype %d", d->dtype);
    if (d->ext != FGNN_EXT_NONE && d->N != d->M)
        FGNN_FAIL(FGNN_EINVAL, "extension %d needs N == M (got N=%d M=%d)", d->ext, d->N, d->M);
    return FGNN_OK;
}

template <typename T, int NET>
static void* pick_agg(int agg) {
    switch (agg) {
        case FGNN_AGG_MAX: return (void*)mpconv_fwd_kernel<T, NET, FGNN_AGG_MAX>;
        case FGNN_AGG_LSE: return (void*)mpconv_fwd_kernel<T, NET, FGNN_AGG_LSE>;
        default: return (void*)mpconv_fwd_kernel<T, NET, FGNN_AGG_MEAN>;
    }
}
template <typename T>
static void* pick_net(int net, int agg) {
    switch (net) {
        case 1: return pick_agg<T, 1>(agg);
        case 4: return pick_agg<T, 4>(agg);
        case 16: return pick_agg<T, 16>(agg);
        default: return pick_agg<T, 0>(agg);
    }
}

extern "C" int64_t fgnn_mpconv_forward_lds_bytes(const fgnn_mpconv_desc* d) {
    if (fgnn_check_desc(d)) return -1;
    FwdParams p;
    return plan_forward(d, &p);
}

extern "C" int fgnn_mpconv_forward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                   const void* etype, const float* filters, const float* bias,
                                   const float* post_scale, const float* post_shift, void* y,
                                   uint8_t* argmax, fgnn_stream_t stream) {
    int rc = fgnn_check_desc(d);
    if (rc) return rc;
    if (!x || !nn_idx || !etype || !filters || !y) FGNN_FAIL(FGNN_EINVAL, "null tensor pointer");
    if ((post_scale == nullptr) != (post_shift == nullptr))
        FGNN_FAIL(FGNN_EINVAL, "post_scale and post_shift must be given together");
    if (d->B == 0) return FGNN_OK;
    {   // LDPC shape family: W-stationary persistent kernel (mpconv_fwd_res.hip)
        static const bool force_generic = getenv("FGNN_FORCE_GENERIC") != nullptr;
        if (!force_generic) {
            rc = fgnn_mpconv_forward_hyper(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, argmax,
                                           stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_forward_b16(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, argmax,
                                         stream, nullptr, nullptr);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_forward_ext(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, argmax, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
            rc = fgnn_mpconv_forward_resident(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y,
                                              argmax, stream);
            if (rc != 0) return rc < 0 ? rc : FGNN_OK;
        }
    }
    FwdParams p;
    p.d = *d;
    p.x = x; p.idx = nn_idx; p.et = etype; p.W = filters; p.bias = bias;
    p.pscale = post_scale; p.pshift = post_shift; p.y = y; p.argmax = argmax;
    const int lds = plan_forward(d, &p);
    if (lds < 0) return lds;
    void* fn = d->dtype == FGNN_F32 ? pick_net<float>(d->net, d->agg) : pick_net<bf16_t>(d->net, d->agg);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel("mpconv_fwd_kernel<%s, %d, %d>", d->dtype ? "bf16_t" : "float",
                     (d->net == 1 || d->net == 4 || d->net == 16) ? d->net : 0, d->agg);
    const int grid = d->B < 4096 ? d->B : 4096;
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(FGNN_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv forward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// Training form of fgnn_mpconv_forward whose epilogue also leaves the BatchNorm batch statistics of the stored output:
// per-workgroup partials [rows][2][nou] (sum, sum of squares) for fgnn_bn_finalize — the BatchNorm behind the operator
// (mp_nn.py:170) then needs no pass of its own over z.  fgnn_mpconv_forward_stats_partials: the number of partial rows
// this descriptor's launch writes, or 0 when the shape has no statistics epilogue (use fgnn_mpconv_forward + fgnn_bn_stats).
extern "C" int fgnn_mpconv_forward_stats_partials(const fgnn_mpconv_desc* d) {
    if (fgnn_check_desc(d) || d->B == 0) return 0;
    if (d->net == 1 && (d->M == 1 || (d->N == 1 && d->k == 1))) return 0;       // the hyper-edge kernels take these
    int grid = 0;
    return fgnn_mpconv_forward_b16(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, &grid) == 1 ? grid : 0;
}

// The BatchNorm finalisation of the NEXT statistics launch on this thread: fgnn_mpconv_forward_stats sets it around its call of
// the kernel families' host entry points (b16 -> sg -> ws), whichever launches picks it up (as the inference addends travel).
static thread_local const fgnn_bn_final* stats_pending_fin = nullptr;
static thread_local void* stats_pending_scratch = nullptr;
void fgnn_stats_pending(const fgnn_bn_final** fin, void** scratch) { *fin = stats_pending_fin; *scratch = stats_pending_scratch; }
// second launch of a 64 -> 128 call: the upper 64 channels of every per-channel vector (num_batches_tracked was counted by the first)
void fgnn_stats_upper_half(FgnnFold* fold, fgnn_bn_final* fin) {
    if (!fold->tickets) return;
    fold->part += 64;
    if (fin->gamma) fin->gamma += 64;
    if (fin->beta) fin->beta += 64;
    if (fin->running_mean) { fin->running_mean += 64; fin->running_var += 64; }
    fin->mean += 64; fin->invstd += 64; fin->scale += 64; fin->shift += 64;
    if (fin->shift_k) fin->shift_k += 64;
    fin->num_batches_tracked = nullptr;
}
int fgnn_bn_finalize_launch(const float* partials, int npartials, int C, const fgnn_bn_final* fin, hipStream_t st);

extern "C" int fgnn_mpconv_forward_stats(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                                         const void* etype, const float* filters, const float* bias, void* y,
                                         uint8_t* argmax, float* stats_partials, const fgnn_bn_final* fin, void* fold_scratch,
                                         fgnn_stream_t stream) {
    int rc = fgnn_check_desc(d);
    if (rc) return rc;
    if (!x || !nn_idx || !etype || !filters || !y || !stats_partials) FGNN_FAIL(FGNN_EINVAL, "null tensor pointer");
    const int rows = fgnn_mpconv_forward_stats_partials(d);
    if (rows == 0)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_forward_stats: this shape has no statistics epilogue");
    if (fin && (!fin->mean || !fin->invstd || !fin->scale || !fin->shift || fin->shift_k || fin->count != (int64_t)d->B * d->M))
        FGNN_FAIL(FGNN_EINVAL, "mpconv_forward_stats: fgnn_bn_final needs its outputs, count == B * M and no shift_k");
    const bool inkernel = fin && fold_scratch && !fgnn_separate_finalisers();
    stats_pending_fin = inkernel ? fin : nullptr;
    stats_pending_scratch = inkernel ? fold_scratch : nullptr;
    rc = fgnn_mpconv_forward_b16(d, x, nn_idx, etype, filters, bias, nullptr, nullptr, y, argmax, stream, stats_partials,
                                 nullptr);
    stats_pending_fin = nullptr;
    stats_pending_scratch = nullptr;
    if (rc == 0) FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_forward_stats: this shape has no statistics epilogue");
    if (rc != 1) return rc;
    if (fin && !inkernel && fgnn_bn_finalize_launch(stats_partials, rows, d->nou, fin, (hipStream_t)stream))
        FGNN_FAIL(FGNN_ELAUNCH, "mpconv_forward_stats finaliser launch: %s", hipGetErrorString(hipGetLastError()));
    return FGNN_OK;
}


// fgnn_mpconv_forward with up to three ADDENDS of y's layout (the caller's running sum, residual, skip term: factor_mpnn_sp.py:139-168
// adds them to the operator's activated output) folded into the kernel's epilogue where the kernel family supports it — the
// third-generation bf16 parity kernel in its inference mode (post_scale / post_shift + ReLU).  Returns 1 when the addends were
// added by the kernel, 0 when y was computed WITHOUT them (the caller adds them: one more pass), < 0 on error.
void fgnn_ws_set_pending_addends(const void* a0, const void* a1, const void* a2);
int fgnn_ws_pending_addends_taken(void);
extern "C" int fgnn_mpconv_forward_addends(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                                           const float* filters, const float* bias, const float* post_scale,
                                           const float* post_shift, const void* addend0, const void* addend1, const void* addend2,
                                           void* y, fgnn_stream_t stream) {
    if (!addend0 && (addend1 || addend2)) FGNN_FAIL(FGNN_EINVAL, "mpconv_forward_addends: addend0 first");
    fgnn_ws_set_pending_addends(addend0, addend1, addend2);
    const int rc = fgnn_mpconv_forward(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, nullptr, stream);
    const int taken = fgnn_ws_pending_addends_taken();
    fgnn_ws_set_pending_addends(nullptr, nullptr, nullptr);
    if (rc != FGNN_OK) return rc;
    return (addend0 && taken) ? 1 : 0;
}
