// mpconv_fwd_hyper.hip — forward of the VF/FV message operator for the two "hyper-edge" calls of a factor graph
// with one factor that touches every variable (the LDPC hyper-factor, /root/reference/train_ldpc.py:60-75), bf16
// channel-fastest storage, single edge type, NO_EXTENSION.  Same maths as mpconv_fwd_b16.hip (reference
// /root/reference/lib/model/mpnn/mp_nn.py:115-175):
//
//  fan-in  (M == 1):  z[o] = agg_j et[j] * P[idx[j], o],  P = x^T W            (k = 96 neighbours, one destination)
//  fan-out (N == 1, k == 1):  z[m, o] = et[m] * P[0, o]                        (96 destinations, one source)
//
// then y = act((z + bias) * post_scale + post_shift).  Both calls move ~12 KB per sample (x in, or y out) and do
// almost no arithmetic, so the shape-generic persistent kernel (LDS staging, per-sample barriers, 8 waves sharing
// one sample) leaves 90 % of the HBM bandwidth unused on them.  Here: ONE WAVE PER SAMPLE, no workgroup barriers
// in the sample loop.  fan-in: the B operand of P^T = W^T x is read straight from global memory (8 consecutive
// channels of a node = one 16-byte load per lane, every byte of x fetched once), W^T fragments stay in registers,
// P goes to a per-wave LDS image as bf16 and the neighbour reduction walks it with lane <-> output channel.
// fan-out: P is 64..128 numbers per sample (lane <-> 4..8 channels, W from LDS), the output rows are streamed
// with 8..16-byte stores.
#include "fgnn_common.h"
#include <stdlib.h>

typedef __bf16 fh_bf16x8 __attribute__((ext_vector_type(8)));

struct FhParams {
    fgnn_mpconv_desc d;
    const uint16_t* x;
    const int64_t* idx;
    const uint16_t* et;
    const float* W;      // [nin][nou]
    const float* bias;
    const float* pscale;
    const float* pshift;
    uint16_t* y;
    uint8_t* argmax;
    int waves;           // waves per workgroup
    int Npad16;
};

extern __shared__ __attribute__((aligned(16))) unsigned char fgnn_lds_fh[];

__device__ __forceinline__ float fh_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float fh_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned fh_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ uint16_t fh_bf16(float a) {
    const __bf16 h = (__bf16)a;
    return __builtin_bit_cast(uint16_t, h);
}

// ----------------------------------------------------------------------------------------
// fan-in: M == 1.  KS2 = nin / 32, OT = nou / 16 output tiles.
// ----------------------------------------------------------------------------------------
template <int AGG, int KS2, int OT>
__global__ __launch_bounds__(512) void mpconv_fwd_fanin_kernel(const FhParams p) {
    constexpr int NIN = 32 * KS2, NOU = 16 * OT, NO = NOU / 64;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int N = d.N, k = d.k;
    uint16_t* Pl = reinterpret_cast<uint16_t*>(fgnn_lds_fh) + (size_t)wave * p.Npad16 * NOU;   // [Npad16][NOU] bf16

    // resident A fragments of P^T = W^T x: A[i = o][k = c] = W[c][o], 8 consecutive c (strided read, once)
    fh_bf16x8 aP[OT][KS2];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = p.W[(int64_t)(32 * ks + 8 * lk + u) * NOU + ot * 16 + li];
            aP[ot][ks] = __builtin_bit_cast(fh_bf16x8, make_uint4(fh_pack2(w8[0], w8[1]), fh_pack2(w8[2], w8[3]),
                                                                  fh_pack2(w8[4], w8[5]), fh_pack2(w8[6], w8[7])));
        }
    float c_bias[NO], c_scale[NO], c_shift[NO];
#pragma unroll
    for (int q = 0; q < NO; ++q) {
        const int o = lane + 64 * q;
        c_bias[q] = p.bias ? p.bias[o] : 0.f;
        c_scale[q] = p.pscale ? p.pscale[o] : 1.f;
        c_shift[q] = p.pscale ? p.pshift[o] : 0.f;
    }
    const int ntile = p.Npad16 / 16;
    const int nwaves = gridDim.x * p.waves;
    for (int b = blockIdx.x * p.waves + wave; b < d.B; b += nwaves) {
        const uint16_t* xb = p.x + (int64_t)b * d.x_sb;
        // ---- P[n][o] for all nodes: B[k = c][j = n] straight from global (16 bytes per lane).  The launch gives every wave ONE
        //      sample (4096 waves at the benched batch), so the kernel's duration is one wave's dependency chain: all the loads of
        //      TB node tiles (the whole sample at 64 input channels) are in flight together (tile by tile: six HBM round trips; 33.3 ->
        //      30.0 us per launch for 51 MB).  Measured with the phases switched off (gpurun_out/r05s): 4 us fixed (W fragments), + 13 us
        //      projection, + 12 us neighbour walk; a persistent grid (256 / 512 workgroups) is slower (44 / 31 us) ----
        constexpr int TB = KS2 == 2 ? 6 : 3;
        for (int nt0 = 0; nt0 < ntile; nt0 += TB) {
            uint4 bx[TB][KS2];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int n = (nt0 + t) * 16 + li;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks)
                    bx[t][ks] = n < N ? *reinterpret_cast<const uint4*>(xb + (int64_t)n * NIN + 32 * ks + 8 * lk) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int n = (nt0 + t) * 16 + li;
                if (nt0 + t < ntile) {
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KS2; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aP[ot][ks], __builtin_bit_cast(fh_bf16x8, bx[t][ks]), acc, 0, 0, 0);
                        // D[i = o 4lk+r][j = n]: four consecutive channels of node n
                        *reinterpret_cast<uint2*>(Pl + n * NOU + ot * 16 + 4 * lk) =
                            make_uint2(fh_pack2(acc[0], acc[1]), fh_pack2(acc[2], acc[3]));
                    }
                }
            }
        }
        // ---- neighbour reduction, lane <-> output channel (wave-private LDS image: no barrier needed, but the
        //      writes above must have landed: same wave, in-order LDS) ----
        float best[NO], ssum[NO];
        int arg[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) { best[q] = 0.f; ssum[q] = 0.f; arg[q] = 0; }
        const int64_t* ib = p.idx + (int64_t)b * d.idx_sb;
        const uint16_t* eb = p.et + (int64_t)b * d.et_sb;
        for (int j0 = 0; j0 < k; j0 += 64) {
            // lane j0 + lane holds its neighbour's node id and weight; broadcast by readlane
            int nid = 0;
            float ew = 0.f;
            if (j0 + lane < k) {
                long long v = ib[(int64_t)(j0 + lane) * d.idx_sk];
                nid = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
                ew = __uint_as_float((unsigned)eb[(int64_t)(j0 + lane) * d.et_sk] << 16);
            }
            const int jn = min(64, k - j0);
            for (int jg = 0; jg < jn; jg += 8) {           // eight neighbours at a time: their P reads are issued together
                float e8[8], pv[8][NO];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = min(jg + u, jn - 1);
                    const int n = __builtin_amdgcn_readlane(nid, j);
                    e8[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ew), j));
#pragma unroll
                    for (int q = 0; q < NO; ++q) pv[u][q] = __uint_as_float((unsigned)Pl[n * NOU + lane + 64 * q] << 16);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (jg + u < jn) {
                        const int jj = j0 + jg + u;
#pragma unroll
                        for (int q = 0; q < NO; ++q) {
                            float v = e8[u] * pv[u][q];
                            if constexpr (AGG == FGNN_AGG_MAX) {
                                if (jj == 0 || v > best[q]) { best[q] = v; arg[q] = jj; }   // strict >: first occurrence
                            } else if constexpr (AGG == FGNN_AGG_LSE) {
                                v *= 3.0f;
                                if (jj == 0) { best[q] = v; ssum[q] = 1.0f; }
                                else if (v > best[q]) { ssum[q] = ssum[q] * expf(best[q] - v) + 1.0f; best[q] = v; }
                                else ssum[q] += expf(v - best[q]);
                            } else {
                                ssum[q] += v;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            float res;
            if constexpr (AGG == FGNN_AGG_MAX) res = best[q];
            else if constexpr (AGG == FGNN_AGG_LSE) res = (1.0f / 3.0f) * (best[q] + logf(ssum[q]));
            else res = ssum[q] / (float)k;
            res = (res + c_bias[q]) * c_scale[q] + c_shift[q];
            if (d.relu) res = fmaxf(res, 0.f);
            const int64_t off = (int64_t)b * d.y_sb + (int64_t)(lane + 64 * q) * d.y_sc;
            p.y[off] = fh_bf16(res);
            if (AGG == FGNN_AGG_MAX && p.argmax) p.argmax[off] = (uint8_t)arg[q];
        }
    }
}

// ----------------------------------------------------------------------------------------
// fan-in, max aggregation, IDENTITY neighbour list (k == N, idx[j] == j: the LDPC hyper-factor listens to every variable in order,
// /root/reference/train_ldpc.py:40-46) — round 5.  The kernel above is one dependency chain per wave (4 us fixed + 13 us projection
// + 12 us neighbour walk at 4096 codewords, gpurun_out/r05s): it writes P to an LDS image and walks the 96 neighbours one by one
// per output-channel lane — ~12 wave instructions per neighbour.  With the neighbours in node order the reduction stays in the MFMA
// accumulators: D[i = channel][j = node] of a node tile gives lane (node li, channel group lk) four channels of ONE node, so the
// lane keeps a running (max, first argmax) over its node position across the tiles — 4 instructions per value — and the 16 node
// lanes meet once per sample through a 5 KB LDS transpose (lane <-> channel reads its 16 candidates: larger value, then smaller
// node).  No P image (the projected values stay f32), no per-neighbour LDS reads or readlanes; 4 waves of 128 registers per
// workgroup, so every sample of a 4096-codeword batch is resident at once.
// ----------------------------------------------------------------------------------------
#define FI_ROWPAD 4      // floats: candidate rows 16 bytes apart in their bank phase
#ifndef FI_TB
#define FI_TB 3          // node tiles whose loads are in flight together at 64 input channels: 6 (the whole sample) costs 166 registers = 2-3 waves per SIMD, 3 costs 118 = 4
#endif

template <int KS2, int OT, bool ARG>
__global__ __launch_bounds__(256) void mpconv_fwd_fanin_id_kernel(const FhParams p) {
    constexpr int NIN = 32 * KS2, NOU = 16 * OT, NO = NOU / 64, TB = KS2 == 2 ? FI_TB : 3, CROW = NOU + FI_ROWPAD;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int N = d.N;
    float* cand = reinterpret_cast<float*>(fgnn_lds_fh) + (size_t)wave * 16 * CROW;                   // [16 node lanes][NOU (+pad)] f32
    uint8_t* carg = reinterpret_cast<uint8_t*>(reinterpret_cast<float*>(fgnn_lds_fh) + 4 * 16 * CROW) + (size_t)wave * 16 * NOU;   // [16][NOU] u8
    const int ntile = p.Npad16 / 16;
    const int nwaves = gridDim.x * 4;
    int b = blockIdx.x * 4 + wave;
    // the first sample's rows go out before the W fragments are fetched (64 strided 4-byte loads per lane: the kernel's fixed cost)
    fh_bf16x8 aP[OT][KS2];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = p.W[(int64_t)(32 * ks + 8 * lk + u) * NOU + ot * 16 + li];
            aP[ot][ks] = __builtin_bit_cast(fh_bf16x8, make_uint4(fh_pack2(w8[0], w8[1]), fh_pack2(w8[2], w8[3]),
                                                                  fh_pack2(w8[4], w8[5]), fh_pack2(w8[6], w8[7])));
        }
    float c_bias[NO], c_scale[NO], c_shift[NO];
#pragma unroll
    for (int q = 0; q < NO; ++q) {
        const int o = lane + 64 * q;
        c_bias[q] = p.bias ? p.bias[o] : 0.f;
        c_scale[q] = p.pscale ? p.pscale[o] : 1.f;
        c_shift[q] = p.pscale ? p.pshift[o] : 0.f;
    }
    for (; b < d.B; b += nwaves) {
        const uint16_t* xb = p.x + (int64_t)b * d.x_sb;
        const uint16_t* eb = p.et + (int64_t)b * d.et_sb;
        float best[OT][4];
        int arg[OT][4];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[ot][r] = -__builtin_huge_valf(); arg[ot][r] = 255; }
        for (int nt0 = 0; nt0 < ntile; nt0 += TB) {
            uint4 bx[TB][KS2];
            float ew[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int n = (nt0 + t) * 16 + li;
                const bool ok = n < N;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks)
                    bx[t][ks] = ok ? *reinterpret_cast<const uint4*>(xb + (int64_t)n * NIN + 32 * ks + 8 * lk) : make_uint4(0, 0, 0, 0);
                ew[t] = ok ? __uint_as_float((unsigned)eb[(int64_t)n * d.et_sk] << 16) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int n = (nt0 + t) * 16 + li;
                if (nt0 + t < ntile) {                               // (wave-uniform)
                    const bool ok = n < N;
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KS2; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aP[ot][ks], __builtin_bit_cast(fh_bf16x8, bx[t][ks]), acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                // acc[r] = P[node n][channel 16 ot + 4 lk + r]
                            const float v = ew[t] * acc[r];
                            const bool take = ok && v > best[ot][r];   // strict >: the lane's first occurrence (its nodes ascend)
                            best[ot][r] = take ? v : best[ot][r];
                            if constexpr (ARG) arg[ot][r] = take ? n : arg[ot][r];
                        }
                    }
                }
            }
        }
        // the 16 node lanes meet: candidates [node lane][channel] through the wave's LDS rows, then lane <-> channel
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            *reinterpret_cast<f32x4*>(cand + li * CROW + 16 * ot + 4 * lk) = (f32x4){best[ot][0], best[ot][1], best[ot][2], best[ot][3]};
            if constexpr (ARG)
                *reinterpret_cast<unsigned*>(carg + li * NOU + 16 * ot + 4 * lk) =
                    (unsigned)arg[ot][0] | ((unsigned)arg[ot][1] << 8) | ((unsigned)arg[ot][2] << 16) | ((unsigned)arg[ot][3] << 24);
        }
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            const int o = lane + 64 * q;
            float bv = cand[o];
            int ba = ARG ? (int)carg[o] : 0;
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                const float v = cand[j * CROW + o];
                if constexpr (ARG) {
                    const int a = (int)carg[j * NOU + o];
                    const bool take = v > bv || (v == bv && a < ba);
                    bv = take ? v : bv;
                    ba = take ? a : ba;
                } else {
                    bv = fmaxf(bv, v);
                }
            }
            float res = (bv + c_bias[q]) * c_scale[q] + c_shift[q];
            if (d.relu) res = fmaxf(res, 0.f);
            const int64_t off = (int64_t)b * d.y_sb + (int64_t)o * d.y_sc;
            p.y[off] = fh_bf16(res);
            if constexpr (ARG) p.argmax[off] = (uint8_t)ba;
        }
    }
}

// ----------------------------------------------------------------------------------------
// fan-out: N == 1, k == 1; y channel-fastest [M][nou].  CH = nou / 16 channels per lane.
// ----------------------------------------------------------------------------------------
template <int NI, int CH>
__global__ __launch_bounds__(512) void mpconv_fwd_fanout_kernel(const FhParams p) {
    constexpr int NIN = 64 * NI, NOU = 16 * CH;
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    float* Wl = reinterpret_cast<float*>(fgnn_lds_fh);             // [NIN][NOU] f32, as in memory
    for (int f = tid; f < NIN * NOU; f += blockDim.x) Wl[f] = p.W[f];
    __syncthreads();
    const int ch0 = li * CH;
    float c_bias[CH], c_scale[CH], c_shift[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
        c_bias[t] = p.bias ? p.bias[ch0 + t] : 0.f;
        c_scale[t] = p.pscale ? p.pscale[ch0 + t] : 1.f;
        c_shift[t] = p.pscale ? p.pshift[ch0 + t] : 0.f;
    }
    const int M = d.M;
    const int nwaves = gridDim.x * p.waves;
    for (int b = blockIdx.x * p.waves + wave; b < d.B; b += nwaves) {
        float xv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            xv[i] = __uint_as_float((unsigned)p.x[(int64_t)b * d.x_sb + (int64_t)(lane + 64 * i) * d.x_sc] << 16);
        float P[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) P[t] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[i]), c));
                const float* wr = Wl + (64 * i + c) * NOU + ch0;
#pragma unroll
                for (int t = 0; t < CH; t += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + t);
                    P[t] = fmaf(s, w4[0], P[t]); P[t + 1] = fmaf(s, w4[1], P[t + 1]);
                    P[t + 2] = fmaf(s, w4[2], P[t + 2]); P[t + 3] = fmaf(s, w4[3], P[t + 3]);
                }
            }
        // the projected row is rounded to bf16 before the edge weight, like the P image of mpconv_fwd_b16.hip
#pragma unroll
        for (int t = 0; t < CH; ++t) P[t] = __uint_as_float((unsigned)fh_bf16(P[t]) << 16);
        const uint16_t* eb = p.et + (int64_t)b * d.et_sb;
        uint16_t* yb = p.y + (int64_t)b * d.y_sb + ch0;
        uint8_t* ab = p.argmax ? p.argmax + (int64_t)b * d.y_sb + ch0 : nullptr;
        for (int m = lk; m < M; m += 4) {
            const float e = __uint_as_float((unsigned)eb[(int64_t)m * d.et_sm] << 16);
            float r[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                float res = (e * P[t] + c_bias[t]) * c_scale[t] + c_shift[t];
                if (d.relu) res = fmaxf(res, 0.f);
                r[t] = res;
            }
            uint16_t* dst = yb + (int64_t)m * d.y_sm;
#pragma unroll
            for (int t = 0; t < CH; t += 4)
                *reinterpret_cast<uint2*>(dst + t) = make_uint2(fh_pack2(r[t], r[t + 1]), fh_pack2(r[t + 2], r[t + 3]));
            if (ab) {
#pragma unroll
                for (int t = 0; t < CH; t += 4) *reinterpret_cast<unsigned*>(ab + (int64_t)m * d.y_sm + t) = 0u;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define FH_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] hyper-edge forward rejects shape: rule %d\n", code); return 0; } while (0)

template <int AGG>
static void* fh_pick_fanin(int KS2, int OT) {
#define FH_CASE(ks, ot) if (KS2 == ks && OT == ot) return (void*)mpconv_fwd_fanin_kernel<AGG, ks, ot>;
    FH_CASE(2, 4) FH_CASE(2, 8) FH_CASE(4, 4)
#undef FH_CASE
    return nullptr;
}

// Returns 1 if launched, 0 if the call is not a bf16 hyper-edge call, <0 on error.
int fgnn_mpconv_forward_hyper(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                              const float* filters, const float* bias, const float* post_scale,
                              const float* post_shift, void* y, uint8_t* argmax, fgnn_stream_t stream) {
    static const bool off = getenv("FGNN_NO_FWD_HYPER") != nullptr;
    if (off) return 0;
    if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->net != 1) FH_REJECT(1);
    if ((d->nin != 64 && d->nin != 128) || (d->nou != 64 && d->nou != 128) || d->nin * d->nou > 64 * 128) FH_REJECT(2);
    const bool fanin = d->M == 1 && d->k >= 1 && d->k <= 255 && d->N >= 1 && d->N <= 128;
    const bool fanout = !fanin && d->N == 1 && d->k == 1;
    if (!fanin && !fanout) FH_REJECT(3);
    FhParams p;
    p.d = *d;
    p.x = (const uint16_t*)x; p.idx = nn_idx; p.et = (const uint16_t*)etype; p.W = filters; p.bias = bias;
    p.pscale = post_scale; p.pshift = post_shift; p.y = (uint16_t*)y; p.argmax = argmax;
    p.Npad16 = fgnn_round_up(d->N, 16);
    void* fn = nullptr;
    int lds = 0, waves = 8;
    if (fanin) {
        if (!(d->x_sc == 1 && d->x_sn == d->nin && d->x_sb % 8 == 0) || ((uintptr_t)x & 15)) FH_REJECT(4);
        const int KS2 = d->nin / 32, OT = d->nou / 16;
        static const bool no_id = getenv("FGNN_NO_FANIN_ID") != nullptr;
        // the caller vouches for an identity neighbour list (FGNN_DESC_IDENTITY_LIST): the reduction stays in the MFMA accumulators
        if (!no_id && (d->reserved & FGNN_DESC_IDENTITY_LIST) && d->agg == FGNN_AGG_MAX && d->k == d->N && d->N <= 128 && d->N >= 2) {
            void* idfn = nullptr;
#define FI_CASE(ks, ot) if (KS2 == ks && OT == ot) idfn = argmax ? (void*)mpconv_fwd_fanin_id_kernel<ks, ot, true> : (void*)mpconv_fwd_fanin_id_kernel<ks, ot, false>;
            FI_CASE(2, 4) FI_CASE(2, 8) FI_CASE(4, 4)
#undef FI_CASE
            if (idfn) {
                int g = (d->B + 3) / 4;
                if (g > 1024) g = 1024;
                const int idlds = 4 * 16 * (d->nou + FI_ROWPAD) * 4 + 4 * 16 * d->nou;
                fgnn_note_kernel("mpconv_fwd_fanin_id_kernel<%d, %d>", KS2, OT);
                void* args[] = {(void*)&p};
                hipError_t e = hipLaunchKernel(idfn, dim3(g), dim3(256), args, idlds, (hipStream_t)stream);
                if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv hyper-edge forward launch: %s", hipGetErrorString(e));
                return 1;
            }
        }
        fn = d->agg == FGNN_AGG_MAX ? fh_pick_fanin<FGNN_AGG_MAX>(KS2, OT)
           : d->agg == FGNN_AGG_LSE ? fh_pick_fanin<FGNN_AGG_LSE>(KS2, OT) : fh_pick_fanin<FGNN_AGG_MEAN>(KS2, OT);
        const int per_wave = p.Npad16 * d->nou * 2;
        while (waves > 1 && waves * per_wave > 64 * 1024) waves /= 2;      // keep >= 2 workgroups per CU
        lds = waves * per_wave;
    } else {
        if (!(d->y_sc == 1 && d->y_sm % 4 == 0 && d->y_sb % 4 == 0 && d->y_sm >= d->nou) || ((uintptr_t)y & 7) ||
            (argmax && ((uintptr_t)argmax & 3))) FH_REJECT(5);
        const int NI = d->nin / 64, CH = d->nou / 16;
        fn = NI == 1 ? (CH == 4 ? (void*)mpconv_fwd_fanout_kernel<1, 4> : (void*)mpconv_fwd_fanout_kernel<1, 8>)
                     : (CH == 4 ? (void*)mpconv_fwd_fanout_kernel<2, 4> : nullptr);
        lds = d->nin * d->nou * 4;
    }
    if (!fn) FH_REJECT(6);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    p.waves = waves;
    int grid = (d->B + waves - 1) / waves;
    if (grid > 1024) grid = 1024;
    if (fanin) fgnn_note_kernel("mpconv_fwd_fanin_kernel<%d, %d, %d>", d->agg, d->nin / 32, d->nou / 16);
    else fgnn_note_kernel("mpconv_fwd_fanout_kernel<%d, %d>", d->nin / 64, d->nou / 16);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(64 * waves), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv hyper-edge forward launch: %s", hipGetErrorString(e));
    return 1;
}
