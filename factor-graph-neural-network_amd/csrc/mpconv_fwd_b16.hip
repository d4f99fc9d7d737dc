// mpconv_fwd_b16.hip — bf16 forward of the VF/FV message operator for the LDPC shape family: bf16
// storage AND bf16 matrix cores (v_mfma_f32_16x16x32_bf16, f32 accumulate) — the HBM-bound regime of
// BASELINE config 3 (AI ~80 FLOP/B << bf16 ridge ~312), where the goal is bytes/s, not FLOP/s.
//
// Same math as mpconv_fwd.hip (reference: /root/reference/lib/model/mpnn/mp_nn.py:115-134); rounding
// points: x, etype, filters and the projected rows P are bf16 (P is rounded once before the gather),
// every sum is f32, the output is rounded to bf16 once.
//
//   * 512-thread workgroups (8 waves), 2 per CU (<= 80 KB LDS, <= 128 VGPRs: 4 waves per SIMD hide the
//     LDS latency of the gather), persistent over samples;
//   * filters live in registers as bf16 A-fragments for the kernel's lifetime (W-stationary);
//   * x is channel-fastest in HBM, so a sample is one dense block moved with 16-byte loads into a
//     padded LDS image whose rows are MFMA B-fragments (ds_read_b128, conflict-free);
//   * P[N, 256] sits in LDS as bf16 (half the LDS traffic of the f32 kernel); the gather reads 8 bytes
//     per (edge, channel), lanes = channels, destinations walk across the 4 waves;
//   * next sample's x / etype / nn_idx are prefetched into registers during the current sample.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#ifndef B16_XPAD
#define B16_XPAD 16
#endif
#ifndef B16_PPAD
#define B16_PPAD 16
#endif
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define B16_THREADS 512
#define B16_WAVES 8
#define B16_XPT 3       // 16-byte x chunks per thread     (nin*N/8   <= 512*3)
#define B16_EPT 3       // etype scalars per thread        (M*k*net   <= 512*3)
#define B16_IPT 1       // nn_idx entries per thread       (M*k       <= 512)
#define B16_PASS_COLS 256

struct B16Params {
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
    float* stats;                         // [grid][2][nou] per-workgroup (sum, sum of squares) of the stored output, or NULL
    int Npad, pass_cols;
    int XSB, PSB;                         // LDS row strides in BYTES (x image, P image)
    int c8shift;                          // log2(nin/8)
    int et_mode;
    unsigned mkmagic;
    int dbg;                              // FGNN_DBG ablation mask (tuning only)
    long long* prof;                      // FGNN_PROF: phase timeline of one sample of workgroup 0 (tuning only)
    int JP;                               // neighbour-list split over waves (1 = off)
    int off_xs, off_ps, off_idx, off_et, off_red;  // byte offsets into LDS
    FgnnFold fold;                        // fold.tickets != NULL: the last workgroup finalises the BatchNorm statistics (fgnn_gridfold.h)
    fgnn_bn_final fin;
};

extern __shared__ __attribute__((aligned(16))) unsigned char fgnn_lds_h[];

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    // native fptrunc -> one v_cvt_pk_bf16_f32 (round-to-nearest-even).  NOT inline asm: an asm
    // statement reading MFMA results directly gets none of the compiler's MFMA->VALU wait states
    // (observed: one node tile of stale P values once the accumulators stopped living in AGPRs).
    bf16x2_t r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}

// message value of one (edge, channel): sum_e etype[e] * P[n][o*net+e], P in bf16
template <int NET>
__device__ __forceinline__ float b16_dot(const float* __restrict__ etp, const unsigned char* __restrict__ prow) {
    if constexpr (NET == 1) {
        const unsigned short u = *reinterpret_cast<const unsigned short*>(prow);
        return etp[0] * __uint_as_float(((unsigned)u) << 16);
    } else {
        const uint2 pk = *reinterpret_cast<const uint2*>(prow);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(etp);
        float v = e4[0] * bf16_lo(pk.x);
        v = fmaf(e4[1], bf16_hi(pk.x), v);
        v = fmaf(e4[2], bf16_lo(pk.y), v);
        v = fmaf(e4[3], bf16_hi(pk.y), v);
        return v;
    }
}

// phase-timeline stamps (tuning aid): compiled in only with -DFGNN_ENABLE_PROF, read with FGNN_PROF=1
#ifdef FGNN_ENABLE_PROF
#define B16_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && tid == 0 && b == 2 * (int)gridDim.x) p.prof[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define B16_STAMP(slot) do { } while (0)
#endif

// KSB = nin/32 MFMA k-steps, SWP = column slabs per wave per pass, NPASS = column passes of <= 256,
// KC = neighbours per destination when known at compile time (3 / 6: the LDPC degrees), 0 = runtime k
template <int NET, int AGG, int KSB, int SWP, int NPASS, int KC>
__global__ __launch_bounds__(B16_THREADS) void mpconv_fwd_b16_kernel(const B16Params p) {
    const fgnn_mpconv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar index math
    const int lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nin = d.nin, nou = d.nou, N = d.N, M = d.M;
    const int k = KC > 0 ? KC : d.k;
    constexpr int net = NET;
    const int ncols = nou * net;
    const int mk = M * k;
    const int XSB = p.XSB, PSB = p.PSB;

    unsigned char* xs = fgnn_lds_h + p.off_xs;        // [Npad][XSB]  bf16 x, node-major
    unsigned char* ps = fgnn_lds_h + p.off_ps;        // [Npad][PSB]  bf16 P of the current pass
    int* idx_s = reinterpret_cast<int*>(fgnn_lds_h + p.off_idx);
    unsigned short* et_h = reinterpret_cast<unsigned short*>(fgnn_lds_h + p.off_et);   // [mk][net] bf16
    float* red = reinterpret_cast<float*>(fgnn_lds_h + p.off_red);                      // JP partials

    const unsigned short* xg = static_cast<const unsigned short*>(p.x);
    const unsigned short* etg = static_cast<const unsigned short*>(p.et);
    unsigned short* yg = static_cast<unsigned short*>(p.y);

    // ---- filters -> bf16 A fragments: areg[pass][q][kk] = W[c = 32kk + 8lk + 0..7][col] ----
    const int slabs_per_pass = p.pass_cols / 16;
    bf16x8_t areg[NPASS][SWP][KSB];
#pragma unroll
    for (int ps_i = 0; ps_i < NPASS; ++ps_i)
#pragma unroll
        for (int q = 0; q < SWP; ++q) {
            const int slab = wave + B16_WAVES * q;
            const int col = ps_i * p.pass_cols + slab * 16 + li;
            const bool ok = slab < slabs_per_pass && col < ncols;
#pragma unroll
            for (int kk = 0; kk < KSB; ++kk) {
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int c = 32 * kk + 8 * lk + 2 * h;
                    const float f0 = (ok && c < nin) ? p.W[(int64_t)c * ncols + col] : 0.f;
                    const float f1 = (ok && c + 1 < nin) ? p.W[(int64_t)(c + 1) * ncols + col] : 0.f;
                    w[h] = pack_bf16(f0, f1);
                }
                areg[ps_i][q][kk] = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
            }
        }

    // ---- prefetch registers ----
    uint4 xr[B16_XPT];
    unsigned short er[B16_EPT];
    long long ir[B16_IPT];                            // raw: clamped at the commit, so the load is not waited for here
    const int xchunks = (nin * N) >> 3;               // 16-byte chunks of the dense sample block
    auto prefetch = [&](int b, int t) {
        const uint4* xb = reinterpret_cast<const uint4*>(xg + (int64_t)b * d.x_sb);
#pragma unroll
        for (int q = 0; q < B16_XPT; ++q) {
            const int f = t + q * B16_THREADS;
            xr[q] = f < xchunks ? xb[f] : make_uint4(0, 0, 0, 0);
        }
        const unsigned short* eb = etg + (int64_t)b * d.et_sb;
#pragma unroll
        for (int q = 0; q < B16_EPT; ++q) {
            const int f = t + q * B16_THREADS;
            er[q] = f < mk * net ? eb[f] : (unsigned short)0;
        }
#pragma unroll
        for (int q = 0; q < B16_IPT; ++q) {
            const int f = t + q * B16_THREADS;
            long long v = 0;
            if (f < mk) {
                const int m = f / k, j = f - m * k;
                v = (p.idx + (int64_t)b * d.idx_sb)[(int64_t)m * d.idx_sm + (int64_t)j * d.idx_sk];
            }
            ir[q] = v;
        }
    };
    auto commit = [&](int t) {
#pragma unroll
        for (int q = 0; q < B16_XPT; ++q) {
            const int f = t + q * B16_THREADS;      // chunk f = (n, c8) = (f >> c8shift, f & mask)
            if (f < xchunks) {
                const int n = f >> p.c8shift, c8 = f - (n << p.c8shift);
                *reinterpret_cast<uint4*>(xs + n * XSB + c8 * 16) = xr[q];
            }
        }
#pragma unroll
        for (int q = 0; q < B16_EPT; ++q) {
            const unsigned f = t + q * B16_THREADS;
            if ((int)f < mk * net) {
                if (p.et_mode == 1 || net == 1) {
                    et_h[f] = er[q];
                } else {
                    const unsigned e = __umulhi(f, p.mkmagic), r = f - e * mk;
                    et_h[r * net + e] = er[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < B16_IPT; ++q) {
            const int f = t + q * B16_THREADS;
            if (f < mk) { const long long w = ir[q]; idx_s[f] = (int)(w < 0 ? 0 : (w >= N ? N - 1 : w)); }
        }
    };

    // zero the padded rows of the x image once (n >= N)
    for (int f = tid; f < (p.Npad - N) * (XSB / 4); f += B16_THREADS)
        reinterpret_cast<unsigned*>(xs + N * XSB)[f] = 0u;

    int b = blockIdx.x;
    if (b < d.B) prefetch(b, tid);
    const int ntile = p.Npad / 16;
    // per-lane epilogue constants of the channel block last finished: survive across samples, so a call whose
    // lanes always finish the same channel (one pass, <= 64 channels per pass) loads them once
    int cc_cached = -1;
    float c_bias = 0.f, c_scale = 1.f, c_shift = 0.f;
    float st0 = 0.f, st1 = 0.f;          // BatchNorm statistics of this lane's channel (stats runs: one channel block)

    for (; b < d.B; b += gridDim.x) {
        int t = tid;
        asm volatile("" : "+v"(t));          // opaque per sample: no cross-iteration hoisting
        B16_STAMP(0);
        commit(t);
        B16_STAMP(1);
        __syncthreads();
        B16_STAMP(2);
        if (b + (int)gridDim.x < d.B && !(p.dbg & 4)) prefetch(b + gridDim.x, t);
        B16_STAMP(3);
        unsigned short* yb = yg + (int64_t)b * d.y_sb;
        uint8_t* ab = p.argmax ? p.argmax + (int64_t)b * d.y_sb : nullptr;

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int o0 = pass * p.pass_cols / net;
            const int otc = min(p.pass_cols / net, nou - o0);
            B16_STAMP(4 + 4 * pass);
            // ---- projection: P^T tile = W^T (cols x nin) . x (nin x nodes), bf16 MFMA, f32 accumulate ----
            for (int tile = 0; tile < ((p.dbg & 1) ? 0 : ntile); ++tile) {
                f32x4 acc[SWP];
#pragma unroll
                for (int q = 0; q < SWP; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const unsigned char* bp = xs + (tile * 16 + li) * XSB + lk * 16;
#pragma unroll
                for (int kk = 0; kk < KSB; ++kk) {
                    const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bp + kk * 64));
#pragma unroll
                    for (int q = 0; q < SWP; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[pass][q][kk], bfrag, acc[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < SWP; ++q) {
                    const int slab = wave + B16_WAVES * q;
                    if (slab < slabs_per_pass) {
                        uint2 pk;
                        pk.x = pack_bf16(acc[q][0], acc[q][1]);
                        pk.y = pack_bf16(acc[q][2], acc[q][3]);
                        *reinterpret_cast<uint2*>(ps + (tile * 16 + li) * PSB + (slab * 16 + 4 * lk) * 2) = pk;
                    }
                }
            }
            __syncthreads();
            B16_STAMP(5 + 4 * pass);

            // ---- gather + edge-type contraction + aggregation ----
            // One wave per (destination m, block of 64 channels), lane = channel.  The neighbour ids and
            // the edge-type weights of m are wave-uniform: ONE LDS read each per block of up to JB
            // neighbours (lane j holds entry j) and v_readlane broadcasts them to scalar registers, so the
            // only per-edge LDS traffic is the 8-byte P row slice per lane; the contraction is two
            // v_dot2c_f32_bf16 per edge with the weights as scalar operands.  Few, high-degree
            // destinations (the degree-96 hyper-factor) split their neighbour list over JP waves and
            // combine the partial results through LDS.
            {
                constexpr int JB = NET == 4 ? 32 : 64;          // neighbours whose ids + weights fit one lane each
                const int nch = (otc + 63) >> 6;
                const int JP = p.JP;
                const int jchunk = (k + JP - 1) / JP;
                const unsigned* et_w = reinterpret_cast<const unsigned*>(et_h);
                auto finish = [&](int m, int cc, int ch, float a, float bsum, int arg) {
                    float res;
                    if constexpr (AGG == FGNN_AGG_MAX) res = a;
                    else if constexpr (AGG == FGNN_AGG_LSE) res = (1.0f / 3.0f) * (a + logf(bsum));
                    else res = bsum / (float)k;
                    if (pass * 8 + cc != cc_cached) { // per-lane channel constants: reload only when the block changes
                        cc_cached = pass * 8 + cc;
                        const int o = o0 + ch;
                        c_bias = p.bias ? p.bias[o] : 0.f;
                        c_scale = p.pscale ? p.pscale[o] : 1.f;
                        c_shift = p.pscale ? p.pshift[o] : 0.f;
                    }
                    res = (res + c_bias) * c_scale + c_shift;
                    if (d.relu) res = fmaxf(res, 0.f);
                    const int off = (o0 + ch) * (int)d.y_sc + m * (int)d.y_sm;
                    const unsigned packed = pack_bf16(res, 0.f);
                    if (!(p.dbg & 8) || res == 1.2345e-30f) yb[off] = (unsigned short)packed;
                    if (p.stats) { const float zr = bf16_lo(packed); st0 += zr; st1 = fmaf(zr, zr, st1); }     // of the value as stored
                    if (AGG == FGNN_AGG_MAX && ab) ab[off] = (uint8_t)arg;
                };
                bool paired = false;
                if constexpr (KC > 0 && NET == 4 && AGG == FGNN_AGG_MAX) {
                    // LDPC parity calls (fixed degree, <= 64 channels per pass): two destinations per wave at a
                    // time, so that the id / weight reads of one overlap the row reads of the other
                    if (JP == 1 && nch == 1) {
                        paired = true;
                        constexpr int UN = 2;                        // destinations in flight per wave (3-4 measured no better)
                        const bool active = lane < otc;
                        const unsigned char* pc = ps + (active ? lane : 0) * 8;
                        for (int m0 = wave; m0 < ((p.dbg & 2) ? 0 : M); m0 += UN * B16_WAVES) {
                            int id[UN];
                            unsigned ew[UN];
#pragma unroll
                            for (int u = 0; u < UN; ++u) {
                                const int m = m0 + u * B16_WAVES < M ? m0 + u * B16_WAVES : m0;
                                id[u] = lane < KC ? idx_s[m * KC + lane] : 0;
                                ew[u] = lane < 2 * KC ? et_w[m * KC * 2 + lane] : 0u;
                            }
                            uint2 pk[UN][KC];
#pragma unroll
                            for (int u = 0; u < UN; ++u)
#pragma unroll
                                for (int j = 0; j < KC; ++j)
                                    pk[u][j] = *reinterpret_cast<const uint2*>(pc + __builtin_amdgcn_readlane(id[u], j) * PSB);
#pragma unroll
                            for (int u = 0; u < UN; ++u) {
                                float best = 0.f;
                                int arg = 0;
#pragma unroll
                                for (int j = 0; j < KC; ++j) {
                                    float v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk[u][j].x),
                                        __builtin_bit_cast(bf16x2_t, (unsigned)__builtin_amdgcn_readlane(ew[u], 2 * j)), 0.f, false);
                                    v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk[u][j].y),
                                        __builtin_bit_cast(bf16x2_t, (unsigned)__builtin_amdgcn_readlane(ew[u], 2 * j + 1)), v, false);
                                    if (j == 0 || v > best) { best = v; arg = j; }      // strict >: first occurrence
                                }
                                if (active && m0 + u * B16_WAVES < M) finish(m0 + u * B16_WAVES, 0, lane, best, 0.f, arg);
                            }
                        }
                    }
                }
                for (int u = wave; u < ((p.dbg & 2) || paired ? 0 : M * nch * JP); u += B16_WAVES) {
                    const int part = u % JP, base = u / JP;
                    const int m = base / nch, cc = base - m * nch;
                    const int ch = cc * 64 + lane;
                    const bool active = ch < otc;
                    const unsigned char* pc = ps + (active ? ch : 0) * (net * 2);
                    const int jlo = KC > 0 ? 0 : part * jchunk;
                    const int jhi = KC > 0 ? KC : min(k, jlo + jchunk);
                    float best = 0.f, mx = -INFINITY, ssum = 0.f;
                    int arg = jlo;
                    for (int jb = jlo; jb < jhi; jb += JB) {
                        const int nb = min(JB, jhi - jb);
                        const int idxv = lane < nb ? idx_s[m * k + jb + lane] : 0;
                        unsigned etv;
                        if constexpr (NET == 4) etv = lane < 2 * nb ? et_w[(m * k + jb) * 2 + lane] : 0u;
                        else etv = lane < nb ? (unsigned)et_h[m * k + jb + lane] << 16 : 0u;
                        auto load_p = [&](int j, uint2& pk) {
                            const int n = __builtin_amdgcn_readlane(idxv, j);
                            if constexpr (NET == 4) pk = *reinterpret_cast<const uint2*>(pc + n * PSB);
                            else pk.x = *reinterpret_cast<const unsigned short*>(pc + n * PSB);
                        };
                        auto consume = [&](int j, const uint2& pk) {
                            float v;
                            if constexpr (NET == 4) {
                                const unsigned e01 = __builtin_amdgcn_readlane(etv, 2 * j);
                                const unsigned e23 = __builtin_amdgcn_readlane(etv, 2 * j + 1);
                                v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk.x),
                                                                    __builtin_bit_cast(bf16x2_t, e01), 0.f, false);
                                v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk.y),
                                                                    __builtin_bit_cast(bf16x2_t, e23), v, false);
                            } else {
                                const float e = __uint_as_float(__builtin_amdgcn_readlane(etv, j));
                                v = e * __uint_as_float(pk.x << 16);
                            }
                            if constexpr (AGG == FGNN_AGG_MAX) {
                                if (jb + j == jlo || v > best) { best = v; arg = jb + j; }   // strict >: first occurrence
                            } else if constexpr (AGG == FGNN_AGG_LSE) {
                                v *= 3.0f;
                                if (v > mx) { ssum = ssum * expf(mx - v) + 1.0f; mx = v; }
                                else ssum += expf(v - mx);
                            } else {
                                ssum += v;
                            }
                        };
                        if constexpr (KC > 0) {                // all KC row slices in flight, then reduce
                            uint2 pk[KC];
#pragma unroll
                            for (int j = 0; j < KC; ++j) load_p(j, pk[j]);
#pragma unroll
                            for (int j = 0; j < KC; ++j) consume(j, pk[j]);
                        } else {
                            int j = 0;
                            for (; j + 3 <= nb; j += 3) {
                                uint2 p0, p1, p2;
                                load_p(j, p0); load_p(j + 1, p1); load_p(j + 2, p2);
                                consume(j, p0); consume(j + 1, p1); consume(j + 2, p2);
                            }
                            for (; j < nb; ++j) {
                                uint2 p0;
                                load_p(j, p0);
                                consume(j, p0);
                            }
                        }
                    }
                    const float a = AGG == FGNN_AGG_MAX ? best : mx;
                    if (JP == 1) {
                        if (active) finish(m, cc, ch, a, ssum, arg);
                    } else {                                   // partial of this wave -> LDS
                        float* r3 = red + ((base * JP + part) * 64 + lane) * 3;
                        r3[0] = jlo < jhi ? a : -INFINITY;
                        r3[1] = ssum;
                        r3[2] = __int_as_float(arg);
                    }
                }
                if (JP > 1) {
                    __syncthreads();
                    for (int base = wave; base < M * nch; base += B16_WAVES) {
                        const int m = base / nch, cc = base - m * nch;
                        const int ch = cc * 64 + lane;
                        const float* r3 = red + (base * JP * 64 + lane) * 3;
                        float a = r3[0], bsum = r3[1];
                        int arg = __float_as_int(r3[2]);
                        for (int q = 1; q < JP; ++q) {
                            if (q * jchunk >= k) break;
                            const float a2 = r3[q * 192], b2 = r3[q * 192 + 1];
                            if constexpr (AGG == FGNN_AGG_MAX) {
                                if (a2 > a) { a = a2; arg = __float_as_int(r3[q * 192 + 2]); }   // later part: strictly larger only
                            } else if constexpr (AGG == FGNN_AGG_LSE) {
                                const float mxx = fmaxf(a, a2);
                                bsum = bsum * expf(a - mxx) + b2 * expf(a2 - mxx);
                                a = mxx;
                            } else {
                                bsum += b2;
                            }
                        }
                        if (ch < otc) finish(m, cc, ch, a, bsum, arg);
                    }
                }
            }
            B16_STAMP(6 + 4 * pass);
            __syncthreads();          // P (and, after the last pass, xs / et_s / idx_s) may be rewritten
            B16_STAMP(7 + 4 * pass);
        }
    }
    if (p.stats) {
        // BatchNorm statistics epilogue (host: one pass, <= 64 output channels, lane <-> channel): fold the 8 waves'
        // per-channel sums in a fixed order and leave this workgroup's partial row for fgnn_bn_finalize
        float* red = reinterpret_cast<float*>(ps);      // P image is free: the loop ended on a barrier
        red[(wave * 2) * 64 + lane] = st0;
        red[(wave * 2 + 1) * 64 + lane] = st1;
        __syncthreads();
        if (tid < 128) {
            const int c = tid & 63, which = tid >> 6;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < B16_WAVES; ++w) sum += red[(w * 2 + which) * 64 + c];
            if (c < nou) fgnn_fold_store(p.stats + ((int64_t)blockIdx.x * 2 + which) * nou + c, sum);
        }
        if (p.fold.tickets) {
            double* sums = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(ps) + B16_WAVES * 2 * 64 * 4);   // (past the fold's own floats)
            if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) fgnn_bn_final_apply(p.fin, nou, sums);
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
template <int NET, int AGG, int KC>
static void* b16_pick_shape(int KSB, int SWP, int NPASS) {
#define B16_CASE(ks, swp, np) \
    if (KSB == ks && SWP == swp && NPASS == np) return (void*)mpconv_fwd_b16_kernel<NET, AGG, ks, swp, np, KC>;
    B16_CASE(2, 1, 1) B16_CASE(2, 2, 1) B16_CASE(2, 2, 2)
    B16_CASE(4, 1, 1) B16_CASE(4, 2, 1)
#undef B16_CASE
    return nullptr;
}
template <int NET>
static void* b16_pick_agg(int agg, int k, int KSB, int SWP, int NPASS) {
    switch (agg) {
        case FGNN_AGG_MAX:
            if (NET == 4 && k == 3) return b16_pick_shape<NET, FGNN_AGG_MAX, (NET == 4 ? 3 : 0)>(KSB, SWP, NPASS);
            if (NET == 4 && k == 6) return b16_pick_shape<NET, FGNN_AGG_MAX, (NET == 4 ? 6 : 0)>(KSB, SWP, NPASS);
            return b16_pick_shape<NET, FGNN_AGG_MAX, 0>(KSB, SWP, NPASS);
        case FGNN_AGG_LSE: return b16_pick_shape<NET, FGNN_AGG_LSE, 0>(KSB, SWP, NPASS);
        default: return b16_pick_shape<NET, FGNN_AGG_MEAN, 0>(KSB, SWP, NPASS);
    }
}

int fgnn_mpconv_forward_sg(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                           const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                           void* y, uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid);

// Returns 1 if launched, 0 if the shape is outside this kernel's family, <0 on error.
// stats != NULL: also write per-workgroup BatchNorm partials (only one-pass shapes with <= 64 output channels: returns 0
// otherwise).  plan_grid != NULL: no launch, *plan_grid = the grid (= number of partial rows) the launch would use.
int fgnn_mpconv_forward_b16(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                            const void* etype, const float* filters, const float* bias,
                            const float* post_scale, const float* post_shift, void* y,
                            uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid) {
    if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE) return 0;
    if (d->net != 1 && d->net != 4) return 0;
    {   // batch-shared graph, fixed degree, max aggregation: the second-generation kernel (mpconv_fwd_sg.hip)
        const int rc = fgnn_mpconv_forward_sg(d, x, nn_idx, etype, filters, bias, post_scale, post_shift, y, argmax, stream,
                                              stats, plan_grid);
        if (rc != 0) return rc;
    }
    const int ncols = d->nou * d->net;
    if (ncols % 16 != 0 || ncols > 512) return 0;
    if (d->nin != 64 && d->nin != 128) return 0;
    // x: dense channel-fastest sample block, 16-byte aligned; y: channel-fastest
    if (!(d->x_sc == 1 && (d->x_sn == d->nin || d->N == 1)) || (d->x_sb % 8) != 0) return 0;
    if (!(d->y_sc == 1 && (d->M == 1 || d->y_sm == d->nou))) return 0;
    const int Npad = fgnn_round_up(d->N, 16);
    const int mk = d->M * d->k;
    if ((d->nin * d->N) / 8 > B16_THREADS * B16_XPT) return 0;
    if (mk * d->net > B16_THREADS * B16_EPT || mk > B16_THREADS * B16_IPT) return 0;
    int et_mode;
    if (d->net == 1) {
        if (!((d->et_sk == 1 || d->k == 1) && (d->et_sm == d->k || d->M == 1))) return 0;
        et_mode = 1;
    } else if (mk == 1) {
        if (d->et_se != 1) return 0;
        et_mode = 1;
    } else if ((d->et_sk == 1 || d->k == 1) && (d->et_sm == d->k || d->M == 1) && d->et_se == mk) {
        et_mode = 0;
    } else if (d->et_se == 1 && d->et_sk == d->net && (d->et_sm == d->k * d->net || d->M == 1)) {
        et_mode = 1;
    } else {
        return 0;
    }
    B16Params p = {};
    p.d = *d;
    p.x = x; p.idx = nn_idx; p.et = etype; p.W = filters; p.bias = bias;
    p.pscale = post_scale; p.pshift = post_shift; p.y = y; p.argmax = argmax; p.stats = stats;
    p.Npad = Npad;
    const int NPASS = (ncols + B16_PASS_COLS - 1) / B16_PASS_COLS;
    p.pass_cols = NPASS == 1 ? ncols : B16_PASS_COLS;
    if (NPASS > 1 && ncols % B16_PASS_COLS != 0) return 0;
    if ((stats || plan_grid) && (NPASS != 1 || d->nou > 64 || d->M * ((d->nou + 63) / 64) * 2 <= B16_WAVES)) return 0;
    const int slabs_per_pass = p.pass_cols / 16;
    const int SWP = (slabs_per_pass + B16_WAVES - 1) / B16_WAVES;
    const int KSB = d->nin / 32;
    p.XSB = d->nin * 2 + B16_XPAD;                       // + padding: rows land on distinct 16-byte bank groups
    p.PSB = p.pass_cols * 2 + B16_PPAD;
    p.c8shift = d->nin == 64 ? 3 : 4;
    p.et_mode = et_mode;
    { const char* e = getenv("FGNN_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.prof = nullptr;
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 64 * 8);
        (void)hipMemset(prof_buf, 0, 64 * 8);
        p.prof = prof_buf;
    }
    p.mkmagic = mk == 1 ? 0u : (unsigned)((0x100000000ULL + mk - 1) / mk);
    int off = 0;
    p.off_xs = off;  off += Npad * p.XSB;                off = fgnn_round_up(off, 16);
    p.off_ps = off;  off += Npad * p.PSB;                off = fgnn_round_up(off, 16);
    p.off_idx = off; off += fgnn_round_up(mk, 4) * 4;
    p.off_et = off;  off += fgnn_round_up(mk * d->net * 2, 16);
    // few high-degree destinations: split each neighbour list over JP waves
    {
        const int otp = p.pass_cols / d->net;
        const int units = d->M * ((otp < d->nou ? otp : d->nou) + 63) / 64;
        p.JP = 1;
        if (d->k >= 16 && units * 2 <= B16_WAVES) p.JP = B16_WAVES / units;
        p.off_red = off;
        if (p.JP > 1) off += units * p.JP * 64 * 3 * 4;
    }
    const int lds = off;
    if (lds > 160 * 1024) return 0;
    void* fn = d->net == 1 ? b16_pick_agg<1>(d->agg, d->k, KSB, SWP, NPASS)
                           : b16_pick_agg<4>(d->agg, d->k, KSB, SWP, NPASS);
    if (!fn) return 0;
    int wg_per_cu = (160 * 1024) / lds;
    if (wg_per_cu > 4) wg_per_cu = 4;
    if (wg_per_cu < 1) wg_per_cu = 1;
    int grid = 256 * wg_per_cu;
    if (grid > d->B) grid = d->B;
    if (plan_grid) { *plan_grid = grid; return 1; }
    {   // the BatchNorm behind the operator, finalised by this launch (fgnn_mpconv_forward_stats set it for this call)
        const fgnn_bn_final* fin = nullptr;
        void* scratch = nullptr;
        fgnn_stats_pending(&fin, &scratch);
        p.fold = fgnn_fold_make(stats, (stats && fin) ? scratch : nullptr, grid, d->nou);
        if (fin) p.fin = *fin;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel("mpconv_fwd_b16_kernel<%d, %d, %d, %d, %d, %d>", d->net, d->agg, KSB, SWP, NPASS,
                     (d->agg == FGNN_AGG_MAX && d->net == 4 && (d->k == 3 || d->k == 6)) ? d->k : 0);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(B16_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv bf16 forward launch: %s", hipGetErrorString(e));
    if (p.prof) {                                     // tuning aid: phase timeline of one sample (shader clocks)
        long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[fgnn prof fwd]");
        for (int i = 0; i < 20; ++i) if (h[i]) fprintf(stderr, " %d:%lld", i, h[i] - h[0]);
        fprintf(stderr, "\n");
    }
    return 1;
}
