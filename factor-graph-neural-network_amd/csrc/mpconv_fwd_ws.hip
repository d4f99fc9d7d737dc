// mpconv_fwd_ws.hip — third-generation bf16 forward of the VF/FV message operator for the LDPC parity-check calls
// (NO_EXTENSION, 4 edge types, max aggregation, degree 3 / 6, graph shared by the batch; reference:
// /root/reference/lib/model/mpnn/mp_nn.py:115-134,160-175).  Same maths and rounding points as mpconv_fwd_sg.hip (x, etype,
// filters and the projected rows P are bf16, every sum is f32, the output is rounded to bf16 once).
//
// What was wrong with the second generation (profiles/r03/pmc_fwd_*.json): every wave ran projection -> barrier -> gather
// -> barrier for ONE sample; waves waited 44 % of their cycles and issued 27 %.  Per sample and CU the resources it needs are
// ~770 matrix-pipe cycles, ~750 VALU cycles and ~2 300 LDS cycles (of which 580 were the edge-type BROADCAST reads and 540
// the x image's write + eight-fold operand re-read) against a byte bound of ~1 900 cycles — it took 5 700–6 600.
//
// This kernel is wave-specialised and software-pipelined over samples (one 1024-thread workgroup per CU, ONE barrier per
// sample):
//   * waves 0-3 (one per SIMD) are PRODUCERS: they project sample t+1 on the matrix cores (v_mfma_f32_32x32x16_bf16, 64
//     columns of W resident per wave, so the x image is re-read 4x instead of 8x) into one of two P images, and they are
//     the only waves that load: x and the edge types of sample t+XBUF go global -> LDS by LDS-DMA (global_load_lds_dwordx4:
//     no staging registers, no ds_write), XBUF samples ahead.  Producers never store, so their vmcnt only counts DMAs.
//   * waves 4-15 are CONSUMERS: they gather sample t from the other P image.  A lane owns FOUR channels of one of FOUR
//     destinations (16 lanes x 4 channels = a destination's 64): per edge a wave reads 4 P rows (two ds_read_b128 per lane)
//     and 4 edge-type rows with ONE ds_read_b64 — the second generation's lane = channel layout spent an 8-cycle broadcast
//     read per edge pair on them —, and stores 8 bytes of y + 4 bytes of argmax per lane instead of 2 + 1.
//   * the 16 lanes that read one P row are exactly a service group of ds_read_b128 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH
//     LDS table), so a group always reads 256 contiguous bytes: conflict-free for ANY neighbour table.  P rows are 512 B,
//     un-padded; the producers' 16-byte stores (8 consecutive nodes per service group) are made conflict-free by XOR-ing
//     the chunk index with (node & 7), which the consumers' precomputed row addresses absorb for free.
//   * the x image is a linear copy (what LDS-DMA can write); the 128-byte row stride would put the MFMA operand reads of
//     32 nodes on 2 bank groups, so the DMA's per-lane SOURCE addresses apply the XOR swizzle (chunk ^ (node >> 1 & 7))
//     instead and the readers undo it.
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ws_f32x16 __attribute__((ext_vector_type(16)));

#define WS_THREADS 1024
#define WS_NPROD 4
#define WS_NCONS 12
#define WS_MAXN 96
#define WS_PROW 512                     // P row: 32 chunks of 16 B = 2 channels x 4 edge types
#define WS_ESZ 3072                     // one edge-type buffer: 3 DMA pieces (M k 8 B <= 3072)
#define WS_EPIECES 3

enum { WS_MODE_TRAIN = 0, WS_MODE_TRAIN_STATS = 1, WS_MODE_AFFINE_RELU = 2, WS_MODE_GENERIC = 3 };

struct WsParams {
    const unsigned short* x;
    const int64_t* idx;
    const unsigned short* et;
    const float* W;
    const float* bias;
    const float* pscale;
    const float* pshift;
    unsigned short* y;
    uint8_t* argmax;
    float* stats;
    const unsigned short* add[3];                 // MODE AFFINE_RELU only: up to three tensors of y's layout added after the activation (or NULL)
    int B, N, M, Npad, relu;
    int y_ld, w_ld, st_ld;                        // row strides (elements) of y / argmax, of W, of a statistics partial row
    long long x_sb, et_sb, y_sb;                  // elements
    long long* prof;                              // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): stage timeline
    FgnnFold fold;                                // fold.tickets != NULL: the last workgroup finalises the BatchNorm statistics (fgnn_gridfold.h)
    fgnn_bn_final fin;
};

// stage-timeline stamps (tuning aid): workgroup 0, waves 0 (producer) and 4 (consumer), one stamp per barrier
#ifdef FGNN_ENABLE_PROF
#define WS_STAMP(slot) do { const int sl_ = (slot) - 28; if (p.prof && blockIdx.x == 0 && lane == 0 && sl_ >= 0 && sl_ < 8) p.prof[wave * 8 + sl_] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP(slot) do { } while (0)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];

__device__ __forceinline__ unsigned ws_pack(float a, float b) {
    ws_bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float ws_dot2(unsigned p, unsigned e, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ws_bf16x2, p), __builtin_bit_cast(ws_bf16x2, e), acc, false);
}
#ifndef WS_OPT_ASMDOT
#define WS_OPT_ASMDOT 0       // 1: seed each message with the three-address v_dot2_f32_bf16 (no v_mov of the bias per message)
#endif
#ifndef WS_OPT_MASKARG
#define WS_OPT_MASKARG 0      // argmax through lane masks in SGPRs (first-occurrence logic on the scalar unit) instead of a v_cmp -> s_nop -> v_cndmask chain
#endif
__device__ __forceinline__ float ws_dot2_seed(unsigned p, unsigned e, float c) {
#if WS_OPT_ASMDOT
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(e), "v"(c));
    return r;
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ws_bf16x2, p), __builtin_bit_cast(ws_bf16x2, e), c, false);
#endif
}
__device__ __forceinline__ float ws_max3(float a, float b, float c) {
    // (not inline asm: a VALU read of a v_dot2c result needs wait states the compiler only inserts for instructions it can see)
    return fmaxf(fmaxf(a, b), c);
}
// one LDS-DMA piece: 64 lanes x 16 bytes, lane l's bytes land at lds_dst + 16 l (tools/ubench/lds_dma_tr.hip).  M0 is
// compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7).  Not counted by the compiler's
// s_waitcnt bookkeeping: the producers wait with ws_wait_dma<>.
__device__ __forceinline__ void ws_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void ws_wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// Dataflow flags in LDS (monotonic counters, one ds_add per wave and sample) instead of a workgroup barrier per sample: a
// barrier makes every wave wait for the slowest of all 16 once per sample (measured: the three consumer waves of a SIMD
// finish ~1 900 / 2 900 / 3 500 cycles into a 3 660-cycle stage, the producers at 2 500); with flags a wave only waits
// for the data it needs.  LDS executes one wave's operations in order, so a ds_add issued after a wave's ds_write /
// ds_read of an image is ordered behind them without a wait.
// (ds instructions through inline asm: a volatile access to a computed LDS address compiles to flat_load + s_waitcnt vmcnt(0),
// i.e. every poll would drain the wave's stores.  The compiler does not count these two in its lgkmcnt bookkeeping; LDS
// returns in order, so an uncounted extra operation only makes its own waits conservative.)
__device__ __forceinline__ void ws_signal(unsigned flag, int lane) {
    if (lane == 0) {
        const unsigned one = 1u;
        asm volatile("ds_add_u32 %0, %1" :: "v"(flag), "v"(one) : "memory");
    }
}
__device__ __forceinline__ void ws_wait_ge(unsigned flag, int need) {
    for (;;) {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= need) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// lane <-> (destination slot s, 16-byte column q) such that the 16 lanes of one ds_read_b128 service group share s
__device__ __forceinline__ void ws_lane_sq(int lane, int& s, int& q) {
    const int h = lane >> 5, l5 = lane & 31;
    const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    s = 2 * h + (g1 ? 1 : 0);
    if (!g1) q = l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12);
    else q = l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16);
}
__device__ __forceinline__ int ws_lane_of(int s, int q) {
    const int h = s >> 1;
    int l5;
    if (!(s & 1)) l5 = q < 4 ? q : (q < 8 ? q + 8 : q + 12);
    else l5 = q < 8 ? q + 4 : (q < 12 ? q + 8 : q + 16);
    return 32 * h + l5;
}

// NIN in {64, 128}; KC = degree (3 or 6); XBUF = x buffers = DMA depth in samples (3, or 2 when 128-channel rows leave no room)
template <int NIN, int KC, int MODE, int XBUF>
__global__ __launch_bounds__(WS_THREADS) void mpconv_fwd_ws_kernel(const WsParams p) {
    constexpr int KS = NIN / 16;                      // MFMA k-steps
    constexpr int C8 = NIN / 8;                       // 16-byte chunks per x row
    constexpr int XROW = NIN * 2;
    constexpr int XBYTES = WS_MAXN * XROW;            // one x buffer (always 96 rows: the DMA pieces cover all of it)
    constexpr int XP = WS_MAXN * C8 / 64 / WS_NPROD;  // x DMA pieces per producer wave and sample: 3 / 6
    constexpr int EBUF = XBUF + 1;
    constexpr int NDMA = XP + 1;                      // DMA instructions per producer wave and sample (x pieces + one edge-type piece)
    constexpr int MAXG = KC == 6 ? 1 : 2;             // destination groups (of 4) per consumer wave: M <= 48 / 96
    constexpr bool WANT_ARG = MODE == WS_MODE_TRAIN || MODE == WS_MODE_TRAIN_STATS || MODE == WS_MODE_GENERIC;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int N = p.N, M = p.M, Npad = p.Npad;
    const int PBYTES = Npad * WS_PROW;
    const int OFF_E = XBUF * XBYTES, OFF_P = OFF_E + EBUF * WS_ESZ;      // DMA targets first: all below 64 KB
    // flags: [0] x / edge types of a sample landed (4 per sample), [1] P image written (4 per sample), [2] / [3] P image 0 / 1
    // consumed (12 per sample of that parity).  A cumulative count is exact only while no wave can run more than ONE signal ahead
    // of the slowest: true for the producers (flag 0 gates them together every sample), not for the consumers, which may be a
    // whole sample ahead of a slow wave — one "consumed" counter over all samples let eleven fast waves' signals for sample t + 1
    // stand in for the twelfth wave's for sample t, and the producers overwrote the image (and edge-type buffer) it was still
    // reading (seen as run-to-run differences in the last destinations of a sample, tools/diag_epilogue.py).  Per image parity
    // the count is exact: nobody can start the next sample of a parity before the producers have rewritten that image.
    const unsigned lds0 = (unsigned)(uintptr_t)ws_lds;
    int* flags_p = reinterpret_cast<int*>(ws_lds + OFF_P + 2 * PBYTES);
    const unsigned flags = lds0 + (unsigned)(OFF_P + 2 * PBYTES);        // LDS byte address: + 0 / 4 / 8
    const int grid = gridDim.x;
    const int cnt = (p.B - (int)blockIdx.x + grid - 1) / grid;       // samples of this workgroup: blockIdx.x + i grid
    if (tid < 4) flags_p[tid] = 0;
    __syncthreads();

    if (wave < WS_NPROD) {
        // =====================================================================================  producers
        const int pw = wave, l31 = lane & 31, lh = lane >> 5;
        // W^T A-fragments of this wave's 64 columns = channels 16 pw .. 16 pw + 15.  Row i = 8 g + 4 h + r of column tile tc is
        // channel 4 (4 pw + 2 tc + h) + g, edge type r: the D fragment of an output lane (node, h) is then the 32 bytes
        // {channels 4 q .. 4 q + 3} x {4 edge types} of q = 4 pw + 2 tc + h — one consumer lane's reads.
        ws_bf16x8 areg[2][KS];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int g = l31 >> 3, h = (l31 >> 2) & 1, r = l31 & 3;
            const float* wc = p.W + (4 * (4 * pw + 2 * tc + h) + g) * 4 + r;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                unsigned w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = 16 * kk + 8 * lh + 2 * u;
                    w[u] = ws_pack(wc[(size_t)c * p.w_ld], wc[(size_t)(c + 1) * p.w_ld]);
                }
                areg[tc][kk] = __builtin_bit_cast(ws_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
        // x image reads: node row l31 (+ 32 per tile), chunk 2 kk + lh, stored at chunk ^ fx(row)
        const int fx = NIN == 64 ? (l31 >> 1) & 7 : l31 & 15;
        unsigned xoff[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) xoff[kk] = (unsigned)(l31 * XROW + (((2 * kk + lh) ^ fx) << 4));
        // P image writes: node row l31 (+ 32 per tile), chunks q and 16 + q at chunk ^ (row & 7)
        unsigned pwo[2];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) pwo[tc] = (unsigned)(l31 * WS_PROW + (((4 * pw + 2 * tc + lh) ^ (l31 & 7)) << 4));
        // DMA sources.  Piece u of this wave fills LDS slots 64 (pw XP + u) + lane of the x buffer; slot -> (row, position),
        // position holds source chunk position ^ fx(row); rows >= N re-read row N - 1 (never gathered).
        unsigned dsrc[XP];
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int slot = 64 * (pw * XP + u) + lane;
            const int row = slot / C8, pos = slot - row * C8;
            const int f = NIN == 64 ? (row >> 1) & 7 : row & 15;
            dsrc[u] = (unsigned)(min(row, N - 1) * XROW + ((pos ^ f) << 4));
        }
        const int ebytes = M * KC * 8;                                    // a multiple of 16 (host)
        const unsigned esrc = (unsigned)min((pw * 64 + lane) * 16, ebytes - 16);
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(p.x);
        const unsigned char* eg = reinterpret_cast<const unsigned char*>(p.et);
        auto dma_batch = [&](int i) {                                     // sample blockIdx.x + i grid -> x buffer i % XBUF, edge-type buffer i % EBUF
            const long long b = (long long)blockIdx.x + (long long)i * grid;
            const unsigned char* xb = xg + b * p.x_sb * 2;
            const unsigned xdst = lds0 + (unsigned)((i % XBUF) * XBYTES + pw * XP * 1024);
#pragma unroll
            for (int u = 0; u < XP; ++u) ws_dma16(xb + dsrc[u], xdst + u * 1024);
            // (wave 3 has no edge-type piece; it re-fetches piece 2 into the same place so that every wave counts NDMA)
            const int ep = pw < WS_EPIECES ? pw : WS_EPIECES - 1;
            const unsigned es = pw < WS_EPIECES ? esrc : (unsigned)min((ep * 64 + lane) * 16, ebytes - 16);
            ws_dma16(eg + b * p.et_sb * 2 + es, lds0 + (unsigned)(OFF_E + (i % EBUF) * WS_ESZ + ep * 1024));
        };
        const int ntile = Npad / 32;
        auto project = [&](int i, int par) {                              // sample i: x buffer i % XBUF -> P image `par`
            const unsigned char* xs = ws_lds + (i % XBUF) * XBYTES;
            unsigned char* ps = ws_lds + OFF_P + par * PBYTES;
            auto store_tile = [&](unsigned char* pp, const ws_f32x16& acc) {
                *reinterpret_cast<uint4*>(pp) = make_uint4(ws_pack(acc[0], acc[1]), ws_pack(acc[2], acc[3]),
                                                           ws_pack(acc[4], acc[5]), ws_pack(acc[6], acc[7]));
                *reinterpret_cast<uint4*>(pp + 256) = make_uint4(ws_pack(acc[8], acc[9]), ws_pack(acc[10], acc[11]),
                                                                 ws_pack(acc[12], acc[13]), ws_pack(acc[14], acc[15]));
            };
            if constexpr (KS == 4) {
                // Two accumulator chains (the wave's two column tiles) alternate on the matrix pipe — a chain alone stalls 64 cycles
                // per link — and the next node tile's operand fragments are in flight under this tile's MFMAs.
                ws_bf16x8 bfr[2][4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    bfr[0][kk] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const uint4*>(xs + xoff[kk]));
#pragma unroll
                for (int tile = 0; tile < 3; ++tile) {
                    if (tile < ntile) {
                        if (tile + 1 < ntile) {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                bfr[(tile + 1) & 1][kk] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const uint4*>(xs + (tile + 1) * 32 * XROW + xoff[kk]));
                        }
                        ws_f32x16 acc0, acc1;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[0][kk], bfr[tile & 1][kk], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[1][kk], bfr[tile & 1][kk], acc1, 0, 0, 0);
                        }
                        store_tile(ps + tile * 32 * WS_PROW + pwo[0], acc0);
                        store_tile(ps + tile * 32 * WS_PROW + pwo[1], acc1);
                    }
                }
            } else {
                // nin = 128: eight fragments + 64 resident A registers + two accumulators do not fit 128 VGPRs, so each column tile
                // re-reads its operand fragments four at a time
#pragma unroll
                for (int tile = 0; tile < 3; ++tile) {
                    if (tile < ntile) {
#pragma unroll
                        for (int tc = 0; tc < 2; ++tc) {
                            ws_f32x16 acc;
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                            for (int grp = 0; grp < KS / 4; ++grp) {
                                ws_bf16x8 bfr[4];
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
                                    bfr[kk] = __builtin_bit_cast(ws_bf16x8, *reinterpret_cast<const uint4*>(xs + tile * 32 * XROW + xoff[grp * 4 + kk]));
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[tc][grp * 4 + kk], bfr[kk], acc, 0, 0, 0);
                            }
                            store_tile(ps + tile * 32 * WS_PROW + pwo[tc], acc);
                        }
                    }
                }
            }
        };
        // ---- the first XBUF samples' inputs go out at once; from then on sample t - 1 + XBUF is requested when sample t's projection starts
        WS_STAMP(0);
#pragma unroll
        for (int i = 0; i < XBUF; ++i)
            if (i < cnt) dma_batch(i);
        for (int t = 0; t < cnt; ++t) {
            // this wave's pieces of sample t have landed when at most the younger batches (t + 1 .. t + XBUF - 2) are outstanding
            if (XBUF > 2 && t + 1 < cnt) ws_wait_dma<(XBUF - 2) * NDMA>();
            else ws_wait_dma<0>();
            ws_signal(flags + 0, lane);
            if (t >= 1) {
                // P image t & 1 held sample t - 2, edge-type buffer (t - 1 + XBUF) % EBUF sample t - 2: consumers done with it
                ws_wait_ge(flags + 8 + 4 * (t & 1), WS_NCONS * (t >> 1));
                if (t - 1 + XBUF < cnt) {
                    ws_wait_ge(flags + 4, WS_NPROD * t);                  // every producer is done reading x buffer (t - 1) % XBUF
                    dma_batch(t - 1 + XBUF);
                }
            }
            ws_wait_ge(flags + 0, WS_NPROD * (t + 1));                    // all four waves' pieces of sample t
            WS_STAMP(4 + 3 * t);
            project(t, t & 1);
            ws_signal(flags + 4, lane);
            WS_STAMP(5 + 3 * t);
        }
        ws_wait_dma<0>();
    } else {
        // =====================================================================================  consumers
        const int cw = wave - WS_NPROD;
        int s, q;
        ws_lane_sq(lane, s, q);
        // LDS byte offsets (from the P image's base) of the first chunk of every P row this lane reads
        unsigned addr[MAXG][KC];
        unsigned eoff[MAXG];                           // byte offset of the destination's edge-type rows in an edge-type buffer
        unsigned yoff[MAXG];                           // element offset of this lane's 4 channels in a sample's y / argmax
        bool valid[MAXG];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int m = (cw + WS_NCONS * g) * 4 + s;
            valid[g] = m < M;
            const int mm = min(m, M - 1);
            const int64_t* ib = p.idx + (int64_t)mm * KC;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                long long v = ib[j];
                v = v < 0 ? 0 : (v >= N ? N - 1 : v);                  // never read outside the image
                const unsigned n = (unsigned)v;
                addr[g][j] = n * WS_PROW + ((q ^ (n & 7u)) << 4);
                asm volatile("" : "+v"(addr[g][j]));                    // (keep it a per-lane register: see mpconv_fwd_sg.hip)
            }
            eoff[g] = (unsigned)(mm * KC * 8);
            yoff[g] = (unsigned)(mm * p.y_ld + 4 * q);
        }
        float c_bias[4], c_scale[4], c_shift[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * q + i;
            c_bias[i] = p.bias ? p.bias[o] : 0.f;
            c_scale[i] = (MODE >= WS_MODE_AFFINE_RELU && p.pscale) ? p.pscale[o] : 1.f;
            c_shift[i] = (MODE >= WS_MODE_AFFINE_RELU && p.pscale) ? p.pshift[o] : 0.f;
        }
        float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};

        auto gather = [&](int i, auto par_c) {                            // sample i out of P image PAR
            constexpr int PAR = decltype(par_c)::value;
            const long long b = (long long)blockIdx.x + (long long)i * grid;
            const unsigned char* pim = ws_lds + OFF_P + PAR * PBYTES;
            const unsigned char* eim = ws_lds + OFF_E + (i % EBUF) * WS_ESZ;
            unsigned short* yb = p.y + b * p.y_sb;
            uint8_t* ab = (WANT_ARG && p.argmax) ? p.argmax + b * p.y_sb : nullptr;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                if ((cw + WS_NCONS * g) * 4 < M) {                        // wave-uniform
                    uint4 pa[KC], pb[KC];
                    uint2 ev[KC];
                    uint2 av[3] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)};
                    if constexpr (MODE == WS_MODE_AFFINE_RELU) {          // the caller's addends: requested first, consumed last
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            if (p.add[a] && valid[g]) av[a] = *reinterpret_cast<const uint2*>(p.add[a] + b * p.y_sb + yoff[g]);
                    }
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const unsigned char* pr = pim + addr[g][j];
                        pa[j] = *reinterpret_cast<const uint4*>(pr);
                        pb[j] = *reinterpret_cast<const uint4*>(pr + 256);
                        ev[j] = *reinterpret_cast<const uint2*>(eim + eoff[g] + j * 8);
                    }
                    // the four channels' chains advance together (edge-major): the compare -> select chain of the argmax is serial
                    // per channel, and left channel-major the compiler pads every link with s_nop instead of another channel's work
                    float v[4][KC];
#pragma unroll
                    for (int j = 0; j < KC; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned w0 = c == 0 ? pa[j].x : (c == 1 ? pa[j].z : (c == 2 ? pb[j].x : pb[j].z));
                            const unsigned w1 = c == 0 ? pa[j].y : (c == 1 ? pa[j].w : (c == 2 ? pb[j].y : pb[j].w));
                            v[c][j] = ws_dot2(w1, ev[j].y, ws_dot2_seed(w0, ev[j].x, c_bias[c]));      // the bias rides in the accumulator
                        }
                    float best[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        best[c] = ws_max3(v[c][0], v[c][1], v[c][2]);
                        if constexpr (KC == 6) best[c] = fmaxf(ws_max3(best[c], v[c][3], v[c][4]), v[c][KC - 1]);
                    }
                    unsigned args = 0u;
                    if (WANT_ARG) {                                       // first occurrence of the maximum (torch.max on CPU)
#if WS_OPT_MASKARG
                        // eq_j = lanes whose message j equals the maximum (v_cmp into an SGPR pair); "first such j" is resolved on
                        // the scalar unit, and the three bits of the index come back as three selects per channel
                        typedef unsigned long long u64;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            u64 none = ~0ull, b0 = 0, b1 = 0, b2 = 0;       // none: no earlier message equals the maximum
#pragma unroll
                            for (int j = 0; j < KC - 1; ++j) {
                                const u64 first = __builtin_amdgcn_fcmpf(v[c][j], best[c], 1 /* oeq */) & none;
                                none &= ~first;
                                if (j & 1) b0 |= first;
                                if (j & 2) b1 |= first;
                                if (j & 4) b2 |= first;
                            }
                            if ((KC - 1) & 1) b0 |= none;                    // nothing before the last slot: it is the last slot
                            if ((KC - 1) & 2) b1 |= none;
                            if ((KC - 1) & 4) b2 |= none;
                            const unsigned u1 = 1u << (8 * c), u2 = 2u << (8 * c), u4 = 4u << (8 * c);
                            args |= (__builtin_amdgcn_inverse_ballot_w64(b0) ? u1 : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b1) ? u2 : 0u);
                            if (KC > 4) args |= __builtin_amdgcn_inverse_ballot_w64(b2) ? u4 : 0u;
                        }
#else
                        int arg[4] = {KC - 1, KC - 1, KC - 1, KC - 1};
#pragma unroll
                        for (int j = KC - 2; j >= 0; --j)
#pragma unroll
                            for (int c = 0; c < 4; ++c) arg[c] = v[c][j] == best[c] ? j : arg[c];
                        args = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
#endif
                    }
                    float res[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float r = best[c];
                        if (MODE >= WS_MODE_AFFINE_RELU) r = fmaf(r, c_scale[c], c_shift[c]);
                        if (MODE == WS_MODE_AFFINE_RELU || (MODE == WS_MODE_GENERIC && p.relu)) r = fmaxf(r, 0.f);
                        res[c] = r;
                    }
                    if constexpr (MODE == WS_MODE_AFFINE_RELU) {
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            if (p.add[a]) {
                                res[0] += __uint_as_float(av[a].x << 16); res[1] += __uint_as_float(av[a].x & 0xffff0000u);
                                res[2] += __uint_as_float(av[a].y << 16); res[3] += __uint_as_float(av[a].y & 0xffff0000u);
                            }
                    }
                    const uint2 packed = make_uint2(ws_pack(res[0], res[1]), ws_pack(res[2], res[3]));
                    if (MODE == WS_MODE_TRAIN_STATS && valid[g]) {        // of the values as stored
                        const float z0 = __uint_as_float(packed.x << 16), z1 = __uint_as_float(packed.x & 0xffff0000u);
                        const float z2 = __uint_as_float(packed.y << 16), z3 = __uint_as_float(packed.y & 0xffff0000u);
                        st0[0] += z0; st1[0] = fmaf(z0, z0, st1[0]);
                        st0[1] += z1; st1[1] = fmaf(z1, z1, st1[1]);
                        st0[2] += z2; st1[2] = fmaf(z2, z2, st1[2]);
                        st0[3] += z3; st1[3] = fmaf(z3, z3, st1[3]);
                    }
                    if (valid[g]) {
                        *reinterpret_cast<uint2*>(yb + yoff[g]) = packed;
                        if (WANT_ARG && ab) *reinterpret_cast<unsigned*>(ab + yoff[g]) = args;
                    }
                }
            }
        };

        for (int t = 0; t < cnt; t += 2) {                                // sample t out of P image t & 1
            ws_wait_ge(flags + 4, WS_NPROD * (t + 1));
            WS_STAMP(4 + 3 * t);
            gather(t, std::integral_constant<int, 0>());
            ws_signal(flags + 8, lane);
            WS_STAMP(5 + 3 * t);
            if (t + 1 < cnt) {
                ws_wait_ge(flags + 4, WS_NPROD * (t + 2));
                WS_STAMP(4 + 3 * (t + 1));
                gather(t + 1, std::integral_constant<int, 1>());
                ws_signal(flags + 12, lane);
                WS_STAMP(5 + 3 * (t + 1));
            }
        }

        if (MODE == WS_MODE_TRAIN_STATS && p.stats) {
            // this lane's sums over its destinations: [12 waves][64 lanes][8] floats in LDS, folded below in a fixed order
            float* red = reinterpret_cast<float*>(ws_lds);
            float* mine = red + (cw * 64 + lane) * 8;
            *reinterpret_cast<f32x4*>(mine) = (f32x4){st0[0], st0[1], st0[2], st0[3]};
            *reinterpret_cast<f32x4*>(mine + 4) = (f32x4){st1[0], st1[1], st1[2], st1[3]};
        }
    }

    if (MODE == WS_MODE_TRAIN_STATS && p.stats) {
        // BatchNorm statistics epilogue: one partial row (sum, sum of squares per channel) per workgroup
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(ws_lds);
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63, q = c >> 2, i = c & 3;
            float sum = 0.f;
            for (int w = 0; w < WS_NCONS; ++w)
#pragma unroll
                for (int s = 0; s < 4; ++s) sum += red[(w * 64 + ws_lane_of(s, q)) * 8 + which * 4 + i];
            fgnn_fold_store(p.stats + ((int64_t)blockIdx.x * 2 + which) * p.st_ld + c, sum);
        }
        if (p.fold.tickets) {
            double* sums = reinterpret_cast<double*>(ws_lds + 32768);     // (past the fold's own [12][64][8] floats; the images are dead)
            if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) fgnn_bn_final_apply(p.fin, 64, sums);
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
template <int NIN, int KC, int XBUF>
static void* ws_pick_mode(int mode) {
#define WS_CASE(m) \
    if (mode == m) return (void*)mpconv_fwd_ws_kernel<NIN, KC, m, XBUF>;
    WS_CASE(WS_MODE_TRAIN) WS_CASE(WS_MODE_TRAIN_STATS) WS_CASE(WS_MODE_AFFINE_RELU) WS_CASE(WS_MODE_GENERIC)
#undef WS_CASE
    return nullptr;
}

// Same contract as fgnn_mpconv_forward_sg (mpconv_fwd_sg.hip), which calls this first: 1 = launched, 0 = shape outside this
// kernel's family (the second-generation kernel takes it), < 0 = error.  grid_out: the number of statistics partial rows.
// Addends of the NEXT inference-mode launch on this thread (fgnn_mpconv_forward_addends sets them around its call of the ordinary
// entry point; `taken` tells it whether this kernel consumed them or the caller has to add them itself).
static thread_local const void* ws_pending_add[3] = {nullptr, nullptr, nullptr};
static thread_local int ws_pending_taken = 0;
void fgnn_ws_set_pending_addends(const void* a0, const void* a1, const void* a2) {
    ws_pending_add[0] = a0; ws_pending_add[1] = a1; ws_pending_add[2] = a2; ws_pending_taken = 0;
}
int fgnn_ws_pending_addends_taken(void) { return ws_pending_taken; }

void fgnn_stats_pending(const fgnn_bn_final** fin, void** scratch);
void fgnn_stats_upper_half(FgnnFold* fold, fgnn_bn_final* fin);

int fgnn_mpconv_forward_ws(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                           const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                           void* y, uint8_t* argmax, fgnn_stream_t stream, float* stats, int* plan_grid, int mode, int split) {
    static const bool off = getenv("FGNN_NO_WS") != nullptr;
    if (off) return 0;
    const int KC = d->k;
    if (d->N > WS_MAXN || d->M > (KC == 6 ? 48 : 96)) return 0;
    if ((d->M * KC) & 1) return 0;                                    // a sample's edge types: whole 16-byte DMA lanes
    if (d->M * KC * 8 > WS_ESZ || d->M * KC * 8 < 16) return 0;
    if ((d->et_sb % 8) != 0 || (d->y_sb % 4) != 0) return 0;
    if (x && ((((uintptr_t)etype) & 15) || (((uintptr_t)y) & 7) || (argmax && (((uintptr_t)argmax) & 3)))) return 0;
    const int Npad = fgnn_round_up(d->N, 32);
    const int xbuf = d->nin == 64 ? 3 : 2;
    const int lds = xbuf * WS_MAXN * d->nin * 2 + 2 * Npad * WS_PROW + (xbuf + 1) * WS_ESZ + 64;      // + the dataflow flags
    if (lds > 160 * 1024) return 0;
    void* fn = nullptr;
    if (d->nin == 64) fn = KC == 6 ? ws_pick_mode<64, 6, 3>(mode) : ws_pick_mode<64, 3, 3>(mode);
    // (nin = 128: 64 resident A registers per producer; what spills under 128 VGPRs is the fragment-loading prologue only — none in the sample loops)
    else if (d->nin == 128) fn = KC == 6 ? ws_pick_mode<128, 6, 2>(mode) : ws_pick_mode<128, 3, 2>(mode);
    if (!fn) return 0;
    int grid = 256;      // one workgroup per CU
    if (grid > d->B) grid = d->B;
    if (plan_grid) { *plan_grid = grid; return 1; }
    WsParams p = {};
    p.x = static_cast<const unsigned short*>(x); p.idx = nn_idx; p.et = static_cast<const unsigned short*>(etype);
    p.W = filters; p.bias = bias; p.pscale = post_scale; p.pshift = post_shift;
    p.y = static_cast<unsigned short*>(y); p.argmax = argmax; p.stats = stats;
    p.B = d->B; p.N = d->N; p.M = d->M; p.Npad = Npad; p.relu = d->relu;
    p.x_sb = d->x_sb; p.et_sb = d->et_sb; p.y_sb = d->y_sb;
    p.y_ld = d->nou; p.w_ld = d->nou * 4; p.st_ld = d->nou;
    {   // the BatchNorm behind the operator, finalised by this launch (fgnn_mpconv_forward_stats set it for this call)
        const fgnn_bn_final* fin = nullptr;
        void* scratch = nullptr;
        fgnn_stats_pending(&fin, &scratch);
        p.fold = fgnn_fold_make(stats, (stats && fin) ? scratch : nullptr, grid, 64, 2 * d->nou, d->nou);
        if (fin) p.fin = *fin;
    }
    for (int a = 0; a < 3; ++a) p.add[a] = nullptr;
    if (mode == WS_MODE_AFFINE_RELU && ws_pending_add[0]) {
        bool ok = true;
        for (int a = 0; a < 3; ++a) ok = ok && !(((uintptr_t)ws_pending_add[a]) & 7);
        if (ok) {
            for (int a = 0; a < 3; ++a) p.add[a] = static_cast<const unsigned short*>(ws_pending_add[a]);
            ws_pending_taken = 1;
        }
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    fgnn_note_kernel(split ? "mpconv_fwd_ws_kernel<%d, %d, %d, %d> x2" : "mpconv_fwd_ws_kernel<%d, %d, %d, %d>", d->nin, KC, mode, xbuf);
    p.prof = nullptr;
#ifdef FGNN_ENABLE_PROF
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 128 * 8);
        (void)hipMemset(prof_buf, 0, 128 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(WS_THREADS), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ws forward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {                                     // tuning aid: per-stage timeline of workgroup 0 (shader clocks)
        long long h[128];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 16; ++w) {                 // slots: 0 / 1 = sample 8 started / finished, 3 / 4 = sample 9, 6 / 7 = sample 10
            fprintf(stderr, "[fgnn prof ws fwd] wave %2d:", w);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %6lld", h[w * 8 + i] ? h[w * 8 + i] - h[0] : -1);
            fprintf(stderr, "\n");
        }
        p.prof = nullptr;
    }
#endif
    if (split) {                                                      // the upper 64 output channels of a 64 -> 128 call
        p.W += 256; p.y += 64;
        for (int a = 0; a < 3; ++a)
            if (p.add[a]) p.add[a] += 64;
        if (p.bias) p.bias += 64;
        if (p.pscale) { p.pscale += 64; p.pshift += 64; }
        if (p.argmax) p.argmax += 64;
        if (p.stats) p.stats += 64;
        fgnn_stats_upper_half(&p.fold, &p.fin);
        e = hipLaunchKernel(fn, dim3(grid), dim3(WS_THREADS), args, lds, (hipStream_t)stream);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ws forward launch (upper half): %s", hipGetErrorString(e));
    }
    return 1;
}
