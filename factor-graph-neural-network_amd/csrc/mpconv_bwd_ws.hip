// mpconv_bwd_ws.hip — third-generation backward of the VF/FV message operator for the LDPC parity-check calls: bf16
// channel-fastest x / gz / etype, 64 -> 64 channels, 4 edge types, max aggregation, degree 3 / 6, ONE neighbour table shared
// by the batch whose transposed incidence has in-degree <= 3 / 6.  Same maths and rounding points as mpconv_bwd_sg.hip
// (autograd through /root/reference/lib/model/mpnn/mp_nn.py:115-175):
//
//     G[(n,q),o]    = gz[m,o] [j == argmax[m,o]]   for the q-th in-edge (m, j) of source node n     (bf16, exact)
//     P[n,o,e]      = sum_c x[n,c] W[c,o*4+e]                                                      (recomputed, bf16)
//     detype[(n,q),e] = sum_o G[(n,q),o] P[n,o,e]
//     dP[n,o,e]     = sum_q G[(n,q),o] etype[(n,q),e]                                              (bf16)
//     dx[n,c]       = sum_col dP[n,col] W[c,col]      dW[c,col] += sum_n x[n,c] dP[n,col]      dbias[o] += sum_(n,q) G[(n,q),o]
//
// Why a third generation (profiles/r03/pmc_bwd_*.json, README): the second one kept the LDS busy 8 700 cycles per sample
// (46 % of the launch, a quarter of it bank conflicts) and issued 7 400 VALU + 2 250 SALU instructions per wave — the
// routing was done per (destination, channel lane) with a wave-private zero-restored image and an even/odd selector MFMA
// (36 LDS cycles per destination), the dP phase re-derived its masks per in-edge and lane, dW transposed its operands in
// registers.  Here the ROUTED gradient is materialised once per sample as the image G, rows sorted by SOURCE node, by
// the threads that stage gz / argmax anyway (3 packed-16-bit VALU ops per dword and slot, conflict-free 16-byte stores):
//   * detype becomes two v_mfma_f32_16x16x32_bf16 per four source nodes (A = 16 rows of G, B = the nodes' P rows kept
//     edge-type-major): 48 MFMAs and 96 ds_read_b128 per sample, no VALU;
//   * dP is one v_mfma_f32_4x4x4_16b_bf16 per node whose B operand is ONE ds_read_b64_tr_b16 of the node's G rows (the
//     hardware transpose gives lane = channel its four in-edges) — no masks, no per-edge reads; a second 4x4x4 with an
//     all-ones row accumulates dbias in the same pass;
//   * dW's node-contracted operands come straight out of the row-major x / dP images through ds_read_b64_tr_b16;
//   * projection, dx and dW use 32x32x16 tiles (half the operand traffic of 16x16x32); x arrives by LDS-DMA.
// LDS work per sample ~3 600 cycles (was 8 700), VALU ~1 200 wave-instructions per sample and CU (was ~9 000).
//
// One 512-thread workgroup per CU (8 waves x 256 VGPRs), four barriers per sample (the P and dP images share their LDS: 49 + 50 KB do not fit
// beside G's 49 KB):
//   1  waves 0-7 project sample s                                   (x image -> P, edge-type-major, XOR-swizzled)
//   2a all waves: detype MFMAs                                      (G, P -> getype staging)
//   2b all waves: dP per source node; staging -> getype             (G, etT -> dP image, row stride 528 B)
//   3  waves 0-1 dx, waves 2,3,6,7 dW (48 MFMAs per SIMD), waves 4-5 build G / etT of sample s+1 from registers prefetched in
//      phase 1 and wait for its x (LDS-DMA)
#include "fgnn_common.h"
#include <stdlib.h>

#define BW_THREADS 512       // 8 waves x 256 VGPRs: the roles' resident state (64 MFMA registers, 16 + 48 of the projection, ~60 of the staging) spills at 128
#define BW_WAVES 8
#define BW_MAXN 96
#define BW_XROW 128           // x image row: 64 bf16, linear (LDS-DMA), 16-byte chunks XOR-swizzled by bw_swz_r(node)
#define BW_GROW 128           // G image row (one in-edge slot of one source node): 64 bf16, chunks swizzled by bw_swz_r(row)
#define BW_PROW 512           // P image row: [4 edge types][64 channels] bf16, chunks swizzled per (node, edge type)
#define BW_DROW 528           // dP image row: [64 channels][4 edge types] bf16 + 16 (conflict-free b128 operand reads)

typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef short bw_s16x4 __attribute__((ext_vector_type(4)));
typedef float bw_f32x16 __attribute__((ext_vector_type(16)));

struct BwParams {
    const uint16_t* x;
    const int64_t* idx;
    const uint16_t* et;
    const float* W;          // [64][256]
    const uint16_t* gz;
    const uint8_t* argmax;
    uint16_t* gx;
    uint16_t* get;
    float* ws;               // per-workgroup slabs [grid][64*256 + 64]
    int B, N, M;
    int y_ld, w_ld, x_ld;        // row strides (elements): gz / argmax, W, and x / gx in MEMORY (64, or 128 when the call's 128 input channels
                                 // run as two launches over their halves)
    int accum;                   // bit 0: gx is ADDED to, bit 1: getype is ADDED to (the second launch of a two-launch call)
    long long x_sb, et_sb, y_sb;     // elements
    const int* tables;       // the transposed incidence, built ONCE per graph by mpconv_bwd_ws_tables_kernel, or NULL (the default): every
                             // workgroup builds its own — 17 400 of a launch's ~23 000 set-up cycles (profiles/r04), but NOT on its critical
                             // path: they overlap the first samples' LDS-DMA; with the tables 90.3 / 79.9 us, without 87.8 / 81.5 (gpurun_out/r05c)
    long long* prof;         // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): phase timeline
};

// phase-timeline stamps (tuning aid): workgroup 0, its fourth sample, every wave: slot k = arrival at / release from the barriers
#ifdef FGNN_ENABLE_PROF
#define BW_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && b == b_begin + 3) p.prof[wave * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define BW_STAMP_G(slot) do { if (p.prof && blockIdx.x == 0 && tid == 0 && (slot) < 128) p.prof[128 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define BW_STAMP(slot) do { } while (0)
#define BW_STAMP_G(slot) do { } while (0)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char bw_lds[];

void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);
void fgnn_launch_slab_reduce_ld(const float* ws, int nslab, int64_t slab_len, int64_t nw, int ncols, int ld, float* gW,
                                float* gbias, hipStream_t st);

__device__ __forceinline__ unsigned bw_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned bw_add2(unsigned a, unsigned b) {      // two bf16 sums, f32 arithmetic, rounded once
    return bw_pack2(__uint_as_float(a << 16) + __uint_as_float(b << 16), __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u));
}
__device__ __forceinline__ bw_bf16x8 bw_frag_f32(const float* p8) {      // 8 consecutive f32 -> one fragment
    const f32x4 a = *reinterpret_cast<const f32x4*>(p8), b = *reinterpret_cast<const f32x4*>(p8 + 4);
    return __builtin_bit_cast(bw_bf16x8, make_uint4(bw_pack2(a[0], a[1]), bw_pack2(a[2], a[3]),
                                                    bw_pack2(b[0], b[1]), bw_pack2(b[2], b[3])));
}
// LDS-DMA piece (64 lanes x 16 B -> lds_dst + 16 lane) and the waits the compiler cannot place (see mpconv_fwd_ws.hip)
__device__ __forceinline__ void bw_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with a UNIFORM 64-bit base (scalar registers) + an unsigned 32-bit per-lane byte offset.  Round 6: dma_x's three per-lane
// sources were carried across the sample loop as 64-bit register pairs, one pair was spilled in the V -> F instance, and its reload —
// a memory operation — waited (vmcnt(0)) behind the two pieces just requested: an HBM round trip on the dW waves at the tail of every
// sample's phase 3, the phase's long pole.
__device__ __forceinline__ void bw_dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void bw_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// phase barrier: this wave's LDS operations done, then s_barrier — NOT __syncthreads(), which would also drain the gx /
// getype stores and the next sample's prefetch loads four times per sample
__device__ __forceinline__ void bw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ds_read_b64_tr_b16: the 16 lanes of a group each name 4 consecutive bf16 of a [4 keys][16 columns] block (lane i: key i >> 2,
// columns 4 (i & 3) ..), lane c of the group receives column c of the four keys (tools/ubench/lds_dma_tr.hip)
__device__ __forceinline__ uint2 bw_tr(unsigned lds_addr) {
    typedef __attribute__((address_space(3))) bw_s16x4 lds_v4;
    const bw_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4*>(static_cast<uintptr_t>(lds_addr)));
    return __builtin_bit_cast(uint2, v);
}
// uniform 64-bit base + UNSIGNED 32-bit per-lane byte offset: the form that compiles to `global_load v, v_off, s[base]` (anything
// else becomes a per-lane 64-bit pointer, hoisted out of the sample loop and spilled)
template <typename T> __device__ __forceinline__ const T* bw_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ T* bw_at(void* base, unsigned byte_off) {
    return reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}
// Chunk swizzles of the 128-byte-row images (x, G) and of the dP image.  Each image is read two ways: 16-byte operand reads
// (ds_read_b128: the 16 lanes of a service group {0-3,12-15,20-27} / ... must land in 16 different 16-byte slots of a 256-byte
// window) and TRANSPOSE reads (ds_read_b64_tr_b16: a 32-lane pass fetches 4 consecutive rows x two adjacent 32-byte segments =
// eight 32-byte pieces that must land in the eight 32-byte bank groups).  (row >> 1) & 7 — the first form — satisfies the b128
// reads only: rows r and r + 2 of a transpose read met in the same banks (26-29 % of the kernel's LDS cycles were conflicts).
//   128-byte rows: s = (b1, b2^b3, b3) of the row index — bit 2 of s must come from b1 (rows r / r + 2 apart by two segments),
//     and s^-1(001) must lie in {001,110,111} (the b128 groups mix chunk g of rows {0-3,12-15} with chunk g^1 of rows {4-11});
//   The dP image keeps its 528-byte rows (b128-conflict-free; its transpose reads stay ~2-way conflicted): 512-byte rows with
//   chunks swizzled by (b1, b0, b3, b2) of the node make both read kinds conflict-free, but every dx fragment address then
//   needs its own XOR — 25 spilled registers at the 256-VGPR limit: 106 / 98 us against 89 / 84 (profiles/r04/README.md).
__device__ __forceinline__ int bw_swz_r(int r) { return (((r >> 1) & 1) << 2) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ int bw_swz_p(int n, int e) { return ((2 * (n & 3) + (e >> 1)) ^ ((n >> 2) & 1)) & 7; }

// LDS layout, fixed per instance (sized for the largest graph the instance takes: no shape-dependent scalar branches or
// offsets in the sample loop — as runtime values they cost ~100 live SGPRs, spilled to VGPR lanes, and a branch per node)
template <int KC> struct BwLayout {
    static constexpr int DEG = KC == 6 ? 3 : 6, QS = DEG <= 4 ? 4 : 8;
    static constexpr int NMAX = KC == 6 ? 96 : 64, MMAX = KC == 6 ? 48 : 96;
    static constexpr int OFF_X = 0;                                            // 3 x [96][128 B]: samples s, s + 1, s + 2
    static constexpr int OFF_G = OFF_X + 3 * BW_MAXN * BW_XROW;                // [NMAX][QS][128 B]
    static constexpr int OFF_PD = OFF_G + NMAX * QS * BW_GROW;                 // P [NMAX][512 B], then dP [NMAX][528 B]
    static constexpr int OFF_ET = OFF_PD + NMAX * BW_DROW;                     // etT [NMAX][4][QS] bf16
    static constexpr int OFF_GST = OFF_ET + NMAX * 4 * QS * 2;                 // getype staging [4][M KC] bf16
    static constexpr int OFF_TAB = OFF_GST + 8 * MMAX * KC + 16;               // [NMAX][QS] int (+ one all -1 row)
    static constexpr int OFF_SLOT = OFF_TAB + NMAX * QS * 4 + 16;              // [M KC] int
    static constexpr int OFF_GT = OFF_SLOT + MMAX * KC * 4;                    // [8 M][KC] unsigned
    static constexpr int OFF_DUMP = OFF_GT + 8 * MMAX * KC * 4;                // 16 bytes nobody reads (pieces of in-edges beyond DEG)
    static constexpr int BYTES = OFF_DUMP + 16;
};

// The transposed incidence of a batch-shared neighbour table, in LDS: tab [NMAX][QS] = edge id m * KC + j of every in-edge slot of
// every source node (-1 = none; + one all -1 row), slot_of [M KC] = G row n * QS + q of every edge, gtab [8 M][KC] = LDS byte address of
// a staging item's 16-byte piece of each of its KC G rows.  cnt: NMAX * 9 ints of zeroed scratch.  Called by all BW_THREADS threads;
// idx_v = the thread's table entry (tid < M KC).  Deterministic: the arrival order of the counters is undone by a sort.
template <int KC, int DEG>
__device__ __forceinline__ void bw_build_tables(long long idx_v, int N, int M, int tid, int* tab, int* slot_of, unsigned* gtab, int* cnt) {
    typedef BwLayout<KC> LY;
    constexpr int QS = LY::QS, NMAX = LY::NMAX;
    const int mk = M * KC;
    int* tmp = cnt + NMAX;
    for (int f = tid; f < NMAX * QS + 4; f += BW_THREADS) tab[f] = -1;            // (+ the all -1 row unused lanes read)
    if (tid < mk) {
        const int r = tid;
        const long long v = idx_v;
        const int n = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        slot_of[r] = -1;
        const int pos = atomicAdd(&cnt[n], 1);                           // arrival order: made deterministic by the sort below
        if (pos < 8) tmp[n * 8 + pos] = r;
    }
    bw_barrier();
    if (tid < N) {                                                        // the node's in-edges in (m, j) order = ascending edge id
        const int n = tid, c = min(cnt[n], 8);
        int e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = q < c ? tmp[n * 8 + q] : 0x7fffffff;
#pragma unroll
        for (int a = 0; a < 8; ++a)                                       // (odd-even transposition sort of 8: 8 passes)
#pragma unroll
            for (int q = a & 1; q + 1 < 8; q += 2) {
                const int lo = min(e[q], e[q + 1]), hi = max(e[q], e[q + 1]);
                e[q] = lo; e[q + 1] = hi;
            }
#pragma unroll
        for (int q = 0; q < DEG; ++q)                                     // (host guarantees in-degree <= DEG)
            if (q < c) { tab[n * QS + q] = e[q]; slot_of[e[q]] = n * QS + q; }
    }
    bw_barrier();
    for (int f = tid; f < 8 * mk; f += BW_THREADS) {                  // (item, j) -> 16-byte piece c8 of G row R, chunk-swizzled
        const int item = f / KC, j = f - item * KC, m = item >> 3, c8 = item & 7;
        const int R = slot_of[m * KC + j];
        gtab[f] = R >= 0 ? (unsigned)(LY::OFF_G + R * BW_GROW + ((c8 ^ bw_swz_r(R)) << 4)) : (unsigned)LY::OFF_DUMP;
    }
    bw_barrier();
}

// The same tables, ONCE per graph, into global memory: out [NMAX QS + 4] tab, then [8 M KC] gtab (one workgroup).
template <int KC, int DEG>
__global__ __launch_bounds__(BW_THREADS) void mpconv_bwd_ws_tables_kernel(const int64_t* idx, int N, int M, int* out) {
    typedef BwLayout<KC> LY;
    constexpr int QS = LY::QS, NMAX = LY::NMAX, MMAX = LY::MMAX;
    __shared__ int tab[NMAX * QS + 4], slot_of[MMAX * KC], cnt[NMAX * 9];
    __shared__ unsigned gtab[8 * MMAX * KC];
    const int tid = threadIdx.x, mk = M * KC;
    for (int f = tid; f < NMAX * 9; f += BW_THREADS) cnt[f] = 0;
    const long long idx_v = tid < mk ? idx[tid] : 0;
    bw_barrier();
    bw_build_tables<KC, DEG>(idx_v, N, M, tid, tab, slot_of, gtab, cnt);
    for (int f = tid; f < NMAX * QS + 4; f += BW_THREADS) out[f] = tab[f];
    for (int f = tid; f < 8 * mk; f += BW_THREADS) out[NMAX * QS + 4 + f] = (int)gtab[f];
}

// KC = destination degree (3 / 6), DEG = in-edge slots per source node the tables are sized for (6 / 3)
// XLD = row stride (elements) of x / gx in memory: 64, or 128 for one half of a 128-channel call (compile-time: as a runtime value
// it cost 14 - 20 spilled registers at the 256-VGPR limit)
// NL = source nodes the detype / dP / dW loops run over (a multiple of 16, N <= NL <= NMAX).  The images stay sized for NMAX rows (the
// projection and dx work in 32-node tiles), but the LDPC F -> V call has 48 factor nodes in a layout sized for 64: a quarter of the
// detype tiles, of the per-node dP MFMAs and of dW's k-steps worked on rows that are zero for the kernel's lifetime.  Rows NL ..
// NMAX - 1 of the dP image then keep what the projection left there (finite P values): only dx's discarded rows n >= N see them.
template <int KC, int DEG, int XLD = 64, int NL = BwLayout<KC>::NMAX>
__global__ __launch_bounds__(BW_THREADS) void mpconv_bwd_ws_kernel(const BwParams p) {
    typedef BwLayout<KC> LY;
    constexpr int QS = LY::QS;                        // G rows per source node (slots >= DEG stay zero)
    constexpr int NMAX = LY::NMAX;                    // node rows every image is sized (and every loop runs) for
    constexpr int OFF_X = LY::OFF_X, OFF_G = LY::OFF_G, OFF_PD = LY::OFF_PD, OFF_ET = LY::OFF_ET, OFF_GST = LY::OFF_GST,
                  OFF_TAB = LY::OFF_TAB, OFF_SLOT = LY::OFF_SLOT, OFF_GT = LY::OFF_GT;
    constexpr int NPG = 16 / QS;                      // source nodes per detype tile
    static_assert(NL % 16 == 0 && NL <= LY::NMAX && (NL / 8) % NPG == 0, "NL");
    constexpr int NG = NL / NPG / 8;                  // detype tiles per wave
    constexpr int NSLOT = KC == 3 ? 2 : 1;            // staging items per staging thread (8 M <= 768 / 384 items, 384 staging threads)
    constexpr int ESLOT = 1;                          // in-edge slots per staging thread (N QS <= 384)
    constexpr int MAXNPW = NL / 8;                    // source nodes per wave in the dP phase (a multiple of 16 / QS: the dP read patterns below)
    const int tid = threadIdx.x;
    BW_STAMP_G(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, lh = lane >> 5, i16 = lane & 15, g4 = lane >> 4;
    const int N = p.N, M = p.M, mk = M * KC;
    constexpr int Npad = NMAX;
    const unsigned lds0 = (unsigned)(uintptr_t)bw_lds;
    int* tab = reinterpret_cast<int*>(bw_lds + OFF_TAB);                // [Npad][QS]  edge id m * KC + j of every in-edge slot, -1 = none
    int* slot_of = reinterpret_cast<int*>(bw_lds + OFF_SLOT);           // [M * KC]    G row n * QS + q of every edge
    unsigned* gtab = reinterpret_cast<unsigned*>(bw_lds + OFF_GT);      // [8 M][KC]   LDS address of a staging item's piece of each of its KC G rows

    BW_STAMP_G(1);
    // ---- roles ----
    // phase 3: waves 0-1 dx (48 MFMAs each), waves 2,3,6,7 dW (24 each; waves w and w + 4 share a SIMD: 48 per SIMD either way),
    // waves 4-5 — the SIMD mates of the dx waves, VALU beside MFMA — stage the next sample and own the LDS-DMA of x
    const bool dx_wave = wave < 2;
    const bool dw_wave = wave == 2 || wave == 3 || wave == 6 || wave == 7;
    const bool build_wave = wave >= 2;                                    // staging: every wave but the two that run dx (one item per thread)
    const bool dma_wave = dw_wave;                                        // the LDS-DMA of x: waves that never store (their vmcnt only counts it)
    const int bl = (wave - 2) * 64 + lane;                                // staging-local thread index 0..383
    const int dwq = wave < 4 ? wave - 2 : wave - 4;                       // 0..3 among the dW waves

    // LDS-DMA of x: pieces dwq + 4 u of the twelve; rows >= N re-read row N - 1 (finite; their dP rows are zero)
    unsigned dsrc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int slot = 64 * (dwq + 4 * u) + lane, row = slot >> 3, pos = slot & 7;
        dsrc[u] = (unsigned)(min(row, N - 1) * (XLD * 2) + ((pos ^ bw_swz_r(row)) << 4));
    }
    constexpr int npieces = 12;                       // both x buffers hold 96 rows; rows >= N are copies of row N - 1

    const int chunk_b = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk_b, b_end = min(p.B, b_begin + chunk_b);
    const unsigned char* xg = reinterpret_cast<const unsigned char*>(p.x);

    auto dma_x = [&](int b, int buf) {                 // the dW waves
        const unsigned char* xb = xg + (int64_t)b * p.x_sb * 2;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int piece = dwq + 4 * u;
            if (piece < npieces) bw_dma16s(xb, dsrc[u], lds0 + (unsigned)(OFF_X + buf * (BW_MAXN * BW_XROW) + piece * 1024));
        }
    };
    // the first two samples' x: requested before anything else (the tables below take ~5 us)
    if (b_begin < b_end && dma_wave) { dma_x(b_begin, 0); if (b_begin + 1 < b_end) dma_x(b_begin + 1, 1); }

    // W [64][256] f32 -> bf16 in LDS once, by all threads with coalesced 32-byte reads (rows of 528 bytes in the still unused G / P
    // region), and every wave's resident fragments from there.  Read straight from memory the fragments are gathers — 32 four-byte
    // loads per lane for the projection's, and for the two dx waves 32 sixteen-byte loads from 32 different rows each: 10 000 cycles
    // on those two waves with the other six waiting at the first barrier (of ~21 000 cycles of set-up per launch).
    constexpr int WROW = 528;
    {
        unsigned char* wst = bw_lds + OFF_G;
        static_assert(64 * WROW <= LY::OFF_ET - LY::OFF_G, "W staging fits the G + P region");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + BW_THREADS * it, row = i >> 5, c8 = i & 31;
            const bw_bf16x8 f = bw_frag_f32(p.W + (int64_t)row * p.w_ld + 8 * c8);
            *reinterpret_cast<bw_bf16x8*>(wst + row * WROW + c8 * 16) = f;
        }
    }
    // the neighbour table (M k <= 288 entries: one per thread): requested now, consumed by the table pass below
    long long idx_v = tid < mk ? p.idx[tid] : 0;
    bw_barrier();
    BW_STAMP_G(40);
    // projection (all eight waves): column tile T = wave = (16-channel block ob, edge-type pair ep); A row i = 8 g + 4 h + r is
    // (edge type 2 ep + (g >> 1), channel 16 ob + 8 h + 4 (g & 1) + r): an output lane (node, h) then holds, per edge type of the
    // pair, EIGHT consecutive channels = one 16-byte chunk of the node's edge-type-major P row
    bw_bf16x8 aP[4];
    unsigned xoff[4], pwo[2];
    {
        const int ob = (wave & 7) >> 1, ep = wave & 1;
        const int g = l31 >> 3, h = (l31 >> 2) & 1, r = l31 & 3;
        const int col = 4 * (16 * ob + 8 * h + 4 * (g & 1) + r) + 2 * ep + (g >> 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            unsigned w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = *reinterpret_cast<const uint16_t*>(bw_lds + OFF_G + (16 * kk + 8 * lh + u) * WROW + col * 2);
            aP[kk] = __builtin_bit_cast(bw_bf16x8, make_uint4(w8[0] | (w8[1] << 16), w8[2] | (w8[3] << 16), w8[4] | (w8[5] << 16), w8[6] | (w8[7] << 16)));
            xoff[kk] = (unsigned)(l31 * BW_XROW + (((2 * kk + lh) ^ bw_swz_r(l31)) << 4));
        }
#pragma unroll
        for (int el = 0; el < 2; ++el)
            pwo[el] = (unsigned)(OFF_PD + l31 * BW_PROW + (2 * ep + el) * 128 + (((2 * ob + lh) ^ bw_swz_p(l31, 2 * ep + el)) << 4));
    }
    // dx (waves 0-1): A of dx^T = W dP^T, channel tile ct = wave: W[32 ct + l31][16 ks + 8 lh ..+7], ks = 0..15 -> RA (64 VGPRs)
    // dW (waves 2,3,6,7): RA = the four 32x32 accumulators (channel tile 0/1) x (column tiles 2 dwi, 2 dwi + 1)
    bw_f32x16 RA[4];
    const int dwi = wave < 4 ? wave - 2 : wave - 4;
    if (dx_wave) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(bw_lds + OFF_G + (32 * wave + l31) * WROW + 32 * (4 * t + sub) + 16 * lh);
#pragma unroll
                for (int u = 0; u < 4; ++u) RA[t][4 * sub + u] = f[u];
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 16; ++u) RA[t][u] = 0.f;
    }
    auto prefetch_g = [&](int b, uint4 (&pg)[NSLOT], uint2 (&pa)[NSLOT]) {      // builder threads: gz / argmax of their staging items
        const unsigned char* gzb = reinterpret_cast<const unsigned char*>(p.gz + (int64_t)b * p.y_sb);
        const unsigned char* amb = p.argmax + (int64_t)b * p.y_sb;
        int blo = bl;                                  // opaque per call: keeps the per-lane offsets (and six 64-bit pointers) from being hoisted
        asm volatile("" : "+v"(blo));
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int item = blo + 384 * sl;
            pg[sl] = make_uint4(0, 0, 0, 0);
            pa[sl] = make_uint2(0, 0);
            if (build_wave && item < 8 * M) {
                const unsigned el = (unsigned)((item >> 3) * p.y_ld + 8 * (item & 7));
                pg[sl] = *bw_at<uint4>(gzb, el * 2u);
                pa[sl] = *bw_at<uint2>(amb, el);
            }
        }
    };
    BW_STAMP_G(41);
    // sample 0's gz / argmax: requested now, consumed below
    uint4 pg0[NSLOT];
    uint2 pa0[NSLOT];
    if (b_begin < b_end) prefetch_g(b_begin, pg0, pa0);

    // ---- setup: zero every image (pad rows of G / etT / dP are read as zeros for the kernel's lifetime), transposed incidence ----
    // (bw_barrier, not __syncthreads: the loads requested above stay in flight across it)
    bw_barrier();                                      // every wave has its fragments: the staged W may go
    BW_STAMP_G(42);
    for (int f = OFF_G / 16 + tid; f < LY::BYTES / 16; f += BW_THREADS) reinterpret_cast<uint4*>(bw_lds)[f] = make_uint4(0, 0, 0, 0);      // (not the x buffers)
    BW_STAMP_G(43);
    bw_barrier();
    BW_STAMP_G(44);
    if (p.tables) {
        // the graph is static: its tables were built once (fgnn_mpconv_backward_tables) — two coalesced copies instead of counters,
        // a sort and three barriers in every workgroup of every launch
        for (int f = tid; f < Npad * QS + 4; f += BW_THREADS) tab[f] = p.tables[f];
        for (int f = tid; f < 8 * mk; f += BW_THREADS) gtab[f] = (unsigned)p.tables[Npad * QS + 4 + f];
        bw_barrier();
        BW_STAMP_G(2);
        BW_STAMP_G(6);
    } else {
        // scratch in the (still unused) P / dP region: in-degree counters and up to 8 edge ids per source node
        int* cnt = reinterpret_cast<int*>(bw_lds + OFF_PD);
        bw_build_tables<KC, DEG>(idx_v, N, M, tid, tab, slot_of, gtab, cnt);
        BW_STAMP_G(6);
        for (int f = tid; f < (NMAX * 9 + 3) / 4; f += BW_THREADS) reinterpret_cast<uint4*>(cnt)[f] = make_uint4(0, 0, 0, 0);      // scratch back to zeros
        bw_barrier();
    }

    BW_STAMP_G(7);
    // builder: in-edge slot (n, q) -> its edge-type row, transposed into etT[n][e][q]
    int et_src[ESLOT];
    unsigned et_dst[ESLOT];
#pragma unroll
    for (int sl = 0; sl < ESLOT; ++sl) {
        et_src[sl] = -1;
        et_dst[sl] = 0;
        const int is = bl + 384 * sl;
        if (build_wave && is < N * QS) {
            const int n = is / QS, q = is - n * QS;
            if (q < DEG) et_src[sl] = tab[n * QS + q];
            et_dst[sl] = (unsigned)(OFF_ET + ((n * 4) * QS + q) * 2);
        }
    }
    // detype tiles of this wave (gi = wave + 8 sl): operand addresses and the output lane's table row, static for the kernel
    unsigned da[NG], db_[NG], dt[NG];
#pragma unroll
    for (int sl = 0; sl < NG; ++sl) {
        const int gi = wave + 8 * sl;
        const int R = gi * 16 + i16;
        da[sl] = (unsigned)(OFF_G + R * BW_GROW + ((g4 ^ bw_swz_r(R)) << 4));
        const int jn = QS == 4 ? (i16 >> 2) : ((i16 & 7) >> 2), e = i16 & 3;
        const int nb = gi * NPG + jn;
        db_[sl] = (unsigned)(OFF_PD + nb * BW_PROW + e * 128 + ((g4 ^ bw_swz_p(nb, e)) << 4));
        // D[i = 4 g4 + r][j = i16] is useful where row and column name the same node: (n, first slot q0) of this lane, or none
        bool ok;
        int n, q0;
        if (QS == 4) { ok = g4 == (i16 >> 2); n = gi * 4 + g4; q0 = 0; }
        else { ok = i16 < 8 && (g4 >> 1) == (i16 >> 2); n = gi * 2 + (g4 >> 1); q0 = 4 * (g4 & 1); }
        dt[sl] = (ok && n < N) ? (unsigned)(OFF_TAB + (n * QS + q0) * 4) : (unsigned)(OFF_TAB + NMAX * QS * 4);
    }
    // dP phase: address pattern of the transpose read of a node's G rows.  Lane i16 of group g4 names (row n QS + 4 h4 + (i16 >> 2),
    // chunk 2 g4 + ((i16 & 3) >> 1), half i16 & 1); the row's swizzle (bits 1-3 of R) depends on the node only through n mod (16 / QS),
    // and a wave's first node is a multiple of 4: one pattern per (i mod 4, h4), the node's base added as a scalar
    constexpr int npw = MAXNPW;
    unsigned gpat[4][QS / 4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int h4 = 0; h4 < QS / 4; ++h4) {
            const int R = v * QS + 4 * h4 + (i16 >> 2);
            const int chunk = 2 * g4 + ((i16 & 3) >> 1);
            gpat[v][h4] = lds0 + (unsigned)(OFF_G + (4 * h4 + (i16 >> 2)) * BW_GROW + ((chunk ^ bw_swz_r(R)) << 4) + 8 * (i16 & 1));
        }
    // dW transpose-read offsets: segment of lane i16 in group g4 = (node 8 (g4 >> 1) + (i16 >> 2) [+ 4], columns 16 (g4 & 1) + 4 (i16 & 3) ..)
    unsigned xa[2][2], db[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int nd = 8 * (g4 >> 1) + (i16 >> 2) + 4 * r;
        const int fx = bw_swz_r(nd);                                      // (+ 16 ks leaves it unchanged)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int chunk = 4 * c2 + 2 * (g4 & 1) + ((i16 & 3) >> 1);
            xa[c2][r] = (unsigned)(nd * BW_XROW + ((chunk ^ fx) << 4) + 8 * (i16 & 1));
        }
        db[r] = lds0 + (unsigned)(OFF_PD + nd * BW_DROW + (64 * dwi + 16 * (g4 & 1) + 4 * (i16 & 3)) * 2);
    }
    bw_s16x4 ones_row0;
    {
        const short o = (lane & 3) == 0 ? (short)0x3f80 : (short)0;
        ones_row0 = (bw_s16x4){o, o, o, o};
    }
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};                // [0]: dbias of channel `lane` over this wave's nodes

    // ---- staging pieces ----
    auto prefetch_e = [&](int b, uint2 (&pe)[ESLOT]) {      // builder threads: the edge-type rows of their in-edge slots
#pragma unroll
        for (int sl = 0; sl < ESLOT; ++sl) {
            pe[sl] = make_uint2(0, 0);
            if (et_src[sl] >= 0) pe[sl] = *bw_at<uint2>(p.et + (int64_t)b * p.et_sb, (unsigned)et_src[sl] * 8u);
        }
    };
    auto build = [&](const uint4 (&pg)[NSLOT], const uint2 (&pa)[NSLOT], const uint2 (&pe)[ESLOT]) {      // prefetched registers -> etT, G
#pragma unroll
        for (int sl = 0; sl < ESLOT; ++sl) {
            if (et_src[sl] >= 0) {
                uint16_t* ew = reinterpret_cast<uint16_t*>(bw_lds + et_dst[sl]);
                ew[0] = (uint16_t)pe[sl].x; ew[QS] = (uint16_t)(pe[sl].x >> 16);
                ew[2 * QS] = (uint16_t)pe[sl].y; ew[3 * QS] = (uint16_t)(pe[sl].y >> 16);
            }
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int item = bl + 384 * sl;
            if (item < 8 * M) {
                // argmax bytes -> 16-bit halves, one-hot per half; slot j keeps a gz half where bit j is set:
                // (onehot << (15 - j)) >> 15 (arithmetic, per half) is the 16-bit mask
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const unsigned gq[4] = {pg[sl].x, pg[sl].y, pg[sl].z, pg[sl].w};
                unsigned gofs[KC];
#pragma unroll
                for (int j = 0; j < KC; ++j) gofs[j] = gtab[item * KC + j];
                u16x2 oh[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned src = d < 2 ? pa[sl].x : pa[sl].y;
                    const unsigned a2 = __builtin_amdgcn_perm(0u, src, (d & 1) ? 0x0c030c02u : 0x0c010c00u) & 0x00070007u;
                    oh[d] = (u16x2){1, 1} << __builtin_bit_cast(u16x2, a2);
                }
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    unsigned w[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const s16x2 msk = __builtin_bit_cast(s16x2, (u16x2)(oh[d] << (u16x2){(unsigned short)(15 - j), (unsigned short)(15 - j)})) >> (s16x2){15, 15};
                        w[d] = gq[d] & __builtin_bit_cast(unsigned, msk);
                    }
                    *reinterpret_cast<uint4*>(bw_lds + gofs[j]) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    };

    BW_STAMP_G(3);
    // ---- pipeline fill: sample b_begin staged ----
    if (b_begin < b_end) {
        if (build_wave) {
            uint2 pe[ESLOT];
            prefetch_e(b_begin, pe); build(pg0, pa0, pe);
        }
        if (dma_wave) bw_wait_vm0();
    }
    bw_barrier();
    constexpr int ntile = NMAX / 32;
    int cur = 0;                                       // x buffer of this sample: (b - b_begin) % 3

    for (int b = b_begin; b < b_end; ++b) {
        const unsigned char* xs = bw_lds + OFF_X + cur * (BW_MAXN * BW_XROW);
        const bool has_next = b + 1 < b_end;
        BW_STAMP(0);
        BW_STAMP_G(8 + (b - b_begin));
        // ================= phase 1: P^T tile of this wave (32 columns) for every node tile, three independent chains =================
        // Every phase below issues ALL its LDS reads first and computes behind a scheduling barrier: with two waves per SIMD
        // nothing else hides an LDS round trip, and left alone the compiler puts each read right in front of its MFMA.
        {
            uint4 xb[3][4];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    xb[t][kk] = t < ntile ? *reinterpret_cast<const uint4*>(xs + t * 32 * BW_XROW + xoff[kk]) : make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bw_f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[t][u] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aP[kk], __builtin_bit_cast(bw_bf16x8, xb[t][kk]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ntile) {
                    unsigned char* pp0 = bw_lds + t * 32 * BW_PROW + pwo[0];
                    unsigned char* pp1 = bw_lds + t * 32 * BW_PROW + pwo[1];
                    *reinterpret_cast<uint4*>(pp0) = make_uint4(bw_pack2(acc[t][0], acc[t][1]), bw_pack2(acc[t][2], acc[t][3]),
                                                                bw_pack2(acc[t][4], acc[t][5]), bw_pack2(acc[t][6], acc[t][7]));
                    *reinterpret_cast<uint4*>(pp1) = make_uint4(bw_pack2(acc[t][8], acc[t][9]), bw_pack2(acc[t][10], acc[t][11]),
                                                                bw_pack2(acc[t][12], acc[t][13]), bw_pack2(acc[t][14], acc[t][15]));
                }
            }
        }
        BW_STAMP(1);
        bw_barrier();
        BW_STAMP(2);
        // Phases 2a / 2b are written once (lambdas) and instantiated in BOTH arms of the role branch below: the staging waves hold the
        // next sample's prefetched words in registers from here to phase 3, and values that flow around a branch merge are copied
        // at the merge — i.e. the wave would wait for its loads right after issuing them (measured: 2 300 cycles per sample).
        auto phase2a = [&]() {
            {
                uint16_t* gst = reinterpret_cast<uint16_t*>(bw_lds + OFF_GST);
                uint4 fa[NG][2], fb[NG][2];
                int4 te[NG];
    #pragma unroll
                for (int sl = 0; sl < NG; ++sl) {
                    fa[sl][0] = *reinterpret_cast<const uint4*>(bw_lds + da[sl]);
                    fb[sl][0] = *reinterpret_cast<const uint4*>(bw_lds + db_[sl]);
                    fa[sl][1] = *reinterpret_cast<const uint4*>(bw_lds + (da[sl] ^ 64u));
                    fb[sl][1] = *reinterpret_cast<const uint4*>(bw_lds + (db_[sl] ^ 64u));
                    te[sl] = *reinterpret_cast<const int4*>(bw_lds + dt[sl]);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[NG];
    #pragma unroll
                for (int sl = 0; sl < NG; ++sl)
                    acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bw_bf16x8, fa[sl][0]), __builtin_bit_cast(bw_bf16x8, fb[sl][0]),
                                                                      (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    #pragma unroll
                for (int sl = 0; sl < NG; ++sl)
                    acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bw_bf16x8, fa[sl][1]), __builtin_bit_cast(bw_bf16x8, fb[sl][1]),
                                                                      acc[sl], 0, 0, 0);
                const int e = i16 & 3;
    #pragma unroll
                for (int sl = 0; sl < NG; ++sl) {
                    const int ed[4] = {te[sl].x, te[sl].y, te[sl].z, te[sl].w};
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ed[r] >= 0) gst[e * mk + ed[r]] = (uint16_t)(bw_pack2(acc[sl][r], 0.f) & 0xffffu);
                }
            }
        };
        auto phase2b = [&]() {
            {
                uint2 bv[MAXNPW][QS / 4], av[MAXNPW][QS / 4];
    #pragma unroll
                for (int i = 0; i < MAXNPW; ++i) {
                    const int n = wave * npw + i;
    #pragma unroll
                    for (int h4 = 0; h4 < QS / 4; ++h4) {
                        // (nodes >= N: their G rows and edge-type rows are zero for the kernel's lifetime -> a zero dP row, what dW needs)
                        bv[i][h4] = bw_tr(gpat[i & 3][h4] + (unsigned)(n * QS * BW_GROW));
                        av[i][h4] = *reinterpret_cast<const uint2*>(bw_lds + OFF_ET + ((n * 4 + (lane & 3)) * QS + 4 * h4) * 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                BW_STAMP(9);
    #pragma unroll
                for (int i = 0; i < MAXNPW; ++i) {
                    {
                        const int n = wave * npw + i;
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                        for (int h4 = 0; h4 < QS / 4; ++h4) {
                            const bw_s16x4 b4 = __builtin_bit_cast(bw_s16x4, bv[i][h4]);
                            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bw_s16x4, av[i][h4]), b4, acc, 0, 0, 0);
                            dbacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones_row0, b4, dbacc, 0, 0, 0);
                        }
                        *reinterpret_cast<uint2*>(bw_lds + OFF_PD + n * BW_DROW + lane * 8) = make_uint2(bw_pack2(acc[0], acc[1]), bw_pack2(acc[2], acc[3]));
                    }
                }
                BW_STAMP(10);
                // edge-type gradient of this sample: staging -> memory, 16 bytes per lane (8 mk bytes; host: mk even, 16-byte aligned)
                const int nvec = (8 * mk) >> 4;
                if (tid < nvec) {
                    uint4 v = *reinterpret_cast<const uint4*>(bw_lds + OFF_GST + tid * 16);
                    if (p.accum & 2) {                         // second launch of a split call: add to what the first one stored
                        const uint4 o = *bw_at<uint4>(p.get + (int64_t)b * 4 * mk, (unsigned)tid * 16u);
                        v = make_uint4(bw_add2(v.x, o.x), bw_add2(v.y, o.y), bw_add2(v.z, o.z), bw_add2(v.w, o.w));
                    }
                    *bw_at<uint4>(p.get + (int64_t)b * 4 * mk, (unsigned)tid * 16u) = v;
                }
            }
        };
        auto phase3_dw = [&](const uint4 (&pg)[NSLOT], const uint2 (&pa)[NSLOT], const uint2 (&pe)[ESLOT]) {
                const unsigned xbase = lds0 + (unsigned)(OFF_X + cur * (BW_MAXN * BW_XROW));
                constexpr int nks = NL / 16;
                uint2 f0[8], f1[8];
                auto load = [&](uint2 (&f)[8], int ks) {   // A: channel tiles 0 / 1 (two reads each); B: this wave's two column tiles
    #pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        f[2 * c2] = bw_tr(xbase + xa[c2][0] + ks * (16 * BW_XROW));
                        f[2 * c2 + 1] = bw_tr(xbase + xa[c2][1] + ks * (16 * BW_XROW));
                    }
    #pragma unroll
                    for (int cj = 0; cj < 2; ++cj) {
                        f[4 + 2 * cj] = bw_tr(db[0] + ks * (16 * BW_DROW) + cj * 64);
                        f[5 + 2 * cj] = bw_tr(db[1] + ks * (16 * BW_DROW) + cj * 64);
                    }
                };
                auto mma = [&](const uint2 (&f)[8]) {
    #pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
    #pragma unroll
                        for (int cj = 0; cj < 2; ++cj)
                            RA[2 * c2 + cj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bw_bf16x8, make_uint4(f[2 * c2].x, f[2 * c2].y, f[2 * c2 + 1].x, f[2 * c2 + 1].y)),
                                __builtin_bit_cast(bw_bf16x8, make_uint4(f[4 + 2 * cj].x, f[4 + 2 * cj].y, f[5 + 2 * cj].x, f[5 + 2 * cj].y)),
                                RA[2 * c2 + cj], 0, 0, 0);
                };
                load(f0, 0);
                if (has_next) build(pg, pa, pe);           // G / etT of sample b + 1, under the first transpose reads
    #pragma unroll 1
                for (int ks = 0; ks + 1 < nks; ks += 2) {  // the next k-step's eight transpose reads are in flight under this one's MFMAs
                    load(f1, ks + 1);
                    mma(f0);
                    if (ks + 2 < nks) load(f0, ks + 2);
                    mma(f1);
                }
                if (nks & 1) mma(f0);                      // (an odd count: the last k-step was loaded by the loop's final iteration)
        };
        if (build_wave) {
            // the next sample's gz / argmax / edge types: requested two phases ahead of their use (past the last sample: a harmless re-read)
            uint4 pg[NSLOT];
            uint2 pa[NSLOT], pe[ESLOT];
            prefetch_g(has_next ? b + 1 : b, pg, pa);
            prefetch_e(has_next ? b + 1 : b, pe);
            phase2a();
            BW_STAMP(3);
            bw_barrier();
            BW_STAMP(4);
            phase2b();
            BW_STAMP(5);
            bw_barrier();
            BW_STAMP(6);
            // ================= phase 3 (waves 2-7): dW behind the staging of the next sample (their readers finished in phase 2b) =================
            if (dw_wave) phase3_dw(pg, pa, pe);
            else if (has_next) build(pg, pa, pe);
        } else {
            phase2a();
            BW_STAMP(3);
            bw_barrier();
            BW_STAMP(4);
            phase2b();
            BW_STAMP(5);
            bw_barrier();
            BW_STAMP(6);
            // ================= phase 3 (waves 0-1): dx =================
            {
                uint16_t* gxb = p.gx + (int64_t)b * p.x_sb;
                    // Software pipeline over the node tiles with no extra registers: the tile's fragments 8..15 are requested when it starts
                // (they land under the first eight MFMAs), fragments 0..7 of the NEXT tile when those eight are done.  Two chains over the
                // even / odd k-steps: one chain of 16 dependent MFMAs would idle the pipe half the time.
                uint4 bf[16];
                const unsigned char* bp0 = bw_lds + OFF_PD + l31 * BW_DROW + lh * 16;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(bp0 + 32 * ks);
#pragma unroll
                for (int nt = 0; nt < ntile; ++nt) {
                    const unsigned char* bp = bp0 + nt * 32 * BW_DROW;
#pragma unroll
                    for (int ks = 8; ks < 16; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(bp + 32 * ks);
                    __builtin_amdgcn_sched_barrier(0);
                    bw_f32x16 ae, ao;
#pragma unroll
                    for (int u = 0; u < 16; ++u) { ae[u] = 0.f; ao[u] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < 16; ks += 2) {
                        if (ks == 8) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (nt + 1 < ntile) {
#pragma unroll
                                for (int k2 = 0; k2 < 8; ++k2) bf[k2] = *reinterpret_cast<const uint4*>(bp + 32 * BW_DROW + 32 * k2);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        f32x4 f0, f1;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { f0[u] = RA[ks >> 2][4 * (ks & 3) + u]; f1[u] = RA[(ks + 1) >> 2][4 * ((ks + 1) & 3) + u]; }
                        ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bw_bf16x8, f0), __builtin_bit_cast(bw_bf16x8, bf[ks]), ae, 0, 0, 0);
                        ao = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bw_bf16x8, f1), __builtin_bit_cast(bw_bf16x8, bf[ks + 1]), ao, 0, 0, 0);
                    }
                    const int n = nt * 32 + l31;
                    if (n < N) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v0 = ae[4 * g] + ao[4 * g], v1 = ae[4 * g + 1] + ao[4 * g + 1], v2 = ae[4 * g + 2] + ao[4 * g + 2],
                                  v3 = ae[4 * g + 3] + ao[4 * g + 3];
                            uint2* dst = bw_at<uint2>(gxb, (unsigned)(n * XLD + 32 * wave + 8 * g + 4 * lh) * 2u);
                            if (p.accum & 1) {
                                const uint2 o = *dst;
                                v0 += __uint_as_float(o.x << 16); v1 += __uint_as_float(o.x & 0xffff0000u);
                                v2 += __uint_as_float(o.y << 16); v3 += __uint_as_float(o.y & 0xffff0000u);
                            }
                            *dst = make_uint2(bw_pack2(v0, v1), bw_pack2(v2, v3));
                        }
                    }
                }
            }
        }
        // x by LDS-DMA, TWO samples ahead and from the tail of phase 3: the memory pipe takes a piece per ~85 cycles (HBM rate), and
        // issued in front of phase 1 the twelve pieces held up the projections of the issuing waves by up to 1 000 cycles.  These
        // waves store nothing, so vmcnt(0) = "x of sample b + 1 (requested a sample ago) has landed".
        if (dma_wave) {
            bw_wait_vm0();
            if (b + 2 < b_end) dma_x(b + 2, cur >= 1 ? cur - 1 : 2);      // buffer (cur + 2) % 3: last read by dW of sample b - 1
        }
        BW_STAMP(7);
        bw_barrier();
        BW_STAMP(8);
        cur = cur == 2 ? 0 : cur + 1;
    }   // samples

    BW_STAMP_G(4);
    // ---- flush dW tiles and dbias into this workgroup's slab (summed by the slab reduce, fixed order) ----
    if (b_begin < b_end) {
        float* slab = p.ws + (int64_t)blockIdx.x * (64 * 256 + 64);
        if (dw_wave) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int cj = 0; cj < 2; ++cj) {
                    const int col = 32 * (2 * dwi + cj) + l31;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int c = 32 * c2 + (u & 3) + 8 * (u >> 2) + 4 * lh;
                        slab[c * 256 + col] = RA[2 * c2 + cj][u];
                    }
                }
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(bw_lds + OFF_X);          // [8 waves][64 channels]
        red[wave * 64 + lane] = dbacc[0];
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int w = 0; w < BW_WAVES; ++w) s += red[w * 64 + tid];
            slab[64 * 256 + tid] = s;
        }
    }
    BW_STAMP_G(5);
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define BW_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] ws backward rejects shape: rule %d\n", code); return 0; } while (0)

// Called first by fgnn_mpconv_backward_sg (mpconv_bwd_sg.hip) for the 64 -> 64 calls: 1 = launched, 0 = not this kernel's
// shape (the second-generation kernel takes it), < 0 = error.
int fgnn_check_desc(const fgnn_mpconv_desc* d);

// Does this descriptor go to the kernel above, and with how many table entries?  (The shape rules of fgnn_mpconv_backward_ws that do
// not depend on pointers.)
static int bw_tables_count(const fgnn_mpconv_desc* d) {
    static const bool off = getenv("FGNN_NO_WS") != nullptr;
    if (off || d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->net != 4 || d->agg != FGNN_AGG_MAX) return 0;
    if (d->nin != 64 || (d->nou != 64 && d->nou != 128) || (d->k != 3 && d->k != 6)) return 0;
    if (d->idx_sb != 0 && d->B > 1) return 0;
    if (!(d->idx_sk == 1 && d->idx_sm == d->k)) return 0;
    const int KC = d->k;
    if (KC == 6 ? (d->N > 96 || d->M > 48) : (d->N > 64 || d->M > 96)) return 0;
    if ((d->M * KC) & 1) return 0;
    return (KC == 6 ? BwLayout<6>::NMAX * BwLayout<6>::QS : BwLayout<3>::NMAX * BwLayout<3>::QS) + 4 + 8 * d->M * KC;
}

// Bytes of the per-graph tables the backward of this descriptor can take (0: the shape does not use any).
extern "C" int64_t fgnn_mpconv_backward_tables_bytes(const fgnn_mpconv_desc* d) {
    if (fgnn_check_desc(d)) return 0;
    return (int64_t)bw_tables_count(d) * 4;
}

// Builds them (one small launch; the in-degree bound travels in d->reserved as for fgnn_mpconv_backward).
extern "C" int fgnn_mpconv_backward_tables(const fgnn_mpconv_desc* d, const int64_t* nn_idx, void* tables, fgnn_stream_t stream) {
    int rc = fgnn_check_desc(d);
    if (rc) return rc;
    if (!nn_idx || !tables) FGNN_FAIL(FGNN_EINVAL, "mpconv_backward_tables: null pointer");
    const int KC = d->k, DEG = KC == 6 ? 3 : 6, indeg = d->reserved & 0xffff;
    if (bw_tables_count(d) == 0 || indeg < 1 || indeg > DEG)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv_backward_tables: not a shape / in-degree of the table-driven backward");
    if (KC == 6) hipLaunchKernelGGL((mpconv_bwd_ws_tables_kernel<6, 3>), dim3(1), dim3(BW_THREADS), 0, (hipStream_t)stream, nn_idx, d->N, d->M, (int*)tables);
    else hipLaunchKernelGGL((mpconv_bwd_ws_tables_kernel<3, 6>), dim3(1), dim3(BW_THREADS), 0, (hipStream_t)stream, nn_idx, d->N, d->M, (int*)tables);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv_backward_tables launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

// The tables of the NEXT backward launch on this thread (fgnn_mpconv_backward_with_tables sets them around its call).
static thread_local const void* bw_pending_tables = nullptr;
void fgnn_bw_set_pending_tables(const void* t) { bw_pending_tables = t; }

int fgnn_mpconv_backward_ws(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                            float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                            fgnn_stream_t stream) {
    static const bool off = getenv("FGNN_NO_WS") != nullptr;       // (tests: the second-generation kernels on the same shapes)
    if (off) BW_REJECT(0);
    const bool split = d->nin == 64 && d->nou == 128;                    // 64 -> 128: two launches over the halves of the output channels
    // 128 -> 64 (round 5): two launches over the halves of the INPUT channels.  Everything the kernel computes is linear in the
    // input-channel block it is given — P = P_lo + P_hi, so detype = sum_o G P splits into two addends (the second launch ADDS to
    // getype); dP depends on G and etype only; dx and dW are per input channel — so the 64-channel kernel runs twice on x / W / gx /
    // gfilters offset by 64 channels / rows (x and gx rows stay 128 apart in memory: x_ld), dbias counted once.  Replaces the
    // first-generation mpconv_bwd_b16_kernel<4,2> for these calls (245 us at 4096 codewords, profiles/r04).
    const bool ksplit = d->nin == 128 && d->nou == 64;
    if (ksplit) {      // (the layout rules fgnn_mpconv_backward_sg checks for its callers)
        if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->agg != FGNN_AGG_MAX || d->net != 4 || (d->k != 3 && d->k != 6)) BW_REJECT(10);
        if ((d->idx_sb != 0 && d->B > 1) || !(d->idx_sk == 1 && d->idx_sm == d->k) || !getype || !argmax || !gbias) BW_REJECT(11);
        if (!(d->x_sc == 1 && d->x_sn == d->nin) || !(d->y_sc == 1 && (d->y_sm == d->nou || d->M == 1))) BW_REJECT(12);
        if (!(d->et_se == 1 && d->et_sk == 4 && (d->et_sm == 4 * d->k || d->M == 1))) BW_REJECT(13);
    }
    if (!ksplit && (d->nin != 64 || (d->nou != 64 && !split))) BW_REJECT(1);
    const int KC = d->k, DEG = KC == 6 ? 3 : 6;
    const int indeg = d->reserved & 0xffff;
    if (indeg < 1 || indeg > DEG) BW_REJECT(2);
    if (KC == 6 ? (d->N > 96 || d->M > 48) : (d->N > 64 || d->M > 96)) BW_REJECT(3);
    if ((d->M * KC) & 1) BW_REJECT(4);                                  // getype leaves as whole 16-byte lanes
    if (((uintptr_t)getype & 15) || ((uintptr_t)x & 15) || ((uintptr_t)gz & 15) || ((uintptr_t)argmax & 7) || ((uintptr_t)gx & 7) ||
        ((uintptr_t)etype & 7)) BW_REJECT(5);
    if ((d->x_sb % 8) != 0 || (d->y_sb % 8) != 0 || (d->et_sb % 4) != 0) BW_REJECT(6);
    const int64_t nw = 64 * 256, slab_len = nw + 64;
    if (!workspace || workspace_bytes < ((split || ksplit) ? 2 : 1) * 256 * slab_len * 4) BW_REJECT(7);

    BwParams p;
    p.x = (const uint16_t*)x; p.idx = nn_idx; p.et = (const uint16_t*)etype; p.W = filters;
    p.gz = (const uint16_t*)gz; p.argmax = argmax; p.gx = (uint16_t*)gx; p.get = (uint16_t*)getype;
    p.ws = (float*)workspace;
    p.B = d->B; p.N = d->N; p.M = d->M;
    p.y_ld = d->nou; p.w_ld = d->nou * 4; p.x_ld = d->nin; p.accum = 0;
    p.x_sb = d->x_sb; p.et_sb = d->et_sb; p.y_sb = d->y_sb;
    p.tables = (bw_pending_tables && bw_tables_count(d) > 0) ? (const int*)bw_pending_tables : nullptr;
    const int off_b = KC == 6 ? BwLayout<6>::BYTES : BwLayout<3>::BYTES;
    static_assert(BwLayout<6>::BYTES <= 160 * 1024 && BwLayout<3>::BYTES <= 160 * 1024, "LDS");
    const bool nl48 = KC == 3 && d->N <= 48;                   // the LDPC F -> V call: 48 factor nodes
    void* fn = ksplit ? (KC == 6 ? (void*)mpconv_bwd_ws_kernel<6, 3, 128> : nl48 ? (void*)mpconv_bwd_ws_kernel<3, 6, 128, 48> : (void*)mpconv_bwd_ws_kernel<3, 6, 128>)
                      : (KC == 6 ? (void*)mpconv_bwd_ws_kernel<6, 3> : nl48 ? (void*)mpconv_bwd_ws_kernel<3, 6, 64, 48> : (void*)mpconv_bwd_ws_kernel<3, 6>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, off_b);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", off_b, hipGetErrorString(e));
    int grid = 256;      // one workgroup per CU (caps of 248 / 240 / 224 to leave CUs to the other stream: no effect, profiles/r05/README.md)
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    fgnn_note_kernel(split ? "mpconv_bwd_ws_kernel<%d, %d> x2" : (ksplit ? "mpconv_bwd_ws_kernel<%d, %d> k2" : "mpconv_bwd_ws_kernel<%d, %d>"), KC, DEG);
    p.prof = nullptr;
#ifdef FGNN_ENABLE_PROF
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 256 * 8);
        (void)hipMemset(prof_buf, 0, 256 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(BW_THREADS), args, off_b, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ws backward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {      // slots: 0 sample start, 1/2 projection done / released, 3/4 detype, 5/6 dP, 7/8 dx | dW | staging
        long long h[256];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[fgnn prof ws bwd] wave %d:", w);
            for (int i = 0; i < 11; ++i) fprintf(stderr, " %6lld", h[w * 16 + i] - h[0]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[fgnn prof ws bwd] kernel: early-dma %lld zero+idx %lld rank %lld gtab %lld consts %lld loop-end %lld end %lld; samples:", h[129] - h[128],
                h[130] - h[128], h[134] - h[128], h[135] - h[128], h[131] - h[128], h[132] - h[128], h[133] - h[128]);
        for (int i = 0; i < 20 && h[136 + i]; ++i) fprintf(stderr, " %lld", h[136 + i] - h[128]);
        fprintf(stderr, "\n[fgnn prof ws bwd] setup (thread 0): W-proj frags %lld, dx frags %lld, prefetch issued %lld, zeroed %lld, barrier %lld\n",
                h[128 + 40] - h[128], h[128 + 41] - h[128], h[128 + 42] - h[128], h[128 + 43] - h[128], h[128 + 44] - h[128]);
    }
#endif
    if (ksplit) {
        fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);                  // rows 0..63 of gfilters, dbias
        p.ws += (int64_t)grid * slab_len;                                                         // (its own slabs: the first launch's fold may be a recorded one)
        p.x += 64; p.W += 64 * (int64_t)p.w_ld; p.gx += 64; p.accum = 2;                          // input channels 64..127: getype accumulates
        e = hipLaunchKernel(fn, dim3(grid), dim3(BW_THREADS), args, off_b, st);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ws backward launch (upper input channels): %s", hipGetErrorString(e));
        fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters + nw, nullptr, st);           // rows 64..127 (dbias was counted by the first launch)
    } else if (!split) fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    else {
        // slab rows are 256 columns of gfilters' 512: lower half, then the second launch on the upper 64 output channels, which ADDS
        // to gx / getype (one more bf16 rounding of those two) and folds its dW / dbias into the upper column / channel blocks
        fgnn_launch_slab_reduce_ld(p.ws, grid, slab_len, nw, 256, 512, gfilters, gbias, st);
        p.ws += (int64_t)grid * slab_len;
        p.W += 256; p.gz += 64; p.argmax += 64; p.accum = 3;
        e = hipLaunchKernel(fn, dim3(grid), dim3(BW_THREADS), args, off_b, st);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv ws backward launch (upper half): %s", hipGetErrorString(e));
        fgnn_launch_slab_reduce_ld(p.ws, grid, slab_len, nw, 256, 512, gfilters + 256, gbias + 64, st);
    }
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
