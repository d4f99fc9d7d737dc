// linear_wgrad_b16.hip — weight / bias gradient of the node-wise (1x1) maps for bf16 activations with
// channel counts that are multiples of 64 (every map of the LDPC model's main path; reference
// /root/reference/lib/model/mpnn/mp_nn_residual.py:25-35, base_model.py:43-90):
//
//     gW[o][c] += sum_r gy[r][o] * x[r][c]        gb[o] += sum_r gy[r][o]          r over R = B*N rows
//
// The contraction runs over ROWS, so both MFMA operands want "8 consecutive rows of one channel" per lane
// while memory is row-major.  Instead of staging through LDS, every lane loads 4 channels (8 bytes) of each
// of its 8 rows straight from global memory and transposes the 8x4 block in registers with v_perm_b32: the
// result is four v_mfma_f32_16x16x32_bf16 fragments, one per channel, whose 16 lanes cover channels
// {4*i + p}.  The channel permutation is harmless (it is undone when the partials are summed) and the
// k-slot order inside a fragment is irrelevant as long as gy and x agree on it, which they do by construction.
//
// One wave owns one 64x64 slice of gW (16 MFMA tiles = 64 accumulator registers) and a share of the rows.
// A 1024-thread workgroup holds S = (Cout/64)*(Cin/64) <= 16 slices x 16/S row-waves; the waves of different
// slices walk the SAME 32-row blocks at the same time, so x / gy lines are fetched from HBM once and re-hit in
// L1/L2.  No LDS, no barriers in the streaming loop.  The op is HBM-bound (bf16 matrix cores are ~5x faster than the stream).  dbias: v_dot2 of
// each fragment dword against (1, 1).  Partials: the row-waves of a slice fold through LDS as a binary tree
// (fixed order), the workgroup writes its slab in REGISTER order (coalesced), and wgb_reduce_kernel sums the
// slabs and undoes the permutation: deterministic, no atomics.
#include "fgnn_common.h"
#include <stdlib.h>

bool fgnn_fold_push(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gb, int kind, int a, int b, int c, int d);   // fold_batch.hip

#define WB_THREADS 1024
#define WB_WAVES 16
#define WB_NACC 68       // 64 gW accumulators + 4 dbias partials per lane

typedef __bf16 wb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));

#define WB_MAXSRC 3
struct WgbParams {
    const uint16_t* x;   // [R][Cin]  bf16
    const uint16_t* gy;  // [R][Cout] bf16 (source 0)
    float* ws;           // [gridDim.x][S][WB_NACC][64]
    int R, Cin, Cout;
    int nso, S, RW;      // output-channel slices, slices per workgroup, row-waves per slice (S * RW == 16)
    // Several gradient tensors contracted with the SAME x in one pass (fgnn_linear_wgrad_multi: the maps that consume one layer
    // state — x is read once instead of once per map).  Source s owns slices [sb[s], sb[s + 1]), ordered sc * nso_s + so inside;
    // slices >= sb[WB_MAXSRC] (S padded to a power of two) belong to nobody: their waves only take part in the fold's barriers.
    const uint16_t* gys[WB_MAXSRC];
    int couts[WB_MAXSRC];
    int sb[WB_MAXSRC + 1];
};

extern __shared__ __attribute__((aligned(16))) float wb_lds[];

// rows r0..r7 each hold channels (c0 c1 | c2 c3) as two dwords: gather channel P's eight values
template <int P>
__device__ __forceinline__ uint4 wb_pack(const uint2 (&r)[8]) {
    constexpr unsigned sel = (P & 1) ? 0x07060302u : 0x05040100u;     // high / low halves of (hi:b, lo:a)
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned a = P < 2 ? r[2 * q].x : r[2 * q].y, b = P < 2 ? r[2 * q + 1].x : r[2 * q + 1].y;
        w[q] = __builtin_amdgcn_perm(b, a, sel);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ float wb_sum8(const uint4& f, float acc) {
    const wb_bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wb_bf16x2, f.x), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wb_bf16x2, f.y), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wb_bf16x2, f.z), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wb_bf16x2, f.w), one, acc, false);
    return acc;
}

__global__ __launch_bounds__(WB_THREADS) void linear_wgrad_b16_kernel(const WgbParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int slice = wave % p.S, rw = wave / p.S;     // slice = sb[source] + sc * nso + so
    const int src = slice >= p.sb[2] ? 2 : (slice >= p.sb[1] ? 1 : 0);
    const bool live = slice < p.sb[WB_MAXSRC];
    const int Cout = src == 2 ? p.couts[2] : (src == 1 ? p.couts[1] : p.couts[0]);
    const int nso = Cout >> 6, sl = slice - (src == 2 ? p.sb[2] : (src == 1 ? p.sb[1] : p.sb[0]));
    const int so = sl % nso, sc = sl / nso;
    const int R = p.R, Cin = p.Cin;
    const uint16_t* gyp = (src == 2 ? p.gys[2] : (src == 1 ? p.gys[1] : p.gys[0])) + so * 64 + 4 * li;
    const uint16_t* xp = p.x + sc * 64 + 4 * li;

    f32x4 acc[4][4];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nblk = live ? (R + 31) / 32 : 0;
    const int stride = gridDim.x * p.RW;
    uint2 rg[8], rx[8];
    auto load = [&](int blk) {
        const int row0 = blk * 32 + 8 * lk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = row0 + j;
            const bool ok = row < R;
            rg[j] = ok ? *reinterpret_cast<const uint2*>(gyp + (int64_t)row * Cout) : make_uint2(0, 0);
            rx[j] = ok ? *reinterpret_cast<const uint2*>(xp + (int64_t)row * Cin) : make_uint2(0, 0);
        }
    };
    // no software prefetch: 16 waves per CU with 8 KB of loads each keep ~128 KB in flight per CU, and a
    // second register set would not fit the 128-VGPR budget of a 1024-thread workgroup
    // (round 5, measured and dropped, gpurun_out/r05p / r05q: (a) staggering the S slice-waves of a row-wave over different blocks so
    // that a 256 x 256 map has 16 blocks instead of one in flight per CU — 250 vs 178 us: the repeats then miss L1 and queue on L2;
    // (b) folding the slabs in 2 KB contiguous pieces instead of 256-byte lines — no change for the wide maps, +5 us for 64 x 64)
    for (int blk = blockIdx.x * p.RW + rw; blk < nblk; blk += stride) {
        load(blk);
        uint4 A[4], Bf[4];
        A[0] = wb_pack<0>(rg); A[1] = wb_pack<1>(rg); A[2] = wb_pack<2>(rg); A[3] = wb_pack<3>(rg);
        Bf[0] = wb_pack<0>(rx); Bf[1] = wb_pack<1>(rx); Bf[2] = wb_pack<2>(rx); Bf[3] = wb_pack<3>(rx);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bs[a] = wb_sum8(A[a], bs[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wb_bf16x8, A[a]),
                                                                    __builtin_bit_cast(wb_bf16x8, Bf[b]), acc[a][b], 0, 0, 0);
        }
    }

    // ---- binary-tree fold of the RW row-waves of each slice (fixed order) ----
    // slot (slice, t) for t < RW/2 holds WB_NACC x 64 floats, thread-private positions
    for (int half = p.RW >> 1; half >= 1; half >>= 1) {
        float* slot = wb_lds + ((int64_t)(slice * (p.RW >> 1) + (rw - half)) * WB_NACC) * 64 + lane;
        if (rw >= half && rw < 2 * half) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slot[((a * 4 + b) * 4 + r) * 64] = acc[a][b][r];
                slot[(64 + a) * 64] = bs[a];
            }
        }
        __syncthreads();
        if (rw < half) {
            const float* src = wb_lds + ((int64_t)(slice * (p.RW >> 1) + rw) * WB_NACC) * 64 + lane;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += src[((a * 4 + b) * 4 + r) * 64];
                bs[a] += src[(64 + a) * 64];
            }
        }
        __syncthreads();
    }
    if (rw == 0 && live) {                             // register-order slab: [slice][q][lane], coalesced
#pragma unroll
        for (int a = 0; a < 4; ++a) {                  // dbias: fold the 4 row-group lanes of a channel
            bs[a] += __shfl_xor(bs[a], 16);
            bs[a] += __shfl_xor(bs[a], 32);
        }
        float* slab = p.ws + (((int64_t)blockIdx.x * p.S + slice) * WB_NACC) * 64 + lane;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[((a * 4 + b) * 4 + r) * 64] = acc[a][b][r];
            slab[(64 + a) * 64] = bs[a];
        }
    }
}

// ----------------------------------------------------------------------------------------
// WIDE maps (8 or 16 slices per workgroup: 128 x 256, 256 x 256, the merged launches over a 256-channel state) — round 6.
// With S slices the register-direct kernel above has 16 / S row-waves, i.e. ONE or TWO 32-row blocks in flight per CU, and every
// wave fetches its own copy of the operands (128 KB of requests for 32 KB of rows): the 256 x 256 map ran at 2.2 TB/s, latency-bound
// (profiles/r05/wbench.log; 64 x 256, four row-waves, ran at 5.0).  Here the workgroup's 16 waves copy each block of rows ONCE,
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, the 1 KB a wave instruction moves lands lane-linear at M0), three
// or four stages deep, and every wave cuts its MFMA fragments out of the row-major image with ds_read_b64_tr_b16 (the 16 lanes of a
// group name a [4 rows][16 channels] block and receive one channel's four rows: two reads = the eight k-values of a lane).
// Bank conflicts: a 32-lane pass of a transpose read touches rows {r..r+3, r+8..r+11} of one 32-byte channel segment; the image
// keeps segment g of row r at position g ^ f(r), f(r) = (r & 3) + 4 (r >> 3 & 1) (128-byte rows: (r >> 1 & 1) + 2 (r >> 3 & 1), the
// row index itself supplying bank bit 2) — eight different 32-byte bank groups.  The swizzle is applied to the DMA's per-lane SOURCE
// addresses; the LDS side stays a linear copy.  One barrier per stage.  A last partial stage goes through guarded loads + ds_write
// (zero rows).  Slabs in NATURAL tile order (fold kind 2): o = 64 so + 16 a + 4 lk + r, c = 64 sc + 16 b + li.
// ----------------------------------------------------------------------------------------
struct WglParams {
    WgbParams b;
    int rows_stage, nst, stage_bytes;         // rows per stage (32 RW), stages, bytes per stage
    int nsrc;
    int off[WB_MAXSRC + 1];                   // byte offset of the x image (0) and of every gy image inside a stage
};

// uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit VGPR address arithmetic in the loop
__device__ __forceinline__ void wl_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint2 wl_tr(unsigned lds_addr) {
    typedef short wl_s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) wl_s16x4 lds_v4;
    const wl_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4*>(static_cast<uintptr_t>(lds_addr)));
    return __builtin_bit_cast(uint2, v);
}
template <int N> __device__ __forceinline__ void wl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wl_wait_vm_n(int n) {          // n = (stages still in flight: 0..2) x (this wave's DMA instructions per stage: 1..3)
    if (n >= 6) wl_wait_vm<6>();
    else if (n >= 4) wl_wait_vm<4>();
    else if (n == 3) wl_wait_vm<3>();
    else if (n == 2) wl_wait_vm<2>();
    else if (n == 1) wl_wait_vm<1>();
    else wl_wait_vm<0>();
}
// swizzle of the 32-byte segments of a row whose pitch is `segs` segments (4, 8 or 16)
__device__ __forceinline__ int wl_f(int row, int segs) {
    return segs == 4 ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) : ((row & 3) | (((row >> 3) & 1) << 2));
}

__global__ __launch_bounds__(WB_THREADS) void linear_wgrad_lds_kernel(const WglParams q) {
    const WgbParams& p = q.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int slice = wave % p.S, rw = wave / p.S;
    const int src = slice >= p.sb[2] ? 2 : (slice >= p.sb[1] ? 1 : 0);
    const bool live = slice < p.sb[WB_MAXSRC];
    const int Cout = src == 2 ? p.couts[2] : (src == 1 ? p.couts[1] : p.couts[0]);
    const int nso = Cout >> 6, sl = slice - (src == 2 ? p.sb[2] : (src == 1 ? p.sb[1] : p.sb[0]));
    const int so = sl % nso, sc = sl / nso;
    const int R = p.R, Cin = p.Cin;
    const unsigned lds0 = (unsigned)(uintptr_t)wb_lds;
    const int ni = q.stage_bytes >> 10;                        // 1 KB DMA instructions per stage: wave w issues w, w + 16, w + 32 (< ni)
    const int ndma = (ni - wave + 15) >> 4;                    // this wave's count (1..3; a stage need not be a multiple of 16 KB)

    // ---- this wave's share of a stage's copy: instructions wave, wave + 16, ... (1 KB each, inside ONE image) ----
    // (everything but the lane's byte offset is wave-uniform and lives in scalar registers; the offset is recomputed per
    //  instruction — a dozen VALU operations against 16 MFMAs per stage — instead of held: the kernel sits at the 128-VGPR limit, and
    //  ONE spilled register is fatal here: its scratch reload is a vector-memory load, in order behind the DMA just issued, so the
    //  wait for it is a wait for the whole prefetch — 125 us instead of 9x us for the 256 x 256 map with 10 spilled registers)
    const char* gbase[3];
    unsigned gpitch[3], ldst[3], gpc0[3], gshift[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gbase[j] = nullptr; gpitch[j] = 0; ldst[j] = 0; gpc0[j] = 0; gshift[j] = 0;
        if (j < ndma) {
            const int byte0 = (wave + 16 * j) * 1024;
            // (selects, not indexed reads: a dynamically indexed by-value argument struct is copied to scratch)
            const int img = (q.nsrc >= 3 && byte0 >= q.off[3]) ? 3 : ((q.nsrc >= 2 && byte0 >= q.off[2]) ? 2 : (byte0 >= q.off[1] ? 1 : 0));
            const int ch = img == 0 ? Cin : (img == 1 ? p.couts[0] : (img == 2 ? p.couts[1] : p.couts[2]));
            const int ioff = img == 0 ? 0 : (img == 1 ? q.off[1] : (img == 2 ? q.off[2] : q.off[3]));
            const int pitch = __builtin_amdgcn_readfirstlane(2 * ch);          // bytes per row: 128 / 256 / 512
            gbase[j] = (const char*)(img == 0 ? (const void*)p.x : (img == 1 ? (const void*)p.gys[0] : (img == 2 ? (const void*)p.gys[1] : (const void*)p.gys[2])));
            gpitch[j] = (unsigned)pitch;
            gshift[j] = pitch == 512 ? 5u : (pitch == 256 ? 4u : 3u);          // log2 of the 16-byte chunks per row
            gpc0[j] = (unsigned)((byte0 - ioff) >> 4);                          // first chunk of the instruction inside its image
            ldst[j] = (unsigned)byte0;
        }
    }
    auto lane_off = [&](int j) -> unsigned {           // byte offset of this lane's 16 bytes from the stage's first row of the image
        const unsigned pc = gpc0[j] + (unsigned)lane;
        const unsigned row = pc >> gshift[j], c = pc & ((1u << gshift[j]) - 1u);
        const unsigned col = (((c >> 1) ^ (unsigned)wl_f((int)row, (int)(gpitch[j] >> 5))) << 5) | ((c & 1u) << 4);
        return row * gpitch[j] + col;
    };
    // ---- fragment addresses inside a stage: tile t of the slice's 64 gy / x channels, rows 32 rw + 8 lk + (li >> 2) (+ 4) ----
    const int kr = 32 * rw + 8 * lk + (li >> 2);
    const int pg = 2 * Cout, px = 2 * Cin;
    unsigned aA[4], aB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        aA[t] = (unsigned)((src == 2 ? q.off[3] : (src == 1 ? q.off[2] : q.off[1])) + kr * pg + (((so * 4 + t) ^ wl_f(kr, pg >> 5)) << 5) + 8 * (li & 3));
        aB[t] = (unsigned)(kr * px + (((sc * 4 + t) ^ wl_f(kr, px >> 5)) << 5) + 8 * (li & 3));
    }

    f32x4 acc[4][4];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int blk, int stage) {
        const unsigned base = lds0 + (unsigned)(stage * q.stage_bytes);
        const int64_t row0 = (int64_t)blk * q.rows_stage;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < ndma) {
                const uint64_t sb = (uint64_t)(uintptr_t)gbase[j] + (uint64_t)row0 * gpitch[j];
                wl_dma16((const void*)(uintptr_t)sb, lane_off(j), __builtin_amdgcn_readfirstlane(base + ldst[j]));
            }
    };
    auto compute = [&](int stage) {
        const unsigned base = lds0 + (unsigned)(stage * q.stage_bytes);
        uint4 A[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint2 a0 = wl_tr(base + aA[t]), a1 = wl_tr(base + aA[t] + 4u * (unsigned)pg);
            A[t] = make_uint4(a0.x, a0.y, a1.x, a1.y);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {              // (one x fragment at a time: 16 + 4 operand registers beside the 64 accumulators)
            const uint2 b0 = wl_tr(base + aB[b]), b1 = wl_tr(base + aB[b] + 4u * (unsigned)px);
            const uint4 Bf = make_uint4(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wb_bf16x8, A[a]),
                                                                    __builtin_bit_cast(wb_bf16x8, Bf), acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) bs[a] = wb_sum8(A[a], bs[a]);
    };

    const int nfull = R / q.rows_stage, tail = R - nfull * q.rows_stage;
    const int G = gridDim.x, me = blockIdx.x;
    const int nb = me < nfull ? (nfull - me + G - 1) / G : 0;              // full stages of this workgroup: me, me + G, ...
    const int depth = q.nst - 1;
    for (int t = 0; t < depth && t < nb; ++t) issue(me + t * G, t);
#pragma nounroll
    for (int k = 0; k < nb; ++k) {
        const int ahead = min(depth - 1, nb - 1 - k);                        // later stages already on their way
        wl_wait_vm_n(ahead * ndma);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everybody's share of stage k landed; stage k - 1 is free
        if (k + depth < nb) issue(me + (k + depth) * G, (k + depth) % q.nst);
        if (live) compute(k % q.nst);
    }
    if (tail > 0 && me == nfull % G) {                                      // the last, partial stage: guarded loads, zero rows
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int64_t row0 = (int64_t)nfull * q.rows_stage;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < ndma) {
                const unsigned lo = lane_off(j);
                const int row = (int)(lo / gpitch[j]);
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < tail) v = *reinterpret_cast<const uint4*>(gbase[j] + row0 * gpitch[j] + lo);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(wb_lds) + ldst[j] + 16 * lane) = v;
            }
        __syncthreads();
        if (live) compute(0);
    }
    __syncthreads();                                                         // the stages are dead: the fold may use their memory

    // ---- binary-tree fold of the RW row-waves of each slice (fixed order), as in linear_wgrad_b16_kernel ----
    for (int half = p.RW >> 1; half >= 1; half >>= 1) {
        float* slot = wb_lds + ((int64_t)(slice * (p.RW >> 1) + (rw - half)) * WB_NACC) * 64 + lane;
        if (rw >= half && rw < 2 * half) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slot[((a * 4 + b) * 4 + r) * 64] = acc[a][b][r];
                slot[(64 + a) * 64] = bs[a];
            }
        }
        __syncthreads();
        if (rw < half) {
            const float* srcp = wb_lds + ((int64_t)(slice * (p.RW >> 1) + rw) * WB_NACC) * 64 + lane;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += srcp[((a * 4 + b) * 4 + r) * 64];
                bs[a] += srcp[(64 + a) * 64];
            }
        }
        __syncthreads();
    }
    if (rw == 0 && live) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bs[a] += __shfl_xor(bs[a], 16);
            bs[a] += __shfl_xor(bs[a], 32);
        }
        float* slab = p.ws + (((int64_t)blockIdx.x * p.S + slice) * WB_NACC) * 64 + lane;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[((a * 4 + b) * 4 + r) * 64] = acc[a][b][r];
            slab[(64 + a) * 64] = bs[a];
        }
    }
}

// gW[o][c] += sum over slabs;  gb[o] += sum over slabs.  One 1024-thread workgroup per 64 consecutive slab elements (one
// accumulator register of one slice, lanes 0..63: a 256-byte line per slab): wave w folds slabs w, w+16, ... — at most 16
// independent line loads, all in flight at once — and wave 0 folds the 16 partial sums in order.  Fixed order: deterministic.
// (The first form gave each 256-thread block 16 elements: 64-byte pieces, 24 us per call for 4 MB of slabs.)
template <bool NATURAL>      // NATURAL: the LDS-staged kernel's tile order (consecutive channels per tile), else the register-direct kernel's (stride 4)
__global__ __launch_bounds__(1024) void wgb_reduce_kernel(const float* __restrict__ ws, int nslab, int64_t slab_len, int nso,
                                                          int Cin, int Cout, float* __restrict__ gW,
                                                          float* __restrict__ gb) {
    __shared__ float part[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;      // slab_len = distance between slabs (a merged launch's slabs hold several maps' slices)
    const int64_t i = (int64_t)blockIdx.x * 64 + l;    // i < this map's S * WB_NACC * 64
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    {
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
    part[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][l];
    const int q = (int)((i >> 6) % WB_NACC), slice = (int)(i / (WB_NACC * 64));
    const int so = slice % nso, sc = slice / nso;
    const int li = l & 15, lk = l >> 4;
    if (q < 64) {
        const int r = q & 3, b = (q >> 2) & 3, a = q >> 4;            // D row = 4*lk + r (tile a), D col = li (tile b)
        const int o = NATURAL ? so * 64 + 16 * a + 4 * lk + r : so * 64 + 4 * (4 * lk + r) + a;
        const int c = NATURAL ? sc * 64 + 16 * b + li : sc * 64 + 4 * li + b;
        gW[(int64_t)o * Cin + c] += s;
    } else if (gb && sc == 0 && lk == 0) {
        gb[NATURAL ? so * 64 + 16 * (q - 64) + li : so * 64 + 4 * li + (q - 64)] += s;          // the main kernel already folded the 4 row-group lanes
    }
}

// ----------------------------------------------------------------------------------------
// one NARROW operand (<= 16 channels: the edge-feature / input maps 2|6|7 -> 64 and the edge-type head 64 -> 4)
// against a 64-channel one.  The narrow side is a single MFMA tile whose lane li carries channel li (8 two-byte
// loads per lane, zero for li >= C); the wide side uses the 8x4 in-register transpose above.  All 16 waves of a
// workgroup are row-waves.  NARROW_X: x is the narrow operand (gW [64][Cin]); else gy is (gW [Cout][64]).
// ----------------------------------------------------------------------------------------
#define WN_NACC 20       // 16 gW accumulators + 4 dbias partials per lane

__device__ __forceinline__ uint4 wb_pack_narrow(const unsigned (&v)[8]) {
    return make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

template <bool NARROW_X>
__global__ __launch_bounds__(WB_THREADS) void linear_wgrad_b16_narrow_kernel(const WgbParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int rw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int R = p.R, Cin = p.Cin, Cout = p.Cout;
    const int Cn = NARROW_X ? Cin : Cout, Cw = NARROW_X ? Cout : Cin;      // Cw: row stride of the wide operand (64 channels used)
    const uint16_t* wide = (NARROW_X ? p.gy : p.x) + 4 * li;
    const uint16_t* nar = (NARROW_X ? p.x : p.gy) + li;
    const bool nar_ok = li < Cn;

    f32x4 acc[4];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nblk = (R + 31) / 32;
    const int stride = gridDim.x * WB_WAVES;
    for (int blk = blockIdx.x * WB_WAVES + rw; blk < nblk; blk += stride) {
        uint2 rwd[8];
        unsigned rn[8];
        const int row0 = blk * 32 + 8 * lk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = row0 + j;
            const bool ok = row < R;
            rwd[j] = ok ? *reinterpret_cast<const uint2*>(wide + (int64_t)row * Cw) : make_uint2(0, 0);
            rn[j] = (ok && nar_ok) ? (unsigned)nar[(int64_t)row * Cn] : 0u;
        }
        const uint4 N = wb_pack_narrow(rn);
        uint4 Wd[4];
        Wd[0] = wb_pack<0>(rwd); Wd[1] = wb_pack<1>(rwd); Wd[2] = wb_pack<2>(rwd); Wd[3] = wb_pack<3>(rwd);
        if (NARROW_X) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {              // D[i][j]: o = 4 i + t, c = j
                bs[t] = wb_sum8(Wd[t], bs[t]);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wb_bf16x8, Wd[t]),
                                                                 __builtin_bit_cast(wb_bf16x8, N), acc[t], 0, 0, 0);
            }
        } else {
            bs[0] = wb_sum8(N, bs[0]);
#pragma unroll
            for (int t = 0; t < 4; ++t)                // D[i][j]: o = i, c = 4 j + t
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wb_bf16x8, N),
                                                                 __builtin_bit_cast(wb_bf16x8, Wd[t]), acc[t], 0, 0, 0);
        }
    }
    // tree fold of the 16 row-waves
    for (int half = WB_WAVES >> 1; half >= 1; half >>= 1) {
        if (rw >= half && rw < 2 * half) {
            float* slot = wb_lds + (rw - half) * WN_NACC * 64 + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[(t * 4 + r) * 64] = acc[t][r];
                slot[(16 + t) * 64] = bs[t];
            }
        }
        __syncthreads();
        if (rw < half) {
            const float* src = wb_lds + rw * WN_NACC * 64 + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += src[(t * 4 + r) * 64];
                bs[t] += src[(16 + t) * 64];
            }
        }
        __syncthreads();
    }
    if (rw == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bs[t] += __shfl_xor(bs[t], 16);
            bs[t] += __shfl_xor(bs[t], 32);
        }
        float* slab = p.ws + (int64_t)blockIdx.x * WN_NACC * 64 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(t * 4 + r) * 64] = acc[t][r];
            slab[(16 + t) * 64] = bs[t];
        }
    }
}

// woff: first channel of the wide operand's 64-channel group this launch covered
template <bool NARROW_X>
__global__ __launch_bounds__(1024) void wgn_reduce_kernel(const float* __restrict__ ws, int nslab, int Cin, int Cout,
                                                          int woff, float* __restrict__ gW, float* __restrict__ gb) {
    __shared__ float part[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int slab_len = WN_NACC * 64;
    const int i = blockIdx.x * 64 + l;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    {
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
    part[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][l];
    const int q = i >> 6, li = l & 15, lk = l >> 4;
    if (q < 16) {
        const int r = q & 3, t = q >> 2, row = 4 * lk + r;
        const int o = NARROW_X ? woff + 4 * row + t : row, c = NARROW_X ? li : woff + 4 * li + t;
        if (o < Cout && c < Cin) gW[(int64_t)o * Cin + c] += s;
    } else if (gb && lk == 0) {
        const int t = q - 16;
        if (NARROW_X) gb[woff + 4 * li + t] += s;
        else if (t == 0 && li < Cout && woff == 0) gb[li] += s;      // the narrow side's bias: once
    }
}

static bool wn_plan(int64_t R, int Cin, int Cout, bool* narrow_x, int* gx) {
    if (Cin <= 16 && Cout % 64 == 0 && Cout <= 256) *narrow_x = true;          // wide side: 64-channel groups, one launch each
    else if (Cout <= 16 && Cin % 64 == 0 && Cin <= 256) *narrow_x = false;
    else return false;
    const int64_t nblk = (R + 31) / 32;
    int64_t g = (nblk + 2 * WB_WAVES - 1) / (2 * WB_WAVES);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *gx = (int)g;
    return true;
}

static bool wb_plan(int64_t R, int Cin, int Cout, int* nso, int* S, int* RW, int* gx) {
    if (Cin % 64 || Cout % 64) return false;
    *nso = Cout / 64;
    *S = *nso * (Cin / 64);
    if (*S > 16 || (*S & (*S - 1))) return false;
    *RW = 16 / *S;
    const int wgs = 256;      // one workgroup per CU (128: no effect, profiles/r05/README.md)
    const int64_t nblk = (R + 31) / 32;
    int64_t g = (nblk + 2 * *RW - 1) / (2 * *RW);                      // >= 2 blocks per row-wave
    if (g > wgs) g = wgs;
    if (g < 1) g = 1;
    *gx = (int)g;
    return true;
}

int64_t fgnn_linear_wgrad_b16_workspace_bytes(int64_t R, int Cin, int Cout) {
    int nso, S, RW, gx;
    bool nx;
    if (wn_plan(R, Cin, Cout, &nx, &gx)) return (int64_t)gx * WN_NACC * 64 * 4;
    if (!wb_plan(R, Cin, Cout, &nso, &S, &RW, &gx)) return 0;
    return (int64_t)gx * S * WB_NACC * 64 * 4;
}

// nsrc gradient tensors against one x: live slices, S padded to a power of two <= 16, row-waves, grid
static bool wb_plan_multi(int64_t R, int Cin, int nsrc, const int32_t* couts, int* sb, int* S, int* RW, int* gx) {
    if (nsrc < 1 || nsrc > WB_MAXSRC || Cin % 64 || Cin < 64 || Cin > 256) return false;
    int live = 0;
    for (int s = 0; s < WB_MAXSRC; ++s) {
        sb[s] = live;
        if (s < nsrc) {
            if (couts[s] % 64 || couts[s] < 64 || couts[s] > 256) return false;
            live += (couts[s] / 64) * (Cin / 64);
        }
    }
    sb[WB_MAXSRC] = live;
    if (live > 16) return false;
    int sp = 1;
    while (sp < live) sp *= 2;
    *S = sp;
    *RW = 16 / sp;
    const int wgs = 256;      // one workgroup per CU (128: no effect, profiles/r05/README.md)
    const int64_t nblk = (R + 31) / 32;
    int64_t g = (nblk + 2 * *RW - 1) / (2 * *RW);                      // >= 2 blocks per row-wave
    if (g > wgs) g = wgs;
    if (g < 1) g = 1;
    *gx = (int)g;
    return true;
}

static int wb_launch(const void* x, int64_t R, int Cin, int nsrc, const void* const* gys, const int32_t* couts, float* const* gWs,
                     float* const* gbs, const int* sb, int S, int RW, int gx, void* workspace, hipStream_t st) {
    WgbParams p = {};
    p.x = (const uint16_t*)x; p.gy = (const uint16_t*)gys[0]; p.ws = (float*)workspace;
    p.R = (int)R; p.Cin = Cin; p.Cout = couts[0]; p.nso = couts[0] / 64; p.S = S; p.RW = RW;
    for (int s = 0; s < WB_MAXSRC; ++s) {
        p.gys[s] = (const uint16_t*)gys[s < nsrc ? s : 0];
        p.couts[s] = couts[s < nsrc ? s : 0];
        p.sb[s] = sb[s];
    }
    p.sb[WB_MAXSRC] = sb[WB_MAXSRC];
    const int lds = S * (RW / 2) * WB_NACC * 64 * 4;                   // 0 when every slice has one row-wave
    // wide maps (8 / 16 slices): rows staged ONCE per workgroup through LDS (linear_wgrad_lds_kernel)
    WglParams q = {};
    bool staged = false;
    static const bool no_lds = getenv("FGNN_WG_NOLDS") != nullptr;      // A/B switch: the register-direct kernel for every shape
    if (S >= 8 && !no_lds && R >= 2048) {
        int ctot = Cin;
        for (int s = 0; s < nsrc; ++s) ctot += couts[s];
        q.b = p; q.nsrc = nsrc;
        q.rows_stage = 32 * RW;
        q.stage_bytes = q.rows_stage * ctot * 2;
        q.nst = q.stage_bytes <= 32 * 1024 ? 4 : 3;
        q.off[0] = 0;
        q.off[1] = q.rows_stage * Cin * 2;
        for (int s = 1; s <= WB_MAXSRC; ++s) if (s < WB_MAXSRC) q.off[s + 1] = q.off[s] + (s - 1 < nsrc ? q.rows_stage * couts[s - 1] * 2 : 0);
        const bool aligned = ((uintptr_t)x & 15) == 0 && ((uintptr_t)gys[0] & 15) == 0 && (nsrc < 2 || ((uintptr_t)gys[1] & 15) == 0) &&
                             (nsrc < 3 || ((uintptr_t)gys[2] & 15) == 0);
        staged = aligned && q.stage_bytes % 1024 == 0 && q.stage_bytes >= 1024 * WB_WAVES && q.stage_bytes <= 3 * 1024 * WB_WAVES &&
                 q.nst * q.stage_bytes <= 150 * 1024 && q.nst * q.stage_bytes >= lds;
    }
    hipError_t e;
    if (staged) {
        void* fn = (void*)linear_wgrad_lds_kernel;
        const int bytes = q.nst * q.stage_bytes;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        void* args[] = {(void*)&q};
        e = hipLaunchKernel(fn, dim3(gx), dim3(WB_THREADS), args, bytes, st);
    } else {
        void* fn = (void*)linear_wgrad_b16_kernel;
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        void* args[] = {(void*)&p};
        e = hipLaunchKernel(fn, dim3(gx), dim3(WB_THREADS), args, lds, st);
    }
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad_b16 launch: %s", hipGetErrorString(e));
    const int64_t stride = (int64_t)S * WB_NACC * 64;                  // distance between the workgroups' slabs
    for (int s = 0; s < nsrc; ++s) {
        const int Ss = sb[s + 1] - sb[s], nso = couts[s] / 64;
        const int64_t len = (int64_t)Ss * WB_NACC * 64;
        const float* ws_s = p.ws + (int64_t)sb[s] * WB_NACC * 64;
        if (fgnn_fold_push(ws_s, gx, stride, len, gWs[s], gbs ? gbs[s] : nullptr, staged ? 2 : 1, Ss, nso, Cin, couts[s])) continue;      // recorded (fold_batch.hip)
        if (staged)
            hipLaunchKernelGGL(wgb_reduce_kernel<true>, dim3((unsigned)(len / 64)), dim3(1024), 0, st, ws_s, gx, stride, nso, Cin, couts[s], gWs[s],
                               gbs ? gbs[s] : nullptr);
        else
            hipLaunchKernelGGL(wgb_reduce_kernel<false>, dim3((unsigned)(len / 64)), dim3(1024), 0, st, ws_s, gx, stride, nso, Cin, couts[s], gWs[s],
                               gbs ? gbs[s] : nullptr);
    }
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad_b16 reduce launch: %s", hipGetErrorString(e));
    return 1;
}

// The weight / bias gradients of nsrc <= 3 node-wise maps that read the SAME rows x [R][Cin] (the maps consuming one layer state:
// conv1 of the blocks' heads, the state's own v2v / f2f map — /root/reference/lib/model/mpnn/factor_mpnn_sp.py:136-168) in ONE
// pass: gW_s [couts[s]][Cin] += gy_s^T x, gb_s [couts[s]] += column sums of gy_s (gb_s may be NULL).  bf16, channel counts in
// multiples of 64 up to 256, sum_s (couts[s] / 64) (Cin / 64) <= 16.  x is read once instead of nsrc times.
extern "C" int64_t fgnn_linear_wgrad_multi_workspace_bytes(int64_t R, int32_t Cin, int32_t nsrc, const int32_t* couts) {
    int sb[WB_MAXSRC + 1], S, RW, gx;
    if (R <= 0 || R > 0x7fffffff || !couts || !wb_plan_multi(R, Cin, nsrc, couts, sb, &S, &RW, &gx)) return -1;
    return (int64_t)gx * S * WB_NACC * 64 * 4;
}

extern "C" int fgnn_linear_wgrad_multi(const void* x, int64_t R, int32_t Cin, int32_t nsrc, const void* const* gy, const int32_t* couts,
                                       float* const* gW, float* const* gb, void* workspace, int64_t workspace_bytes,
                                       fgnn_stream_t stream) {
    int sb[WB_MAXSRC + 1], S, RW, gx;
    if (!x || !gy || !couts || !gW || !workspace) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: null pointer");
    if (R <= 0 || R > 0x7fffffff) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: bad sizes");
    if (!wb_plan_multi(R, Cin, nsrc, couts, sb, &S, &RW, &gx))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "linear_wgrad_multi: Cin=%d with %d sources is outside the kernel's family", Cin, nsrc);
    for (int s = 0; s < nsrc; ++s) {
        if (!gy[s] || !gW[s]) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: null pointer (source %d)", s);
        if ((uintptr_t)gy[s] & 7) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: gy[%d] not 8-byte aligned", s);
    }
    if ((uintptr_t)x & 7) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: x not 8-byte aligned");
    if (workspace_bytes < (int64_t)gx * S * WB_NACC * 64 * 4) FGNN_FAIL(FGNN_EINVAL, "linear_wgrad_multi: workspace too small");
    const int rc = wb_launch(x, R, Cin, nsrc, gy, couts, gW, gb, sb, S, RW, gx, workspace, (hipStream_t)stream);
    return rc < 0 ? rc : FGNN_OK;
}

// Returns 1 if launched, 0 if the shape is outside this kernel's family, <0 on error.
int fgnn_linear_wgrad_b16(const void* x, const void* gy, int64_t R, int Cin, int Cout, float* gW, float* gb,
                          void* workspace, int64_t workspace_bytes, fgnn_stream_t stream) {
    int nso, S, RW, gx;
    bool nx;
    if (wn_plan(R, Cin, Cout, &nx, &gx)) {
        if (((uintptr_t)(nx ? gy : x) & 7) || ((uintptr_t)(nx ? x : gy) & 1)) return 0;
        if (workspace_bytes < (int64_t)gx * WN_NACC * 64 * 4) return 0;
        WgbParams p;
        p.x = (const uint16_t*)x; p.gy = (const uint16_t*)gy; p.ws = (float*)workspace;
        p.R = (int)R; p.Cin = Cin; p.Cout = Cout; p.nso = 1; p.S = 1; p.RW = WB_WAVES;
        const int lds = (WB_WAVES / 2) * WN_NACC * 64 * 4;
        hipStream_t st = (hipStream_t)stream;
        const int wide_c = nx ? Cout : Cin;
        for (int woff = 0; woff < wide_c; woff += 64) {     // the kernel reads 64 channels from the pointer, rows at the full stride
            if (nx) {
                p.gy = (const uint16_t*)gy + woff;
                hipLaunchKernelGGL(linear_wgrad_b16_narrow_kernel<true>, dim3(gx), dim3(WB_THREADS), lds, st, p);
                hipLaunchKernelGGL(wgn_reduce_kernel<true>, dim3(WN_NACC), dim3(1024), 0, st, p.ws, gx, Cin, Cout, woff, gW, gb);
            } else {
                p.x = (const uint16_t*)x + woff;
                hipLaunchKernelGGL(linear_wgrad_b16_narrow_kernel<false>, dim3(gx), dim3(WB_THREADS), lds, st, p);
                hipLaunchKernelGGL(wgn_reduce_kernel<false>, dim3(WN_NACC), dim3(1024), 0, st, p.ws, gx, Cin, Cout, woff, gW, gb);
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "linear_wgrad_b16 narrow launch: %s", hipGetErrorString(e));
        return 1;
    }
    if (!wb_plan(R, Cin, Cout, &nso, &S, &RW, &gx)) return 0;
    if (((uintptr_t)x & 7) || ((uintptr_t)gy & 7)) return 0;
    if (workspace_bytes < (int64_t)gx * S * WB_NACC * 64 * 4) return 0;
    const int sb[WB_MAXSRC + 1] = {0, S, S, S};
    const void* gys[1] = {gy};
    const int32_t couts[1] = {Cout};
    float* gWs[1] = {gW};
    float* gbs[1] = {gb};
    return wb_launch(x, R, Cin, 1, gys, couts, gWs, gbs, sb, S, RW, gx, workspace, (hipStream_t)stream);
}
