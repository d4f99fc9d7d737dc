// block_tail.hip — TRAINING-mode fusion of the tail of `mp_conv_residual`: everything behind the message
// operator,
//
//     a2 = act2(BN2(e))            e [R][64] = the operator's aggregate + bias, R = B * M destination rows (bf16)
//     z3 = W2 a2 + b2              the block's second 1x1 map, 64 -> Cout in {64, 128, 256}
//     out = act3(BN3(z3)) + addends
//
// (/root/reference/lib/model/mpnn/mp_nn.py:165-175 epilogue, mp_nn_residual.py:31-35,49-51 conv2 + BatchNorm + LeakyReLU),
// WITHOUT ever writing the Cout-wide pre-BatchNorm tensor z3 (or, in the backward, reading it back): batch-statistics
// BatchNorm needs a grid-wide reduction between producing z3 and normalising it, so the staged path stores z3 (bf16),
// re-reads it to normalise, and in the backward re-reads it twice more and writes / re-reads its gradient three times —
// 10 passes over the widest tensor of the block.  The K = 64 product is cheap on the matrix cores (13 GFLOP for the widest
// call, ~6 us at a third of the bf16 peak), so every pass that needs z3 RECOMPUTES it from the 64-channel e instead:
//
//   MODE 0 "stats"   : e -> a2 -> z3 -> per-workgroup (sum z3, sum z3^2) partials          (reads 128 bytes per row, writes nothing)
//   MODE 1 "apply"   : e -> a2 -> z3 -> out = act3(z3 s3 + t3) + addends                    (z3 never stored; optionally stores
//                                                                                            a2 for the weight-gradient kernel)
//   MODE 2 "reduce"  : e, gout -> z3 -> g' = gout act3'(.) -> partials (sum g', sum g' z3)  (BatchNorm3's backward sums)
//   MODE 3 "grad"    : e, gout -> z3 -> gz3 = s3 g' + A z3 + B (BatchNorm3's input gradient, closed form per channel)
//                      -> ga2 = gz3 W2 (64 channels); gz3 itself is stored (bf16) only for callers that still run conv2's weight
//                      gradient as its own kernel
//
// conv2's WEIGHT GRADIENT without a stored gz3 / a2 (round 6).  gW2[o][c] = sum_rows gz3[row][o] a2[row][c], and gz3 is affine in
// quantities the "reduce" pass already holds: with acc = z3 - b2 = W2 a2 (bf16 operands, as every mode computes it),
//     gz3 = s3 g' + A acc + K,   K = A b2 + Bc        =>       gW2 = diag(s3) M1 + diag(A) (W2b Gram) + K v^T
//     M1[o][c] = sum_rows g'[row][o] a2[row][c]     Gram[c'][c] = sum_rows a2[row][c'] a2[row][c]     v[c] = sum_rows a2[row][c]
// (sum acc a2^T = W2b Gram exactly, W2b = the bf16 W2 the matrix cores see).  MODE 2 with NA == 1 accumulates the three moments
// beside BatchNorm3's sums: per 16-row tile a wave writes its g' and a2 fragments to a private 4 KB LDS tile (row-major, 32-byte
// segments swizzled by (row >> 1) & 3) and reads them back TRANSPOSED (ds_read_b64_tr_b16: lane = channel, four rows) as the operands
// of v_mfma_f32_16x16x16_bf16 with K = rows: 16 MFMAs for its slab of M1, 16 / CG for its share of Gram.  The constants A, Bc only
// exist after the pass's grid-wide sums — which is why the staged form needed gz3 in memory — but the moments do not depend on them:
// block_tail_wgrad_combine_kernel applies them to the FOLDED moments (a [Cout][64] problem).  Gone per block: gz3 written (R Cout
// bf16) and read back, a2 written by the forward and read back, one weight-gradient launch with its slabs; conv2's bias gradient is
// identically zero in front of a batch-statistics BatchNorm (sum gz3 = 0) and is no longer formed from rounding noise.
//
// Layout trick shared by all modes (as csrc/linear_fwd_b16.hip): the product is computed TRANSPOSED, D[i = out channel]
// [j = row] = W2 a2^T, with the MFMA row <-> channel permutation that hands lane (row, lk) the 16 CONSECUTIVE channels
// 16 lk .. 16 lk + 15 of its row and 64-channel slab: the e operand is one contiguous 32-byte run of the row per lane,
// gout / addends / out are contiguous 32-byte runs, and the D fragment is — after the per-element BatchNorm algebra and a
// bf16 pack — directly the B operand of the next product (ga2^T = W2^T gz3^T), no cross-lane movement anywhere.
// A wave owns 16-row tiles and walks all Cout / 64 slabs of its tile (W2 / W2^T fragments and the per-channel constants
// come from LDS, stored in fragment order: conflict-free 16-byte reads; the stream stays HBM-bound).
#include "fgnn_common.h"
#include "fgnn_gridfold.h"
#include <stdlib.h>

#define BT_THREADS 256
#define BT_WAVES 4
#define BT_MAXGRID 1024      // = BN_MAXPART of bnact.hip: the statistics partials use its workspace layout and finalisers

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));

struct BtParams {
    const uint16_t* e;       // [R][64] bf16
    const float* s2;         // [64] BatchNorm2 scale  (gamma * invstd)
    const float* t2;         // [64] BatchNorm2 shift  (beta - mean * scale)
    const float* W2;         // [Cout][64] f32
    const float* b2;         // [Cout] or NULL
    const float* s3;         // [Cout] BatchNorm3 scale                         (modes 1, 2, 3)
    const float* t3;         // [Cout] BatchNorm3 shift                         (modes 1, 2, 3)
    const float* ga;         // [Cout] A  = -s3 k2 invstd3                      (mode 3)
    const float* gb;         // [Cout] Bc = -s3 k1 + s3 k2 invstd3 mean3        (mode 3)
    const uint16_t* gout;    // [R][Cout] bf16 upstream gradient                (modes 2, 3)
    const uint16_t* add0;    // [R][Cout] addends of the output or NULL         (mode 1)
    const uint16_t* add1;
    const uint16_t* add2;
    uint16_t* out;           // mode 1: out [R][Cout];  mode 3: gz3 [R][Cout]
    uint16_t* out2;          // mode 1: a2 [R][64] or NULL;  mode 3: ga2 [R][64]
    float* part;             // [grid][2][Cout] partial sums                    (modes 0, 2)
    float* part2;            // [grid][2][64] BatchNorm2's backward sums (sum g2', sum g2' e), or NULL      (mode 3)
    int R, Cout;
    float slope2, slope3;
    int aperiod[3];          // mode 1: rows of out per addend row (1 = a tensor of out's shape; m = a per-sample row broadcast over m nodes)
    // the reducing modes finalise their own sums in the last workgroup (fgnn_gridfold.h) when fold.tickets != NULL:
    FgnnFold fold;
    fgnn_bn_final fin;       // mode 0: BatchNorm3's forward statistics (shift_k = b2)
    const float* mean3; const float* invstd3; const float* gamma3;     // mode 2: -> A, Bc [Cout] (the grad pass's constants), gweight3 / gbias3 +=
    float* A; float* Bc; float* gweight3; float* gbias3;
    const float* mean2; const float* invstd2;                           // mode 3: BatchNorm2's sums -> dsum2 [2][64], gweight2 / gbias2 +=
    float* dsum2; float* gweight2; float* gbias2;
    float* mom;              // mode 2 with NA == 1 (round 6): per (workgroup, row group) slabs [Cout*64 + 64*64 + 64] of the MOMENTS conv2's weight
                             // gradient is a closed form of (M1 = sum g' a2^T, Gram = sum a2 a2^T, v = sum a2): see block_tail_wgrad_* below
};

extern __shared__ __attribute__((aligned(16))) unsigned char bt_lds[];

__device__ __forceinline__ unsigned bt_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
template <typename T> __device__ __forceinline__ const T* bt_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float bt_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bt_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void bt_unpack8(const uint4 q, float (&v)[8]) {
    v[0] = bt_lo(q.x); v[1] = bt_hi(q.x); v[2] = bt_lo(q.y); v[3] = bt_hi(q.y);
    v[4] = bt_lo(q.z); v[5] = bt_hi(q.z); v[6] = bt_lo(q.w); v[7] = bt_hi(q.w);
}
__device__ __forceinline__ float bt_act(float v, float slope) { return v > 0.f ? v : v * slope; }
// Scalar f32 VALU only in this file (and -fno-slp-vectorize in the Makefile): beside a bf16 MFMA stream a packed-f32 op
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) costs ~22 extra cycles (MI355X_MICROARCH.md, per-instruction constants) — the
// first version of the statistics pass, written with packed pairs, ran at 1 000 cycles per tile and slab instead of ~250.
// Leaky ReLU for 0 <= slope <= 1 as max(v, slope v) (no compare / select); Z = the slope is 0: plain ReLU, one instruction.
template <bool Z> __device__ __forceinline__ float bt_act2(float v, float slope) { return Z ? fmaxf(v, 0.f) : fmaxf(v, v * slope); }

// CG = Cout / 64 slabs, NA = addends (mode 1), SL2Z = the first activation is a plain ReLU (slope2 == 0)
template <int MODE, int CG, int NA, bool SL2Z>
__global__ __launch_bounds__(BT_THREADS, MODE == 1 ? 3 : 2) void block_tail_kernel(const BtParams p) {
    constexpr int COUT = 64 * CG;
    constexpr int NCST = MODE == 0 ? 1 : (MODE == 3 ? 4 : 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int R = p.R;
    // W2 / W2^T live in LDS in FRAGMENT order — one 16-byte slot per (slab, tile, k-step, lane) — so that a wave's operand
    // read is 64 consecutive slots: conflict-free whatever the row permutation (a [row][k] image puts rows 16 apart, which
    // the permutation hands to one ds_read_b128 lane group, on the same banks)
    uint4* Wf = reinterpret_cast<uint4*>(bt_lds);                                        // [CG][4 ot][2 ks][64]   z3^T = W2 a2^T
    uint4* Tf = Wf + CG * 8 * 64;                                                        // [CG][4 t][2 s][64]     ga2^T = W2^T gz3^T (mode 3)
    float* cst = reinterpret_cast<float*>(Tf + (MODE == 3 ? CG * 8 * 64 : 0));           // [NCST][COUT]
    float* red = cst + NCST * COUT;                                                      // [BT_WAVES][2][COUT] (modes 0, 2)
    for (int f = tid; f < CG * 8 * 64; f += BT_THREADS) {
        const int fl = f & 63, fs = (f >> 6) & 1, ft = (f >> 7) & 3, sl = f >> 9;
        const int fi = fl & 15, fk = fl >> 4;
        const int pr = 16 * (fi >> 2) + (fi & 3) + 4 * ft;
        {   // A[i = out channel 64 sl + pr][k = input channels 16 fk + 8 fs .. + 7]
            const float* wp = p.W2 + (int64_t)(64 * sl + pr) * 64 + 16 * fk + 8 * fs;
            const f32x4 a = *reinterpret_cast<const f32x4*>(wp), b = *reinterpret_cast<const f32x4*>(wp + 4);
            Wf[f] = make_uint4(bt_pack2(a[0], a[1]), bt_pack2(a[2], a[3]), bt_pack2(b[0], b[1]), bt_pack2(b[2], b[3]));
        }
        if (MODE == 3) {   // A[i = input channel pr][k = out channels 64 sl + 16 fk + 8 fs .. + 7]
            const float* wp = p.W2 + (int64_t)(64 * sl + 16 * fk + 8 * fs) * 64 + pr;
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = wp[u * 64];
            Tf[f] = make_uint4(bt_pack2(w[0], w[1]), bt_pack2(w[2], w[3]), bt_pack2(w[4], w[5]), bt_pack2(w[6], w[7]));
        }
    }
    for (int o = tid; o < COUT; o += BT_THREADS) {
        const float b = p.b2 ? p.b2[o] : 0.f;
        if (MODE == 0) cst[o] = b;
        else {
            const float s = p.s3[o];
            cst[o] = s;
            cst[COUT + o] = fmaf(b, s, p.t3[o]);                 // pre3 = (acc + b) s3 + t3 = acc s3 + (b s3 + t3)
            if (MODE == 3) {
                const float a = p.ga[o];
                cst[2 * COUT + o] = a;
                cst[3 * COUT + o] = fmaf(a, b, p.gb[o]);         // gz3 = s3 g' + A (acc + b) + Bc
            }
        }
    }
    // BatchNorm2 affine of this lane's 16 input channels (16 lk .. 16 lk + 15)
    // (mode 3 is short of registers: there the affine is re-read from LDS every tile — 8 broadcast reads)
    constexpr bool S2REG = MODE != 3 && !(MODE == 2 && NA == 1);        // (the moments variant of mode 2 needs its registers for 80 - 128 accumulators)
    float* s2l = red + ((MODE == 0 || MODE == 2) ? BT_WAVES * 2 * COUT : 0);             // [2][64] (mode 3)
    float s2[S2REG ? 16 : 1], t2[S2REG ? 16 : 1];
    if (S2REG) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { s2[c] = p.s2[16 * lk + c]; t2[c] = p.t2[16 * lk + c]; }
    } else if (tid < 64) { s2l[tid] = p.s2[tid]; s2l[64 + tid] = p.t2[tid]; }
    __syncthreads();

    // MFMA row i of output tile ot  <->  channel 16 (i >> 2) + 4 ot + (i & 3) of the slab: D then gives lane (row li, lk)
    // the channels 16 lk + 4 ot + r; k-step ks, k-group lk  <->  input channels 16 lk + 8 ks .. + 7.
    // Modes 0 / 1 / 2 have no dependency across slabs: a wave is BOUND to one slab (its W2 fragments and per-channel
    // constants stay in registers, the CG waves of a row group share the e rows through L1); mode 3 sums ga2 over the
    // slabs, so there a wave walks all of them with the fragments coming from LDS.
    constexpr bool BOUND = MODE != 3;                       // (mode 0 walking all slabs — one a2 prologue per tile instead of one per
                                                            // tile and slab — measured SLOWER: 60 vs 46 us at Cout 256, 393 k rows: fewer,
                                                            // fatter waves expose the MFMA -> statistics dependency)
    const int cg = BOUND ? wave % CG : 0, rg = BOUND ? wave / CG : wave;
    constexpr int NRG = BOUND ? BT_WAVES / CG : BT_WAVES;
    constexpr bool WREG = BOUND && !(MODE == 2 && NA == 1);      // (the moments variant of mode 2: fragments and constants from LDS per tile, as mode 3)
    uint4 aW[WREG ? 4 : 1][2];
    f32x4 c0[WREG ? 4 : 1], c1[WREG ? 4 : 1];
    if (WREG) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            aW[ot][0] = Wf[((cg * 4 + ot) * 2 + 0) * 64 + lane];
            aW[ot][1] = Wf[((cg * 4 + ot) * 2 + 1) * 64 + lane];
            c0[ot] = *reinterpret_cast<const f32x4*>(cst + 64 * cg + 16 * lk + 4 * ot);
            c1[ot] = MODE == 0 ? c0[ot] : *reinterpret_cast<const f32x4*>(cst + COUT + 64 * cg + 16 * lk + 4 * ot);
        }
    }
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) { s0[ot] = (f32x4){0.f, 0.f, 0.f, 0.f}; s1[ot] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // moments of conv2's weight gradient (mode 2, NA == 1): this wave's slab of M1, its share of Gram (NTC column tiles), v
    constexpr bool MOM = MODE == 2 && NA == 1;
    constexpr int NTC = MOM ? 4 / CG : 1;
    f32x4 m1[MOM ? 4 : 1][MOM ? 4 : 1], gr[MOM ? 4 : 1][NTC];
    float vs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < (MOM ? 4 : 1); ++a) {
#pragma unroll
        for (int b = 0; b < (MOM ? 4 : 1); ++b) m1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NTC; ++b) gr[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the wave's transposition tiles: g' [16 rows][64 o] and a2 [16 rows][64 c] bf16, behind the statistics fold's rows
    const unsigned mom_tile = (unsigned)(CG * 8 * 64 * 16 + NCST * COUT * 4 + BT_WAVES * 2 * COUT * 4 + 512) + (unsigned)wave * 4096u;      // byte offset in bt_lds (behind s2l)
    const unsigned mom_w = mom_tile + (unsigned)(li * 128 + ((lk ^ ((li >> 1) & 3)) << 5));                 // this lane's 32 bytes of row li (segment lk)
    const int mom_kr = 4 * lk + (li >> 2);                                                                  // transpose reads: key row of lane (group lk, i = li)
    const unsigned mom_r = (unsigned)(uintptr_t)bt_lds + mom_tile + (unsigned)(mom_kr * 128 + 8 * (li & 3)), mom_f = (unsigned)((mom_kr >> 1) & 3);

    const int ntile = (R + 15) / 16;
    const int stride = gridDim.x * NRG;
    // Every tile's global loads are asked for DEPTH tiles ahead (a ring of register slots): a wave that only ever has the
    // next tile in flight runs at one memory latency per tile (the statistics pass, 2 KB per tile, measured 36 us for 50 MB).
    // The loop body is STRAIGHT-LINE code — loads are unconditional (rows past the end are clamped to the last row and masked
    // where they are used, a wave's tile count is rounded up to a multiple of DEPTH): with loads under exec-masked branches
    // the compiler's s_waitcnt placement falls back to vmcnt(0) at every merge and the ring buys nothing.
    constexpr int DEPTH = MODE == 0 ? 4 : ((MODE == 2 && NA != 1) ? 2 : 1);
    const int first = blockIdx.x * NRG + rg;
    const int mine = first < ntile ? (ntile - first + stride - 1) / stride : 0;           // tiles of this wave
    // (uniform 64-bit base of the tile + a 32-bit lane offset: per-lane 64-bit pointers were what the moments variant spilled, DESIGN 4.14;
    // rows past the end are clamped to the last row, tiles past the end to the last tile — masked where they are used)
    auto load_e = [&](int it, uint4 (&q)[2]) {
        const int tl = min(first + it * stride, ntile - 1);
        const int rl = min(li, R - 1 - tl * 16);
        const uint16_t* eb = p.e + (int64_t)tl * 16 * 64;
        const unsigned off = (unsigned)((rl * 64 + 16 * lk) * 2);
        q[0] = *bt_at<uint4>(eb, off);
        q[1] = *bt_at<uint4>(eb, off + 16u);
    };
    auto load_g = [&](int it, uint4 (&gp)[2]) {           // a slab-bound wave knows its slab: the upstream gradient rides along (mode 2)
        const int tl = min(first + it * stride, ntile - 1);
        const int rl = min(li, R - 1 - tl * 16);
        const uint16_t* gb = p.gout + (int64_t)tl * 16 * COUT;
        const unsigned off = (unsigned)((rl * COUT + 64 * cg + 16 * lk) * 2);
        gp[0] = *bt_at<uint4>(gb, off);
        gp[1] = *bt_at<uint4>(gb, off + 16u);
    };
    // A slot is refilled right AFTER its last use, into the same registers: a refill issued while the old value is still
    // live gets registers of its own and a copy at the loop's back edge — a copy that has to wait for the load.
    uint4 ring_e[DEPTH][2], ring_g[MODE == 2 ? DEPTH : 1][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                     // (issued in slot order, like the refills: the loop header's wait is the
        load_e(d, ring_e[d]);                             //  worst case over both ways into the loop)
        if (MODE == 2) load_g(d, ring_g[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int base = 0; base < mine; base += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int it = base + d;
        const int tile = first + it * stride;
        const int row = tile * 16 + li;
        const bool ok = it < mine && row < R;
        const bool partial = it >= mine || tile * 16 + 15 >= R;     // wave-uniform
        if (MODE == 3) asm volatile("" ::: "memory");       // LDS operands are re-read per tile, not hoisted into (spilled) registers
        uint4 (&eq)[2] = ring_e[d];
        const uint4 zq4 = make_uint4(0, 0, 0, 0);
        const int crow = min(row, R - 1);                   // (address arithmetic of masked rows stays inside the tensors)
        // ---- a2 = act2(e s2 + t2), packed to bf16: the B operand of z3^T = W2 a2^T (and the tensor the staged path stores) ----
        bt_bf16x8 a2f[2];
        unsigned pos2 = 0u;                                // bit c: pre2 > 0 for channel 16 lk + c of this row (mode 3)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned w[4] = {eq[ks].x, eq[ks].y, eq[ks].z, eq[ks].w};
            unsigned o[4];
            f32x4 sl4[2], tl4[2];
            if (!S2REG) {
                sl4[0] = *reinterpret_cast<const f32x4*>(s2l + 16 * lk + 8 * ks); sl4[1] = *reinterpret_cast<const f32x4*>(s2l + 16 * lk + 8 * ks + 4);
                tl4[0] = *reinterpret_cast<const f32x4*>(s2l + 64 + 16 * lk + 8 * ks); tl4[1] = *reinterpret_cast<const f32x4*>(s2l + 64 + 16 * lk + 8 * ks + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = 8 * ks + 2 * u;
                const float sc0 = S2REG ? s2[S2REG ? c : 0] : sl4[u >> 1][2 * (u & 1)], sc1 = S2REG ? s2[S2REG ? c + 1 : 0] : sl4[u >> 1][2 * (u & 1) + 1];
                const float sh0 = S2REG ? t2[S2REG ? c : 0] : tl4[u >> 1][2 * (u & 1)], sh1 = S2REG ? t2[S2REG ? c + 1 : 0] : tl4[u >> 1][2 * (u & 1) + 1];
                const float v0 = bt_act2<SL2Z>(fmaf(bt_lo(w[u]), sc0, sh0), p.slope2), v1 = bt_act2<SL2Z>(fmaf(bt_hi(w[u]), sc1, sh1), p.slope2);
                o[u] = bt_pack2(v0, v1);
                if (MODE == 3) pos2 |= (v0 > 0.f ? 1u : 0u) << c | (v1 > 0.f ? 1u : 0u) << (c + 1);
            }
            uint4 q = make_uint4(o[0], o[1], o[2], o[3]);
            if (partial && !ok) q = make_uint4(0, 0, 0, 0);      // rows past the end contribute nothing to any sum (last tile only)
            a2f[ks] = __builtin_bit_cast(bt_bf16x8, q);
            if (MODE == 1 && p.out2 && ok && cg == 0) *reinterpret_cast<uint4*>(p.out2 + (int64_t)row * 64 + 16 * lk + 8 * ks) = q;
        }
        if (MODE != 3) {                                    // e of this slot is consumed: ask for the tile DEPTH ahead (pinned here)
            load_e(it + DEPTH, ring_e[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 g2acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) g2acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {           // sums of z3 - b2 (the finaliser adds the bias back: K = b2); rows past the end are zero columns
            f32x4 acc[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, aW[WREG ? ot : 0][ks]), a2f[ks], acc[ot], 0, 0, 0);
            }
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s0[ot][r] += acc[ot][r];
                    s1[ot][r] = fmaf(acc[ot][r], acc[ot][r], s1[ot][r]);
                }
        }
#pragma unroll 1
        for (int sl = BOUND ? cg : 0; sl < (MODE == 0 ? 0 : (BOUND ? cg + 1 : CG)); ++sl) {
            const int o_base = 64 * sl;
            // the slab's upstream gradient / addends: 32 contiguous bytes of this lane's row each, asked for before the product
            uint4 gq[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            uint4 aq[3][2];
            const int64_t eoff = (int64_t)crow * COUT + o_base + 16 * lk;
            if (MODE == 2) { gq[0] = ok ? ring_g[d][0] : zq4; gq[1] = ok ? ring_g[d][1] : zq4; }
            if (MODE == 3) {
                const uint4 g0 = *reinterpret_cast<const uint4*>(p.gout + eoff), g1 = *reinterpret_cast<const uint4*>(p.gout + eoff + 8);
                gq[0] = ok ? g0 : zq4;
                gq[1] = ok ? g1 : zq4;
            }
            if (MODE == 1) {
                const uint16_t* const ads[3] = {p.add0, p.add1, p.add2};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    aq[a][0] = aq[a][1] = zq4;
                    if (a < NA) {
                        const int64_t aoff = p.aperiod[a] > 1 ? (int64_t)(crow / p.aperiod[a]) * COUT + o_base + 16 * lk : eoff;
                        aq[a][0] = *reinterpret_cast<const uint4*>(ads[a] + aoff); aq[a][1] = *reinterpret_cast<const uint4*>(ads[a] + aoff + 8);
                    }
                }
            }
            f32x4 acc[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint4 a = WREG ? aW[WREG ? ot : 0][ks] : Wf[((sl * 4 + ot) * 2 + ks) * 64 + lane];
                    acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, a), a2f[ks], acc[ot], 0, 0, 0);
                }
            }
            // acc[ot][r] = z3 - b2 of channel o_base + 16 lk + 4 ot + r, row li
            const float* cp = cst + o_base + 16 * lk;
            if (MODE == 0) {
            } else if (MODE == 1) {
                float y[16];
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * ot + r] = bt_act2<false>(fmaf(acc[ot][r], c0[WREG ? ot : 0][r], c1[WREG ? ot : 0][r]), p.slope3);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    float v[8];
                    bt_unpack8(aq[a][0], v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] += v[u];
                    bt_unpack8(aq[a][1], v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[8 + u] += v[u];
                }
                if (ok) {
                    uint16_t* op = p.out + eoff;
                    *reinterpret_cast<uint4*>(op) = make_uint4(bt_pack2(y[0], y[1]), bt_pack2(y[2], y[3]), bt_pack2(y[4], y[5]), bt_pack2(y[6], y[7]));
                    *reinterpret_cast<uint4*>(op + 8) = make_uint4(bt_pack2(y[8], y[9]), bt_pack2(y[10], y[11]), bt_pack2(y[12], y[13]), bt_pack2(y[14], y[15]));
                }
            } else {
                float g[16];
                {
                    float v[8];
                    bt_unpack8(gq[0], v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = v[u];
                    bt_unpack8(gq[1], v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[8 + u] = v[u];
                }
                if (MODE == 2) {
                    load_g(it + DEPTH, ring_g[d]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float gz[16];
                float gp16[MOM ? 16 : 1];
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    f32x4 s, t, ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
                    if (MODE == 2 && WREG) { s = c0[WREG ? ot : 0]; t = c1[WREG ? ot : 0]; }
                    else if (MODE == 2) {
                        s = *reinterpret_cast<const f32x4*>(cp + 4 * ot);
                        t = *reinterpret_cast<const f32x4*>(cp + COUT + 4 * ot);
                    } else {
                        s = *reinterpret_cast<const f32x4*>(cp + 4 * ot);
                        t = *reinterpret_cast<const f32x4*>(cp + COUT + 4 * ot);
                        ca = *reinterpret_cast<const f32x4*>(cp + 2 * COUT + 4 * ot);
                        cb = *reinterpret_cast<const f32x4*>(cp + 3 * COUT + 4 * ot);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = fmaf(acc[ot][r], s[r], t[r]);
                        const float ge = pre > 0.f ? g[4 * ot + r] : g[4 * ot + r] * p.slope3;
                        if (MODE == 2) {
                            s0[ot][r] += ge;                                       // rows >= R carry g = 0
                            s1[ot][r] = fmaf(ge, acc[ot][r], s1[ot][r]);
                            if (MOM) gp16[MOM ? 4 * ot + r : 0] = ge;
                        } else gz[4 * ot + r] = fmaf(s[r], ge, fmaf(ca[r], acc[ot][r], cb[r]));
                    }
                }
                if (MOM) {
                    // g' (channels 64 cg + 16 lk .. + 15 of row li) and a2 (channels 16 lk .. + 15) -> the wave's LDS tiles; back transposed
                    typedef short bt_s16x4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) bt_s16x4 lds_s4;
                    uint4* wg = reinterpret_cast<uint4*>(bt_lds + mom_w);
                    wg[0] = make_uint4(bt_pack2(gp16[0], gp16[1]), bt_pack2(gp16[2], gp16[3]), bt_pack2(gp16[4], gp16[5]), bt_pack2(gp16[6], gp16[7]));
                    wg[1] = make_uint4(bt_pack2(gp16[8], gp16[9]), bt_pack2(gp16[10], gp16[11]), bt_pack2(gp16[12], gp16[13]), bt_pack2(gp16[14], gp16[15]));
                    uint4* wa = reinterpret_cast<uint4*>(bt_lds + mom_w + 2048u);
                    wa[0] = __builtin_bit_cast(uint4, a2f[0]);
                    wa[1] = __builtin_bit_cast(uint4, a2f[1]);
                    bt_s16x4 fa[4], fb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned off = ((unsigned)t ^ mom_f) << 5;
                        fa[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s4*>(static_cast<uintptr_t>(mom_r + off)));
                        fb[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s4*>(static_cast<uintptr_t>(mom_r + 2048u + off)));
                    }
#pragma unroll
                    for (int to = 0; to < 4; ++to)
#pragma unroll
                        for (int tc = 0; tc < 4; ++tc)
                            m1[MOM ? to : 0][MOM ? tc : 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[to], fb[tc], m1[MOM ? to : 0][MOM ? tc : 0], 0, 0, 0);
#pragma unroll
                    for (int tq = 0; tq < 4; ++tq)
#pragma unroll
                        for (int u = 0; u < NTC; ++u)
                        {   // (selects, not an indexed register array: the column tile cg NTC + u is wave-uniform but not a constant)
                            const int tcol = cg * NTC + u;
                            const bt_s16x4 bsel = tcol == 0 ? fb[0] : (tcol == 1 ? fb[1] : (tcol == 2 ? fb[2] : fb[3]));
                            gr[MOM ? tq : 0][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fb[tq], bsel, gr[MOM ? tq : 0][u], 0, 0, 0);
                        }
                    if (cg == 0) {
                        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
                        const v2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const uint2 w = __builtin_bit_cast(uint2, fb[t]);
                            vs[t] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w.x), one, vs[t], false);
                            vs[t] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w.y), one, vs[t], false);
                        }
                    }
                }
                if (MODE == 3) {
                    const uint4 q0 = make_uint4(bt_pack2(gz[0], gz[1]), bt_pack2(gz[2], gz[3]), bt_pack2(gz[4], gz[5]), bt_pack2(gz[6], gz[7]));
                    const uint4 q1 = make_uint4(bt_pack2(gz[8], gz[9]), bt_pack2(gz[10], gz[11]), bt_pack2(gz[12], gz[13]), bt_pack2(gz[14], gz[15]));
                    if (ok && p.out) {                          // (NULL: conv2's weight gradient comes from the moments, nobody reads gz3)
                        *reinterpret_cast<uint4*>(p.out + eoff) = q0;
                        *reinterpret_cast<uint4*>(p.out + eoff + 8) = q1;
                    }
                    // ga2^T += W2^T[:, slab] gz3^T[slab]: k-step s, k-group lk <-> out channels o_base + 16 lk + 8 s .. + 7 = q_s as it stands
                    const uint4 zq = make_uint4(0, 0, 0, 0);
                    const bt_bf16x8 b0 = __builtin_bit_cast(bt_bf16x8, ok ? q0 : zq), b1 = __builtin_bit_cast(bt_bf16x8, ok ? q1 : zq);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint4* wp = Tf + ((sl * 4 + t) * 2) * 64 + lane;
                        g2acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, wp[0]), b0, g2acc[t], 0, 0, 0);
                        g2acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, wp[64]), b1, g2acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (MODE == 3 && p.part2 && ok) {     // BatchNorm2's backward sums over this row: g2' = ga2 act2'(pre2), against the raw e
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned w0 = t < 2 ? (t == 0 ? eq[0].x : eq[0].z) : (t == 2 ? eq[1].x : eq[1].z);
                const unsigned w1 = t < 2 ? (t == 0 ? eq[0].y : eq[0].w) : (t == 2 ? eq[1].y : eq[1].w);
                const float ev[4] = {bt_lo(w0), bt_hi(w0), bt_lo(w1), bt_hi(w1)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gg = (pos2 >> (4 * t + r)) & 1u ? g2acc[t][r] : g2acc[t][r] * p.slope2;
                    s0[t][r] += gg;
                    s1[t][r] = fmaf(gg, ev[r], s1[t][r]);
                }
            }
        }
        if (MODE == 3 && ok) {       // g2acc[t][r] = ga2 of channel 16 lk + 4 t + r of row li
            uint16_t* gp = p.out2 + (int64_t)row * 64 + 16 * lk;
            *reinterpret_cast<uint4*>(gp) = make_uint4(bt_pack2(g2acc[0][0], g2acc[0][1]), bt_pack2(g2acc[0][2], g2acc[0][3]),
                                                       bt_pack2(g2acc[1][0], g2acc[1][1]), bt_pack2(g2acc[1][2], g2acc[1][3]));
            *reinterpret_cast<uint4*>(gp + 8) = make_uint4(bt_pack2(g2acc[2][0], g2acc[2][1]), bt_pack2(g2acc[2][2], g2acc[2][3]),
                                                           bt_pack2(g2acc[3][0], g2acc[3][1]), bt_pack2(g2acc[3][2], g2acc[3][3]));
        }
        if (MODE == 3) load_e(it + DEPTH, ring_e[d]);
      }
    }
    if (MOM && NRG > 1) {
        // the row groups of the workgroup fold into row group 0 through LDS, one at a time in a fixed order (the fragments and tiles are
        // dead): one slab per workgroup instead of NRG — at Cout 64 the slabs would otherwise outweigh the tensors the moments replace
        constexpr int PER = 64 + 16 * NTC + 4;                               // floats per lane: m1, gr, vs
        float* fb = reinterpret_cast<float*>(bt_lds) + (size_t)cg * PER * 64 + lane;
        __syncthreads();
        for (int w = 1; w < NRG; ++w) {
            if (rg == w) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) fb[((a * 4 + b) * 4 + r) * 64] = m1[MOM ? a : 0][MOM ? b : 0][r];
#pragma unroll
                    for (int u = 0; u < NTC; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) fb[(64 + (a * NTC + u) * 4 + r) * 64] = gr[MOM ? a : 0][u][r];
                    fb[(64 + 16 * NTC + a) * 64] = vs[a];
                }
            }
            __syncthreads();
            if (rg == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) m1[MOM ? a : 0][MOM ? b : 0][r] += fb[((a * 4 + b) * 4 + r) * 64];
#pragma unroll
                    for (int u = 0; u < NTC; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gr[MOM ? a : 0][u][r] += fb[(64 + (a * NTC + u) * 4 + r) * 64];
                    vs[a] += fb[(64 + 16 * NTC + a) * 64];
                }
            }
            __syncthreads();
        }
    }
    if (MOM && rg == 0) {
        // this wave's part of the workgroup's slab: M1 rows of its slab, its Gram column tiles, v (slab 0's wave)
        float* slab = p.mom + (int64_t)blockIdx.x * (COUT * 64 + 64 * 64 + 64);
#pragma unroll
        for (int to = 0; to < 4; ++to)
#pragma unroll
            for (int tc = 0; tc < 4; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[(64 * cg + 16 * to + 4 * lk + r) * 64 + 16 * tc + li] = m1[MOM ? to : 0][MOM ? tc : 0][r];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
            for (int u = 0; u < NTC; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[COUT * 64 + (16 * tq + 4 * lk + r) * 64 + 16 * (cg * NTC + u) + li] = gr[MOM ? tq : 0][u][r];
        if (cg == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = vs[t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lk == 0) slab[COUT * 64 + 64 * 64 + 16 * t + li] = v;
            }
        }
    }
    if (MODE == 0 || MODE == 2 || (MODE == 3 && p.part2)) {
        // per-workgroup partial sums of every channel: fold the 16 rows of the tiles, then the row groups (mode 3: BatchNorm2's
        // sums over the 64 input channels, 16 lk + 4 t + r)
        constexpr int CF = MODE == 3 ? 64 : COUT;
        float* redf = MODE == 3 ? reinterpret_cast<float*>(bt_lds) : red;     // mode 3: the fragments are no longer needed
        if (MODE == 3) __syncthreads();
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s0[ot][r], b = s1[ot][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
                if (li == 0) {
                    const int o = 64 * cg + 16 * lk + 4 * ot + r;
                    redf[(rg * 2) * CF + o] = a;
                    redf[(rg * 2 + 1) * CF + o] = b;
                }
            }
        __syncthreads();
        float* dst = MODE == 3 ? p.part2 : p.part;
        for (int f = tid; f < 2 * CF; f += BT_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NRG; ++w) s += redf[w * 2 * CF + f];
            fgnn_fold_store(dst + (int64_t)blockIdx.x * 2 * CF + f, s);
        }
        if (p.fold.tickets) {            // no finaliser launch: the last workgroup folds every workgroup's row and finalises
            double* sums = reinterpret_cast<double*>(bt_lds);      // (the fragments are dead; the fold starts with a barrier)
            if (fgnn_grid_fold(p.fold, sums, blockIdx.x)) {
                if (MODE == 0) fgnn_bn_final_apply(p.fin, COUT, sums);
                else if (MODE == 3) fgnn_bn_bwd_final_apply(64, sums, p.mean2, p.invstd2, p.dsum2, p.gweight2, p.gbias2);
                else {
                    // BatchNorm3's backward sums and the per-channel constants of the closed-form input gradient:
                    //   S0 = sum g', S1 = sum g' (z3 - b2)   ->   dbeta = S0,  dgamma = invstd (S1 + (b2 - mean) S0)
                    //   gz3 = s3 (g' - dbeta / R - zhat dgamma / R) = s3 g' + A z3 + Bc,  A = -s3 invstd dgamma / R,  Bc = -s3 dbeta / R - A mean
                    for (int c = tid; c < COUT; c += BT_THREADS) {
                        const double S0 = sums[c], S1 = sums[COUT + c];
                        const double mu = (double)p.mean3[c], is = (double)p.invstd3[c], bb = p.b2 ? (double)p.b2[c] : 0.0;
                        const double dbeta = S0, dgamma = is * (S1 + (bb - mu) * S0);
                        const double s3 = (double)p.gamma3[c] * is, n = (double)p.R;
                        const double Ac = -s3 * is * dgamma / n;
                        p.A[c] = (float)Ac;
                        p.Bc[c] = (float)(-s3 * dbeta / n - Ac * mu);
                        if (p.gbias3) p.gbias3[c] += (float)dbeta;
                        if (p.gweight3) p.gweight3[c] += (float)dgamma;
                    }
                }
            }
        }
    }
}

// BatchNorm3's backward sums from the "reduce" partials, and the per-channel constants of the closed-form input gradient:
//   S0 = sum g', S1 = sum g' (z3 - b2)   ->   dbeta = S0,  dgamma = invstd (S1 + (b2 - mean) S0)
//   gz3 = s3 (g' - dbeta / R - zhat dgamma / R) = s3 g' + A z3 + Bc,   A = -s3 invstd dgamma / R,   Bc = -s3 dbeta / R - A mean
// 4 channels x 64 partial groups per block, folded in f64 in a fixed order.
__global__ __launch_bounds__(256) void block_tail_bwd_final_kernel(const float* ws, int nwg, int C, int64_t R, const float* b2,
                                                                   const float* mean, const float* invstd, const float* gamma,
                                                                   float* A, float* Bc, float* gweight, float* gbias) {
    __shared__ double r0[256], r1[256];
    const int cc = threadIdx.x & 3, pg = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cc;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int w = pg; w < nwg; w += 64) { a += (double)ws[(int64_t)w * 2 * C + c]; b += (double)ws[(int64_t)w * 2 * C + C + c]; }
    r0[threadIdx.x] = a;
    r1[threadIdx.x] = b;
    __syncthreads();
    if (pg != 0 || c >= C) return;
    double S0 = 0.0, S1 = 0.0;
    for (int q = 0; q < 64; ++q) { S0 += r0[q * 4 + cc]; S1 += r1[q * 4 + cc]; }
    const double mu = (double)mean[c], is = (double)invstd[c], bb = b2 ? (double)b2[c] : 0.0;
    const double dbeta = S0, dgamma = is * (S1 + (bb - mu) * S0);
    const double s3 = (double)gamma[c] * is, n = (double)R;
    const double Ac = -s3 * is * dgamma / n;
    A[c] = (float)Ac;
    Bc[c] = (float)(-s3 * dbeta / n - Ac * mu);
    if (gbias) gbias[c] += (float)dbeta;
    if (gweight) gweight[c] += (float)dgamma;
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
// Workgroup cap = ONE resident wave of workgroups on the 256 CUs: the statistics / reduce / grad modes hold 2 workgroups per CU
// (__launch_bounds__(256, 2)) -> 512, the apply mode 3 -> 768.  (768 for every mode, the first setting, left the reducing modes a
// half-empty second wave: 13.70 -> 13.56 ms per LDPC step with 512, four interleaved A/B runs, gpurun_out/r05ar.)
static int bt_plan(int64_t R, int Cout, int* grid, bool bound = true, bool apply = false) {
    if (R <= 0 || R > 0x7fffffff || (Cout != 64 && Cout != 128 && Cout != 256)) return -1;
    const int64_t ntile = (R + 15) / 16;
    const int nrg = bound ? BT_WAVES / (Cout / 64) : BT_WAVES;       // row groups per workgroup (modes 0-2: a wave is bound to a slab)
    int64_t g = (ntile + 2 * nrg - 1) / (2 * nrg);                    // >= 2 tiles per wave
    const int maxg = apply ? 768 : 512;      // one resident wave of workgroups: 3 per CU for the apply mode, 2 for the others (round 5: -0.13 ms)
    if (g > maxg) g = maxg;
    if (g > BT_MAXGRID) g = BT_MAXGRID;
    if (g < 1) g = 1;
    *grid = (int)g;
    return 0;
}

extern "C" int fgnn_block_tail_partials(int64_t R, int Cout) {
    int grid;
    return bt_plan(R, Cout, &grid) ? 0 : grid;
}

// rows of [2][64] BatchNorm2 partials fgnn_block_tail_backward writes when asked to (0 = shape not supported)
extern "C" int fgnn_block_tail_backward_partials(int64_t R, int Cout) {
    int grid;
    return bt_plan(R, Cout, &grid, false) ? 0 : grid;
}

template <int MODE, int NA = 0>
static int bt_launch(const BtParams& p, int grid, hipStream_t st) {
    const int CG = p.Cout / 64;
    void* fn;
    if (p.slope2 == 0.f) fn = CG == 1 ? (void*)block_tail_kernel<MODE, 1, NA, true> : (CG == 2 ? (void*)block_tail_kernel<MODE, 2, NA, true> : (void*)block_tail_kernel<MODE, 4, NA, true>);
    else fn = CG == 1 ? (void*)block_tail_kernel<MODE, 1, NA, false> : (CG == 2 ? (void*)block_tail_kernel<MODE, 2, NA, false> : (void*)block_tail_kernel<MODE, 4, NA, false>);
    const int ncst = MODE == 0 ? 1 : (MODE == 3 ? 4 : 2);
    int lds = CG * 8 * 64 * 16 * (MODE == 3 ? 2 : 1) + ncst * p.Cout * 4 + ((MODE == 0 || MODE == 2) ? BT_WAVES * 2 * p.Cout * 4 : 128 * 4)
              + ((MODE == 2 && NA == 1) ? 512 + BT_WAVES * 4096 : 0);
    if (MODE == 2 && NA == 1 && CG < 4) {                  // the in-workgroup fold of the moments: CG waves x (64 + 16 (4 / CG) + 4) floats per lane
        const int fold = CG * (64 + 16 * (4 / CG) + 4) * 64 * 4;
        if (fold > lds) lds = fold;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BT_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "block_tail launch (mode %d): %s", MODE, hipGetErrorString(e));
    return FGNN_OK;
}

static int bt_check(const char* who, const void* e, const float* s2, const float* t2, const float* W2, int64_t R, int Cout, int* grid, bool apply = false) {
    if (!e || !s2 || !t2 || !W2) FGNN_FAIL(FGNN_EINVAL, "%s: null pointer", who);
    if (bt_plan(R, Cout, grid, true, apply) || ((uintptr_t)e & 15) || ((uintptr_t)W2 & 7))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "%s: Cout=%d / alignment outside the fused block tail's family", who, Cout);
    return FGNN_OK;
}

int fgnn_bn_finalize_launch(const float* partials, int npartials, int C, const fgnn_bn_final* fin, hipStream_t st);
int fgnn_bn_bwd_final_raw_launch(const float* partials, int npartials, int C, const float* mean, const float* invstd, float* dsum,
                                 float* gweight, float* gbias, hipStream_t st);

extern "C" int fgnn_block_tail_stats(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                                     const float* b2, int64_t R, int Cout, float* partials, const fgnn_bn_final* fin,
                                     void* fold_scratch, fgnn_stream_t stream) {
    int grid, rc;
    if ((rc = bt_check("block_tail_stats", e, scale2, shift2, W2, R, Cout, &grid))) return rc;
    if (!partials || !fin || !fin->mean || !fin->invstd || !fin->scale || !fin->shift) FGNN_FAIL(FGNN_EINVAL, "block_tail_stats: null pointer");
    if (fin->shift_k) FGNN_FAIL(FGNN_EINVAL, "block_tail_stats: shift_k is this kernel's own (b2)");
    if (fin->count != R || (fin->population != 0 && fin->population < R)) FGNN_FAIL(FGNN_EINVAL, "block_tail_stats: fin->count must be R");
    BtParams p = {};
    p.e = (const uint16_t*)e; p.s2 = scale2; p.t2 = shift2; p.W2 = W2; p.b2 = b2; p.part = partials;
    p.R = (int)R; p.Cout = Cout; p.slope2 = slope2;
    p.fin = *fin;
    p.fin.shift_k = b2;                       // the kernel sums z3 - b2
    const bool inkernel = fold_scratch && !fgnn_separate_finalisers();
    p.fold = fgnn_fold_make(partials, inkernel ? fold_scratch : nullptr, grid, Cout);
    if ((rc = bt_launch<0>(p, grid, (hipStream_t)stream))) return rc;
    if (!inkernel && fgnn_bn_finalize_launch(partials, grid, Cout, &p.fin, (hipStream_t)stream))
        FGNN_FAIL(FGNN_ELAUNCH, "block_tail_stats finaliser launch: %s", hipGetErrorString(hipGetLastError()));
    return FGNN_OK;
}

extern "C" int fgnn_block_tail_apply(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                                     const float* b2, const float* scale3, const float* shift3, float slope3,
                                     const void* addend0, const void* addend1, const void* addend2, const int32_t* addend_period,
                                     void* out, void* a2_out, int64_t R, int Cout, fgnn_stream_t stream) {
    int grid, rc;
    if ((rc = bt_check("block_tail_apply", e, scale2, shift2, W2, R, Cout, &grid, true))) return rc;
    if (!scale3 || !shift3 || !out) FGNN_FAIL(FGNN_EINVAL, "block_tail_apply: null pointer");
    if (((uintptr_t)out | (uintptr_t)a2_out | (uintptr_t)addend0 | (uintptr_t)addend1 | (uintptr_t)addend2) & 15)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "block_tail_apply: misaligned operand");
    BtParams p = {};
    p.e = (const uint16_t*)e; p.s2 = scale2; p.t2 = shift2; p.W2 = W2; p.b2 = b2; p.s3 = scale3; p.t3 = shift3;
    const void* ads[3] = {addend0, addend1, addend2};      // packed to the front: the kernel is specialised on their number
    int per[3] = {1, 1, 1};
    int na = 0;
    for (int a = 0; a < 3; ++a)
        if (ads[a]) {
            const int pr = addend_period ? addend_period[a] : 1;
            if (pr < 1) FGNN_FAIL(FGNN_EINVAL, "block_tail_apply: addend period < 1");
            ads[na] = ads[a]; per[na] = pr; ++na;
        }
    p.add0 = (const uint16_t*)(na > 0 ? ads[0] : nullptr); p.add1 = (const uint16_t*)(na > 1 ? ads[1] : nullptr);
    p.add2 = (const uint16_t*)(na > 2 ? ads[2] : nullptr);
    for (int a = 0; a < 3; ++a) p.aperiod[a] = a < na ? per[a] : 1;
    p.out = (uint16_t*)out; p.out2 = (uint16_t*)a2_out; p.R = (int)R; p.Cout = Cout; p.slope2 = slope2; p.slope3 = slope3;
    hipStream_t st = (hipStream_t)stream;
    switch (na) {
        case 0: return bt_launch<1, 0>(p, grid, st);
        case 1: return bt_launch<1, 1>(p, grid, st);
        case 2: return bt_launch<1, 2>(p, grid, st);
        default: return bt_launch<1, 3>(p, grid, st);
    }
}

// ----------------------------------------------------------------------------------------
// conv2's weight gradient from the moments of the "reduce" pass (see the top of this file)
// ----------------------------------------------------------------------------------------
// moments buffer (floats): [nslab][slab_len] slabs | A [Cout] | Bc [Cout] | T [slab_len] (the folded moments), slab_len = Cout*64 + 64*64 + 64,
// nslab = workgroups of the reduce pass
static int64_t bt_mom_slab_len(int Cout) { return (int64_t)Cout * 64 + 64 * 64 + 64; }
static int bt_mom_nslab(int64_t R, int Cout) {
    int grid;
    if (bt_plan(R, Cout, &grid)) return 0;
    return grid;                                           // one slab per workgroup of the reduce pass (its row groups fold in LDS)
}
extern "C" int64_t fgnn_block_tail_moments_bytes(int64_t R, int Cout) {
    const int n = bt_mom_nslab(R, Cout);
    return n ? ((int64_t)(n + 1) * bt_mom_slab_len(Cout) + 2 * Cout) * 4 : 0;
}

// gW2[o][c] += s3[o] M1[o][c] + A[o] sum_c' bf16(W2[o][c']) Gram[c'][c] + (A[o] b2[o] + Bc[o]) v[c]   from the FOLDED moments T
__global__ __launch_bounds__(256) void block_tail_wgrad_combine_kernel(const float* __restrict__ T, const float* __restrict__ W2,
                                                                       const float* __restrict__ b2, const float* __restrict__ s3,
                                                                       const float* __restrict__ A, const float* __restrict__ Bc,
                                                                       float* __restrict__ gW2, int Cout) {
    __shared__ float G[64 * 64];
    __shared__ float w[4][64];
    const int tid = threadIdx.x, ol = tid >> 6, c = tid & 63, o = blockIdx.x * 4 + ol;
    const float* Gp = T + (int64_t)Cout * 64;
    for (int f = tid; f < 64 * 64; f += 256) G[f] = Gp[f];
    {
        const __bf16 h = (__bf16)W2[(int64_t)o * 64 + c];                   // the operand the matrix cores multiplied a2 with
        w[ol][c] = __uint_as_float((unsigned)__builtin_bit_cast(uint16_t, h) << 16);
    }
    __syncthreads();
    float m2 = 0.f;
#pragma unroll 16
    for (int q = 0; q < 64; ++q) m2 = fmaf(w[ol][q], G[q * 64 + c], m2);
    const float a = A[o], k = fmaf(a, b2 ? b2[o] : 0.f, Bc[o]);
    gW2[(int64_t)o * 64 + c] += fmaf(s3[o], T[(int64_t)o * 64 + c], fmaf(a, m2, k * Gp[64 * 64 + c]));
}

void fgnn_launch_slab_store(const float* ws, int nslab, int64_t slab_len, float* out, hipStream_t st);     // mpconv_bwd_res.hip

// Fold the moments fgnn_block_tail_backward_moments left and ADD conv2's weight gradient to gW2 [Cout][64] (two short launches; nothing
// in a backward pass reads a weight gradient: callers park this call like any other weight-gradient launch).
extern "C" int fgnn_block_tail_wgrad_finish(void* moments, int64_t moments_bytes, int64_t R, int Cout, const float* W2, const float* b2,
                                            const float* scale3, float* gW2, fgnn_stream_t stream) {
    const int nslab = bt_mom_nslab(R, Cout);
    if (!moments || !W2 || !scale3 || !gW2) FGNN_FAIL(FGNN_EINVAL, "block_tail_wgrad_finish: null pointer");
    if (!nslab || moments_bytes < fgnn_block_tail_moments_bytes(R, Cout)) FGNN_FAIL(FGNN_EINVAL, "block_tail_wgrad_finish: bad sizes");
    const int64_t len = bt_mom_slab_len(Cout);
    float* m = (float*)moments;
    float* A = m + (int64_t)nslab * len;
    float* Bc = A + Cout;
    float* T = Bc + Cout;
    hipStream_t st = (hipStream_t)stream;
    fgnn_launch_slab_store(m, nslab, len, T, st);
    hipLaunchKernelGGL(block_tail_wgrad_combine_kernel, dim3(Cout / 4), dim3(256), 0, st, T, W2, b2, scale3, A, Bc, gW2, Cout);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "block_tail_wgrad_finish launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

static int bt_backward(const void* e, const float* scale2, const float* shift2, float slope2,
                       const float* W2, const float* b2, const float* mean3, const float* invstd3,
                       const float* gamma3, const float* scale3, const float* shift3, float slope3,
                       const void* gout, void* gz3, void* ga2, float* gweight3, float* gbias3,
                       const float* mean2, const float* invstd2, float* gweight2, float* gbias2, float* bn2_dsum,
                       int64_t R, int Cout, void* workspace, int64_t workspace_bytes, void* fold_scratch,
                       float* moments, int64_t moments_bytes, fgnn_stream_t stream) {
    int grid, rc;
    if ((rc = bt_check("block_tail_backward", e, scale2, shift2, W2, R, Cout, &grid))) return rc;
    if (!mean3 || !invstd3 || !gamma3 || !scale3 || !shift3 || !gout || (!gz3 && !moments) || !ga2 || !workspace)
        FGNN_FAIL(FGNN_EINVAL, "block_tail_backward: null pointer");
    if (moments && moments_bytes < fgnn_block_tail_moments_bytes(R, Cout)) FGNN_FAIL(FGNN_EINVAL, "block_tail_backward: moments buffer too small");
    if (bn2_dsum && (!mean2 || !invstd2)) FGNN_FAIL(FGNN_EINVAL, "block_tail_backward: bn2_dsum needs mean2 / invstd2");
    if (((uintptr_t)gout | (uintptr_t)gz3 | (uintptr_t)ga2 | (uintptr_t)moments) & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "block_tail_backward: misaligned operand");
    if (workspace_bytes < ((int64_t)BT_MAXGRID * 2 * Cout + 2 * Cout + (int64_t)BT_MAXGRID * 128) * 4)
        FGNN_FAIL(FGNN_EINVAL, "block_tail_backward: workspace too small");
    float* ws = (float*)workspace;
    float* A = ws + (int64_t)BT_MAXGRID * 2 * Cout;
    float* Bc = A + Cout;
    float* part2 = Bc + Cout;                              // [grid3][2][64] BatchNorm2's partial rows
    if (moments) {     // the per-channel constants outlive this call's shared workspace: fgnn_block_tail_wgrad_finish reads them later
        A = moments + (int64_t)bt_mom_nslab(R, Cout) * bt_mom_slab_len(Cout);
        Bc = A + Cout;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool inkernel = fold_scratch && !fgnn_separate_finalisers();
    BtParams p = {};
    p.e = (const uint16_t*)e; p.s2 = scale2; p.t2 = shift2; p.W2 = W2; p.b2 = b2; p.s3 = scale3; p.t3 = shift3;
    p.gout = (const uint16_t*)gout; p.part = ws; p.R = (int)R; p.Cout = Cout; p.slope2 = slope2; p.slope3 = slope3;
    p.mean3 = mean3; p.invstd3 = invstd3; p.gamma3 = gamma3; p.A = A; p.Bc = Bc; p.gweight3 = gweight3; p.gbias3 = gbias3;
    p.fold = fgnn_fold_make(ws, inkernel ? fold_scratch : nullptr, grid, Cout);
    p.mom = moments;
    if ((rc = moments ? bt_launch<2, 1>(p, grid, st) : bt_launch<2>(p, grid, st))) return rc;
    int grid3;
    (void)bt_plan(R, Cout, &grid3, false);
    if (!inkernel)
        hipLaunchKernelGGL(block_tail_bwd_final_kernel, dim3((Cout + 3) / 4), dim3(256), 0, st, ws, grid, Cout, R, b2, mean3,
                           invstd3, gamma3, A, Bc, gweight3, gbias3);
    p.ga = A; p.gb = Bc; p.out = (uint16_t*)gz3; p.out2 = (uint16_t*)ga2; p.part = nullptr; p.part2 = bn2_dsum ? part2 : nullptr;
    p.mean2 = mean2; p.invstd2 = invstd2; p.dsum2 = bn2_dsum; p.gweight2 = gweight2; p.gbias2 = gbias2;
    p.fold = fgnn_fold_make(part2, (inkernel && bn2_dsum) ? fold_scratch : nullptr, grid3, 64);
    if ((rc = bt_launch<3>(p, grid3, st))) return rc;
    if (!inkernel && bn2_dsum && fgnn_bn_bwd_final_raw_launch(part2, grid3, 64, mean2, invstd2, bn2_dsum, gweight2, gbias2, st))
        FGNN_FAIL(FGNN_ELAUNCH, "block_tail_backward finaliser launch: %s", hipGetErrorString(hipGetLastError()));
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "block_tail_backward launch: %s", hipGetErrorString(e2));
    return FGNN_OK;
}

extern "C" int fgnn_block_tail_backward(const void* e, const float* scale2, const float* shift2, float slope2,
                                        const float* W2, const float* b2, const float* mean3, const float* invstd3,
                                        const float* gamma3, const float* scale3, const float* shift3, float slope3,
                                        const void* gout, void* gz3, void* ga2, float* gweight3, float* gbias3,
                                        const float* mean2, const float* invstd2, float* gweight2, float* gbias2, float* bn2_dsum,
                                        int64_t R, int Cout, void* workspace, int64_t workspace_bytes, void* fold_scratch,
                                        fgnn_stream_t stream) {
    if (!gz3) FGNN_FAIL(FGNN_EINVAL, "block_tail_backward: null pointer");
    return bt_backward(e, scale2, shift2, slope2, W2, b2, mean3, invstd3, gamma3, scale3, shift3, slope3, gout, gz3, ga2, gweight3, gbias3,
                       mean2, invstd2, gweight2, gbias2, bn2_dsum, R, Cout, workspace, workspace_bytes, fold_scratch, nullptr, 0, stream);
}

// The same, with the MOMENTS of conv2's weight gradient accumulated by the reduce pass into `moments`
// (fgnn_block_tail_moments_bytes(R, Cout) bytes, owned by the caller until fgnn_block_tail_wgrad_finish has run); gz3 may then be NULL.
extern "C" int fgnn_block_tail_backward_moments(const void* e, const float* scale2, const float* shift2, float slope2,
                                                const float* W2, const float* b2, const float* mean3, const float* invstd3,
                                                const float* gamma3, const float* scale3, const float* shift3, float slope3,
                                                const void* gout, void* gz3, void* ga2, float* gweight3, float* gbias3,
                                                const float* mean2, const float* invstd2, float* gweight2, float* gbias2, float* bn2_dsum,
                                                int64_t R, int Cout, void* workspace, int64_t workspace_bytes, void* fold_scratch,
                                                void* moments, int64_t moments_bytes, fgnn_stream_t stream) {
    if (!moments) FGNN_FAIL(FGNN_EINVAL, "block_tail_backward_moments: null pointer");
    return bt_backward(e, scale2, shift2, slope2, W2, b2, mean3, invstd3, gamma3, scale3, shift3, slope3, gout, gz3, ga2, gweight3, gbias3,
                       mean2, invstd2, gweight2, gbias2, bn2_dsum, R, Cout, workspace, workspace_bytes, fold_scratch, (float*)moments,
                       moments_bytes, stream);
}

// ----------------------------------------------------------------------------------------
// HEAD of the block, backward (round 3): conv1 -> BatchNorm1 -> LeakyReLU in front of the operator
// (/root/reference/lib/model/mpnn/mp_nn_residual.py:25-29,42-44).  Given z1 = conv1's output [R][64] and ga1 = the operator's
// input gradient [R][64]:
//     gz1 = kc g' + A z1 + B      BatchNorm1's input gradient in closed form per channel (kc = gamma invstd, g' = ga1 act'(pre),
//                                 A = -kc invstd dgamma / R, B = kc (invstd mean dgamma - dbeta) / R)
//     gx  = gz1 W1                conv1's input gradient, [R][Cin], Cin in {64, 128, 256}
// in ONE pass: the staged path wrote gz1 (bn_apply_kernel<1>) and read it back in the input-gradient GEMM.  gz1 is still stored once
// (the weight-gradient kernel reads it); the sums dbeta / dgamma come from the BatchNorm reduction pass in front of this kernel.
// Same transposed product and lane <-> 16-consecutive-channels layout as above: the computed gz1 fragment IS the B operand.
// ----------------------------------------------------------------------------------------
struct BhParams {
    const uint16_t* z;       // [R][64] bf16 conv1 output (BatchNorm1's input)
    const uint16_t* g;       // [R][64] bf16 gradient of BatchNorm1's activated output
    const float* mean; const float* invstd; const float* gamma; const float* beta;    // [64]
    const float* dsum;       // [2][64] dbeta, dgamma
    const float* W;          // [64][Cin] f32 conv1 weight
    uint16_t* gz;            // [R][64] out
    uint16_t* gx;            // [R][Cin] out
    int R, Cin;
    float slope, inv_r;
};

template <int CG>
__global__ __launch_bounds__(BT_THREADS, 3) void block_head_bwd_kernel(const BhParams p) {
    constexpr int CIN = 64 * CG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int R = p.R;
    uint4* Wf = reinterpret_cast<uint4*>(bt_lds);                                        // [CG][4 ot][2 ks][64]   gx^T = W1^T gz1^T
    float* cst = reinterpret_cast<float*>(Wf + CG * 8 * 64);                             // [5][64]  s1, t1, kc, A, B
    for (int f = tid; f < CG * 8 * 64; f += BT_THREADS) {
        const int fl = f & 63, fs = (f >> 6) & 1, ft = (f >> 7) & 3, sl = f >> 9;
        const int fi = fl & 15, fk = fl >> 4;
        const int pr = 16 * (fi >> 2) + (fi & 3) + 4 * ft;
        // A[i = input channel 64 sl + pr][k = BatchNorm channels 16 fk + 8 fs .. + 7] = W1[k][i]
        const float* wp = p.W + (int64_t)(16 * fk + 8 * fs) * CIN + 64 * sl + pr;
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = wp[(int64_t)u * CIN];
        Wf[f] = make_uint4(bt_pack2(w[0], w[1]), bt_pack2(w[2], w[3]), bt_pack2(w[4], w[5]), bt_pack2(w[6], w[7]));
    }
    if (tid < 64) {
        const float m = p.mean[tid], is = p.invstd[tid], kc = p.gamma[tid] * is;
        const float ke = p.dsum[tid] * p.inv_r, kf = p.dsum[64 + tid] * p.inv_r;
        cst[tid] = kc;                                    // pre = z kc + (beta - mean kc)
        cst[64 + tid] = p.beta[tid] - m * kc;
        cst[128 + tid] = kc;
        cst[192 + tid] = -kc * kf * is;                   // gz = kc (g' - ke - (z - m) is kf) = kc g' + A z + B
        cst[256 + tid] = kc * (kf * is * m - ke);
    }
    __syncthreads();
    const int cg = wave % CG, rg = wave / CG;
    constexpr int NRG = BT_WAVES / CG;
    uint4 aW[4][2];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        aW[ot][0] = Wf[((cg * 4 + ot) * 2 + 0) * 64 + lane];
        aW[ot][1] = Wf[((cg * 4 + ot) * 2 + 1) * 64 + lane];
    }
    float s1[16], t1[16], ka[16], kb[16];                // (kc == s1)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        s1[c] = cst[16 * lk + c]; t1[c] = cst[64 + 16 * lk + c]; ka[c] = cst[192 + 16 * lk + c]; kb[c] = cst[256 + 16 * lk + c];
    }
    const int ntile = (R + 15) / 16;
    const int stride = gridDim.x * NRG;
    const int first = blockIdx.x * NRG + rg;
    const int mine = first < ntile ? (ntile - first + stride - 1) / stride : 0;
    auto load2 = [&](int it, uint4 (&zq)[2], uint4 (&gq)[2]) {
        const int row = min((first + it * stride) * 16 + li, R - 1);
        const uint16_t* zp = p.z + (int64_t)row * 64 + 16 * lk;
        const uint16_t* gp = p.g + (int64_t)row * 64 + 16 * lk;
        zq[0] = *reinterpret_cast<const uint4*>(zp); zq[1] = *reinterpret_cast<const uint4*>(zp + 8);
        gq[0] = *reinterpret_cast<const uint4*>(gp); gq[1] = *reinterpret_cast<const uint4*>(gp + 8);
    };
    // two tiles in flight per wave (a ring of two register slots, refilled right behind their last use; rows past the end are
    // clamped, so the loads are unconditional and the loop body stays straight-line code)
    constexpr int HD = 2;
    uint4 rz[HD][2], rgq[HD][2];
#pragma unroll
    for (int d = 0; d < HD; ++d) { load2(d, rz[d], rgq[d]); __builtin_amdgcn_sched_barrier(0); }
    for (int base = 0; base < mine; base += HD) {
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        const int it = base + d;
        uint4 (&zq)[2] = rz[d];
        uint4 (&gq)[2] = rgq[d];
        const int row = (first + it * stride) * 16 + li;
        const bool ok = it < mine && row < R;
        bt_bf16x8 gzf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float z[8], g[8];
            bt_unpack8(zq[ks], z);
            bt_unpack8(gq[ks], g);
            unsigned o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = 8 * ks + 2 * u + h;
                    const float pre = fmaf(z[2 * u + h], s1[c], t1[c]);
                    const float ge = pre > 0.f ? g[2 * u + h] : g[2 * u + h] * p.slope;
                    v[h] = fmaf(s1[c], ge, fmaf(ka[c], z[2 * u + h], kb[c]));
                }
                o[u] = bt_pack2(v[0], v[1]);
            }
            uint4 q = make_uint4(o[0], o[1], o[2], o[3]);
            if (!ok) q = make_uint4(0, 0, 0, 0);
            gzf[ks] = __builtin_bit_cast(bt_bf16x8, q);
            if (ok && cg == 0) *reinterpret_cast<uint4*>(p.gz + (int64_t)row * 64 + 16 * lk + 8 * ks) = q;
        }
        load2(it + HD, rz[d], rgq[d]);
        __builtin_amdgcn_sched_barrier(0);
        if (!p.gx) continue;                              // gz1 only: conv1's input gradient is formed elsewhere (fgnn_linear_multi_forward)
        f32x4 acc[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, aW[ot][ks]), gzf[ks], acc[ot], 0, 0, 0);
        }
        if (ok) {     // acc[ot][r] = gx of input channel 64 cg + 16 lk + 4 ot + r, row li
            uint16_t* op = p.gx + (int64_t)row * CIN + 64 * cg + 16 * lk;
            *reinterpret_cast<uint4*>(op) = make_uint4(bt_pack2(acc[0][0], acc[0][1]), bt_pack2(acc[0][2], acc[0][3]),
                                                      bt_pack2(acc[1][0], acc[1][1]), bt_pack2(acc[1][2], acc[1][3]));
            *reinterpret_cast<uint4*>(op + 8) = make_uint4(bt_pack2(acc[2][0], acc[2][1]), bt_pack2(acc[2][2], acc[2][3]),
                                                          bt_pack2(acc[3][0], acc[3][1]), bt_pack2(acc[3][2], acc[3][3]));
        }
      }
    }
}

int fgnn_bn_backward_sums_bf16(const void* x, const void* gy, int64_t R, int C, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, float slope, float* gweight, float* gbias,
                               void* workspace, void* fold_scratch, hipStream_t st, const float** dsum_out);

// BatchNorm1 + activation backward and conv1's input gradient (see above).  z1 / ga1 [R][64] bf16, W1 [64][Cin] f32 (Cin in
// {64, 128, 256}), gz1 [R][64] and gx [R][Cin] bf16 out (gx NULL: only gz1 — the caller multiplies it by W1 together with the state's
// other gradients, fgnn_linear_multi_forward), gweight / gbias [64] ACCUMULATED into (BatchNorm1's parameter gradients).
// workspace: fgnn_bn_workspace_bytes(R, 64).
extern "C" int fgnn_block_head_backward(const void* z1, const void* ga1, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, float slope, const float* W1, void* gz1,
                                        void* gx, float* gweight, float* gbias, int64_t R, int Cin, void* workspace,
                                        int64_t workspace_bytes, void* fold_scratch, fgnn_stream_t stream) {
    if (!z1 || !ga1 || !mean || !invstd || !gamma || !beta || !W1 || !gz1 || !workspace)
        FGNN_FAIL(FGNN_EINVAL, "block_head_backward: null pointer");
    int grid;
    if (bt_plan(R, Cin, &grid, true, true) || (((uintptr_t)z1 | (uintptr_t)ga1 | (uintptr_t)gz1 | (uintptr_t)gx) & 15))      // (3 workgroups per CU: the 768 cap)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "block_head_backward: Cin=%d / alignment outside the fused block head's family", Cin);
    if (workspace_bytes < (int64_t)BT_MAXGRID * 2 * 64 * 4 + 2 * 64 * 4) FGNN_FAIL(FGNN_EINVAL, "block_head_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const float* dsum = nullptr;
    if (fgnn_bn_backward_sums_bf16(z1, ga1, R, 64, mean, invstd, gamma, beta, slope, gweight, gbias, workspace, fold_scratch, st, &dsum))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "block_head_backward: BatchNorm reduction plan failed");
    BhParams p = {};
    p.z = (const uint16_t*)z1; p.g = (const uint16_t*)ga1; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
    p.dsum = dsum; p.W = W1; p.gz = (uint16_t*)gz1; p.gx = (uint16_t*)gx; p.R = (int)R; p.Cin = Cin; p.slope = slope;
    p.inv_r = 1.0f / (float)R;
    const int CG = Cin / 64;
    void* fn = CG == 1 ? (void*)block_head_bwd_kernel<1> : (CG == 2 ? (void*)block_head_bwd_kernel<2> : (void*)block_head_bwd_kernel<4>);
    const int lds = CG * 8 * 64 * 16 + 5 * 64 * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    }
    fgnn_note_kernel("block_head_bwd_kernel<%d>", CG);
    void* args[] = {(void*)&p};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(BT_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "block_head_backward launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
