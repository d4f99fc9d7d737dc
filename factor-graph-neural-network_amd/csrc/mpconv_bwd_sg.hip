// mpconv_bwd_sg.hip — second-generation backward of the VF/FV message operator for the LDPC parity-check calls:
// bf16 channel-fastest x / gz / etype, 64 -> 64 channels, 4 edge types, max aggregation, fixed degree (3 or 6), ONE
// neighbour table shared by the batch whose transposed incidence is regular enough (in-degree <= 3 or 6).
// Same maths and rounding points as mpconv_bwd_b16.hip (autograd through /root/reference/lib/model/mpnn/mp_nn.py:115-175):
//
//     P[n,o,e]      = sum_c x[n,c] W[c,o*4+e]                                   (recomputed, bf16 in LDS)
//     detype[e,m,j] = sum_o gz[m,o] [j == argmax[m,o]] P[idx[m,j],o,e]
//     dP[n,o,e]     = sum_{(m,j): idx[m,j]=n} gz[m,o] [j == argmax[m,o]] etype[m,j,e]     (bf16 in LDS)
//     dx[n,c]       = sum_col dP[n,col] W[c,col]          dW[c,col] += sum_n x[n,c] dP[n,col]       dbias[o] += sum_m gz[m,o]
//
// The first-generation kernel spends ~1 740 VALU instructions per wave and sample (profiles/r01), two thirds of them in
// the two routing phases (work lists with per-item index arithmetic, byte-wise argmax tests, per-edge LDS traffic).
// This kernel is organised around what the shared graph makes constant:
//   * routing by SOURCE node (dP): every wave owns a few nodes; the (destination row, slot) of each of their in-edges
//     sits in scalar registers for the kernel's lifetime (transposed incidence built once per workgroup in LDS, in
//     (m, j) order — fixed summation order, no atomics anywhere).  Per in-edge and channel lane: one 4-byte LDS read of
//     {gz as the high half | one-hot argmax in the low bits}, a bit-field extract, an AND and two packed FMAs;
//   * routing by DESTINATION (detype): lane = channel reads only the routed P row (v_perm picks the node id out of the
//     destination's packed neighbour bytes), forms the 4 products, and parks them as bf16 in a wave-private,
//     zero-initialised [slot j][edge-type pair][channel] image (one dword = two edge types).  The sum over channels is
//     then an MFMA whose A operand is an even / odd selector (4 instructions per destination) — no cross-lane shuffles;
//   * the three GEMM-shaped phases run on v_mfma_f32_16x16x32_bf16 as before (dW takes its node-contracted operands by
//     reading 8 rows x 4 columns per lane and transposing in registers with v_perm_b32).
// One 1024-thread workgroup (16 waves) per CU: the LDS images of a sample (x double-buffered 2 x 14 KB, P / dP 50 KB —
// one buffer, used in turn —, gz+argmax 12..25 KB, the per-wave detype images 55 KB) do not leave room for two.
#include "fgnn_common.h"
#include <stdlib.h>

#define BS_THREADS 1024
#define BS_WAVES 16
#ifndef BS_XSB
#define BS_XSB 160           // x image row stride, bytes: 64 bf16 + 32 = 10 slots (2 mod 4, see BS_ZCS): the projection's operand reads are
                             // conflict-free; the dW phase's transposed reads (rows 8 apart) become 2-way — net 132.1 -> 128.5 us (V->F)
#endif
#ifndef BS_PSB
#define BS_PSB 528           // P / dP image row stride, bytes (256 bf16 + 16)
#endif
#define BS_GSB 256           // ga row: 64 dwords {gz bf16 << 16 | argmax << 8 | 1 << argmax}
#define BS_ZCS 288           // detype image: column (slot j, edge-type pair) stride in bytes: 64 dwords {e even | e odd} + 32 = 18 16-byte
                             // slots.  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, ... (two k-groups mixed): the
                             // 16 rows of an operand tile are conflict-free only at a stride of 2 mod 4 slots (tools/lds_swizzle.py);
                             // at 17 slots (272 B) every read cost 2x: 140.7 -> 132.8 us (V->F), 148.6 -> 136.4 us (F->V)
#define BS_MAXN 96

typedef __bf16 bs_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bs_f32x2 __attribute__((ext_vector_type(2)));

struct BsParams {
    const uint16_t* x;
    const int64_t* idx;
    const uint16_t* et;
    const float* W;          // [64][256]
    const uint16_t* gz;
    const uint8_t* argmax;
    uint16_t* gx;
    uint16_t* get;
    float* ws;               // per-workgroup slabs [grid][64*256 + 64]
    int B, N, M, Npad, NPW, DPW;
    int y_ld, w_ld, accum;           // row strides (elements) of gz / argmax and of W; accum: gx and getype are ADDED to (the 64 -> 128
                                     // calls run as two launches over the halves of the output channels: y_ld = 128, w_ld = 512)
    long long x_sb, et_sb, y_sb;     // elements
    int off_xs, off_pd, off_ga, off_z, off_es, off_tab, off_idx, off_red, off_gst;      // byte offsets into LDS
    int zbytes;              // bytes of one wave's detype image
    long long* prof;         // FGNN_PROF (builds with -DFGNN_ENABLE_PROF only): phase timeline
};

extern __shared__ __attribute__((aligned(16))) unsigned char bs_lds[];

#ifdef FGNN_ENABLE_PROF
#define BS_STAMP(slot) do { if (p.prof && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && b == b_begin + 3) p.prof[(wave >> 2) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define BS_STAMP(slot) do { } while (0)
#endif

void fgnn_launch_slab_reduce(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gbias,
                             hipStream_t st);
void fgnn_launch_slab_reduce_ld(const float* ws, int nslab, int64_t slab_len, int64_t nw, int ncols, int ld, float* gW,
                                float* gbias, hipStream_t st);

// uniform 64-bit base + UNSIGNED 32-bit per-lane byte offset: the form the compiler turns into `global_load v, v_off, s[base]`
// (a signed or 64-bit per-lane offset becomes a per-lane 64-bit pointer: two VGPRs each, hoisted out of the sample loop)
template <typename T> __device__ __forceinline__ const T* bs_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ T* bs_at(void* base, unsigned byte_off) {
    return reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}
__device__ __forceinline__ float bs_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bs_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned bs_pack2(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    const v2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ bs_bf16x8 bs_frag_f32(const float* p8) {      // 8 consecutive f32 -> one fragment
    const f32x4 a = *reinterpret_cast<const f32x4*>(p8), b = *reinterpret_cast<const f32x4*>(p8 + 4);
    return __builtin_bit_cast(bs_bf16x8, make_uint4(bs_pack2(a[0], a[1]), bs_pack2(a[2], a[3]),
                                                    bs_pack2(b[0], b[1]), bs_pack2(b[2], b[3])));
}
// rows r0..r7 each hold columns (c0 c1 | c2 c3) as two dwords: gather column P's eight values
template <int P>
__device__ __forceinline__ bs_bf16x8 bs_tr(const uint2 (&r)[8]) {
    constexpr unsigned sel = (P & 1) ? 0x07060302u : 0x05040100u;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned a = P < 2 ? r[2 * q].x : r[2 * q].y, b = P < 2 ? r[2 * q + 1].x : r[2 * q + 1].y;
        w[q] = __builtin_amdgcn_perm(b, a, sel);
    }
    return __builtin_bit_cast(bs_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}
__device__ __forceinline__ bs_bf16x8 bs_tr_dyn(const uint2 (&r)[8], int P) {
    switch (P) {
        case 0: return bs_tr<0>(r);
        case 1: return bs_tr<1>(r);
        case 2: return bs_tr<2>(r);
        default: return bs_tr<3>(r);
    }
}

// KC = destination degree (neighbour slots per destination), DEG = in-edges per source node the tables are sized for,
// NPW = source nodes per wave (ceil(N / 16)), DPW = destinations per wave (ceil(M / 16))
template <int KC, int DEG, int NPW, int DPW, int GSL>
__global__ __launch_bounds__(BS_THREADS, 4) void mpconv_bwd_sg_kernel(const BsParams p) {
    constexpr int NIN = 64, NCOLS = 256, NOU = 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int N = p.N, M = p.M, Npad = p.Npad;
    const int mk = M * KC;

    unsigned char* xs0 = bs_lds + p.off_xs;                              // 2 x [Npad][XSB]  bf16 x (double-buffered)
    unsigned char* pd = bs_lds + p.off_pd;                               // [Npad][PSB]      bf16 P, then dP
    unsigned char* ga = bs_lds + p.off_ga;                               // [M][GSB]         {gz | argmax}
    unsigned char* zb = bs_lds + p.off_z + wave * p.zbytes;              // this wave's detype image [KC*2][ZCS]
    // this wave's edge-type image [4 e][NPW nodes][QS in-edge slots] bf16: the A operands of the dP products (slots >= DEG stay 0)
    constexpr int QS = DEG > 4 ? 8 : 4;
    unsigned char* etw = bs_lds + p.off_es + wave * (4 * NPW * QS * 2);
    int* tab = reinterpret_cast<int*>(bs_lds + p.off_tab);               // [N][DEG]  m * 256 + j of every in-edge
    int* idx_s = reinterpret_cast<int*>(bs_lds + p.off_idx);             // [M * KC]
    const int xs_bytes = Npad * BS_XSB;

    // ---- zero the LDS images: padding rows are read by the matrix cores, the detype images rely on their zeros ----
    for (int f = tid; f < (p.off_tab - p.off_xs) / 4; f += BS_THREADS) reinterpret_cast<unsigned*>(bs_lds + p.off_xs)[f] = 0u;
    // ---- neighbour table -> LDS, then the transposed incidence in (m, j) order (deterministic, no atomics) ----
    for (int r = tid; r < mk; r += BS_THREADS) {
        long long v = p.idx[r];
        idx_s[r] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
    }
    for (int f = tid; f < N * DEG; f += BS_THREADS) tab[f] = 7;          // padding entry: slot 7 never matches an argmax
    __syncthreads();
    for (int r = tid; r < mk; r += BS_THREADS) {
        const int n = idx_s[r];
        int pos = 0;
        for (int q = 0; q < r; ++q) pos += idx_s[q] == n ? 1 : 0;        // rank among the in-edges of n
        const int m = r / KC, j = r - m * KC;
        if (pos < DEG) tab[n * DEG + pos] = m * BS_GSB + j;              // (host guarantees in-degree <= DEG)
    }
    __syncthreads();

    // ---- per-wave constants ----
    // in-edges of this wave's source nodes: row offset of the destination in ga (m * 256) + slot j, in SGPRs
    const int n0 = wave * NPW;
    int ent2[NPW][DEG / 2 + 1];                       // two 16-bit entries per register
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
        for (int q = 0; q < DEG; q += 2) {
            const int n = n0 + i;
            const int e0 = n < N ? tab[n * DEG + q] : 7;
            const int e1 = (n < N && q + 1 < DEG) ? tab[n * DEG + q + 1] : 7;
            ent2[i][q >> 1] = __builtin_amdgcn_readfirstlane(e0 | (e1 << 16));
        }
    // lane l < NPW*DEG fetches the edge-type row of in-edge l of this wave: element offset inside etype[b]
    int et_goff = 0;
    {
        const int i = lane / DEG, q = lane - i * DEG;
        if (lane < NPW * DEG && n0 + i < N) {
            const int e = tab[(n0 + i) * DEG + q];
            const int m = e >> 8, j = e & 0xff;
            et_goff = j < KC ? (m * KC + j) * 4 : 0;                      // padding entries read row 0 (their gz mask is 0)
        }
    }
    // neighbour ids of this wave's destinations, one byte each (N <= 96), for the routed-row select of the detype phase
    const int m0 = wave * DPW;
    unsigned nb[DPW][2];
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
        nb[d][0] = nb[d][1] = 0u;
        if (m0 + d < M) {
#pragma unroll
            for (int j = 0; j < KC; ++j) nb[d][j >> 2] |= (unsigned)idx_s[(m0 + d) * KC + j] << (8 * (j & 3));
        }
        // kept in vector registers: as wave-uniform values they are spilled to VGPR lanes with ~60 other scalars and come back
        // through v_readlane + s_nop in front of every use
        asm volatile("" : "+v"(nb[d][0]), "+v"(nb[d][1]));
    }
    // W fragments.  aP: A of P^T = W^T x (wave = 16-column slab): A[i = col][k = c] = W[c][16 wave + i], 8 consecutive c.
    //               aT: A of dx^T = W dP^T (wave & 3 = 16-channel tile): A[i = c][k = col] = W[ct*16 + i][col], 8 consecutive cols.
    //               Resident (32 VGPRs): re-reading it from L2 per sample had every CU of the chip hammering the same 64 KB —
    //               17 000 cycles per sample in the dx phase (profiles/r02).
    bs_bf16x8 aP[2];
    {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wp = p.W + (int64_t)(32 * ks + 8 * lk) * p.w_ld + wave * 16 + li;
            alignas(16) float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = wp[(int64_t)u * p.w_ld];
            aP[ks] = bs_frag_f32(w8);
        }
    }
    // The last two phases of a sample both only READ the dP image, so they run side by side on the two halves of the
    // workgroup: waves 0-7 accumulate dW (each re-reads x^T and its dP^T columns: halving the waves that do halves that
    // LDS traffic), waves 8-15 produce dx.  The 32 registers each needs for the whole kernel are ONE array R:
    //   waves 0-7 : R[2 pa + s] = dW accumulator of A slot pa (channel 4 i + pa) x column slot 2 (wave & 1) + s of group wave >> 1
    //   waves 8-15: R[ks]       = A fragment of dx^T = W dP^T: W[ct*16 + i][32 ks + 8 lk ..+7] as bf16 (ct = wave & 3)
    const bool dw_wave = wave < 8;
    const int ct = wave & 3;
    f32x4 R[8];
    if (dw_wave) {
#pragma unroll
        for (int t = 0; t < 8; ++t) R[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            R[ks] = __builtin_bit_cast(f32x4, bs_frag_f32(p.W + (int64_t)(ct * 16 + li) * p.w_ld + 32 * ks + 8 * lk));
    }
    // A operand of the channel sums: row 0 adds the even k (first edge type of a pair), row 1 the odd k, other rows nothing
    bs_bf16x8 evod;
    {
        const unsigned w = li == 0 ? 0x00003f80u : (li == 1 ? 0x3f800000u : 0u);
        evod = __builtin_bit_cast(bs_bf16x8, make_uint4(w, w, w, w));
    }
    float gbacc = 0.f;                                // dbias of channel `lane` over this wave's destinations (detype phase)

    // ---- prefetch registers (raw chunks; decoded at the commit) ----
    uint4 px;                                         // one 16-byte x chunk (N * 8 <= 768 chunks)
    // gz / argmax of item tid = eight consecutive channels (m = item >> 3, channels 8 (item & 7) .. + 7): one 16-byte and one
    // 8-byte load per lane of the first M * 8 threads (<= 768).  The vector-memory path issues a wave's load in ~16 cycles
    // whatever its width: uint2 + dword loads on all 16 waves cost ~0.8 k cycles more per sample than these.
    uint4 pg = make_uint4(0, 0, 0, 0);
    uint2 pa = make_uint2(0, 0);
    uint2 pe;                                         // edge-type row (4 bf16) of in-edge `lane` of this wave, next sample's
    const int xchunks = N * 8, gitems = M * 8;
    auto prefetch = [&](int b, int t) {
        const unsigned utid = (unsigned)t;
        const int lane = t & 63;
        px = make_uint4(0, 0, 0, 0);
        if (t < xchunks) px = *bs_at<uint4>(p.x + (int64_t)b * p.x_sb, utid * 16u);
        const uint16_t* gzb = p.gz + (int64_t)b * p.y_sb;
        const uint8_t* amb = p.argmax + (int64_t)b * p.y_sb;
        pg = make_uint4(0, 0, 0, 0);
        pa = make_uint2(0, 0);
        if (t < gitems) {
            const unsigned el = (utid >> 3) * (unsigned)p.y_ld + (utid & 7u) * 8u;      // row m = item >> 3, channels 8 (item & 7) ..
            pg = *bs_at<uint4>(gzb, el * 2u);
            pa = *bs_at<uint2>(amb, el);
        }
        pe = make_uint2(0, 0);
        if (lane < NPW * DEG) pe = *bs_at<uint2>(p.et + (int64_t)b * p.et_sb, (unsigned)et_goff * 2u);
    };
    auto commit = [&](unsigned char* xs, int t) {
        if (t < xchunks) *reinterpret_cast<uint4*>(xs + (t >> 3) * BS_XSB + (t & 7) * 16) = px;
        if (t < gitems) {
            const unsigned gq[4] = {pg.x, pg.y, pg.z, pg.w};
            unsigned w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned a = ((u < 4 ? pa.x : pa.y) >> (8 * (u & 3))) & 7u;      // slots 0..5 (a corrupt byte cannot reach the gz bits)
                const unsigned g2 = gq[u >> 1];
                const unsigned hi = (u & 1) ? (g2 & 0xffff0000u) : (g2 << 16);
                w[u] = hi | (a << 8) | (1u << a);
            }
            unsigned char* gp = ga + (t >> 3) * BS_GSB + (t & 7) * 32;
            *reinterpret_cast<uint4*>(gp) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(gp + 16) = make_uint4(w[4], w[5], w[6], w[7]);
        }
        {   // in-edge (node i, slot q) of this wave -> etw[e][i][q]: wave-private, read by this wave's dP phase only (program order)
            const int l = t & 63;
            if (l < NPW * DEG) {
                const int i = l / DEG, q = l - i * DEG;
                uint16_t* ew = reinterpret_cast<uint16_t*>(etw) + i * QS + q;
                ew[0] = (uint16_t)pe.x; ew[NPW * QS] = (uint16_t)(pe.x >> 16);
                ew[2 * NPW * QS] = (uint16_t)pe.y; ew[3 * NPW * QS] = (uint16_t)(pe.y >> 16);
            }
        }
    };

    const int ntile = Npad / 16;                      // 16-node tiles: 2..6
    const int nkst = Npad / 32;                       // 32-node k-steps of dW
    const int chunk = (p.B + gridDim.x - 1) / gridDim.x;
    const int b_begin = blockIdx.x * chunk;
    const int b_end = min(p.B, b_begin + chunk);
    __syncthreads();                                  // (tables read above; zeros in place)
    if (b_begin < b_end) {
        prefetch(b_begin, tid);
        commit(xs0, tid);
    }
    int cur = 0;
    // VALU issue goes to the older waves first: in every phase waves 8-15 finish ~1 000 cycles behind waves 0-7, and their
    // dx tiles are the longer half of the closing MFMA phase.  A static priority for the younger half evens that out (142 -> 138 us).
    if (wave >= 8) __builtin_amdgcn_s_setprio(3);

    for (int b = b_begin; b < b_end; ++b) {
        // per-lane offsets are re-derived every sample from an opaque copy of the thread id: left to itself the compiler
        // hoists ~20 per-lane 64-bit global pointers out of the loop and spills them (scratch round trips in the hot loop)
        int t = tid;
        asm volatile("" : "+v"(t));
        const int lane = t & 63, li = t & 15, lk = (t >> 4) & 3;
        unsigned char* xs = xs0 + cur * xs_bytes;
        BS_STAMP(0);
        __syncthreads();                              // B_a: x / ga / es of this sample are staged; pd is free
        BS_STAMP(1);
        // the next sample's loads go out here, under the projection's MFMAs, while the vector-memory path is idle (issued
        // after the dP phase, behind this sample's stores, the same five loads took 1 300-1 900 cycles to issue); they are
        // consumed after B_d, ~10 000 cycles from now
        if (b + 1 < b_end) prefetch(b + 1, t);

        // ---- P^T slab (wave = 16-column slab = channels 4 wave .. +3): D[i = col][j = node] ----
        for (int nt0 = 0; nt0 < ntile; nt0 += 2) {       // two node tiles in flight (ntile is even: Npad is a multiple of 32)
            f32x4 acc[2];
            bs_bf16x8 bf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned char* bp = xs + ((nt0 + u) * 16 + li) * BS_XSB + lk * 16;
                bf[u][0] = __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(bp));
                bf[u][1] = __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(bp + 64));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aP[0], bf[u][0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aP[1], bf[u][1], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                *reinterpret_cast<uint2*>(pd + ((nt0 + u) * 16 + li) * BS_PSB + (4 * wave + lk) * 8) =
                    make_uint2(bs_pack2(acc[u][0], acc[u][1]), bs_pack2(acc[u][2], acc[u][3]));
        }
        BS_STAMP(2);
        __syncthreads();                              // B_b: P complete
        BS_STAMP(3);

        // ---- detype: lane = channel; the routed row's products go to the wave's [slot][edge type][channel] image, the sum
        //      over channels is an all-ones MFMA ----
        {
            // the sample's edge-type gradient (4 x mk bf16, contiguous in memory) is collected in LDS and leaves as 16-byte
            // stores after the barrier: two-byte stores scattered over it kept the vector-memory queue busy for ~1 500 cycles
            // (the next sample's staging below waits on that queue)
            uint16_t* gb = reinterpret_cast<uint16_t*>(bs_lds + p.off_gst);
            // stage 1, all destinations of the wave: {gz | argmax} of channel `lane`, then the routed P row (two dependent LDS
            // reads per destination, all in flight together).  Destinations past M (small graphs) read row M - 1 with gz forced
            // to 0 and store nothing.
            unsigned dwv[DPW];
            uint2 pkv[DPW];
#pragma unroll
            for (int d = 0; d < DPW; ++d) {
                const int mm = min(m0 + d, M - 1);
                dwv[d] = *reinterpret_cast<const unsigned*>(ga + mm * BS_GSB + lane * 4);
            }
#pragma unroll
            for (int d = 0; d < DPW; ++d) {
                const unsigned jst = (dwv[d] >> 8) & 7u;
                const unsigned n = __builtin_amdgcn_perm(nb[d][1], nb[d][0], 0x0c0c0c00u | jst);
                pkv[d] = *reinterpret_cast<const uint2*>(pd + n * BS_PSB + lane * 8);
            }
            // stage 2: the phase is bound by LDS cycles, most of them the 16-column operand reads of the image (4 x 8 cycles per
            // pass for 2 KC real columns).  With 3 slots TWO destinations share a pass: columns 0-5 and 6-11.
            constexpr int DPP = KC == 3 ? 2 : 1;          // destinations per pass
#pragma unroll
            for (int d0 = 0; d0 < DPW; d0 += DPP) {
                unsigned* zw[DPP];
#pragma unroll
                for (int u = 0; u < DPP; ++u) {
                    const int d = d0 + u;
                    const bool live = m0 + d < M;
                    const unsigned jst = (dwv[d] >> 8) & 7u;
                    const uint2 pk = pkv[d];
                    const float g = live ? __uint_as_float(dwv[d] & 0xffff0000u) : 0.f;
                    gbacc += g;
                    const unsigned c01 = bs_pack2(g * bs_lo(pk.x), g * bs_hi(pk.x));
                    const unsigned c23 = bs_pack2(g * bs_lo(pk.y), g * bs_hi(pk.y));
                    // column (destination u, slot jst, pair 0) <- {e0 | e1}, (.., pair 1) <- {e2 | e3}, dword `lane` of each
                    zw[u] = reinterpret_cast<unsigned*>(zb + (u * KC + jst) * (2 * BS_ZCS) + lane * 4);
                    zw[u][0] = c01;
                    zw[u][BS_ZCS / 4] = c23;
                }
                asm volatile("" ::: "memory");            // the image is re-read below through another type: keep the stores
                // D[0][col] = sum over channels of the even halves, D[1][col] of the odd halves; K = 64 channels x 2
                const unsigned char* zr = zb + li * BS_ZCS + lk * 16;
                bs_bf16x8 fr[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fr[ks] = __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(zr + 64 * ks));
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < DPP; ++u) {           // (issued right behind the reads: LDS runs a wave's operations in order)
                    zw[u][0] = 0u;
                    zw[u][BS_ZCS / 4] = 0u;
                }
                asm volatile("" ::: "memory");
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};      // two chains of two instead of one of four
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(evod, fr[0], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(evod, fr[2], s1, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(evod, fr[1], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(evod, fr[3], s1, 0, 0, 0);
                // column li of the image is (destination u = li / 2 KC, slot j, pair li & 1); rows 0 / 1 of D are its two edge types
                const int u = li >= 2 * KC ? 1 : 0, lc = li - 2 * KC * u;
                const int m = m0 + d0 + u;
                if (lk == 0 && li < 2 * KC * DPP && m < M) {
                    const int j = lc >> 1, e0 = 2 * (lc & 1);
                    const __bf16 h0 = (__bf16)(s0[0] + s1[0]), h1 = (__bf16)(s0[1] + s1[1]);
                    gb[e0 * mk + m * KC + j] = __builtin_bit_cast(uint16_t, h0);
                    gb[(e0 + 1) * mk + m * KC + j] = __builtin_bit_cast(uint16_t, h1);
                }
            }
        }
        BS_STAMP(4);
        __syncthreads();                              // B_c: every wave is done reading P
        BS_STAMP(5);
        {   // edge-type gradient of this sample: LDS -> memory, 16 bytes per lane (8 mk bytes in all; a 2-byte tail loop for odd sizes)
            const unsigned char* gsrc = bs_lds + p.off_gst;
            unsigned char* gdst = reinterpret_cast<unsigned char*>(p.get + (int64_t)b * 4 * mk);
            const int nbytes = 8 * mk, nvec = ((uintptr_t)gdst & 15) ? 0 : nbytes >> 4;
            if (t < nvec) {
                uint4 v = *reinterpret_cast<const uint4*>(gsrc + t * 16);
                if (p.accum) {                         // second launch of a split call: add to what the first one stored
                    const uint4 o = *bs_at<uint4>(gdst, (unsigned)t * 16u);
                    v = make_uint4(bs_pack2(bs_lo(v.x) + bs_lo(o.x), bs_hi(v.x) + bs_hi(o.x)), bs_pack2(bs_lo(v.y) + bs_lo(o.y), bs_hi(v.y) + bs_hi(o.y)),
                                   bs_pack2(bs_lo(v.z) + bs_lo(o.z), bs_hi(v.z) + bs_hi(o.z)), bs_pack2(bs_lo(v.w) + bs_lo(o.w), bs_hi(v.w) + bs_hi(o.w)));
                }
                *bs_at<uint4>(gdst, (unsigned)t * 16u) = v;
            }
            for (int f = nvec * 8 + t; f < 4 * mk; f += BS_THREADS) {
                uint16_t v = *reinterpret_cast<const uint16_t*>(gsrc + f * 2);
                if (p.accum) v = (uint16_t)(bs_pack2(bs_lo(v) + bs_lo(*bs_at<uint16_t>(gdst, (unsigned)f * 2u)), 0.f) & 0xffffu);
                *bs_at<uint16_t>(gdst, (unsigned)f * 2u) = v;
            }
        }

        // ---- dP: wave owns source nodes n0 .. n0 + NPW - 1, lane = channel.  For one node dP[n][o][e] = sum_q et_q[e] gzm_q[o] over its
        //      in-edges q is a [4 e x DEG] x [DEG x 64 o] product: ONE v_mfma_f32_4x4x4_16b_bf16 per four in-edges (16 independent
        //      4x4x4 blocks = 16 groups of 4 channels; A = the node's edge types, the same in every block; B = the masked gz of
        //      lane = channel, four in-edges in the lane; D = the 4 edge types of channel `lane`, i.e. the lane's 8 bytes of the dP
        //      row).  Per in-edge and lane: one LDS read, a bit-field extract, an AND (the FMA form took 10 VALU instructions per
        //      in-edge: 3 300 cycles of the sample on the vector ALUs) ----
        {
            typedef short bs_s16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int n = n0 + i;
                if (n < N) {
                    unsigned gm[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) gm[q] = 0u;
#pragma unroll
                    for (int q = 0; q < DEG; ++q) {
                        const int e = (ent2[i][q >> 1] >> (16 * (q & 1))) & 0xffff;
                        const unsigned dw = *reinterpret_cast<const unsigned*>(ga + (e & ~0xff) + lane * 4);
                        gm[q] = dw & (unsigned)__builtin_amdgcn_sbfe((int)dw, e & 0xff, 1);         // gz in the high half where this edge won the max
                    }
                    const unsigned char* ap = etw + ((lane & 3) * NPW + i) * (QS * 2);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g4 = 0; g4 < (DEG + 3) / 4; ++g4) {
                        const uint2 av = *reinterpret_cast<const uint2*>(ap + 8 * g4);
                        const uint2 bv = make_uint2(__builtin_amdgcn_perm(gm[4 * g4 + 1], gm[4 * g4], 0x07060302u),
                                                    __builtin_amdgcn_perm(gm[4 * g4 + 3], gm[4 * g4 + 2], 0x07060302u));
                        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bs_s16x4, av), __builtin_bit_cast(bs_s16x4, bv), acc, 0, 0, 0);
                    }
                    *reinterpret_cast<uint2*>(pd + n * BS_PSB + lane * 8) = make_uint2(bs_pack2(acc[0], acc[1]), bs_pack2(acc[2], acc[3]));
                }
            }
        }
        BS_STAMP(6);
        __syncthreads();                              // B_d: dP complete; ga / es of this sample are no longer read
        BS_STAMP(7);

        // ---- stage the next sample (its loads were issued a whole sample ago) ----
        if (b + 1 < b_end) {
            commit(xs0 + (cur ^ 1) * xs_bytes, t);
            BS_STAMP(8);
        }
        BS_STAMP(9);
        if (!dw_wave) {
            // ---- dx^T tiles: D[i = c][j = n] = W[c][:] . dP[n][:], channel tile ct, node tiles ((wave - 8) >> 2) + 2 i ----
            uint16_t* gxb = p.gx + (int64_t)b * p.x_sb;
            uint2 old[3];                             // second launch of a split call: what the first one stored, fetched ahead of the products
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int n = (((wave - 8) >> 2) + 2 * i) * 16 + li;
                old[i] = make_uint2(0u, 0u);
                if (p.accum && n < N) old[i] = *bs_at<uint2>(gxb, (unsigned)(n * NIN + ct * 16 + 4 * lk) * 2u);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int nt = ((wave - 8) >> 2) + 2 * i;
                if (nt < ntile) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const unsigned char* bp = pd + (nt * 16 + li) * BS_PSB + lk * 16;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bs_bf16x8, R[ks]),
                            __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(bp + 64 * ks)), acc, 0, 0, 0);
                    const int n = nt * 16 + li;
                    if (n < N)
                        *bs_at<uint2>(gxb, (unsigned)(n * NIN + ct * 16 + 4 * lk) * 2u) =
                            make_uint2(bs_pack2(acc[0] + bs_lo(old[i].x), acc[1] + bs_hi(old[i].x)),
                                       bs_pack2(acc[2] + bs_lo(old[i].y), acc[3] + bs_hi(old[i].y)));
                }
            }
        } else {
            // ---- dW: contraction over nodes.  A = x^T (four column slots of the 64 channels), B = dP^T column slots
            //      2 (wave & 1), 2 (wave & 1) + 1 of the 64-column group wave >> 1 ----
            for (int kst = 0; kst < nkst; ++kst) {
                const int row0 = 32 * kst + 8 * lk;
                bs_bf16x8 bfr0, bfr1;
                {
                    uint2 rd[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        rd[j] = *reinterpret_cast<const uint2*>(pd + (row0 + j) * BS_PSB + 128 * (wave >> 1) + 8 * li);
                    if (wave & 1) { bfr0 = bs_tr<2>(rd); bfr1 = bs_tr<3>(rd); }
                    else { bfr0 = bs_tr<0>(rd); bfr1 = bs_tr<1>(rd); }
                }
                uint2 rx[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) rx[j] = *reinterpret_cast<const uint2*>(xs + (row0 + j) * BS_XSB + 8 * li);
                {
                    const bs_bf16x8 a0 = bs_tr<0>(rx);
                    R[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bfr0, R[0], 0, 0, 0);
                    R[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bfr1, R[1], 0, 0, 0);
                }
                {
                    const bs_bf16x8 a1 = bs_tr<1>(rx);
                    R[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bfr0, R[2], 0, 0, 0);
                    R[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bfr1, R[3], 0, 0, 0);
                }
                {
                    const bs_bf16x8 a2 = bs_tr<2>(rx);
                    R[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bfr0, R[4], 0, 0, 0);
                    R[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bfr1, R[5], 0, 0, 0);
                }
                {
                    const bs_bf16x8 a3 = bs_tr<3>(rx);
                    R[6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, bfr0, R[6], 0, 0, 0);
                    R[7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, bfr1, R[7], 0, 0, 0);
                }
            }
        }
        BS_STAMP(10);
        cur ^= 1;
    }   // samples

    // ---- flush dW tiles and dbias into this workgroup's slab (summed by the slab reduce, fixed order) ----
    if (b_begin < b_end) {
        float* slab = p.ws + (int64_t)blockIdx.x * ((int64_t)NIN * NCOLS + NOU);
        if (dw_wave) {
#pragma unroll
            for (int pa_ = 0; pa_ < 4; ++pa_)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int col = 64 * (wave >> 1) + 4 * li + 2 * (wave & 1) + sl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(int64_t)(4 * (4 * lk + r) + pa_) * NCOLS + col] = R[2 * pa_ + sl][r];
                }
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(bs_lds + p.off_red);         // [16 waves][64 channels]
        red[wave * 64 + lane] = gbacc;
        __syncthreads();
        if (tid < NOU) {
            float s = 0.f;
            for (int w = 0; w < BS_WAVES; ++w) s += red[w * 64 + tid];
            slab[(int64_t)NIN * NCOLS + tid] = s;
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
int fgnn_mpconv_backward_ws(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                            float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                            fgnn_stream_t stream);

#define BS_REJECT(code) do { if (getenv("FGNN_TRACE")) fprintf(stderr, "[fgnn] sg backward rejects shape: rule %d\n", code); return 0; } while (0)

// Returns 1 if launched, 0 if the call is outside this kernel's family, <0 on error.  d->reserved carries the largest
// in-degree of the (batch-shared) neighbour table as the caller measured it (0 = unknown: not this kernel).
int fgnn_mpconv_backward_sg(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                            const float* filters, const void* gz, const uint8_t* argmax, void* gx, void* getype,
                            float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                            fgnn_stream_t stream) {
    static const bool off = getenv("FGNN_NO_SG") != nullptr;
    if (off) BS_REJECT(0);
    if (d->dtype != FGNN_BF16 || d->ext != FGNN_EXT_NONE || d->agg != FGNN_AGG_MAX || d->net != 4) BS_REJECT(1);
    // 64 -> 128: the output channels are independent and every gradient is a sum over them, so the call runs as TWO launches
    // of this 64 -> 64 kernel over the halves of gz / argmax / W's columns; the second one ADDS to gx and getype (one more bf16
    // rounding of those two) and its dW / dbias land in the upper halves of gfilters' columns / gbias.
    static const bool no_split = getenv("FGNN_SG_NOSPLIT") != nullptr;
    const bool split = d->nin == 64 && d->nou == 128 && !no_split;
    if (d->nin == 128 && d->nou == 64) {              // third generation, two launches over the input-channel halves (mpconv_bwd_ws.hip); 0 = not its shape
        const int r = fgnn_mpconv_backward_ws(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias, workspace,
                                              workspace_bytes, stream);
        if (r != 0) return r;
    }
    if (d->nin != 64 || (d->nou != 64 && !split)) BS_REJECT(2);
    if (d->k != 3 && d->k != 6) BS_REJECT(3);
    if (d->idx_sb != 0 && d->B > 1) BS_REJECT(4);
    if (!(d->idx_sk == 1 && d->idx_sm == d->k)) BS_REJECT(5);
    if (!getype || !argmax || !gbias) BS_REJECT(6);
    if (d->N < 1 || d->N > BS_MAXN || d->M < 1 || d->M > 96) BS_REJECT(7);
    const int indeg = d->reserved & 0xffff;
    const int KC = d->k, DEG = KC == 6 ? 3 : 6;        // LDPC: degree-6 checks <-> degree-3 variables
    if (indeg < 1 || indeg > DEG) BS_REJECT(8);
    if (!(d->x_sc == 1 && d->x_sn == d->nin && d->x_sb % 8 == 0)) BS_REJECT(9);
    if (!(d->y_sc == 1 && (d->y_sm == d->nou || d->M == 1) && d->y_sb % 8 == 0)) BS_REJECT(10);
    if (!(d->et_se == 1 && d->et_sk == 4 && (d->et_sm == 4 * d->k || d->M == 1) && d->et_sb % 4 == 0)) BS_REJECT(11);
    if (((uintptr_t)x & 15) || ((uintptr_t)gz & 15) || ((uintptr_t)etype & 7) || ((uintptr_t)argmax & 7) ||
        ((uintptr_t)gx & 7)) BS_REJECT(12);
    const int64_t nw = (int64_t)d->nin * 64 * 4, slab_len = nw + 64;                  // of ONE launch (64 output channels)
    if (!workspace || workspace_bytes < (split ? 2 : 1) * 256 * slab_len * 4) BS_REJECT(13);
    const int NPW = (d->N + BS_WAVES - 1) / BS_WAVES, DPW = (d->M + BS_WAVES - 1) / BS_WAVES;
    void* fn = nullptr;
    const int GSL = (d->M * 16 + BS_THREADS - 1) / BS_THREADS;
    if (KC == 6 && NPW <= 6 && DPW <= 3 && GSL == 1) fn = (void*)mpconv_bwd_sg_kernel<6, 3, 6, 3, 1>;
    else if (KC == 3 && NPW <= 3 && DPW <= 6) fn = GSL == 1 ? (void*)mpconv_bwd_sg_kernel<3, 6, 3, 6, 1> : (void*)mpconv_bwd_sg_kernel<3, 6, 3, 6, 2>;
    if (!fn) BS_REJECT(14);

    {                                                 // third generation first (mpconv_bwd_ws.hip; also splits 64 -> 128); 0 = not its shape
        const int r = fgnn_mpconv_backward_ws(d, x, nn_idx, etype, filters, gz, argmax, gx, getype, gfilters, gbias, workspace,
                                              workspace_bytes, stream);
        if (r != 0) return r;
    }
    BsParams p;
    p.x = (const uint16_t*)x; p.idx = nn_idx; p.et = (const uint16_t*)etype; p.W = filters;
    p.gz = (const uint16_t*)gz; p.argmax = argmax; p.gx = (uint16_t*)gx; p.get = (uint16_t*)getype;
    p.ws = (float*)workspace;
    p.B = d->B; p.N = d->N; p.M = d->M; p.Npad = fgnn_round_up(d->N, 32);
    p.NPW = KC == 6 ? 6 : 3; p.DPW = KC == 6 ? 3 : 6;
    p.x_sb = d->x_sb; p.et_sb = d->et_sb; p.y_sb = d->y_sb;
    p.y_ld = d->nou; p.w_ld = d->nou * 4; p.accum = 0;
    int off_b = 0;
    auto take = [&](int bytes) { const int o = off_b; off_b = fgnn_round_up(off_b + bytes, 16); return o; };
    p.off_xs = take(2 * p.Npad * BS_XSB);
    p.off_pd = take(p.Npad * BS_PSB);
    p.off_ga = take(d->M * BS_GSB);
    p.zbytes = (KC == 3 ? 2 : 1) * KC * 2 * BS_ZCS;                      // 3 slots: two destinations per image
    p.off_z = take(BS_WAVES * p.zbytes + 16 * BS_ZCS);                  // + slack: the 16-column tile reads past column 2 KC
    p.off_es = take(BS_WAVES * 4 * p.NPW * (DEG > 4 ? 16 : 8));         // per-wave [4 e][NPW][8 | 4 slots] bf16 edge-type images
    p.off_tab = take(d->N * DEG * 4);
    p.off_idx = take(d->M * KC * 4);
    p.off_gst = take(8 * d->M * KC);
    p.off_red = p.off_xs;                                               // dbias partials: after the last sample (16 KB)
    const int lds = off_b > 16384 ? off_b : 16384;
    if (lds > 160 * 1024) BS_REJECT(15);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    int grid = 256;
    if (grid > d->B) grid = d->B;
    const int chunk = (d->B + grid - 1) / grid;
    grid = (d->B + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    fgnn_note_kernel(split ? "mpconv_bwd_sg_kernel<%d, %d, %d, %d, %d> x2" : "mpconv_bwd_sg_kernel<%d, %d, %d, %d, %d>", KC, DEG, p.NPW, p.DPW, GSL);
    p.prof = nullptr;
#ifdef FGNN_ENABLE_PROF
    static long long* prof_buf = nullptr;
    if (getenv("FGNN_PROF")) {
        if (!prof_buf) (void)hipMalloc(&prof_buf, 64 * 8);
        (void)hipMemset(prof_buf, 0, 64 * 8);
        p.prof = prof_buf;
    }
#endif
    void* args[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(BS_THREADS), args, lds, st);
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv sg backward launch: %s", hipGetErrorString(e));
#ifdef FGNN_ENABLE_PROF
    if (p.prof) {                                     // tuning aid: phase timeline of one sample, waves 0 / 4 / 8 / 12 (shader clocks)
        long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, p.prof, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w) {
            fprintf(stderr, "[fgnn prof sg bwd] wave %d:", 4 * w);
            for (int i = 0; i < 11; ++i) fprintf(stderr, " %lld", h[w * 16 + i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (!split) fgnn_launch_slab_reduce(p.ws, grid, slab_len, nw, gfilters, gbias, st);
    else {
        // slab rows are 256 columns of gfilters' 512: lower half, then the second launch on the upper 64 output channels
        fgnn_launch_slab_reduce_ld(p.ws, grid, slab_len, nw, 256, 512, gfilters, gbias, st);
        p.ws += (int64_t)grid * slab_len;            // (its own slabs: the first launch's fold may be a recorded one, fold_batch.hip)
        p.W += 256; p.gz += 64; p.argmax += 64; p.accum = 1;
        e = hipLaunchKernel(fn, dim3(grid), dim3(BS_THREADS), args, lds, st);
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv sg backward launch (upper half): %s", hipGetErrorString(e));
        fgnn_launch_slab_reduce_ld(p.ws, grid, slab_len, nw, 256, 512, gfilters + 256, gbias + 64, st);
    }
    e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "mpconv backward helper launch: %s", hipGetErrorString(e));
    return 1;
}
