// sum_n.hip — out = a_0 + a_1 + ... + a_{n-1} over dense same-layout tensors (n <= 8), one pass.
// The FGNN layer feeds every variable / factor state into 3-5 consumers (v2v map, two V->F blocks, the residual,
// a skip link: /root/reference/lib/model/mpnn/factor_mpnn_sp.py:139-170), so its gradient is a sum of 3-5 tensors;
// autograd adds them pairwise (n-1 kernels, 3(n-1) passes over memory), this kernel reads n and writes 1.
#include "fgnn_common.h"

#define SN_MAX 8
struct SnParams {
    const void* in[SN_MAX];
    void* out;
    int64_t nchunks;     // 16-byte chunks
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void sum_n_kernel(const SnParams p) {
    constexpr int EPC = 16 / sizeof(T);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.nchunks; i += stride) {
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < SN_MAX; ++q) {
            if (q < p.n) {
                const uint4 v = reinterpret_cast<const uint4*>(p.in[q])[i];
                if constexpr (sizeof(T) == 2) {
                    acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
                    acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
                    acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
                    acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
                } else {
                    acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y);
                    acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
                }
            }
        }
        uint4 o;
        if constexpr (sizeof(T) == 2) {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            b2 h;
            h[0] = (__bf16)acc[0]; h[1] = (__bf16)acc[1]; o.x = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[2]; h[1] = (__bf16)acc[3]; o.y = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[4]; h[1] = (__bf16)acc[5]; o.z = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[6]; h[1] = (__bf16)acc[7]; o.w = __builtin_bit_cast(unsigned, h);
        } else {
            o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
        }
        reinterpret_cast<uint4*>(p.out)[i] = o;
    }
}

// out = sum of n dense arrays of `numel` elements (numel * elem size a multiple of 16, 16-byte aligned pointers).
extern "C" int fgnn_sum_n(const void* const* inputs, int n, int64_t numel, int dtype, void* out, fgnn_stream_t stream) {
    if (!inputs || !out || n < 1 || n > SN_MAX) FGNN_FAIL(FGNN_EINVAL, "sum_n: 1..%d inputs", SN_MAX);
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "sum_n: unknown dtype %d", dtype);
    const int64_t bytes = numel * (dtype == FGNN_F32 ? 4 : 2);
    if (numel < 0 || bytes % 16) FGNN_FAIL(FGNN_EUNSUPPORTED, "sum_n: size must be a multiple of 16 bytes");
    if (numel == 0) return FGNN_OK;
    SnParams p;
    for (int q = 0; q < SN_MAX; ++q) {
        p.in[q] = q < n ? inputs[q] : nullptr;
        if (q < n && (!inputs[q] || ((uintptr_t)inputs[q] & 15))) FGNN_FAIL(FGNN_EINVAL, "sum_n: null / unaligned input %d", q);
    }
    if ((uintptr_t)out & 15) FGNN_FAIL(FGNN_EINVAL, "sum_n: unaligned output");
    p.out = out; p.nchunks = bytes / 16; p.n = n;
    int64_t g = (p.nchunks + 255) / 256;
    if (g > 4096) g = 4096;
    if (dtype == FGNN_F32) hipLaunchKernelGGL(sum_n_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(sum_n_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "sum_n launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}


// ----------------------------------------------------------------------------------------
// node_sum: out[b][c] = sum_m g[b][m][c] over dense channel-fastest rows — the backward of a per-sample row BROADCAST over the
// sample's M nodes.  The LDPC hyper-factor's message to the variables (/root/reference/train_ldpc.py:40-46,82-88: one source node,
// hetype == 1, hnn_idx_f2v == 0) is the same C-vector for all 96 variables of a codeword, so this build carries it as [B][C] and
// lets the consumer add it as a broadcast (fgnn_block_tail_apply / fgnn_bn_apply, addend_period); its gradient is this sum.
// A thread owns one 16-byte channel chunk of one sample and walks the sample's M rows (f32 accumulation, one pass over g).
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void node_sum_kernel(const T* g, T* out, int64_t B, int M, int C) {
    constexpr int EPC = 16 / sizeof(T);
    const int cpr = C / EPC;
    const int64_t total = B * cpr;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t b = t / cpr;
        const int ch = (int)(t - b * cpr);
        const uint4* src = reinterpret_cast<const uint4*>(g + (b * M) * C) + ch;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll 8
        for (int m = 0; m < M; ++m) {
            const uint4 v = src[(int64_t)m * cpr];
            if constexpr (sizeof(T) == 2) {
                acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
                acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
                acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
                acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
            } else {
                acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y);
                acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
            }
        }
        uint4 o;
        if constexpr (sizeof(T) == 2) {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            b2 h;
            h[0] = (__bf16)acc[0]; h[1] = (__bf16)acc[1]; o.x = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[2]; h[1] = (__bf16)acc[3]; o.y = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[4]; h[1] = (__bf16)acc[5]; o.z = __builtin_bit_cast(unsigned, h);
            h[0] = (__bf16)acc[6]; h[1] = (__bf16)acc[7]; o.w = __builtin_bit_cast(unsigned, h);
        } else {
            o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
        }
        reinterpret_cast<uint4*>(out + b * C)[ch] = o;
    }
}

// out [B][C] = sum over the M rows of each sample of g [B][M][C] (dense, 16-byte aligned, C * elem size a multiple of 16).
extern "C" int fgnn_node_sum(const void* g, void* out, int64_t B, int M, int C, int dtype, fgnn_stream_t stream) {
    if (!g || !out) FGNN_FAIL(FGNN_EINVAL, "node_sum: null pointer");
    if (dtype != FGNN_F32 && dtype != FGNN_BF16) FGNN_FAIL(FGNN_EINVAL, "node_sum: unknown dtype %d", dtype);
    const int epc = dtype == FGNN_F32 ? 4 : 8;
    if (B < 0 || M < 1 || C < 1 || C % epc || ((uintptr_t)g & 15) || ((uintptr_t)out & 15))
        FGNN_FAIL(FGNN_EUNSUPPORTED, "node_sum: C must be a multiple of %d and the tensors 16-byte aligned", epc);
    if (B == 0) return FGNN_OK;
    int64_t grid = (B * (C / epc) + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (dtype == FGNN_F32) hipLaunchKernelGGL(node_sum_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                                              (const float*)g, (float*)out, B, M, C);
    else hipLaunchKernelGGL(node_sum_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                            (const bf16_t*)g, (bf16_t*)out, B, M, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "node_sum launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}


// ----------------------------------------------------------------------------------------
// out[s][n] = [ a[s][n] | b[s][n] ]: two arrays of samples x `inner` chunks (ua / ub 16-byte units each, any sample / chunk strides:
// slices of a larger activation qualify) interleaved chunk by chunk —
// torch.cat of two channel-fastest activations along the node axis (inner = 1, chunk = nodes x channels:
// /root/reference/lib/model/mpnn/factor_mpnn.py:104-107, variables and factors of one type stacked for a block) or along the channel
// axis (inner = nodes, chunk = channels: factor_mpnn.py:116, the blocks' messages in front of the merge map).  torch runs one strided
// copy kernel per input (~9 us each at the synthetic-PGM shapes: 100 launches = 0.9 ms of BASELINE config 5's step); this is one
// 16-byte-per-lane pass.
// ----------------------------------------------------------------------------------------
struct CcParams {
    const uint4* a;
    const uint4* b;
    uint4* out;
    unsigned ua, ub;     // chunk sizes in 16-byte units
    unsigned inner;      // chunks per sample
    int64_t sab, san, sbb, sbn;      // strides (16-byte units) of a / b: per sample, per chunk within the sample
    int64_t total;       // samples * inner * (ua + ub)
};

__global__ __launch_bounds__(256) void concat_pair_kernel(const CcParams p) {
    const unsigned u = p.ua + p.ub;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += stride) {
        const int64_t o = i / u;
        const unsigned r = (unsigned)(i - o * u);
        const int64_t s = o / p.inner;
        const unsigned n = (unsigned)(o - s * p.inner);
        p.out[i] = r < p.ua ? p.a[s * p.sab + n * p.san + r] : p.b[s * p.sbb + n * p.sbn + (r - p.ua)];
    }
}

// the node-axis form for inputs whose ROWS are strided (a channel slice of a wider channel-fastest activation — what torch.cat's own
// backward hands on): out[s][r][:] = r < ra ? a[s][r][:] : b[s][r - ra][:], rows of uc 16-byte units
struct CrParams {
    const uint4* a;
    const uint4* b;
    uint4* out;
    unsigned uc, ra, rb;
    int64_t sas, sar, sbs, sbr;      // strides in 16-byte units: per sample, per row
    int64_t total;                   // samples * (ra + rb) * uc
};

__global__ __launch_bounds__(256) void concat_rows_kernel(const CrParams p) {
    const unsigned rows = p.ra + p.rb;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += stride) {
        const int64_t row = i / p.uc;
        const unsigned u = (unsigned)(i - row * p.uc);
        const int64_t s = row / rows;
        const unsigned r = (unsigned)(row - s * rows);
        p.out[i] = r < p.ra ? p.a[s * p.sas + r * p.sar + u] : p.b[s * p.sbs + (r - p.ra) * p.sbr + u];
    }
}

extern "C" int fgnn_concat_rows(const void* a, const void* b, void* out, int64_t samples, int64_t rows_a, int64_t rows_b, int64_t row_bytes,
                                int64_t a_sample_stride_bytes, int64_t a_row_stride_bytes, int64_t b_sample_stride_bytes,
                                int64_t b_row_stride_bytes, fgnn_stream_t stream) {
    if (!a || !b || !out) FGNN_FAIL(FGNN_EINVAL, "concat_rows: null pointer");
    if (samples < 0 || rows_a < 1 || rows_b < 1 || rows_a + rows_b > 0x7fffffff || row_bytes <= 0 || row_bytes > (int64_t)1 << 34 ||
        (row_bytes | a_sample_stride_bytes | a_row_stride_bytes | b_sample_stride_bytes | b_row_stride_bytes) % 16 ||
        a_sample_stride_bytes < 0 || a_row_stride_bytes < 0 || b_sample_stride_bytes < 0 || b_row_stride_bytes < 0)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "concat_rows: rows and strides must be non-negative multiples of 16 bytes");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "concat_rows: pointers must be 16-byte aligned");
    if (samples == 0) return FGNN_OK;
    CrParams p;
    p.a = (const uint4*)a; p.b = (const uint4*)b; p.out = (uint4*)out;
    p.uc = (unsigned)(row_bytes / 16); p.ra = (unsigned)rows_a; p.rb = (unsigned)rows_b;
    p.sas = a_sample_stride_bytes / 16; p.sar = a_row_stride_bytes / 16; p.sbs = b_sample_stride_bytes / 16; p.sbr = b_row_stride_bytes / 16;
    p.total = samples * (rows_a + rows_b) * (int64_t)p.uc;
    int64_t grid = (p.total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(concat_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "concat_rows launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}

extern "C" int fgnn_concat_pair(const void* a, const void* b, void* out, int64_t samples, int64_t inner, int64_t chunk_a_bytes,
                                int64_t chunk_b_bytes, int64_t a_sample_stride_bytes, int64_t a_chunk_stride_bytes,
                                int64_t b_sample_stride_bytes, int64_t b_chunk_stride_bytes, fgnn_stream_t stream) {
    if (!a || !b || !out) FGNN_FAIL(FGNN_EINVAL, "concat_pair: null pointer");
    if (samples < 0 || inner < 1 || inner > 0x7fffffff || chunk_a_bytes <= 0 || chunk_b_bytes <= 0 ||
        chunk_a_bytes > (int64_t)1 << 34 || chunk_b_bytes > (int64_t)1 << 34 ||
        (chunk_a_bytes | chunk_b_bytes | a_sample_stride_bytes | a_chunk_stride_bytes | b_sample_stride_bytes | b_chunk_stride_bytes) % 16 ||
        a_sample_stride_bytes < 0 || a_chunk_stride_bytes < 0 || b_sample_stride_bytes < 0 || b_chunk_stride_bytes < 0)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "concat_pair: chunks and strides must be non-negative multiples of 16 bytes");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) FGNN_FAIL(FGNN_EUNSUPPORTED, "concat_pair: pointers must be 16-byte aligned");
    if (samples == 0) return FGNN_OK;
    CcParams p;
    p.a = (const uint4*)a; p.b = (const uint4*)b; p.out = (uint4*)out;
    p.ua = (unsigned)(chunk_a_bytes / 16); p.ub = (unsigned)(chunk_b_bytes / 16);
    p.inner = (unsigned)inner;
    p.sab = a_sample_stride_bytes / 16; p.san = a_chunk_stride_bytes / 16; p.sbb = b_sample_stride_bytes / 16; p.sbn = b_chunk_stride_bytes / 16;
    p.total = samples * inner * (int64_t)(p.ua + p.ub);
    int64_t grid = (p.total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(concat_pair_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "concat_pair launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
