// ldpc_decode.hip — the reference's classical baseline, MacKay's probability-domain sum-product decoder
// (`zb2x` -> `bndecode`, /root/reference/lib/data/MNC/bnd/bnd.cpp:150-371), for a batch of received words
// (SURVEY §8f rank 4).  The reference decodes one word per pybind11 call in double precision; here one wavefront
// owns one word: the per-edge messages (dqc, pc0, pc1) live in LDS, a lane runs one check's forward/backward
// products (horizontal pass) and one or two variables' downward/upward products (vertical pass) in exactly the
// reference's operation order, in float64 with contraction off — so pseudo-posteriors, hard decisions, violated-check
// counts and iteration counts equal the compiled reference's bit for bit (tests/test_ldpc_datapath_gpu.py).
// Defaults of `bnd_defaults`: clip 0.9999999999, tinydiv 1e-40, no fudge; target syndrome 0.
#pragma clang fp contract(off)
#include "fgnn_common.h"
#include <stdint.h>

#define DEC_MAXE 1024     // edges of the code (LDS: 3 doubles each)
#define DEC_MAXD 16       // variables per check / checks per variable

struct DecParams {
    const double* bias;   // [B][N]  P(bit = 1)
    const int* col_ptr;   // [N+1]   edges of variable n: col_ptr[n] .. col_ptr[n+1]-1, in the alist's order
    const int* row_ptr;   // [M+1]
    const int* row_edge;  // [E]     a check's edges in increasing variable order
    const int* row_var;   // [E]     ... and their variables
    uint8_t* x;           // [B][N]  hard decisions
    double* q1;           // [B][N]  pseudo-posteriors, or NULL
    int* viol;            // [B]     violated checks at exit (0 = decoded)
    int* iters;           // [B]     iterations run
    int N, M, E, loops;
    double tinydiv, clip;
};

__global__ __launch_bounds__(64) void ldpc_decode_kernel(const DecParams p) {
    __shared__ double dqc[DEC_MAXE], pc0[DEC_MAXE], pc1[DEC_MAXE];
    __shared__ uint8_t xs[DEC_MAXE];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int N = p.N, M = p.M;
    const double* bias = p.bias + b * N;
    for (int n = lane; n < N; n += 64) {
        const double d = 1.0 - 2.0 * bias[n];                                   // bnd_load_dqc
        for (int e = p.col_ptr[n]; e < p.col_ptr[n + 1]; ++e) dqc[e] = d;
    }
    for (int n = lane; n < N; n += 64) if (p.q1) p.q1[b * N + n] = 0.0;
    __syncthreads();
    int viol = M, it = 0;
    for (it = 1; it <= p.loops; ++it) {
        for (int m = lane; m < M; m += 64) {                                    // horizontal pass
            const int r0 = p.row_ptr[m], L = p.row_ptr[m + 1] - r0;
            double dpf[DEC_MAXD + 1];
            int ed[DEC_MAXD];
            dpf[0] = 1.0;
#pragma unroll
            for (int l = 0; l < DEC_MAXD; ++l)
                if (l < L) { ed[l] = p.row_edge[r0 + l]; dpf[l + 1] = dqc[ed[l]] * dpf[l]; }
            double dpr = 1.0;
#pragma unroll
            for (int l = DEC_MAXD - 1; l >= 0; --l)
                if (l < L) {
                    const double t = dqc[ed[l]];
                    const double dpc = dpf[l] * dpr * 0.5;
                    dpr = t * dpr;
                    pc0[ed[l]] = 0.5 + dpc;
                    pc1[ed[l]] = 0.5 - dpc;
                }
        }
        __syncthreads();
        for (int n = lane; n < N; n += 64) {                                    // vertical pass
            const int e0 = p.col_ptr[n], U = p.col_ptr[n + 1] - e0;
            const double bn = bias[n];
            double qt0[DEC_MAXD + 1], qt1[DEC_MAXD + 1];
            qt0[0] = 1.0 - bn;
            qt1[0] = bn;
#pragma unroll
            for (int u = 0; u < DEC_MAXD; ++u)
                if (u < U) { qt0[u + 1] = qt0[u] * pc0[e0 + u]; qt1[u + 1] = qt1[u] * pc1[e0 + u]; }
            double q = -1.0;
#pragma unroll
            for (int u = 0; u <= DEC_MAXD; ++u)
                if (u == U) {
                    const double s = qt0[u] + qt1[u];
                    if (s > p.tinydiv) q = qt1[u] / s;
                }
            if (q >= 0.0) {
                if (p.q1) p.q1[b * N + n] = q;
                xs[n] = q >= 0.5 ? 1 : 0;
            } else if (it == 1) {
                xs[n] = 0;                                                      // never assigned: the reference's q1 is 0
            }
            double qb0 = 1.0, qb1 = 1.0;
#pragma unroll
            for (int u = DEC_MAXD - 1; u >= 0; --u)
                if (u < U) {
                    const double a0 = pc0[e0 + u], a1 = pc1[e0 + u];
                    const double qc0 = qt0[u] * qb0, qc1 = qt1[u] * qb1;
                    qb0 = qb0 * a0;
                    qb1 = qb1 * a1;
                    const double s = qc0 + qc1, d = qc0 - qc1;
                    double v = 0.0;
                    if (s > p.tinydiv) {
                        v = d / s;
                        if (v > p.clip) v = p.clip;
                        else if (v < -p.clip) v = -p.clip;
                    }
                    dqc[e0 + u] = v;
                }
        }
        __syncthreads();
        viol = 0;                                                               // bnd_score_state
        for (int m0 = 0; m0 < M; m0 += 64) {
            const int m = m0 + lane;
            int par = 0;
            if (m < M)
                for (int r = p.row_ptr[m]; r < p.row_ptr[m + 1]; ++r) par ^= xs[p.row_var[r]];
            viol += __popcll(__ballot(par != 0));
        }
        if (viol == 0) break;
    }
    if (it > p.loops) it = p.loops;
    for (int n = lane; n < N; n += 64) p.x[b * N + n] = xs[n];
    if (lane == 0) { p.viol[b] = viol; p.iters[b] = it; }
}

// Decodes B words of an (N variables, M checks, E edges) code.  bias [B][N] float64 = P(bit = 1).  The code's incidence
// comes as CSR-like device tables: col_ptr [N+1] (edge e = col_ptr[n] + u is variable n's u-th check in the alist's
// order), row_ptr [M+1] with row_edge / row_var [E] listing each check's edges and variables in increasing variable
// order.  At most 16 edges per variable / per check, E <= 1024.  x [B][N] hard decisions, q1 [B][N] float64
// pseudo-posteriors (or NULL), viol [B] violated checks at exit, iters [B] iterations run (<= loops).
extern "C" int fgnn_ldpc_decode(const double* bias, const int32_t* col_ptr, const int32_t* row_ptr,
                                const int32_t* row_edge, const int32_t* row_var, int64_t B, int N, int M, int E,
                                int loops, uint8_t* x, double* q1, int32_t* viol, int32_t* iters, fgnn_stream_t stream) {
    if (B < 0 || N < 1 || M < 1 || E < 1 || N > DEC_MAXE || E > DEC_MAXE || loops < 1)
        FGNN_FAIL(FGNN_EUNSUPPORTED, "ldpc_decode: N=%d M=%d E=%d (N, E <= %d) loops=%d", N, M, E, DEC_MAXE, loops);
    if (B == 0) return FGNN_OK;
    if (!bias || !col_ptr || !row_ptr || !row_edge || !row_var || !x || !viol || !iters)
        FGNN_FAIL(FGNN_EINVAL, "ldpc_decode: null pointer");
    DecParams p = {bias, col_ptr, row_ptr, row_edge, row_var, x, q1, viol, iters, N, M, E, loops, 1e-40, 0.9999999999};
    fgnn_note_kernel("ldpc_decode_kernel");
    hipLaunchKernelGGL(ldpc_decode_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "ldpc_decode launch: %s", hipGetErrorString(e));
    return FGNN_OK;
}
