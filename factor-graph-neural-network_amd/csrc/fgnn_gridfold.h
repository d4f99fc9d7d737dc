// fgnn_gridfold.h — grid-wide deterministic reductions WITHOUT a finaliser launch.
//
// Every batch-statistics BatchNorm of a training step (reference: torch.nn.BatchNorm2d behind conv1 / conv2 of mp_conv_residual,
// /root/reference/lib/model/mpnn/mp_nn_residual.py:25-35, and mp_conv_v2.bn, mp_nn.py:57-58,170-173) needs per-channel sums over ALL
// rows between the kernel that produces a tensor and the kernel that normalises it.  Rounds 1-4 wrote one partial row per
// workgroup and folded them in a separate 4-10 us kernel: 160 such launches per LDPC step, each a node of the replayed hipGraph
// (profiles/r04/train_step_timeline.txt).  This file is the alternative round 5 built and measured (OFF by default: it lost, see
// bnact.hip): the PRODUCING kernel folds its own partials, the workgroup that finishes last does it ("last block done", two levels
// so that no workgroup ever reads more than FGNN_FOLD_GROUP + FGNN_FOLD_MAXGROUPS rows):
//
//   level 1: workgroups are grouped by row index, FGNN_FOLD_GROUP consecutive rows per group; the last workgroup of a group to
//            arrive (a ticket counter per group) sums the group's rows in row order into one f64 row of `part2`;
//   level 2: the last GROUP to finish (one more ticket) sums the part2 rows in group order -> `sums` in LDS, and the caller's
//            finaliser (BatchNorm statistics, backward sums, ...) runs in that one workgroup.
//
// Fixed grouping + fixed summation order = bit-reproducible whatever the arrival order.
//
// Visibility across the 8 XCDs' L2s WITHOUT cache maintenance: the partial rows, the second-level rows and the tickets are only ever
// touched with AGENT-SCOPE relaxed atomics (gfx950: the sc1 bit on the store / load / RMW: written through to, and read from, the
// memory side of the non-coherent L2s); a workgroup waits for its stores to be acknowledged (s_waitcnt vmcnt(0)) and meets at a
// barrier before ONE thread takes the ticket, and the reading workgroup issues its loads only after it has seen the last ticket.
// The first form of this file used __threadfence() (agent-scope release / acquire fences: buffer_wbl2 + buffer_inv by every wave
// of every workgroup): each one writes back the XCD's whole L2 under the kernel's own output stream — the LDPC step went from
// 15.3 to 29.7 ms (gpurun_out/r05a).  The ticket counters are zero between launches (the last arrival resets them); they live
// in a small zero-initialised buffer per (device, stream) owned by the host side (ops._fold_scratch), so kernels of one stream —
// which never overlap — share it.
#pragma once
#include "fgnn_common.h"

#define FGNN_FOLD_GROUP 16          // partial rows per first-level group
#define FGNN_FOLD_MAXGROUPS 64      // => up to 1024 partial rows (= BN_MAXPART)
#define FGNN_FOLD_MAXJ 512          // slots per row (2 x 256 channels)
#define FGNN_FOLD_TICKET_BYTES 512  // 1 + FGNN_FOLD_MAXGROUPS counters, padded
#define FGNN_FOLD_BYTES (FGNN_FOLD_TICKET_BYTES + FGNN_FOLD_MAXGROUPS * FGNN_FOLD_MAXJ * 8)

struct FgnnFold {
    float* part;          // partial rows: slot j of row w at part[w * row_stride + (j / ch) * half_stride + (j % ch)], j < 2 * ch
    double* part2;        // [FGNN_FOLD_MAXGROUPS][2 * ch] second-level rows
    unsigned* tickets;    // [1 + FGNN_FOLD_MAXGROUPS], zero between launches; NULL = no in-kernel fold (the host launches a finaliser)
    int rows;             // partial rows of this launch (= its workgroups)
    int row_stride, half_stride, ch;
};

static inline FgnnFold fgnn_fold_make(float* part, void* scratch, int rows, int ch, int row_stride = 0, int half_stride = 0) {
    FgnnFold f;
    f.part = part;
    f.tickets = (unsigned*)scratch;
    f.part2 = scratch ? (double*)((char*)scratch + FGNN_FOLD_TICKET_BYTES) : nullptr;
    f.rows = rows; f.ch = ch;
    f.row_stride = row_stride ? row_stride : 2 * ch;
    f.half_stride = half_stride ? half_stride : ch;
    return f;
}

// Default: producers only write their partial rows and the host launches the small finaliser kernels (bnact.hip) behind them.
// FGNN_INKERNEL_FINALISERS=1 / fgnn_set_inkernel_finalisers(1): the fold below runs in the producer instead (fewer graph nodes, but
// measured SLOWER on MI355X: see bnact.hip).
int fgnn_separate_finalisers(void);

// host-side plumbing shared by the translation units (not C ABI)
void fgnn_stats_pending(const fgnn_bn_final** fin, void** scratch);          // mpconv_fwd.hip: the finalisation the NEXT operator-forward launch carries
void fgnn_stats_upper_half(FgnnFold* fold, fgnn_bn_final* fin);               // second launch of a 64 -> 128 call
int fgnn_bn_finalize_launch(const float* partials, int npartials, int C, const fgnn_bn_final* fin, hipStream_t st);     // bnact.hip (stand-alone finalisers)
int fgnn_bn_bwd_final_raw_launch(const float* partials, int npartials, int C, const float* mean, const float* invstd, float* dsum,
                                 float* gweight, float* gbias, hipStream_t st);

#ifdef __HIPCC__
// A partial-row element: written through to the memory side (agent scope), whichever XCD the writer runs on.
__device__ __forceinline__ void fgnn_fold_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL threads of EVERY workgroup of the launch, after the workgroup's partial row `row` has been written WITH
// fgnn_fold_store (by any of its threads; no barrier needed in between — this function waits for the stores and synchronises).
// Returns true in exactly ONE workgroup, the last to arrive, with sums[0 .. 2 ch) (LDS, >= 2 * ch doubles, provided by the
// caller) = the fold of all rows.  All threads of a workgroup get the same answer.  The workgroup's own LDS may be reused for
// `sums` as long as nothing else reads it afterwards.
__device__ __forceinline__ bool fgnn_grid_fold(const FgnnFold& f, double* sums, int row) {
    __shared__ unsigned fgnn_fold_flag;
    const int tid = threadIdx.x, T = blockDim.x;
    const int J = 2 * f.ch;
    const int g = row / FGNN_FOLD_GROUP;
    const int ng = (f.rows + FGNN_FOLD_GROUP - 1) / FGNN_FOLD_GROUP;
    const int gfirst = g * FGNN_FOLD_GROUP;
    const int gsize = f.rows - gfirst < FGNN_FOLD_GROUP ? f.rows - gfirst : FGNN_FOLD_GROUP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial-row stores have been acknowledged by the memory side
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.tickets + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fgnn_fold_flag = old == (unsigned)(gsize - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!fgnn_fold_flag) return false;
    for (int j = tid; j < J; j += T) {                 // level 1: this group's rows, in row order
        const int off = (j / f.ch) * f.half_stride + (j % f.ch);
        float v[FGNN_FOLD_GROUP];
#pragma unroll
        for (int w = 0; w < FGNN_FOLD_GROUP; ++w)
            v[w] = w < gsize ? __hip_atomic_load(f.part + (int64_t)(gfirst + w) * f.row_stride + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < FGNN_FOLD_GROUP; ++w) a += (double)v[w];
        __hip_atomic_store(f.part2 + (int64_t)g * J + j, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(f.tickets + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (nobody else touches this counter any more in this launch)
        const unsigned old = __hip_atomic_fetch_add(f.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fgnn_fold_flag = old == (unsigned)(ng - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!fgnn_fold_flag) return false;
    for (int j = tid; j < J; j += T) {                 // level 2: the groups' rows, in group order
        double a = 0.0;
#pragma unroll 8
        for (int q = 0; q < ng; ++q) a += __hip_atomic_load(f.part2 + (int64_t)q * J + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sums[j] = a;
    }
    if (tid == 0) __hip_atomic_store(f.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
}

// ---- the finalisers that run in the last workgroup (the same arithmetic as the round 1-4 finaliser kernels) -----------------------

// Forward batch statistics from sums[c] = sum (y - K), sums[ch + c] = sum (y - K)^2, K = fin.shift_k (or 0): mean / invstd /
// scale / shift, running statistics (unbiased variance over fin.population rows), num_batches_tracked += 1.
__device__ __forceinline__ void fgnn_bn_final_apply(const fgnn_bn_final& fin, int ch, const double* sums) {
    if (fin.num_batches_tracked && threadIdx.x == 0) *fin.num_batches_tracked += 1;      // BatchNorm2d.num_batches_tracked
    for (int c = threadIdx.x; c < ch; c += blockDim.x) {
        const double n = (double)fin.count;
        const double m0 = sums[c] / n;                 // mean of (y - K)
        double var = sums[ch + c] / n - m0 * m0;
        if (var < 0.0) var = 0.0;
        const float mu = (float)(m0 + (fin.shift_k ? (double)fin.shift_k[c] : 0.0));
        const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
        fin.mean[c] = mu;
        fin.invstd[c] = is;
        const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
        fin.scale[c] = g * is;
        fin.shift[c] = b - mu * g * is;
        if (fin.running_mean) {
            const double np = (double)(fin.population > 0 ? fin.population : fin.count);
            const double unbiased = np > 1.0 ? var * np / (np - 1.0) : var;
            fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * mu;
            fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
        }
    }
}

// Backward sums: sums[c] = sum g (= dbeta), sums[ch + c] = sum g xhat (= dgamma) — or, with mean / invstd given, sum g x with the RAW
// x: dgamma = invstd (S1 - mean S0).  Written to dsum [2][ch] and ACCUMULATED into gweight / gbias (may be NULL).
__device__ __forceinline__ void fgnn_bn_bwd_final_apply(int ch, const double* sums, const float* mean, const float* invstd,
                                                        float* dsum, float* gweight, float* gbias) {
    for (int c = threadIdx.x; c < ch; c += blockDim.x) {
        const double s0 = sums[c];
        const double s1 = mean ? (double)invstd[c] * (sums[ch + c] - (double)mean[c] * s0) : sums[ch + c];
        dsum[c] = (float)s0;
        dsum[ch + c] = (float)s1;
        if (gbias) gbias[c] += (float)s0;
        if (gweight) gweight[c] += (float)s1;
    }
}
#endif
