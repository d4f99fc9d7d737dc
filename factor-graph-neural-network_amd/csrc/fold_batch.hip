// fold_batch.hip — the parameter-gradient slab folds of a backward pass as ONE launch at its end (round 5).
//
// Every gradient kernel of this library that sums over the batch rows — the message operator's filter / bias gradient
// (autograd through /root/reference/lib/model/mpnn/mp_nn.py:115-175), the node-wise maps' weight gradient (mp_nn_residual.py:25-35,
// base_model.py:43-90) — writes one partial slab per workgroup and a small kernel folds the slabs, in a fixed order, into the
// caller's accumulators.  Nothing in a backward pass READS a parameter gradient, yet each fold was its own launch right behind its
// producer: ~100 launches of 5-15 us per LDPC training step, 0.9 ms of the step's wall time (gpurun_out/r05t/timeline: wgb_reduce_kernel
// 64 x, bres_reduce_kernel 35 x), a third of them in the main stream's dependent chain.  With deferral switched on
// (fgnn_fold_defer(1), per call, by a caller that owns the slabs until the flush) the producers only RECORD their fold; one
// fgnn_fold_flush() launches a single kernel over all recorded jobs (a table of <= 48 jobs in the kernel arguments; jobs whose
// targets overlap — a parameter used twice — go to separate launches, in recording order, so every sum keeps its order).
// Fixed summation order per element (bit-reproducible); not the immediate folds' order: the two agree to rounding.
#include "fgnn_common.h"
#include <mutex>
#include <vector>

#define FB_MAXJOBS 48

struct FgnnFoldJob {
    const float* ws;
    float* gW;
    float* gb;
    int64_t slab_len, nw;
    int nslab, kind;        // kind 0: gW[(i / a) * b + i % a] (a = ncols, b = ld); kind 1 / 2: the node-wise maps' register-order slabs (2: the LDS-staged kernel's natural tile order)
                            // (kind 1: nw = the elements to fold, slab_len = the distance between slabs — a merged launch's slabs
                            // hold several maps' slices, linear_wgrad_b16.hip::wb_launch; kind 0: slab_len is both)
    int a, b, c, d;         // kind 1: a = S, b = nso, c = Cin, d = Cout
};

struct FbBatch {
    FgnnFoldJob job[FB_MAXJOBS];
    int first[FB_MAXJOBS + 1];      // first workgroup of each job
    int njobs;
};

static std::mutex g_mu;
static std::vector<FgnnFoldJob> g_jobs;
static int g_defer = 0;

extern "C" int fgnn_fold_defer(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int was = g_defer;
    g_defer = on ? 1 : 0;
    return was;
}

extern "C" int fgnn_fold_pending(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_jobs.size();
}

extern "C" void fgnn_fold_discard(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_jobs.clear();
}

// Called by the producers' host code instead of launching their fold: true = recorded (deferral is on), false = launch it yourself.
bool fgnn_fold_push(const float* ws, int nslab, int64_t slab_len, int64_t nw, float* gW, float* gb, int kind, int a, int b, int c, int d) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_defer || (slab_len & 3) || ((uintptr_t)ws & 15)) return false;       // (16-byte loads across slabs: odd slab lengths fold at once)
    FgnnFoldJob j = {ws, gW, gb, slab_len, nw, nslab, kind, a, b, c, d};
    g_jobs.push_back(j);
    return true;
}

#define WB_NACC 68          // linear_wgrad_b16.hip: 64 gW accumulators + 4 dbias partials per lane

// 1024 threads = 8 slab groups x 128 lanes of 16 bytes: a workgroup folds 512 CONSECUTIVE elements (2 KB per slab) — the slabs were
// written a whole backward pass ago and come from HBM, where 256-byte pieces 64-280 KB apart (the immediate folds' shape, fine for
// slabs still in the infinity cache) open a DRAM page per line.  Group g walks slabs g, g+8, ... (four independent loads in flight
// per lane); group 0 adds the 8 partial sums in order.  NOT the immediate folds' summation order (16 groups): recorded and
// immediate gradients agree to rounding, each is bit-reproducible run to run.
#define FB_LANES 128
#define FB_GROUPS 8
#define FB_PER_WG (4 * FB_LANES)

__global__ __launch_bounds__(1024) void fold_batch_kernel(const FbBatch t) {
    __shared__ f32x4 part[FB_GROUPS][FB_LANES];
    int j = 0;
    while (j + 1 < t.njobs && (int)blockIdx.x >= t.first[j + 1]) ++j;          // (<= 48 scalar steps)
    const FgnnFoldJob& q = t.job[j];
    const int l = threadIdx.x & (FB_LANES - 1), g = threadIdx.x / FB_LANES;
    const int64_t i0 = (int64_t)((int)blockIdx.x - t.first[j]) * FB_PER_WG + 4 * l;      // slab_len % 4 == 0 for every producer
    const bool in = i0 < (q.kind != 0 ? q.nw : q.slab_len);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (in) {
        const float* base = q.ws + i0;
        int w = g;
        for (; w + 3 * FB_GROUPS < q.nslab; w += 4 * FB_GROUPS) {
            s0 += *reinterpret_cast<const f32x4*>(base + (int64_t)w * q.slab_len);
            s1 += *reinterpret_cast<const f32x4*>(base + (int64_t)(w + FB_GROUPS) * q.slab_len);
            s2 += *reinterpret_cast<const f32x4*>(base + (int64_t)(w + 2 * FB_GROUPS) * q.slab_len);
            s3 += *reinterpret_cast<const f32x4*>(base + (int64_t)(w + 3 * FB_GROUPS) * q.slab_len);
        }
        for (; w < q.nslab; w += FB_GROUPS) s0 += *reinterpret_cast<const f32x4*>(base + (int64_t)w * q.slab_len);
    }
    part[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0 || !in) return;
    f32x4 s = part[0][l];
#pragma unroll
    for (int u = 1; u < FB_GROUPS; ++u) s += part[u][l];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const int64_t i = i0 + e4;
        if (q.kind == 0) {
            const int64_t o = q.a == q.b ? i : (i / q.a) * q.b + i % q.a;
            if (i < q.nw) q.gW[o] += s[e4];
            else if (q.gb) q.gb[i - q.nw] += s[e4];
        } else {
            const int nso = q.b, Cin = q.c;
            const int e = (int)(i & 63);
            const int r4 = (int)((i >> 6) % WB_NACC), slice = (int)(i / (WB_NACC * 64));
            const int so = slice % nso, sc = slice / nso;
            const int li = e & 15, lk = e >> 4;
            if (r4 < 64) {
                const int r = r4 & 3, b = (r4 >> 2) & 3, a = r4 >> 4;
                const bool nat = q.kind == 2;
                const int o = nat ? so * 64 + 16 * a + 4 * lk + r : so * 64 + 4 * (4 * lk + r) + a;
                const int c = nat ? sc * 64 + 16 * b + li : sc * 64 + 4 * li + b;
                q.gW[(int64_t)o * Cin + c] += s[e4];
            } else if (q.gb && sc == 0 && lk == 0) {
                q.gb[q.kind == 2 ? so * 64 + 16 * (r4 - 64) + li : so * 64 + 4 * li + (r4 - 64)] += s[e4];
            }
        }
    }
}

static bool fb_overlap(const FgnnFoldJob& x, const FgnnFoldJob& y) {
    auto ext = [](const FgnnFoldJob& j) -> int64_t { return j.kind == 0 ? (j.a == j.b ? j.nw : (j.nw / j.a) * (int64_t)j.b) : (int64_t)j.c * j.d; };
    const float* x0 = x.gW; const float* x1 = x.gW + ext(x);
    const float* y0 = y.gW; const float* y1 = y.gW + ext(y);
    if (x0 < y1 && y0 < x1) return true;
    if (x.gb && y.gb) {                                   // bias vectors: at most a few hundred floats; compare generously
        const float* a0 = x.gb; const float* a1 = x.gb + 512;
        const float* b0 = y.gb; const float* b1 = y.gb + 512;
        if (a0 < b1 && b0 < a1) return true;
    }
    return false;
}

// Launch the recorded folds on `stream` (which must be ordered behind every producer) and forget them.  Returns FGNN_OK.
extern "C" int fgnn_fold_flush(fgnn_stream_t stream) {
    std::vector<FgnnFoldJob> jobs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        jobs.swap(g_jobs);
    }
    hipStream_t st = (hipStream_t)stream;
    std::vector<char> done(jobs.size(), 0);
    size_t left = jobs.size();
    while (left) {
        FbBatch t;
        t.njobs = 0;
        int blocks = 0;
        std::vector<size_t> skipped;                      // jobs passed over in this round: a later job that overlaps one of them must wait too
        for (size_t n = 0; n < jobs.size() && t.njobs < FB_MAXJOBS; ++n) {
            if (done[n]) continue;
            bool clash = false;
            for (int m = 0; m < t.njobs && !clash; ++m) clash = fb_overlap(jobs[n], t.job[m]);
            for (size_t m = 0; m < skipped.size() && !clash; ++m) clash = fb_overlap(jobs[n], jobs[skipped[m]]);
            if (clash) { skipped.push_back(n); continue; }
            t.first[t.njobs] = blocks;
            t.job[t.njobs++] = jobs[n];
            blocks += (int)(((jobs[n].kind != 0 ? jobs[n].nw : jobs[n].slab_len) + FB_PER_WG - 1) / FB_PER_WG);
            done[n] = 1;
            --left;
        }
        t.first[t.njobs] = blocks;
        if (blocks > 0) hipLaunchKernelGGL(fold_batch_kernel, dim3(blocks), dim3(1024), 0, st, t);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) FGNN_FAIL(FGNN_ELAUNCH, "fold_batch launch: %s", hipGetErrorString(e));
    }
    return FGNN_OK;
}
