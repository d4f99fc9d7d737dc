// fgnn_api.hip — error plumbing and shape-only helpers of the C ABI (include/fgnn_hip.h).
#include "fgnn_common.h"

static thread_local char g_err[512] = "";

void fgnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// process-wide (diagnostic only): autograd runs the backward on its own thread and tests ask from the main one
static char g_kernel[160] = "";

void fgnn_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

extern "C" const char* fgnn_last_error(void) { return g_err; }
extern "C" const char* fgnn_last_kernel(void) { return g_kernel; }
extern "C" int fgnn_abi_version(void) { return FGNN_ABI_VERSION; }

// x read once, etype read once, indices read once (int64 as passed; a batch-shared
// graph is read once), y written once, filters + bias/BN vectors once.
extern "C" int64_t fgnn_mpconv_algorithmic_bytes(const fgnn_mpconv_desc* d) {
    if (!d) return -1;
    const int64_t s = d->dtype == FGNN_BF16 ? 2 : 4;
    const int64_t B = d->B, mk = (int64_t)d->M * d->k;
    const int64_t R = d->ext == FGNN_EXT_NONE ? d->nin : 2 * d->nin;
    int64_t bytes = s * B * ((int64_t)d->nin * d->N + (int64_t)d->nou * d->M);
    bytes += s * (d->et_sb == 0 ? 1 : B) * d->net * mk;
    bytes += 8 * (d->idx_sb == 0 ? 1 : B) * mk;
    bytes += 4 * (R * d->nou * d->net + 5 * (int64_t)d->nou);
    return bytes;
}

// diagnostic: one device timestamp (the 100 MHz constant clock) written by a one-thread kernel on `stream` — placed inside a
// captured step it tells when that point of the stream is reached in a replay WITHOUT a profiler attached (tools/stamps.py)
__global__ void fgnn_stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

extern "C" int fgnn_stamp(void* dst, void* stream) {
    if (!dst) { fgnn_set_error("fgnn_stamp: null destination"); return FGNN_EINVAL; }
    hipLaunchKernelGGL(fgnn_stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long*>(dst));
    return FGNN_OK;
}

// diagnostic: one thread spins until the device clock has advanced by `ticks` (100 MHz).  At the head of a captured step it gives
// the HOST a head start: under rocprofv3 the host enqueues a multi-branch graph more slowly than the device runs its short kernels,
// and a kernel trace then shows the host's enqueue order instead of the graph's schedule; with every node already
// in its queue when the spin ends, the trace shows what an unprofiled replay does.
__global__ void fgnn_spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int fgnn_spin(int64_t ticks, void* stream) {
    if (ticks < 0 || ticks > 100000000) { fgnn_set_error("fgnn_spin: 0 .. 1e8 ticks (1 s)"); return FGNN_EINVAL; }
    hipLaunchKernelGGL(fgnn_spin_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), (unsigned long long)ticks);
    return FGNN_OK;
}
