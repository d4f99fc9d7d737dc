// Shared device/host helpers for the FGNN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "fgnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define FGNN_THREADS 256
#define FGNN_WAVES 4

// ---- storage-type helpers: kernels compute in f32, store f32 or bf16 ------------------
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float fgnn_ld(const float* p) { return *p; }
__device__ __forceinline__ float fgnn_ld(const bf16_t* p) {
    return __uint_as_float(((uint32_t)p->v) << 16);
}
__device__ __forceinline__ void fgnn_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void fgnn_st(bf16_t* p, float v) {
    // native fptrunc: v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN preserved)
    const __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
}

// four consecutive elements (16-B / 8-B aligned)
__device__ __forceinline__ void fgnn_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void fgnn_st4(bf16_t* p, f32x4 v) {
    typedef __bf16 bf16x4_n __attribute__((ext_vector_type(4)));
    const bf16x4_n h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4_n*>(p) = h;
}
__device__ __forceinline__ f32x4 fgnn_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 fgnn_ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                   __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}

// ---- host-side error plumbing ---------------------------------------------------------
void fgnn_set_error(const char* fmt, ...);
void fgnn_note_kernel(const char* fmt, ...);   // records which kernel a dispatch chose (fgnn_last_kernel)
#define FGNN_FAIL(code, ...) do { fgnn_set_error(__VA_ARGS__); return (code); } while (0)

static inline int fgnn_round_up(int v, int m) { return (v + m - 1) / m * m; }
