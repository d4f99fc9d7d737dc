// What can a second wave on the same SIMD issue while the first streams v_mfma_f32_16x16x4_f32?
// Waves 0-3 (one per SIMD): MFMA stream (mfma_iters x 64 MFMAs).  Waves 4-7: `kind` work: 0 = v_fma_f32 chain x8 independent,
// 1 = v_pk_fma_f32, 2 = ds_read_b128 (conflict-free), 3 = v_mov/int adds.  Prints cycles of both with the partner on / off.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, bool BF16>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int mfma_iters, int other_iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (!BF16) {
            float a[16], b[16];
            for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
            f32x4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[(kk + i) & 15], acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        } else {
            typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            bf8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f * i); }
            f32x16 acc[2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[kk & 1], 0, 0, 0);
            }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        }
    } else {
        if (KIND == 0) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
            for (int it = 0; it < other_iters; ++it)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
            for (int i = 0; i < 8; ++i) s += v[i];
        } else if (KIND == 1) {
            f32x2 v[8];
            for (int i = 0; i < 8; ++i) v[i] = (f32x2){(float)threadIdx.x + i, 1.f};
            const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
            for (int it = 0; it < other_iters; ++it)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
            for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
        } else if (KIND == 2) {
            f32x4 v[8];
            const float* base = lds + (threadIdx.x & 63) * 4;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"((unsigned)(size_t)base), "n"(0));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            for (int i = 0; i < 8; ++i) s += v[i][0];
        } else {
            int v[8];
            for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
            for (int it = 0; it < other_iters; ++it)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(3));
            for (int i = 0; i < 8; ++i) s += v[i];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int KIND, bool BF16>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    long long h[8];
    const int OI = 50;                       // 50 x 64 = 3200 partner instructions
    const int MI = BF16 ? 200 : 100;         // 6400 MFMAs of 32 cycles
    int cfg[3][2] = {{MI, 0}, {0, OI}, {MI, OI}};
    printf("%-34s", name);
    for (int c = 0; c < 3; ++c) {
        k<KIND, BF16><<<256, 512>>>(out, cyc, cfg[c][0], cfg[c][1]);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("  [mfma %6lld  other %6lld]", h[0], h[4]);
    }
    printf("   (mfma only | other only | both)\n");
}

int main() {
    run<0, false>("f32 MFMA + v_fma_f32");
    run<1, false>("f32 MFMA + v_pk_fma_f32");
    run<2, false>("f32 MFMA + ds_read_b128");
    run<3, false>("f32 MFMA + v_add_u32");
    run<0, true>("bf16 MFMA + v_fma_f32");
    run<1, true>("bf16 MFMA + v_pk_fma_f32");
    run<2, true>("bf16 MFMA + ds_read_b128");
    return 0;
}
