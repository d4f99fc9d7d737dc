// Issue rate of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD: NACC independent accumulators, round-robin, all operands
// distinct registers.  Prints shader cycles per MFMA (s_memtime).  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, long long* cyc, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[(kk + i) & 15], acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int WAVES>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 64 * WAVES * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    k<NACC, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NACC, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * NACC;
    printf("%-24s %6.1f cycles/MFMA/wave  (%.1f per SIMD)  wall %.1f us -> %.1f TF\n", name, h / n, h / n / ((WAVES + 3) / 4),
           ms * 1e3, 256.0 * WAVES * n * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
    run<1, 4>("1 acc, 1 wave/SIMD");
    run<2, 4>("2 acc, 1 wave/SIMD");
    run<4, 4>("4 acc, 1 wave/SIMD");
    run<8, 4>("8 acc, 1 wave/SIMD");
    run<2, 8>("2 acc, 2 waves/SIMD");
    run<4, 8>("4 acc, 2 waves/SIMD");
    return 0;
}
