// Issue-rate microbenchmark of the VALU instructions the gather is made of (gfx950).  One wave per SIMD
// (256-thread blocks x 256 CUs), 8 independent chains per lane, N iterations; prints cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, unsigned seed, int waves_per_simd) {
    float a[8]; unsigned u = seed + threadIdx.x, w = seed * 3 + threadIdx.x;
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
    float b = 1.0001f, c = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#define OP_DOT2C(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u), "v"(w));
#define OP_DOT2(i)  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u), "v"(w));
#define OP_FMA(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_FMAC(i)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX3(i)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX(i)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_CMPSEL(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define OP_CVT(i)   asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define OP_LSHL(i)  asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(u), "v"(w));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&b), "v"(*(double*)&c));
        if (OP == 0) { REP8(OP_DOT2C) } else if (OP == 1) { REP8(OP_DOT2) } else if (OP == 2) { REP8(OP_FMA) }
        else if (OP == 3) { REP8(OP_FMAC) } else if (OP == 4) { REP8(OP_MAX3) } else if (OP == 5) { REP8(OP_MAX) }
        else if (OP == 6) { REP8(OP_CMPSEL) } else if (OP == 7) { REP8(OP_CVT) } else if (OP == 8) { REP8(OP_MOV) }
        else if (OP == 9) { REP8(OP_ADD) } else if (OP == 10) { REP8(OP_LSHL) } else if (OP == 11) { REP8(OP_ANDOR) }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int threads, float* out, long long* cyc, int per_iter) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 7u, threads / 256);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 7u, threads / 256); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double insts = (double)ITER * per_iter;
    printf("%-22s %d waves/SIMD: %6.2f memtime-ticks per wave-instruction (%.1f us total)\n", name, threads / 256, h / insts, ms * 1e3);
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_dot2c_f32_bf16", threads, out, cyc, 8); run<1>("v_dot2_f32_bf16 (VOP3P)", threads, out, cyc, 8);
        run<2>("v_fma_f32", threads, out, cyc, 8); run<3>("v_fmac_f32", threads, out, cyc, 8);
        run<4>("v_max3_f32", threads, out, cyc, 8); run<5>("v_max_f32", threads, out, cyc, 8);
        run<6>("v_cmp+v_cndmask (2)", threads, out, cyc, 16); run<7>("v_cvt_pk_bf16_f32", threads, out, cyc, 8);
        run<8>("v_mov_b32", threads, out, cyc, 8); run<9>("v_add_u32", threads, out, cyc, 8);
        run<10>("v_lshlrev_b32", threads, out, cyc, 8); run<11>("v_and_or_b32", threads, out, cyc, 8);
    }
    return 0;
}
