// Layout check of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 products per instruction) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_4x4x4_layout.hip -o /tmp/m4 && /tmp/m4
// Assumed: lane l -> block l / 4; A: row i = l % 4, four k in the lane; B: column j = l % 4, four k in the lane;
//          D: column j = l % 4, rows i = 0..3 in the lane's four registers.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][4][4] (b,i,k), B [16][4][4] (b,k,j), D [16][4][4] (b,i,j)
    const int l = threadIdx.x, b = l >> 2, r = l & 3;
    bf16x4 a, bb;
    for (int kk = 0; kk < 4; ++kk) { a[kk] = (__bf16)A[(b * 4 + r) * 4 + kk]; bb[kk] = (__bf16)B[(b * 4 + kk) * 4 + r]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + r] = c[i];
}
int main() {
    float hA[256], hB[256], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[(b * 4 + i) * 4 + kk] * hB[(b * 4 + kk) * 4 + j];
        if (fabsf(s - hD[(b * 4 + i) * 4 + j]) > 1e-3f) ++bad;
    }
    printf("4x4x4 layout %s (%d mismatches)\n", bad ? "WRONG" : "as assumed", bad);
    return bad != 0;
}
