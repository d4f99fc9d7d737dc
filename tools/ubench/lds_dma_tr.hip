// Semantics probe (tuning aid, not product code): (1) global_load_lds_dwordx4 — where lane l's 16 bytes land relative to M0;
// (2) ds_read_b64_tr_b16 — which (lane, element) of the 16-lane group's loaded 4x16 block each result element comes from.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_tr.hip -o tools/ubench/lds_dma_tr && tools/ubench/lds_dma_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void k_dma(const uint32_t* g, uint32_t* out, int base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const int src_chunk = (lane * 7 + 3) & 63;                       // a permutation of the 64 16-byte chunks
    const uint32_t* gsrc = g + src_chunk * 4;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k_tr(uint16_t* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) reinterpret_cast<uint16_t*>(lds)[i] = (uint16_t)i;
    __syncthreads();
    // mode 0: lane l points at elements 4l .. 4l+3 (dense); mode 1: rows of 64 elements: lane (g = l>>4, i = l&15) -> row (i>>2), cols 16g + 4(i&3)
    unsigned el = mode == 0 ? 4u * lane : (unsigned)(((lane & 15) >> 2) * 64 + 16 * (lane >> 4) + 4 * (lane & 3));
    unsigned addr = (unsigned)(uintptr_t)lds + el * 2u;
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(r.x & 0xffff); out[lane * 4 + 1] = (uint16_t)(r.x >> 16);
    out[lane * 4 + 2] = (uint16_t)(r.y & 0xffff); out[lane * 4 + 3] = (uint16_t)(r.y >> 16);
}

int main() {
    uint32_t* g; uint32_t* out; uint16_t* o16;
    hipMalloc(&g, 4096 * 4); hipMalloc(&out, 4096 * 4); hipMalloc(&o16, 256 * 2);
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int base : {0, 2048, 100000}) {
        const int ldsb = base > 16384 ? 140 * 1024 : 16384;
        hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        // for the large base only the first 16 KB is dumped: re-point the dump window by running with base inside it
        k_dma<<<1, 64, ldsb>>>(g, out, base > 16384 ? 4096 : base);
        hipMemcpy(h.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
        printf("DMA base %d: ", base);
        int first = -1, n = 0, ok = 1;
        for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeefu) { if (first < 0) first = i; ++n; }
        printf("first written dword %d, %d dwords written; ", first, n);
        for (int l = 0; l < 64 && first >= 0; ++l) {
            const int want = ((l * 7 + 3) & 63) * 4;
            for (int d = 0; d < 4; ++d) if (h[first + l * 4 + d] != (uint32_t)(want + d)) ok = 0;
        }
        printf("lane l -> M0 + 16 l: %s\n", ok ? "YES" : "NO");
    }
    for (int mode = 0; mode < 2; ++mode) {
        k_tr<<<1, 64, 16384>>>(o16, mode);
        std::vector<uint16_t> r(256);
        hipMemcpy(r.data(), o16, 512, hipMemcpyDeviceToHost);
        printf("TR mode %d (value = element index the result came from):\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
