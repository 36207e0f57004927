// Microbenchmark: how fast can a workgroup stream rows of 2 KB into LDS with global_load_lds when one wave
// instruction covers (A) 16 rows x 64 bytes, (B) 8 rows x 128 bytes, (C) 4 rows x 256 bytes or (D) one row x 1 KB?
// Same bytes in flight (ring of S stages of 8 KB per workgroup, 2 workgroups per CU), nothing but the loads and the
// barriers of the band-cost kernel's k loop.  Build: hipcc -O3 --offload-arch=gfx950 dma_pattern.hip -o dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// A "chunk" = 128 rows x 2 KB (ky = 4 layers x 32 rows), read as NK = 32 k-slabs of 64 B per row = 8 KB per slab.
// 512 threads = 8 waves; a slab is 8 wave-instructions of 1 KB.  SEG = contiguous bytes of one row per instruction.
template <int SEG, int S>
__global__ __launch_bounds__(512, 4) void k_dma(const char* __restrict__ src, long chunks, int* sink) {
    __shared__ __attribute__((aligned(1024))) char ring[S * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = SEG / 16;        // lanes per row segment
    constexpr int RPI = 64 / LPR;        // rows per instruction
    constexpr int SPI = SEG / 64;        // k-slabs covered by one instruction's row segment
    // a group of SPI slabs = SPI * 8 KB is fetched by 8 * SPI instructions; instruction q of the group covers rows
    // [q * RPI, (q + 1) * RPI) of the 128-row chunk... wave w issues instructions w, w + 8, ...
    int acc = 0;
    for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const char* base = src + c * (128l * 2048);
        constexpr int NG = 32 / SPI;     // groups per chunk
        constexpr int GS = S / SPI;      // groups in the ring
        auto issue = [&](int g) {
#pragma unroll
            for (int q = 0; q < SPI; q++) {
                const int ins = wave + 8 * q;            // 0 .. 8 * SPI - 1
                const int row = ins * RPI + lane / LPR;  // 0 .. 127
                const char* p = base + (long)row * 2048 + (long)g * SEG + (lane % LPR) * 16;
                char* dst = ring + ((g % GS) * SPI * 8192) + ins * 1024;
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)dst, 16, 0, 0);
            }
        };
#pragma unroll
        for (int g = 0; g < GS - 1; g++) issue(g);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int younger = (NG - 1 - g) < (GS - 2) ? (NG - 1 - g) : (GS - 2);
            if (younger >= 6) wait_vm<6 * SPI>();
            else if (younger == 5) wait_vm<5 * SPI>();
            else if (younger == 4) wait_vm<4 * SPI>();
            else if (younger == 3) wait_vm<3 * SPI>();
            else if (younger == 2) wait_vm<2 * SPI>();
            else if (younger == 1) wait_vm<1 * SPI>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (g + GS - 1 < NG) issue(g + GS - 1);
            acc += *reinterpret_cast<volatile int*>(ring + ((g % GS) * SPI * 8192) + tid * 4);
        }
        __syncthreads();
    }
    if (acc == 0x12345678) *sink = acc;
}

template <int SEG, int S>
static double run(const char* d, long chunks, int* sink, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_dma<SEG, S><<<grid, 512>>>(d, chunks, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 3; i++) k_dma<SEG, S><<<grid, 512>>>(d, chunks, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return 3.0 * chunks * 128 * 2048 / (ms * 1e-3) / 1e12;
}

int main() {
    const long chunks = 96 * 1024;   // 96 Ki chunks x 256 KB = 24 GiB
    char* d; int* sink;
    if (hipMalloc(&d, chunks * 128 * 2048) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(d, 1, chunks * 128 * 2048);
    const int grid = 512;            // 2 workgroups per CU, persistent over the chunks
    printf("16 rows x  64 B per instruction, 8 stages: %.2f TB/s\n", run<64, 8>(d, chunks, sink, grid));
    printf(" 8 rows x 128 B per instruction, 8 stages: %.2f TB/s\n", run<128, 8>(d, chunks, sink, grid));
    printf(" 4 rows x 256 B per instruction, 8 stages: %.2f TB/s\n", run<256, 8>(d, chunks, sink, grid));
    printf("16 rows x  64 B per instruction, 5 stages (32 KB in flight, as the 256 B shape): %.2f TB/s\n", run<64, 5>(d, chunks, sink, grid));
    printf(" 8 rows x 128 B per instruction, 6 stages (32 KB in flight): %.2f TB/s\n", run<128, 6>(d, chunks, sink, grid));
    printf("16 rows x  64 B per instruction, 6 stages: %.2f TB/s\n", run<64, 6>(d, chunks, sink, grid));
    printf("grid 1024: 16 x 64 B, 8 stages: %.2f TB/s\n", run<64, 8>(d, chunks, sink, 1024));
    return 0;
}
