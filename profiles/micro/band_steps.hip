// Microbenchmark: the level-0 band-cost kernel (speech-vecalign_amd/csrc/svx_band.hip, k_band_costs3) rebuilt step by
// step around its k loop, to see which step costs the bandwidth.  Same shape as the kernel: 512 threads, two
// workgroups per CU, a chunk = 32 rows x 4 layers per side of 2 KB bf16 rows; the target rows travel global -> LDS by
// LDS-DMA through a ring of S = 8 stages of 8 KB filled and drained in groups of G = 2 slabs (one barrier and one
// counted s_waitcnt per group), the source rows go to registers (one 16-byte load per lane per slab).
//
//   step 0  LDS-DMA of the target rows only                      (one 64-byte slab of 128 rows per stage)
//   step 1  + the source rows' register loads
//   step 2  + the compute phase: 8 ds_read_b128 + 8 MFMA 16x16x32 per wave and slab
//   step 3  + the epilogue: a 20 KB output image per chunk written from LDS in 16-byte runs
//   step 4  + the halo: chunks advance by 18 rows of the 32 they stage (the kernel's mean at band 14), so every input
//           byte is fetched 1.78 times, all but the first time from L2 (the kernel's real access pattern)
//
// Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/band_steps.hip -o /tmp/band_steps && /tmp/band_steps [pairs]
// Output of the round-3 run: profiles/r03_micro_band_steps.txt.
//
// History: round 2's last GPU call ran an earlier, uncommitted version of this program whose third variant aborted
// with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (gpurun_out/band_steps.txt); that source was lost, so its two
// printed figures (6.41 / 5.65 TB/s) were withdrawn and this program re-derives them.  Every address below is formed
// from an index that is checked against the allocation on the host (check_extents) before the first launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512, ROWS = 32, LAYERS = 4, ROWB = 2048, SLAB = 64, NK = ROWB / SLAB;  // 32 slabs per row
constexpr int S = 8, G = 2, STAGE = LAYERS * ROWS * SLAB;                                         // 8 KB per stage
constexpr int DOC = 4096;                                                                          // rows per layer
constexpr int OUT_BYTES = 37 * 10 * 14 * 4;                                                        // one chunk's costs
constexpr int STRIDE_HALO = 18;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int swz(int r) { return (0x1320 >> (4 * ((r >> 2) & 3))) & 3; }

// chunks per pair and the first row of chunk c (both sides move together: the path is near the diagonal)
__host__ __device__ inline int chunks_per_pair(bool halo) { return halo ? (DOC - ROWS) / STRIDE_HALO + 1 : DOC / ROWS; }
__host__ __device__ inline int chunk_row0(bool halo, int c) { return halo ? c * STRIDE_HALO : c * ROWS; }

__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {  // XCD-contiguous work map, as in the kernel
    const unsigned q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// src: [pairs][2 sides][LAYERS][DOC][ROWB] bytes; out: [pairs][chunks][OUT_BYTES]
template <bool XREG, bool COMPUTE, bool EPI, bool HALO>
__global__ __launch_bounds__(THREADS, 4) void k_steps(const char* __restrict__ src, long pairs, char* __restrict__ out, int* sink) {
    __shared__ __attribute__((aligned(1024))) char ring[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lkg = lane >> 4;
    const int cpp = chunks_per_pair(HALO);
    const long total = pairs * cpp;
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i][0] = acc[i][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int guard = 0;
    for (long it = blockIdx.x; it < total; it += gridDim.x) {
        const long item = xcd_remap((unsigned)it, (unsigned)total);   // (total < 2^32: checked on the host)
        const long pair = item / cpp;
        const int c = (int)(item % cpp), row0 = chunk_row0(HALO, c);
        const char* base = src + pair * (2l * LAYERS * DOC * ROWB);
        // this wave's DMA piece of a target slab: rows 16 wave .. 16 wave + 15 of the 128-row stage
        const char* ysrc;
        {
            const int r = 16 * wave + (lane >> 2), yl = r / ROWS, loc = r % ROWS;
            ysrc = base + ((long)(LAYERS + yl) * DOC + row0 + loc) * ROWB + ((lane & 3) ^ swz(lane >> 2)) * 16;
        }
        typedef const __attribute__((address_space(1))) char* gcp_t;
        gcp_t xsrc;
        {
            const int xl = wave >> 1, xt = wave & 1;
            xsrc = (gcp_t)(gptr_t)(base + ((long)xl * DOC + row0 + 16 * xt + lrow) * ROWB + lkg * 16);
        }
        u32x4_t afr[S];
        auto issue_group = [&](int grp) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int k = grp * G + u;
                __builtin_amdgcn_global_load_lds((gptr_t)(ysrc + (long)k * SLAB), (lptr_t)(ring + (k % S) * STAGE + wave * 1024), 16, 0, 0);
            }
            if (XREG) {
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int k = grp * G + u;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(afr[k % S]) : "v"(xsrc + (long)k * SLAB) : "memory");
                }
            }
        };
        constexpr int GS = S / G, NG = NK / G, OPS = (XREG ? 2 : 1) * G;
#pragma unroll
        for (int grp = 0; grp < GS - 1; grp++) issue_group(grp);
#pragma unroll
        for (int grp = 0; grp < NG; grp++) {
            const int younger = (NG - 1 - grp) < (GS - 2) ? (NG - 1 - grp) : (GS - 2);
            if (younger >= 2) wait_vm<2 * OPS>();
            else if (younger == 1) wait_vm<OPS>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            u32x4_t fa[G];
#pragma unroll
            for (int u = 0; u < G; u++) fa[u] = XREG ? afr[(grp * G + u) % S] : (u32x4_t){1u, 2u, 3u, 4u};
            if (grp + GS - 1 < NG) issue_group(grp + GS - 1);
            if (COMPUTE) {
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const unsigned sb = (unsigned)(size_t)(lptr_t)(ring + ((grp * G + u) % S) * STAGE) + lrow * SLAB + 16 * (lkg ^ swz(lrow));
                    u32x4_t b[8];
                    asm volatile(
                        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                        "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
                        : "v"(sb)
                        : "memory");
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[u]), __builtin_bit_cast(bf16x8_t, b[2 * i]), acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[u]), __builtin_bit_cast(bf16x8_t, b[2 * i + 1]), acc[i][1], 0, 0, 0);
                    }
                }
            } else {
                asm volatile("" ::"v"(fa[0]), "v"(fa[G - 1]));   // the fragments are "used" (their loads are waited for), nothing more
                // (read by inline assembly: for a C++ read the compiler, which cannot tell this stage from the ones the DMA is
                //  still filling, drains the whole DMA queue with s_waitcnt vmcnt(0) first)
                int got;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"((unsigned)(size_t)(lptr_t)(ring + ((grp * G) % S) * STAGE) + tid * 4) : "memory");
                guard += got;
            }
        }
        __syncthreads();   // the ring has drained: its first 20 KB become the output image
        if (EPI) {
            float* Fs = reinterpret_cast<float*>(ring);
            // a lane's accumulator elements that fall into the band go to the image: 5180 of the 16384 here, as in the kernel
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int at = (((wave * 4 + i) * 2 + j) * 4 + r) * 64 + lane;
                        if (at < OUT_BYTES / 4) Fs[at] = acc[i][j][r];
                    }
            __syncthreads();
            float4* dst = reinterpret_cast<float4*>(out + item * (long)OUT_BYTES);
            const float4* sv = reinterpret_cast<const float4*>(Fs);
            for (int i = tid; i < OUT_BYTES / 16; i += THREADS) dst[i] = sv[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][0] = acc[i][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) t += acc[i][0][0] + acc[i][1][3];
    if (guard == 0x12345678 || t == 1.2345f) *sink = guard;
}

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u;
        h ^= h >> 15;
        // two bf16 values in [0.5, 2) with random mantissas and signs
        const uint32_t lo = 0x3f00u | (h & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
        p[i] = lo | (hi << 16);
    }
}

// Every index the kernel forms, at its extremes, against the two allocations.
static bool check_extents(long pairs, size_t src_bytes, size_t out_bytes) {
    for (int halo = 0; halo < 2; halo++) {
        const int cpp = chunks_per_pair(halo);
        const int last_row0 = chunk_row0(halo, cpp - 1);
        if (last_row0 + ROWS > DOC) { printf("extent: chunk rows leave the document (halo=%d)\n", halo); return false; }
        const size_t last = ((size_t)(pairs - 1) * 2 * LAYERS * DOC + (size_t)(2 * LAYERS - 1) * DOC + last_row0 + ROWS - 1) * ROWB + ROWB;
        if (last > src_bytes) { printf("extent: source read past the allocation (halo=%d)\n", halo); return false; }
        if ((size_t)pairs * cpp * OUT_BYTES > out_bytes) { printf("extent: output past the allocation (halo=%d)\n", halo); return false; }
        if ((double)pairs * cpp >= 4294967296.0) { printf("extent: work items do not fit 32 bits\n"); return false; }
    }
    if (OUT_BYTES > S * STAGE || OUT_BYTES % 16 != 0) { printf("extent: output image does not fit the ring\n"); return false; }
    return true;
}

template <bool XREG, bool COMPUTE, bool EPI, bool HALO>
static void run(const char* what, const char* src, long pairs, char* out, int* sink, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_steps<XREG, COMPUTE, EPI, HALO><<<grid, THREADS>>>(src, pairs, out, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); exit(2); }
    const int reps = 3;
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) k_steps<XREG, COMPUTE, EPI, HALO><<<grid, THREADS>>>(src, pairs, out, sink);
    hipEventRecord(b);
    if (hipEventSynchronize(b) != hipSuccess) { printf("%s: run failed\n", what); exit(2); }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double sides = XREG ? 2.0 : 1.0;                                                // (step 0 reads the target side only)
    const double unique = (double)pairs * sides * LAYERS * DOC * ROWB;                    // every input byte once
    const double staged = (double)pairs * chunks_per_pair(HALO) * sides * LAYERS * ROWS * ROWB;   // bytes entering the CUs
    printf("%-58s %7.3f ms  %5.2f TB/s of input bytes, %5.2f TB/s into the CUs  (= %5.2f ms per 1024 pairs)\n", what, ms,
           unique / (ms * 1e-3) / 1e12, staged / (ms * 1e-3) / 1e12, ms * 1024.0 / pairs);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long pairs = argc > 1 ? atol(argv[1]) : 256;   // 64 MiB of inputs per pair
    const size_t src_bytes = (size_t)pairs * 2 * LAYERS * DOC * ROWB;
    const size_t out_bytes = (size_t)pairs * chunks_per_pair(true) * OUT_BYTES;
    if (pairs < 1 || pairs > 2048 || !check_extents(pairs, src_bytes, out_bytes)) return 1;
    char *src, *out; int* sink;
    if (hipMalloc(&src, src_bytes) != hipSuccess || hipMalloc(&out, out_bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    k_fill<<<4096, 256>>>(reinterpret_cast<uint32_t*>(src), src_bytes / 4);   // bf16 values of ordinary size (the clock depends on the data)
    hipMemset(out, 0, out_bytes);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fill failed\n"); return 1; }
    const int grid = 512;   // two persistent workgroups per CU
    printf("band-cost kernel step by step: %ld pairs (%.1f GiB of inputs), %d workgroups of %d threads, ring %d x %d KB in groups of %d\n",
           pairs, src_bytes / 1073741824.0, grid, THREADS, S, STAGE / 1024, G);
    run<false, false, false, false>("0 LDS-DMA of the target rows only (half the bytes)", src, pairs, out, sink, grid);
    run<true, false, false, false>("1 + source rows into registers", src, pairs, out, sink, grid);
    run<true, true, false, false>("2 + 8 ds_read_b128 + 8 MFMA per wave and slab", src, pairs, out, sink, grid);
    run<true, true, true, false>("3 + epilogue (20 KB image per chunk, 16-byte stores)", src, pairs, out, sink, grid);
    run<true, true, true, true>("4 + halo (chunks advance 18 of their 32 rows)", src, pairs, out, sink, grid);
    run<true, false, false, true>("1h: step 1 with the halo pattern (no compute, no epilogue)", src, pairs, out, sink, grid);
    hipFree(src); hipFree(out); hipFree(sink);
    return 0;
}
