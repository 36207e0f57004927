// Microbenchmark: the access pattern of the pyramid passes (speech-vecalign_amd/csrc/svx_rows.hip, k_pyramid) -- one
// wave per 2 KB row, 16 consecutive rows per wave, 4 waves per workgroup, a few rows in flight per wave -- with next to
// no arithmetic, to see what the PATTERN can stream and whether the path of the bytes matters:
//   A  global_load_dwordx4 into registers, DEPTH rows in flight per wave            (what k_pyramid does)
//   B  LDS-DMA (global_load_lds_dwordx4) into a per-wave LDS ring of DEPTH rows, then ds_read_b128, counted s_waitcnt
//   C  A with non-temporal loads
// and the same three with a streaming store of every second row pair (the level-1 pass writes 0.75 bytes per byte read).
// Build / run: hipcc -O3 --offload-arch=gfx950 profiles/micro/row_stream.hip -o /tmp/row_stream && /tmp/row_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int ROWB = 2048, RPW = 16, WAVES = 4;   // bytes per row, rows per wave, waves per workgroup

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: registers, 1: LDS-DMA, 2: registers, non-temporal.  WRITE: fp32 "pair sums" of every two rows go out (4 KB per 4 KB read)
template <int MODE, int DEPTH, bool WRITE, int PAD_KB = 0, int PAT = 0>
__global__ __launch_bounds__(64 * WAVES) void k_rows(const char* __restrict__ src, long rows, float* __restrict__ dst, float* __restrict__ sums) {
    __shared__ __attribute__((aligned(1024))) char lds[MODE == 1 ? WAVES * DEPTH * ROWB : 16];
    __shared__ char pad[PAD_KB > 0 ? PAD_KB * 1024 : 16];   // PAD_KB: limits the workgroups per CU (160 KB of LDS) like the pyramid kernels' registers do
    if (PAD_KB > 0 && threadIdx.x == 1023) pad[src[0] & 15] = 1;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long r0 = ((long)blockIdx.x * WAVES + w) * RPW;
    if (r0 >= rows) return;
    const char* base = src + r0 * ROWB + lane * 16;
    float acc = 0.f;
    float ps[8];
    if (MODE == 1) {
        char* mine = lds + w * DEPTH * ROWB;
        auto issue = [&](int r) {
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (long)r * ROWB), (lptr_t)(mine + (r % DEPTH) * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (long)r * ROWB + 1024), (lptr_t)(mine + (r % DEPTH) * ROWB + 1024), 16, 0, 0);
        };
#pragma unroll
        for (int r = 0; r < DEPTH; r++) issue(r);
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            constexpr int PER = 2 + 0;   // vector-memory operations per row ahead of this one (the stores are counted below)
            const int ahead = (RPW - 1 - r) < (DEPTH - 1) ? (RPW - 1 - r) : (DEPTH - 1);
            // stores issued since this row's DMA: iterations j in [max(r - DEPTH, 0), r - 1] with j odd issued four each
            // (an iteration issues its DMA first, then its stores)
            const int lo = r - DEPTH > 0 ? r - DEPTH : 0;
            const int st = WRITE ? 4 * (r / 2 - lo / 2) : 0;
            const int n = ahead * PER + st;
            if (n >= 16) wait_vm<16>(); else if (n == 15) wait_vm<15>(); else if (n == 14) wait_vm<14>(); else if (n == 13) wait_vm<13>(); else if (n == 12) wait_vm<12>(); else if (n == 11) wait_vm<11>(); else if (n == 10) wait_vm<10>(); else if (n == 9) wait_vm<9>(); else if (n == 8) wait_vm<8>(); else if (n == 7) wait_vm<7>(); else if (n == 6) wait_vm<6>(); else if (n == 5) wait_vm<5>(); else if (n == 4) wait_vm<4>(); else if (n == 3) wait_vm<3>(); else if (n == 2) wait_vm<2>(); else if (n == 1) wait_vm<1>(); else wait_vm<0>();
            u32x4_t a, b;
            const unsigned at = (unsigned)(size_t)(lptr_t)(mine + (r % DEPTH) * ROWB) + lane * 16;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(at) : "memory");
            if (r + DEPTH < RPW) issue(r + DEPTH);
            const float v[8] = {__uint_as_float(a.x << 16), __uint_as_float(a.y << 16), __uint_as_float(a.z << 16), __uint_as_float(a.w << 16),
                                __uint_as_float(b.x << 16), __uint_as_float(b.y << 16), __uint_as_float(b.z << 16), __uint_as_float(b.w << 16)};
#pragma unroll
            for (int i = 0; i < 8; i++) { acc += v[i]; ps[i] = (r & 1) ? ps[i] + v[i] : v[i]; }
            if (WRITE && (r & 1)) {
                // PAT 0: an instruction's 64 lanes write 1 KB contiguous; PAT 1: a lane owns 32 contiguous bytes and writes them
                // with two instructions (each instruction then covers 16-byte pieces at a 32-byte stride: k_pyramid's stores)
                float* o = dst + ((r0 + r) / 2) * 1024 + (PAT == 1 ? lane * 8 : lane * 4);
                __builtin_nontemporal_store((f32x4_t){ps[0], ps[1], ps[2], ps[3]}, (__attribute__((address_space(1))) f32x4_t*)o);
                __builtin_nontemporal_store((f32x4_t){ps[4], ps[5], ps[6], ps[7]}, (__attribute__((address_space(1))) f32x4_t*)(o + (PAT == 1 ? 4 : 256)));
                __builtin_nontemporal_store((f32x4_t){ps[1], ps[0], ps[3], ps[2]}, (__attribute__((address_space(1))) f32x4_t*)(o + 512));
                __builtin_nontemporal_store((f32x4_t){ps[5], ps[4], ps[7], ps[6]}, (__attribute__((address_space(1))) f32x4_t*)(o + (PAT == 1 ? 516 : 768)));
            }
        }
    } else {
        typedef const __attribute__((address_space(1))) u32x4_t* gv_t;
        u32x4_t ra[DEPTH], rb[DEPTH];
        auto load = [&](int r, u32x4_t& a, u32x4_t& b) {
            if (MODE == 2) {
                a = __builtin_nontemporal_load((gv_t)(base + (long)r * ROWB));
                b = __builtin_nontemporal_load((gv_t)(base + (long)r * ROWB + 1024));
            } else {
                a = *(gv_t)(base + (long)r * ROWB);
                b = *(gv_t)(base + (long)r * ROWB + 1024);
            }
        };
#pragma unroll
        for (int r = 0; r < DEPTH; r++) load(r, ra[r], rb[r]);
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const u32x4_t a = ra[r % DEPTH], b = rb[r % DEPTH];
            const float v[8] = {__uint_as_float(a.x << 16), __uint_as_float(a.y << 16), __uint_as_float(a.z << 16), __uint_as_float(a.w << 16),
                                __uint_as_float(b.x << 16), __uint_as_float(b.y << 16), __uint_as_float(b.z << 16), __uint_as_float(b.w << 16)};
            if (r + DEPTH < RPW) load(r + DEPTH, ra[r % DEPTH], rb[r % DEPTH]);
#pragma unroll
            for (int i = 0; i < 8; i++) { acc += v[i]; ps[i] = (r & 1) ? ps[i] + v[i] : v[i]; }
            if (WRITE && (r & 1)) {
                // PAT 0: an instruction's 64 lanes write 1 KB contiguous; PAT 1: a lane owns 32 contiguous bytes and writes them
                // with two instructions (each instruction then covers 16-byte pieces at a 32-byte stride: k_pyramid's stores)
                float* o = dst + ((r0 + r) / 2) * 1024 + (PAT == 1 ? lane * 8 : lane * 4);
                __builtin_nontemporal_store((f32x4_t){ps[0], ps[1], ps[2], ps[3]}, (__attribute__((address_space(1))) f32x4_t*)o);
                __builtin_nontemporal_store((f32x4_t){ps[4], ps[5], ps[6], ps[7]}, (__attribute__((address_space(1))) f32x4_t*)(o + (PAT == 1 ? 4 : 256)));
                __builtin_nontemporal_store((f32x4_t){ps[1], ps[0], ps[3], ps[2]}, (__attribute__((address_space(1))) f32x4_t*)(o + 512));
                __builtin_nontemporal_store((f32x4_t){ps[5], ps[4], ps[7], ps[6]}, (__attribute__((address_space(1))) f32x4_t*)(o + (PAT == 1 ? 516 : 768)));
            }
        }
    }
    if (acc == 1.2345f) sums[0] = acc;
}

template <int MODE, int DEPTH, bool WRITE, int PAD_KB = 0, int PAT = 0>
static void run(const char* what, const char* src, long rows, float* dst, float* sums) {
    const int grid = (int)((rows + WAVES * RPW - 1) / (WAVES * RPW));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_rows<MODE, DEPTH, WRITE, PAD_KB, PAT><<<grid, 64 * WAVES>>>(src, rows, dst, sums);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: failed\n", what); exit(2); }
    hipEventRecord(a);
    for (int i = 0; i < 3; i++) k_rows<MODE, DEPTH, WRITE, PAD_KB, PAT><<<grid, 64 * WAVES>>>(src, rows, dst, sums);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double rd = (double)rows * ROWB, wr = WRITE ? (double)rows * ROWB : 0.0;   // fp32 pair sums: 4 KB per two 2 KB rows
    printf("%-62s %7.3f ms  read %5.2f TB/s  read+write %5.2f TB/s\n", what, ms, rd / (ms * 1e-3) / 1e12, (rd + wr) / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const long rows = 8l << 20;   // 16 GiB of 2 KB rows
    char* src; float *dst, *sums;
    if (hipMalloc(&src, rows * ROWB) != hipSuccess || hipMalloc(&dst, rows * ROWB) != hipSuccess || hipMalloc(&sums, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0x3f, rows * ROWB);
    hipDeviceSynchronize();
    printf("row stream: %ld rows of %d bytes, one wave per row, %d rows per wave, %d waves per workgroup\n", rows, ROWB, RPW, WAVES);
    run<0, 4, false>("A  registers, 4 rows in flight", src, rows, dst, sums);
    run<0, 6, false>("A  registers, 6 rows in flight", src, rows, dst, sums);
    run<2, 4, false>("C  registers, non-temporal, 4 rows in flight", src, rows, dst, sums);
    run<1, 4, false>("B  LDS-DMA ring per wave, 4 rows in flight", src, rows, dst, sums);
    run<1, 3, false>("B  LDS-DMA ring per wave, 3 rows in flight", src, rows, dst, sums);
    run<0, 4, false, 36>("A  registers, 4 rows in flight, 4 workgroups (16 waves) per CU", src, rows, dst, sums);
    run<0, 6, false, 36>("A  registers, 6 rows in flight, 4 workgroups per CU", src, rows, dst, sums);
    run<0, 8, false, 36>("A  registers, 8 rows in flight, 4 workgroups per CU", src, rows, dst, sums);
    run<2, 4, false, 36>("C  non-temporal, 4 rows in flight, 4 workgroups per CU", src, rows, dst, sums);
    run<0, 4, false, 50>("A  registers, 4 rows in flight, 3 workgroups (12 waves) per CU", src, rows, dst, sums);
    run<0, 4, true, 36>("A  4 rows in flight, 4 workgroups per CU, + streaming stores", src, rows, dst, sums);
    run<0, 4, true>("A  registers, 4 rows in flight, + streaming stores", src, rows, dst, sums);
    run<0, 4, true, 36, 1>("A  4 workgroups per CU, + stores of 32 bytes per lane as 2 x 16", src, rows, dst, sums);
    run<2, 4, true>("C  non-temporal loads, + streaming stores", src, rows, dst, sums);
    run<1, 4, true>("B  LDS-DMA, 4 rows in flight, + streaming stores", src, rows, dst, sums);
    return 0;
}
