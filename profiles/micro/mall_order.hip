// Microbenchmark: would the level-1 pyramid pass read its level-0 rows out of the Infinity Cache if it ran, side by side,
// right behind the level-0 pass instead of after the whole batch?  (the level-1 pass re-reads the 67 MB per
// pair the level-0 pass has read, because the level-1 column mean is only known at the end of the level-0 pass.)
// Two passes with the pyramid's access pattern (one wave per 2 KB row, 16 rows per wave, 4 rows in flight):
//   A(side): read every row of the side                           (the level-0 pass)
//   B(side): read every row again, write 0.75 bytes per byte read (the level-1 pass)
// in two orders:
//   batch        A over all sides, then B over all sides (two launches)        -- what the product does
//   interleaved  ONE launch, workgroups in the order A(0) .. A(L-1), then A(s+L), B(s) alternating: B(s) starts when about
//                L sides' worth of other traffic has passed since A(s)
// No dependencies are enforced (the numbers are junk); this measures bandwidth only.
// Build / run: hipcc -O3 --offload-arch=gfx950 profiles/micro/mall_order.hip -o /tmp/mall_order && /tmp/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4_t* gv_t;

constexpr int ROWB = 2048, RPW = 16, WAVES = 4, DEPTH = 4;
constexpr int ROWS_PER_WG = RPW * WAVES;

struct Unit { int side; int write; };

template <bool NT>
__device__ __forceinline__ void st16(float* o, f32x4_t v) {
    if (NT) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) f32x4_t*)o);
    else *(f32x4_t*)o = v;
}

template <bool NT>
__device__ __forceinline__ void do_rows(const char* src, long r0, float* dst, bool write, float* sums, int lane) {
    const char* base = src + r0 * ROWB + lane * 16;
    u32x4_t ra[DEPTH], rb[DEPTH];
    float acc = 0.f, ps[8];
#pragma unroll
    for (int r = 0; r < DEPTH; r++) { ra[r] = *(gv_t)(base + (long)r * ROWB); rb[r] = *(gv_t)(base + (long)r * ROWB + 1024); }
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const u32x4_t a = ra[r % DEPTH], b = rb[r % DEPTH];
        const float v[8] = {__uint_as_float(a.x << 16), __uint_as_float(a.y << 16), __uint_as_float(a.z << 16), __uint_as_float(a.w << 16),
                            __uint_as_float(b.x << 16), __uint_as_float(b.y << 16), __uint_as_float(b.z << 16), __uint_as_float(b.w << 16)};
        if (r + DEPTH < RPW) { ra[r % DEPTH] = *(gv_t)(base + (long)(r + DEPTH) * ROWB); rb[r % DEPTH] = *(gv_t)(base + (long)(r + DEPTH) * ROWB + 1024); }
#pragma unroll
        for (int i = 0; i < 8; i++) { acc += v[i]; ps[i] = (r & 1) ? ps[i] + v[i] : v[i]; }
        if (write && (r & 1)) {   // 3 KB per row pair (4 KB read)
            float* o = dst + ((r0 + r) / 2) * 768 + lane * 4;
            st16<NT>(o, (f32x4_t){ps[0], ps[1], ps[2], ps[3]});
            st16<NT>(o + 256, (f32x4_t){ps[4], ps[5], ps[6], ps[7]});
            st16<NT>(o + 512, (f32x4_t){ps[1], ps[0], ps[3], ps[2]});
        }
    }
    if (acc == 123.456f) sums[0] = acc;
}

// one launch over a list of units; wgs_per_side workgroups each
template <bool NT>
__global__ __launch_bounds__(64 * WAVES) void k_units(const char* __restrict__ src, float* __restrict__ dst, const Unit* __restrict__ units,
                                                     int wgs_per_side, long rows_per_side, float* sums) {
    const int u = blockIdx.x / wgs_per_side, b = blockIdx.x % wgs_per_side;
    const Unit un = units[u];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r0 = (long)un.side * rows_per_side + ((long)b * WAVES + w) * RPW;
    do_rows<NT>(src, r0, dst, un.write != 0, sums, lane);
}

static bool g_nt = false;
static float run(const char* src, float* dst, const std::vector<Unit>& order, int wgs_per_side, long rows_per_side, float* sums, Unit* dunits, int reps) {
    hipMemcpy(dunits, order.data(), order.size() * sizeof(Unit), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int i = 0; i < reps; i++) {
        hipEventRecord(e0);
        if (g_nt) k_units<true><<<(int)order.size() * wgs_per_side, 64 * WAVES>>>(src, dst, dunits, wgs_per_side, rows_per_side, sums);
        else k_units<false><<<(int)order.size() * wgs_per_side, 64 * WAVES>>>(src, dst, dunits, wgs_per_side, rows_per_side, sums);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int sides = argc > 1 ? atoi(argv[1]) : 128;
    const int side_mb = argc > 2 ? atoi(argv[2]) : 32;
    g_nt = argc > 3 && atoi(argv[3]) != 0;
    const long rows_per_side = (long)side_mb * 1024 * 1024 / ROWB;
    const int wgs_per_side = (int)(rows_per_side / ROWS_PER_WG);
    const size_t in_bytes = (size_t)sides * rows_per_side * ROWB, out_bytes = in_bytes / 4 * 3;
    char* src; float *dst, *sums; Unit* dunits;
    hipMalloc(&src, in_bytes); hipMalloc(&dst, out_bytes); hipMalloc(&sums, 64); hipMalloc(&dunits, sizeof(Unit) * (2 * sides + 8));
    hipMemset(src, 0x3c, in_bytes); hipMemset(dst, 0, out_bytes);
    printf("pass order vs the Infinity Cache: %d sides of %d MB (%ld rows of 2 KB, %d workgroups per side and pass); A reads a side, B reads it again and writes 0.75x with %s stores\n",
           sides, side_mb, rows_per_side, wgs_per_side, g_nt ? "non-temporal" : "plain");
    const double gb_a = in_bytes / 1e9, gb_b = (in_bytes + out_bytes) / 1e9;
    std::vector<Unit> oa, ob;
    for (int s = 0; s < sides; s++) { oa.push_back({s, 0}); ob.push_back({s, 1}); }
    const float ta = run(src, dst, oa, wgs_per_side, rows_per_side, sums, dunits, 3);
    const float tb = run(src, dst, ob, wgs_per_side, rows_per_side, sums, dunits, 3);
    printf("batch order        A all sides %7.3f ms (%.2f TB/s)   B all sides %7.3f ms (%.2f TB/s)   A + B %7.3f ms\n", ta, gb_a / ta, tb, gb_b / tb, ta + tb);
    for (int L : {1, 2, 3, 4, 6, 8}) {
        std::vector<Unit> o;
        for (int s = 0; s < L && s < sides; s++) o.push_back({s, 0});
        for (int s = 0; s < sides; s++) {
            if (s + L < sides) o.push_back({s + L, 0});
            o.push_back({s, 1});
        }
        const float t = run(src, dst, o, wgs_per_side, rows_per_side, sums, dunits, 3);
        printf("interleaved, lag %d  one launch  %7.3f ms   = %.3f of batch order  (bytes moved / time %.2f TB/s)\n", L, t, t / (ta + tb), (gb_a + gb_b) / t);
    }
    return 0;
}
