// svx_band.hip -- band ("sparse") costs of the fused pipeline, second generation.
//
// Reference semantics: make_sparse_costs, svecalign/vecalign/dp_core.pyx:165-267 (cost formula :256-263).
//
// Why a second kernel.  The first one (svx_costs.hip) stages 48 rows x (kx + ky) layers per workgroup through
// VGPRs in 128/256-byte k-slabs with two workgroup barriers per slab; with > 100 KB of LDS only one workgroup
// fits a CU, so nothing overlaps its global loads, its LDS traffic and its epilogue (rocprofv3, round 1: 55 % of
// the wave cycles parked on s_waitcnt / barriers, 36 % LDS bank-conflict cycles, 0.34 of HBM peak).  This one is
// built around keeping bytes in flight:
//   * smaller chunks: up to 2 (ROWS - band) + 1 path points whose cells touch <= ROWS (32) consecutive rows per
//     side, 256 threads, <= 76 KB of LDS -> two workgroups per CU, so one's prologue / epilogue hides under the
//     other's k loop (the extra halo rows are L2 hits: neighbouring chunks run on the same XCD);
//   * rows travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR stop-over, no ds_write) in 64-byte
//     k-slabs = one MFMA k-step, into a ring of S stages: S - 1 slabs are in flight while one is multiplied, one
//     workgroup barrier per slab, and the barrier does not drain the DMA queue (counted s_waitcnt vmcnt);
//   * every DMA lane names its own 16 source bytes, so the LDS image is XOR-swizzled for free: the 16 lanes of
//     each ds_read_b128 service group hit 16 different bank quads (no bank conflicts);
//   * units = (alignment type, 16-row x tile), accumulators stay in registers over the whole k loop; tiles the
//     band does not touch are skipped;
//   * epilogue: every lane tests its accumulator elements for band membership, evaluates the cost in double like
//     the reference and drops it into an LDS image of the [point][type][cell] block, which then leaves as one
//     contiguous run of 16-byte stores.
// fp32 rows (deeper pyramid levels, fp32 inputs) use v_mfma_f32_16x16x4_f32 on the same LDS image: a lane's
// 16-byte piece holds 4 floats, used as element j of the j-th of four MFMAs (exact fp32 products).
#include <stdlib.h>
#include <string.h>

#include "svx_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int B2_THREADS = 256;
constexpr int B2_WAVES = 4;
constexpr int B2_TB = 16;          // band cells per workgroup
constexpr int B2_SLAB = 64;        // bytes of a row per k-slab
constexpr int B2_MAXPASS = 8;
constexpr int B2_MAXSLOT = 20;     // overlap layers (both sides) one pass can stage
constexpr int B2_MAXTPP = 16;      // alignment types per pass

__device__ const uint4 band_zero16[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

// one pass = a run of consecutive alignment types and the overlap layers ("slots") they read
struct BandPass {
    int t0, nt, nslot;
    int slot_info[B2_MAXSLOT];   // side << 8 | layer   (ints: scalar loads from the kernel-argument segment)
    int type_slots[B2_MAXTPP];   // x slot | y slot << 8
};
struct BandPlan {
    int npass;
    BandPass pass[B2_MAXPASS];
};

__device__ __forceinline__ float cost_formula2(float sumx, int p, int q, float n0, float n1) {
#pragma clang fp contract(off)
    // dp_core.pyx:259-260, evaluated in double like the generated C, stored to float
    return (float)((((2.0 * (double)p) * (double)q) * (1.0 - (double)sumx)) / ((1e-6 + (double)n0) + (double)n1));
}

__device__ __forceinline__ unsigned xcd_remap2(unsigned id, unsigned n) {
    const unsigned q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// one k-slab (16 bytes per lane of A and of B) into a 16 x 16 accumulator tile
template <typename E>
__device__ __forceinline__ void mma_slab16(f32x4_t& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mma_slab16<ElemBF16>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slab16<ElemF16>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slab16<ElemF32>(f32x4_t& acc, const uint4& a, const uint4& b) {
    // lane (row r, group g) holds floats 4g .. 4g+3 of the 16-float slab; MFMA j multiplies element j of every
    // lane: the four lanes of a row supply k = j, 4 + j, 8 + j, 12 + j -- A and B use the same map
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Band2Args {
    const void* v0;  // [k0][n][d]
    const void* v1;
    int n, m, d;
    const float* inv0;  // [k][n] or null
    const float* inv1;
    const float* nrm0;  // [k][n]
    const float* nrm1;
    const int* path;    // [A][2]
    int A, W, T;
    float* costs;       // [A][T][2W]
    int* boff;          // [A]
    int* status;
};

// LDS image of a stage: row r (slot * ROWS + loc) holds its 64-byte slab as four 16-byte pieces; piece kg sits at
// 64 r + 16 (kg ^ h[(r >> 2) & 3]), h = {0, 2, 3, 1}: the 16 lanes of every ds_read_b128 service group (rows
// {0-3, 12-15} of one k-group with rows {4-11} of the next) then cover all 16 bank quads.
__device__ __forceinline__ int swz(int row_in_tile) { return (0x1320 >> (4 * ((row_in_tile >> 2) & 3))) & 3; }

template <typename E, int ROWS, int NSLOT, int UPW, int S>
struct Band2 {
    static constexpr int XT = ROWS / 16;                    // 16-row tiles per side
    static constexpr int STAGE = NSLOT * ROWS * B2_SLAB;    // bytes
    static constexpr int NQ = NSLOT * ROWS / 16;            // 1-KiB DMA pieces per slab
    static constexpr int PW = NQ / B2_WAVES;                // ... per wave
    static constexpr int TPP = B2_WAVES * UPW / XT;         // alignment types per pass
    static constexpr int TAMAX = 2 * (ROWS - 1) + 1;        // path points per chunk
    static constexpr int FS_FLOATS = (2 * (ROWS - B2_TB) + 1) * B2_TB * TPP;
    static constexpr int KEL = B2_SLAB / (int)sizeof(typename E::storage);  // elements per slab
    static_assert(NQ % B2_WAVES == 0, "DMA pieces must divide over the waves");
    static_assert(TPP <= B2_MAXTPP && NSLOT <= B2_MAXSLOT, "pass limits");
    static_assert(S >= 2 && S <= 4, "ring depth");
    static_assert(PW * (S - 2) <= 63, "vmcnt is a 6-bit counter");
};

template <typename E, int ROWS, int NSLOT, int UPW, int S>
__device__ void band2_block(const Band2Args& g, const SvxTypes& ty, const BandPlan& plan, int a0, int TAe, int chunk_b,
                            char* st0, char* st1, char* st2, char* st3, float* Fs, int* spx, int* spy, float* snrm, float* sinv,
                            int* ltx, int* lty, int* tmask, int* lslot) {
    using C = Band2<E, ROWS, NSLOT, UPW, S>;
    using St = typename E::storage;
    constexpr int XT = C::XT, PW = C::PW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // an SGPR: everything indexed by it stays scalar
    const int B = 2 * g.W, b0 = chunk_b * B2_TB;
    const int TBe = (B - b0) < B2_TB ? (B - b0) : B2_TB;
    const int esz = (int)sizeof(St);
    const int rowbytes = g.d * esz;
    const int NK = (rowbytes + B2_SLAB - 1) / B2_SLAB;
    char* stages[4] = {st0, st1, st2, st3};

    // ---- the chunk's path points
    if (tid < TAe) {
        const int2 p = reinterpret_cast<const int2*>(g.path)[a0 + tid];
        spx[tid] = p.x;
        spy[tid] = p.y;
    }
    for (int t = tid; t < ty.n; t += B2_THREADS) {
        ltx[t] = ty.x[t];
        lty[t] = ty.y[t];
    }
    if (tid < XT) tmask[tid] = 0;
    __syncthreads();
    if (tid < TAe) {
        bool ok = (spx[tid] + spy[tid] == a0 + tid);
        if (tid > 0) ok = ok && spx[tid] >= spx[tid - 1] && spy[tid] >= spy[tid - 1];
        if (!ok && g.status) *g.status = SVX_ERR_PATH;
        if (chunk_b == 0) g.boff[a0 + tid] = spy[tid] - g.W;
    }
    const int xlo = spx[0], ylo = spy[0];
    const int X0 = xlo + g.W - (b0 + TBe - 1);
    const int Y0 = ylo - g.W + b0;
    int NXn = spx[TAe - 1] - xlo + TBe, NYn = spy[TAe - 1] - ylo + TBe;
    // (a path that is not unit-step could ask for more rows than are staged: the status flag is already set, stay in bounds)
    NXn = NXn < 1 ? 1 : (NXn > ROWS ? ROWS : NXn);
    NYn = NYn < 1 ? 1 : (NYn > ROWS ? ROWS : NYn);
    // tiles that hold band cells
    if (tid < TAe) {
        const int ya = spy[tid];
        int mk[XT];
#pragma unroll
        for (int i = 0; i < XT; i++) mk[i] = 0;
        for (int bi = 0; bi < TBe; bi++) {
            const int yloc = ya - g.W + b0 + bi - Y0, xloc = (a0 + tid) - (ya - g.W + b0 + bi) - X0;
            if (xloc >= 0 && xloc < ROWS && yloc >= 0 && yloc < ROWS) {
#pragma unroll
                for (int i = 0; i < XT; i++)
                    if ((xloc >> 4) == i) mk[i] |= 1 << (yloc >> 4);
            }
        }
#pragma unroll
        for (int i = 0; i < XT; i++)
            if (mk[i]) atomicOr(&tmask[i], mk[i]);
    }

    const int lrow = lane & 15, lkg = lane >> 4;
    const int loff = lrow * B2_SLAB + 16 * (lkg ^ swz(lrow));   // this lane's fragment inside a 16-row tile

    for (int pi = 0; pi < plan.npass; pi++) {
        const BandPass& ps = plan.pass[pi];
        const int ntp = ps.nt, nslot = ps.nslot;
        // the pass's slot table -> LDS (one round trip to the kernel-argument segment instead of one per use)
        if (tid < NSLOT) lslot[tid] = tid < nslot ? ps.slot_info[tid] : 0;
        __syncthreads();  // slot table and tile masks complete / the previous pass is done with the stages and Fs
        // ---- per-row scalars of this pass (needed in the epilogue only: the loads ride out the k loop in registers)
        constexpr int SPT = (NSLOT * ROWS + B2_THREADS - 1) / B2_THREADS;
        float r_nrm[SPT], r_inv[SPT];
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int r = tid + i * B2_THREADS;
            r_nrm[i] = 0.f;
            r_inv[i] = 1.f;
            if (r < nslot * ROWS) {
                const int slot = r / ROWS, loc = r % ROWS;
                const int side = lslot[slot] >> 8, layer = lslot[slot] & 255;
                const int gi = (side ? Y0 : X0) + loc, nn = side ? g.m : g.n;
                if (loc < (side ? NYn : NXn) && gi >= 0 && gi < nn) {
                    const size_t o = (size_t)layer * nn + gi;
                    r_nrm[i] = (side ? g.nrm1 : g.nrm0)[o];
                    const float* iv = side ? g.inv1 : g.inv0;
                    if (iv) r_inv[i] = iv[o];
                }
            }
        }
        // ---- this lane's DMA sources: piece q = wave + 4 i covers rows 16 q .. 16 q + 15, lane -> (row, 16-byte piece)
        const char* src[PW];
        unsigned live = 0;
#pragma unroll
        for (int i = 0; i < PW; i++) {
            const int q = wave + B2_WAVES * i;
            const int r = 16 * q + (lane >> 2);
            const int slot = r / ROWS, loc = r % ROWS;
            const int piece = (lane & 3) ^ swz(lane >> 2);
            src[i] = reinterpret_cast<const char*>(band_zero16);
            if (slot < nslot) {
                const int side = lslot[slot] >> 8, layer = lslot[slot] & 255;
                const int gi = (side ? Y0 : X0) + loc, nn = side ? g.m : g.n;
                if (loc < (side ? NYn : NXn) && gi >= 0 && gi < nn) {
                    src[i] = reinterpret_cast<const char*>(side ? g.v1 : g.v0) + ((size_t)layer * nn + gi) * rowbytes + piece * 16;
                    live |= 1u << i;
                }
            }
        }
        const int piece_byte = ((lane & 3) ^ swz(lane >> 2)) * 16;
        auto issue = [&](int k, char* stage) {
#pragma unroll
            for (int i = 0; i < PW; i++) {
                const char* s = src[i] + (size_t)k * B2_SLAB;
                if (!((live >> i) & 1u) || k * B2_SLAB + piece_byte >= rowbytes) s = reinterpret_cast<const char*>(band_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(stage + (wave + B2_WAVES * i) * 1024), 16, 0, 0);
            }
        };
        // ---- units of this wave: (type of the pass, x tile); every unit multiplies its x tile with all XT y tiles
        // (with two tiles per side the band touches all four; the unused corner of a larger staging area is masked)
        f32x4_t acc[UPW][XT];
        int aoff[UPW], boffb[UPW], umask[UPW];
        const int nunits = ntp * XT;
#pragma unroll
        for (int s = 0; s < UPW; s++) {
#pragma unroll
            for (int j = 0; j < XT; j++) acc[s][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int u = wave + B2_WAVES * s;
            const int tl = u < nunits ? u / XT : 0, xt = u % XT;
            const int tsl = ps.type_slots[tl];
            aoff[s] = ((tsl & 255) * ROWS + xt * 16) * B2_SLAB + loff;
            boffb[s] = ((tsl >> 8) * ROWS) * B2_SLAB + loff;
            umask[s] = u < nunits ? (XT <= 2 ? (1 << XT) - 1 : __builtin_amdgcn_readfirstlane(tmask[xt])) : 0;
        }
        auto mma = [&](const char* stage) {
            uint4 fa[UPW], fb[UPW][XT];
#pragma unroll
            for (int s = 0; s < UPW; s++) {
                if (umask[s]) {  // scalar
                    fa[s] = *reinterpret_cast<const uint4*>(stage + aoff[s]);
#pragma unroll
                    for (int j = 0; j < XT; j++)
                        if (XT <= 2 || ((umask[s] >> j) & 1)) fb[s][j] = *reinterpret_cast<const uint4*>(stage + boffb[s] + j * 16 * B2_SLAB);
                }
            }
#pragma unroll
            for (int s = 0; s < UPW; s++) {
                if (umask[s]) {
#pragma unroll
                    for (int j = 0; j < XT; j++)
                        if (XT <= 2 || ((umask[s] >> j) & 1)) mma_slab16<E>(acc[s][j], fa[s], fb[s][j]);
                }
            }
        };
        // ---- k loop: slab k is multiplied out of stage k % S while slabs k+1 .. k+S-2 are in flight
#pragma unroll
        for (int k = 0; k < S - 1; k++)
            if (k < NK) issue(k, stages[k]);
        auto step = [&](int k, const char* rd, char* wr) {
            if (k < NK) {
                const int younger = (NK - 1 - k) < (S - 2) ? (NK - 1 - k) : (S - 2);  // slabs issued after slab k
                if (younger >= 2) wait_vm<2 * PW>();
                else if (younger == 1) wait_vm<PW>();
                else wait_vm<0>();
                __builtin_amdgcn_s_barrier();   // every wave's pieces of slab k have landed; stage (k-1) % S is free
                asm volatile("" ::: "memory");
                if (k + S - 1 < NK) issue(k + S - 1, wr);
                mma(rd);
            }
        };
        for (int k0 = 0; k0 < NK; k0 += S) {
            if (S == 2) {
                step(k0, st0, st1);
                step(k0 + 1, st1, st0);
            } else if (S == 3) {
                step(k0, st0, st2);
                step(k0 + 1, st1, st0);
                step(k0 + 2, st2, st1);
            } else {
                step(k0, st0, st3);
                step(k0 + 1, st1, st0);
                step(k0 + 2, st2, st1);
                step(k0 + 3, st3, st2);
            }
        }
        // ---- epilogue
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int r = tid + i * B2_THREADS;
            if (r < NSLOT * ROWS) {
                snrm[r] = r_nrm[i];
                sinv[r] = r_inv[i];
            }
        }
        __syncthreads();
        const float inf = __builtin_inff();
#pragma unroll
        for (int s = 0; s < UPW; s++) {
            const int u = wave + B2_WAVES * s;
            if (u >= nunits) continue;
            const int tl = u / XT, xt = u % XT;
            const int p = ltx[ps.t0 + tl], q = lty[ps.t0 + tl];
            const int xs = (ps.type_slots[tl] & 255) * ROWS, ys = (ps.type_slots[tl] >> 8) * ROWS;
#pragma unroll
            for (int j = 0; j < XT; j++) {
                if (!((umask[s] >> j) & 1)) continue;
                const int yloc = 16 * j + lrow, yy = Y0 + yloc;
                const float ny = snrm[ys + yloc], iy = sinv[ys + yloc];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int xloc = 16 * xt + 4 * lkg + r, xx = X0 + xloc;
                    const int ai = xx + yy - a0;
                    if (ai < 0 || ai >= TAe) continue;
                    const int bi = yy - (spy[ai] - g.W) - b0;
                    if (bi < 0 || bi >= TBe) continue;
                    float c = inf;
                    if (xx >= 0 && xx < g.n && yy >= 0 && yy < g.m) {
                        const float sumx = acc[s][j][r] * sinv[xs + xloc] * iy;
                        c = cost_formula2(sumx, p, q, snrm[xs + xloc], ny);
                    }
                    Fs[(ai * ntp + tl) * TBe + bi] = c;
                }
            }
        }
        __syncthreads();
        // ---- write-out: [a][type][b].  One band chunk and all types in this pass: the block is one contiguous run.
        if (TBe == B && ntp == g.T && ((g.T * B) & 3) == 0) {
            const int nv = TAe * ntp * TBe / 4;
            float4* dst = reinterpret_cast<float4*>(g.costs + (size_t)a0 * g.T * B);
            const float4* srcv = reinterpret_cast<const float4*>(Fs);
            for (int i = tid; i < nv; i += B2_THREADS) dst[i] = srcv[i];
        } else {
            const int nout = TAe * ntp * TBe;
            for (int idx = tid; idx < nout; idx += B2_THREADS) {
                const int ai = idx / (ntp * TBe);
                const int rem = idx - ai * (ntp * TBe);
                const int tl = rem / TBe, bi = rem - tl * TBe;
                g.costs[((size_t)(a0 + ai) * g.T + ps.t0 + tl) * B + (b0 + bi)] = Fs[idx];
            }
        }
    }
}

// depth 0 uses the final types on the raw rows; deeper levels use (1,1) on the normalised fp32 layer 0
template <typename E, bool LV0, int ROWS, int NSLOT, int UPW, int S>
__global__ __launch_bounds__(B2_THREADS) void k_band_costs2(const SvxPairDev* __restrict__ pairs, int depth, SvxTypes ty, BandPlan plan,
                                                            int W, int nchunk_b, int per_pair) {
    using C = Band2<E, ROWS, NSLOT, UPW, S>;
    // the stages are separate LDS objects: the compiler then knows that the ds_reads of one stage never touch a
    // stage an LDS-DMA is still filling, and does not drain the DMA queue in front of them
    __shared__ __attribute__((aligned(1024))) char st0[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st1[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st2[S >= 3 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(1024))) char st3[S >= 4 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(16))) float Fs[C::FS_FLOATS];
    __shared__ int spx[C::TAMAX + 1], spy[C::TAMAX + 1];
    __shared__ float snrm[NSLOT * ROWS], sinv[NSLOT * ROWS];
    __shared__ int ltx[SVX_MAX_TYPES + 2], lty[SVX_MAX_TYPES + 2];
    __shared__ int tmask[C::XT];
    __shared__ int lslot[NSLOT];
    const unsigned wg = xcd_remap2(blockIdx.x, gridDim.x);
    const SvxPairDev& P = pairs[wg / per_pair];
    const int item = wg % per_pair;
    if (depth > P.L || (depth == P.L && P.L > 0)) return;  // refined levels only (or level 0 when L == 0)
    const SvxLevel& Lv = P.lev[depth];
    Band2Args g;
    g.A = *Lv.path_len;
    const int chunk_a = item / nchunk_b;
    if (g.A <= 0 || chunk_a >= *Lv.nchunks) return;
    const int a0 = Lv.cstart[chunk_a], TAe = Lv.cstart[chunk_a + 1] - a0;
    if (TAe <= 0 || TAe > C::TAMAX) return;
    g.n = Lv.n[0];
    g.m = Lv.n[1];
    g.d = P.d;
    g.v0 = LV0 ? P.v[0] : (const void*)Lv.P[0];
    g.v1 = LV0 ? P.v[1] : (const void*)Lv.P[1];
    g.inv0 = LV0 ? Lv.inv[0] : nullptr;
    g.inv1 = LV0 ? Lv.inv[1] : nullptr;
    g.nrm0 = Lv.nrm[0];
    g.nrm1 = Lv.nrm[1];
    g.path = Lv.path;
    g.W = W;
    g.T = ty.n;
    g.costs = Lv.costs;
    g.boff = Lv.boff;
    g.status = P.status;
    band2_block<E, ROWS, NSLOT, UPW, S>(g, ty, plan, a0, TAe, item % nchunk_b, st0, st1, st2, st3, Fs, spx, spy, snrm, sinv, ltx, lty, tmask, lslot);
}

// ------------------------------------------------------------------------------ third generation (16-bit rows)
// For type sets with at most 4 overlap layers per side and 10 types (alignment_max_size <= 5: the benchmark
// configuration).  Half of a chunk's bytes -- the source rows -- never enter LDS: wave w owns source layer w >> 1,
// x tile w & 1, and streams its 16 rows x 64 bytes per k-slab straight from global memory into a register ring
// of S MFMA A-fragments; only the target rows go through the LDS ring (ky x 32 rows x 64 bytes = 8 KB per stage).
// S = 8 stages, filled and drained in groups of G = 2 slabs (128 adjacent bytes of every row per group, one barrier
// per group): three groups = 96 KB per workgroup are in flight, and two workgroups share a CU (the output image
// reuses the ring once the k loop has drained, 68 KB of LDS per workgroup).  Measured on the 1024-pair batch:
// 6 stages, slab by slab: 17.9 ms; this shape: 17.15 ms.
constexpr int B3_THREADS = 512, B3_ROWS = 32, B3_S = 8, B3_G = 2, B3_KX = 4, B3_KY = 4, B3_TMAX = 10;
constexpr int B3_STAGE = B3_KY * B3_ROWS * B2_SLAB;  // 8 KB
constexpr int B3_FS = (2 * (B3_ROWS - B2_TB) + 1) * B2_TB * B3_TMAX;

struct BandPlan3 {
    int kx, ky;
    int wt_n[B3_KX];            // types per source layer
    int wt_type[B3_KX][B3_KY];  // their indices in the type list
    int wt_yl[B3_KX][B3_KY];    // ... and target layers
};

// NK = d * 2 / 64 k-slabs, a compile-time constant: the k loop is straight-line code, in which the compiler's
// s_waitcnt bookkeeping for the fragment loads is exact (across a loop's back edge it falls back to vmcnt(0)).
// AASM: the fragment loads are issued by inline assembly (the compiler's own bookkeeping drains the queue with
// s_waitcnt vmcnt(0) once per ring revolution; here the counted wait in front of the barrier is the only one).
// Safe because nothing but this step's MFMAs reads a fragment, and they also consume the target fragments that an
// inline-assembly statement behind the barrier delivers -- volatile statements keep their order.
template <typename E, int NK, bool AASM, int G>
__device__ void band3_block(const Band2Args& g, const BandPlan3& plan, int a0, int TAe, int chunk_b, char* ring, float* Fs, int* spx,
                            int* spy, float* snrm, float* sinv) {
    using St = typename E::storage;
    static_assert(sizeof(St) == 2, "16-bit rows");
    constexpr int ROWS = B3_ROWS, S = B3_S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = wave >> 1, xt = wave & 1;
    const int B = 2 * g.W, b0 = chunk_b * B2_TB;
    const int TBe = (B - b0) < B2_TB ? (B - b0) : B2_TB;
    constexpr int rowbytes = NK * B2_SLAB;  // (the launcher checked d * 2 == NK * 64)
    const int kx = plan.kx, ky = plan.ky;

    if (tid < TAe) {
        const int2 p = reinterpret_cast<const int2*>(g.path)[a0 + tid];
        spx[tid] = p.x;
        spy[tid] = p.y;
    }
    __syncthreads();
    if (tid < TAe) {
        bool ok = (spx[tid] + spy[tid] == a0 + tid);
        if (tid > 0) ok = ok && spx[tid] >= spx[tid - 1] && spy[tid] >= spy[tid - 1];
        if (!ok && g.status) *g.status = SVX_ERR_PATH;
        if (chunk_b == 0) g.boff[a0 + tid] = spy[tid] - g.W;
    }
    const int xlo = spx[0], ylo = spy[0];
    const int X0 = xlo + g.W - (b0 + TBe - 1);
    const int Y0 = ylo - g.W + b0;
    int NXn = spx[TAe - 1] - xlo + TBe, NYn = spy[TAe - 1] - ylo + TBe;
    NXn = NXn < 1 ? 1 : (NXn > ROWS ? ROWS : NXn);
    NYn = NYn < 1 ? 1 : (NYn > ROWS ? ROWS : NYn);

    // ---- per-row scalars (rows [0, kx*32): source layers, [128, 128 + ky*32): target layers); used in the epilogue
    float r_nrm = 0.f, r_inv = 1.f;
    if (tid < 2 * B3_KX * ROWS) {
        const int side = tid >= B3_KX * ROWS, rr = tid - side * B3_KX * ROWS;
        const int layer = rr / ROWS, loc = rr % ROWS;
        const int gi = (side ? Y0 : X0) + loc, nn = side ? g.m : g.n;
        if (layer < (side ? ky : kx) && loc < (side ? NYn : NXn) && gi >= 0 && gi < nn) {
            const size_t o = (size_t)layer * nn + gi;
            r_nrm = (side ? g.nrm1 : g.nrm0)[o];
            const float* iv = side ? g.inv1 : g.inv0;
            if (iv) r_inv = iv[o];
        }
    }
    const int lrow = lane & 15, lkg = lane >> 4;
    const int loff = lrow * B2_SLAB + 16 * (lkg ^ swz(lrow));
    // ---- this wave's DMA piece of a target slab (rows 16 wave .. 16 wave + 15 of the stage) and its own source rows
    const char* ysrc = reinterpret_cast<const char*>(band_zero16);
    bool ylive = false;
    {
        const int r = 16 * wave + (lane >> 2), yl = r / ROWS, loc = r % ROWS;
        const int gi = Y0 + loc;
        if (yl < ky && loc < NYn && gi >= 0 && gi < g.m) {
            ysrc = reinterpret_cast<const char*>(g.v1) + ((size_t)yl * g.m + gi) * rowbytes + ((lane & 3) ^ swz(lane >> 2)) * 16;
            ylive = true;
        }
    }
    // (address-space-1 pointers throughout: a generic pointer would make these flat loads, which the compiler orders
    //  against every pending LDS-DMA with s_waitcnt vmcnt(0))
    typedef const __attribute__((address_space(1))) char* gcp_t;
    const gcp_t zsrc = (gcp_t)(gptr_t) reinterpret_cast<const char*>(band_zero16);
    gcp_t xsrc = zsrc;
    bool xlive = false;
    {
        const int loc = 16 * xt + lrow, gi = X0 + loc;
        if (xl < kx && loc < NXn && gi >= 0 && gi < g.n) {
            xsrc = (gcp_t)(gptr_t) reinterpret_cast<const char*>(g.v0) + ((size_t)xl * g.n + gi) * rowbytes + lkg * 16;
            xlive = true;
        }
    }
    auto issue_y = [&](int k, char* stage) {
        const char* s2 = ylive ? ysrc + (size_t)k * B2_SLAB : reinterpret_cast<const char*>(band_zero16);
        __builtin_amdgcn_global_load_lds((gptr_t)s2, (lptr_t)(stage + wave * 1024), 16, 0, 0);
    };
    auto load_x = [&](int k) -> uint4 {
        // (every wave issues this load, also for dead rows: the counted waits below assume two operations per slab)
        const gcp_t s2 = xlive ? xsrc + (size_t)k * B2_SLAB : zsrc;
        typedef uint32_t u32x4a_t __attribute__((ext_vector_type(4)));
        u32x4a_t v;
        if constexpr (AASM) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(s2) : "memory");
        else v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4a_t*>(s2);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    // ---- this wave's types: source layer xl against target layers wt_yl[xl][..]
    static_assert(B3_KY == 4 && 16 * B2_SLAB == 1024, "the fragment reads below are written out for four target layers");
    const int ntw = xl < kx ? plan.wt_n[xl] : 0;
    unsigned boffb[B3_KY];
#pragma unroll
    for (int i = 0; i < B3_KY; i++) boffb[i] = (unsigned)(((i < ntw ? plan.wt_yl[xl < B3_KX ? xl : 0][i] : 0) * ROWS) * B2_SLAB + loff);
    f32x4_t acc[B3_KY][2];
#pragma unroll
    for (int i = 0; i < B3_KY; i++) acc[i][0] = acc[i][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    uint4 afr[S];
    // The target fragments are read by inline assembly, reads and their s_waitcnt in one statement.  Left to the
    // compiler, some of the ds_reads lose the alias scope that tells them apart from the stages still being filled
    // by LDS-DMA, and it drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of them once per ring revolution.
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    auto mma = [&](const char* stage, const uint4& fa) {
        const unsigned sb = (unsigned)(size_t)(lptr_t)const_cast<char*>(stage);
        u32x4_t b00, b01, b10, b11, b20, b21, b30, b31;
        if (ntw >= 4) {
            asm volatile(
                "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:1024\n\t"
                "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:1024\n\tds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:1024\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11), "=&v"(b20), "=&v"(b21), "=&v"(b30), "=&v"(b31)
                : "v"(sb + boffb[0]), "v"(sb + boffb[1]), "v"(sb + boffb[2]), "v"(sb + boffb[3])
                : "memory");
        } else if (ntw == 3) {
            asm volatile(
                "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:1024\n\t"
                "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11), "=&v"(b20), "=&v"(b21)
                : "v"(sb + boffb[0]), "v"(sb + boffb[1]), "v"(sb + boffb[2])
                : "memory");
        } else if (ntw == 2) {
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11)
                : "v"(sb + boffb[0]), "v"(sb + boffb[1])
                : "memory");
        } else if (ntw == 1) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b00), "=&v"(b01)
                         : "v"(sb + boffb[0])
                         : "memory");
        }
        if (ntw >= 1) {
            mma_slab16<E>(acc[0][0], fa, make_uint4(b00.x, b00.y, b00.z, b00.w));
            mma_slab16<E>(acc[0][1], fa, make_uint4(b01.x, b01.y, b01.z, b01.w));
        }
        if (ntw >= 2) {
            mma_slab16<E>(acc[1][0], fa, make_uint4(b10.x, b10.y, b10.z, b10.w));
            mma_slab16<E>(acc[1][1], fa, make_uint4(b11.x, b11.y, b11.z, b11.w));
        }
        if (ntw >= 3) {
            mma_slab16<E>(acc[2][0], fa, make_uint4(b20.x, b20.y, b20.z, b20.w));
            mma_slab16<E>(acc[2][1], fa, make_uint4(b21.x, b21.y, b21.z, b21.w));
        }
        if (ntw >= 4) {
            mma_slab16<E>(acc[3][0], fa, make_uint4(b30.x, b30.y, b30.z, b30.w));
            mma_slab16<E>(acc[3][1], fa, make_uint4(b31.x, b31.y, b31.z, b31.w));
        }
    };
    // ---- k loop over groups of G slabs (G * 64 bytes of every row: the group's requests to a row are adjacent and
    // issued back to back).  Per group every wave issues G DMA pieces and G fragment loads, so a group's 2 G
    // operations are followed by 2 G per younger group: one counted wait in front of the barrier covers them all.
    static_assert(S % G == 0 && NK % G == 0, "whole groups");
    constexpr int GS = S / G, NG = NK / G;  // groups in the ring, groups in a row
    static_assert(GS >= 3, "ring depth");
    auto issue_group = [&](int grp) {
#pragma unroll
        for (int u = 0; u < G; u++) issue_y(grp * G + u, ring + ((grp * G + u) % S) * B3_STAGE);
#pragma unroll
        for (int u = 0; u < G; u++) afr[(grp * G + u) % S] = load_x(grp * G + u);
    };
#pragma unroll
    for (int grp = 0; grp < GS - 1; grp++)
        if (grp < NG) issue_group(grp);
    // (fully unrolled: the slab numbers, the ring slots and the wait counts are constants in every copy of the body)
#pragma unroll
    for (int grp = 0; grp < NG; grp++) {
        constexpr int OPS = 2 * G;
        const int younger = (NG - 1 - grp) < (GS - 2) ? (NG - 1 - grp) : (GS - 2);  // groups issued after this one
        if (younger >= 6) wait_vm<6 * OPS>();
        else if (younger == 5) wait_vm<5 * OPS>();
        else if (younger == 4) wait_vm<4 * OPS>();
        else if (younger == 3) wait_vm<3 * OPS>();
        else if (younger == 2) wait_vm<2 * OPS>();
        else if (younger == 1) wait_vm<1 * OPS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // every wave's pieces of the group have landed; the group before it is free
        asm volatile("" ::: "memory");
        uint4 fa[G];
#pragma unroll
        for (int u = 0; u < G; u++) fa[u] = afr[(grp * G + u) % S];
        if (grp + GS - 1 < NG) issue_group(grp + GS - 1);
#pragma unroll
        for (int u = 0; u < G; u++) mma(ring + ((grp * G + u) % S) * B3_STAGE, fa[u]);
    }
    // ---- epilogue
    if (tid < 2 * B3_KX * ROWS) {
        snrm[tid] = r_nrm;
        sinv[tid] = r_inv;
    }
    __syncthreads();
    const float inf = __builtin_inff();
    const int T = g.T;
#pragma unroll
    for (int i = 0; i < B3_KY; i++) {
        if (i >= ntw) continue;
        const int t = plan.wt_type[xl < B3_KX ? xl : 0][i], yl = plan.wt_yl[xl < B3_KX ? xl : 0][i];
        const int p = xl + 1, q = yl + 1;
        const int xs = xl * ROWS, ys = B3_KX * ROWS + yl * ROWS;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int yloc = 16 * j + lrow, yy = Y0 + yloc;
            const float ny = snrm[ys + yloc], iy = sinv[ys + yloc];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int xloc = 16 * xt + 4 * lkg + r, xx = X0 + xloc;
                const int ai = xx + yy - a0;
                if (ai < 0 || ai >= TAe) continue;
                const int bi = yy - (spy[ai] - g.W) - b0;
                if (bi < 0 || bi >= TBe) continue;
                float c = inf;
                if (xx >= 0 && xx < g.n && yy >= 0 && yy < g.m) {
                    const float sumx = acc[i][j][r] * sinv[xs + xloc] * iy;
                    c = cost_formula2(sumx, p, q, snrm[xs + xloc], ny);
                }
                Fs[(ai * T + t) * TBe + bi] = c;
            }
        }
    }
    __syncthreads();
    if (TBe == B && ((T * B) & 3) == 0) {
        const int nv = TAe * T * TBe / 4;
        float4* dst = reinterpret_cast<float4*>(g.costs + (size_t)a0 * T * B);
        const float4* srcv = reinterpret_cast<const float4*>(Fs);
        for (int i = tid; i < nv; i += B3_THREADS) dst[i] = srcv[i];
    } else {
        const int nout = TAe * T * TBe;
        for (int idx = tid; idx < nout; idx += B3_THREADS) {
            const int ai = idx / (T * TBe);
            const int rem = idx - ai * (T * TBe);
            const int tl = rem / TBe, bi = rem - tl * TBe;
            g.costs[((size_t)(a0 + ai) * T + tl) * B + (b0 + bi)] = Fs[idx];
        }
    }
}

template <typename E, int NK, bool AASM, int G>
__global__ __launch_bounds__(B3_THREADS, 4) void k_band_costs3(const SvxPairDev* __restrict__ pairs, BandPlan3 plan, int T, int W, int nchunk_b,
                                                               int per_pair) {
    // One ring of B3_S stages; the output image takes its place once the k loop has drained (B3_FS floats fit in it).
    __shared__ __attribute__((aligned(1024))) char ring[B3_S * B3_STAGE];
    static_assert(B3_FS * sizeof(float) <= sizeof(ring), "output image inside the ring");
    float* Fs = reinterpret_cast<float*>(ring);
    __shared__ int spx[2 * B3_ROWS], spy[2 * B3_ROWS];
    __shared__ float snrm[2 * B3_KX * B3_ROWS], sinv[2 * B3_KX * B3_ROWS];
    const unsigned wg = xcd_remap2(blockIdx.x, gridDim.x);
    const SvxPairDev& P = pairs[wg / per_pair];
    const int item = wg % per_pair;
    const SvxLevel& Lv = P.lev[0];
    Band2Args g;
    g.A = *Lv.path_len;
    const int chunk_a = item / nchunk_b;
    if (g.A <= 0 || chunk_a >= *Lv.nchunks) return;
    const int a0 = Lv.cstart[chunk_a], TAe = Lv.cstart[chunk_a + 1] - a0;
    if (TAe <= 0 || TAe > 2 * (B3_ROWS - 1) + 1) return;
    g.n = Lv.n[0];
    g.m = Lv.n[1];
    g.d = P.d;
    g.v0 = P.v[0];
    g.v1 = P.v[1];
    g.inv0 = Lv.inv[0];
    g.inv1 = Lv.inv[1];
    g.nrm0 = Lv.nrm[0];
    g.nrm1 = Lv.nrm[1];
    g.path = Lv.path;
    g.W = W;
    g.T = T;
    g.costs = Lv.costs;
    g.boff = Lv.boff;
    g.status = P.status;
    band3_block<E, NK, AASM, G>(g, plan, a0, TAe, item % nchunk_b, ring, Fs, spx, spy, snrm, sinv);
}

bool make_plan3(const SvxTypes& ty, BandPlan3* plan) {
    memset(plan, 0, sizeof(*plan));
    if (ty.n < 1 || ty.n > B3_TMAX) return false;
    for (int t = 0; t < ty.n; t++) {
        const int lx = ty.x[t] - 1, ly = ty.y[t] - 1;
        if (lx >= B3_KX || ly >= B3_KY) return false;
        if (plan->wt_n[lx] >= B3_KY) return false;
        plan->wt_type[lx][plan->wt_n[lx]] = t;
        plan->wt_yl[lx][plan->wt_n[lx]] = ly;
        plan->wt_n[lx]++;
        if (lx + 1 > plan->kx) plan->kx = lx + 1;
        if (ly + 1 > plan->ky) plan->ky = ly + 1;
    }
    return true;
}

int band_version() {
    const char* env = getenv("SVX_BAND_V");
    return env ? atoi(env) : 3;
}

// Passes: consecutive types are packed while they fit the per-pass limits (types and distinct overlap layers).
bool make_plan(const SvxTypes& ty, int tpp, int nslot_max, BandPlan* plan) {
    plan->npass = 0;
    int t = 0;
    while (t < ty.n) {
        if (plan->npass >= B2_MAXPASS) return false;
        BandPass& ps = plan->pass[plan->npass];
        memset(&ps, 0, sizeof(ps));
        ps.t0 = t;
        int xs[SVX_MAX_TYPES + 2], ys[SVX_MAX_TYPES + 2];  // layer -> slot + 1
        memset(xs, 0, sizeof(xs));
        memset(ys, 0, sizeof(ys));
        while (t < ty.n && ps.nt < tpp) {
            const int lx = ty.x[t] - 1, ly = ty.y[t] - 1;
            const int need = (xs[lx] ? 0 : 1) + (ys[ly] ? 0 : 1);
            if (ps.nslot + need > nslot_max) break;
            if (!xs[lx]) { ps.slot_info[ps.nslot] = lx; xs[lx] = ++ps.nslot; }
            if (!ys[ly]) { ps.slot_info[ps.nslot] = (1 << 8) | ly; ys[ly] = ++ps.nslot; }
            ps.type_slots[ps.nt] = (xs[lx] - 1) | ((ys[ly] - 1) << 8);
            ps.nt++;
            t++;
        }
        if (ps.nt == 0) return false;
        plan->npass++;
    }
    return true;
}

struct Variant {
    int rows, nslot, upw, s;
};

// Kernel shape for a type set: the smallest staging area that takes all its layers in one pass where possible.
bool choose_variant(const SvxTypes& ty, int depth, Variant* v, BandPlan* plan) {
    if (ty.n <= 0) return false;
    int kx = 0, ky = 0;
    for (int t = 0; t < ty.n; t++) {
        if (ty.x[t] > kx) kx = ty.x[t];
        if (ty.y[t] > ky) ky = ty.y[t];
    }
    (void)depth;
    // one alignment type on one layer per side (the deeper pyramid levels, alignment_max_size 2): 64 rows per side,
    // so that a chunk holds up to 101 path points and only 1.28 x the rows it needs travel from L2
    const Variant opts[4] = {{64, 2, 1, 4}, {32, 8, 5, 3}, {32, 12, 8, 3}, {32, 20, 8, 2}};
    for (int i = 0; i < 4; i++) {
        const int tpp = B2_WAVES * opts[i].upw / (opts[i].rows / 16);
        const bool last = i == 3;
        if (!last && (kx + ky > opts[i].nslot || ty.n > tpp)) continue;
        if (!make_plan(ty, tpp, opts[i].nslot, plan)) continue;
        *v = opts[i];
        return true;
    }
    return false;
}

}  // namespace

// Does the second-generation kernel handle this level?  (out: rows a chunk may span per side and the most path
// points of a chunk, for k_chunk_path.)  SVX_BAND_V1=1 keeps the first-generation kernel for A/B measurements.
// third generation: level 0, 16-bit rows of 256 / 512 / 1024 elements, <= 4 layers per side, <= 10 types
static bool use_v3(const SvxTypes& types, int depth, int dtype, int d, BandPlan3* plan3) {
    const int nk3 = d / 32;  // 64-byte slabs of a 16-bit row
    return depth == 0 && dtype != SVX_F32 && band_version() >= 3 && d % 32 == 0 && (nk3 == 32 || nk3 == 16 || nk3 == 8) &&
           make_plan3(types, plan3);
}

bool svxl_band2_limits(const SvxTypes& types, int W, int depth, int dtype, int d, int* lim, int* tamax) {
    const char* env = getenv("SVX_BAND_V1");
    if ((env && atoi(env) != 0) || band_version() <= 1) return false;
    const char* deep = getenv("SVX_BAND_DEEP_V1");
    if (depth > 0 && deep && atoi(deep) != 0) return false;
    Variant v;
    BandPlan plan;
    BandPlan3 plan3;
    if (use_v3(types, depth, dtype, d, &plan3)) v.rows = B3_ROWS;
    else if (!choose_variant(types, depth, &v, &plan)) return false;
    const int B = 2 * W, tbe = B < B2_TB ? B : B2_TB;
    *lim = v.rows - tbe;
    *tamax = 2 * (v.rows - tbe) + 1;
    return true;
}

int svxl_band_costs2_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int max_A, const SvxTypes& types, int W,
                           int dtype, int d) {
    if (n_pairs <= 0 || max_A <= 0) return SVX_OK;
    BandPlan3 plan3;
    const int nk3 = d / 32;
    if (use_v3(types, depth, dtype, d, &plan3)) {
        // (same chunk limits as the 32-row shapes of the second generation: svxl_band2_limits)
        const int B3 = 2 * W, tbe3 = B3 < B2_TB ? B3 : B2_TB, lim3 = B3_ROWS - tbe3;
        const int nca3 = (max_A + lim3) / (lim3 + 1), ncb3 = (B3 + B2_TB - 1) / B2_TB;
        dim3 grid3((unsigned)(nca3 * ncb3) * (unsigned)n_pairs);
        const char* aenv = getenv("SVX_BAND_ASMLOAD");
        const bool aasm = aenv ? atoi(aenv) != 0 : true;
#define CALL3(E, NKT)                                                                                                             \
    do {                                                                                                                          \
        if (aasm) hipLaunchKernelGGL((k_band_costs3<E, NKT, true, B3_G>), grid3, dim3(B3_THREADS), 0, ctx->stream, pairs, plan3, types.n, W, ncb3, nca3 * ncb3); \
        else hipLaunchKernelGGL((k_band_costs3<E, NKT, false, B3_G>), grid3, dim3(B3_THREADS), 0, ctx->stream, pairs, plan3, types.n, W, ncb3, nca3 * ncb3); \
    } while (0)
#define CALL3_E(E)                      \
    do {                                \
        if (nk3 == 32) CALL3(E, 32);    \
        else if (nk3 == 16) CALL3(E, 16); \
        else CALL3(E, 8);               \
    } while (0)
        if (dtype == SVX_F16) CALL3_E(ElemF16);
        else CALL3_E(ElemBF16);
#undef CALL3_E
#undef CALL3
        SVX_LAUNCH_CHECK(ctx, "k_band_costs3");
        return SVX_OK;
    }
    Variant v;
    BandPlan plan;
    if (!choose_variant(types, depth, &v, &plan)) return svx_fail(ctx, SVX_ERR_ARG, "band costs: no kernel shape for %d alignment types", types.n);
    const int B = 2 * W, tbe = B < B2_TB ? B : B2_TB;
    const int lim = v.rows - tbe;
    const int nca = (max_A + lim) / (lim + 1), ncb = (B + B2_TB - 1) / B2_TB;  // a chunk holds at least lim + 1 points
    const int per_pair = nca * ncb;
    dim3 grid((unsigned)per_pair * (unsigned)n_pairs);
    hipStream_t st = ctx->stream;
#define CALL2(E, LV0, ROWS, NSLOT, UPW, S) \
    hipLaunchKernelGGL((k_band_costs2<E, LV0, ROWS, NSLOT, UPW, S>), grid, dim3(B2_THREADS), 0, st, pairs, depth, types, plan, W, ncb, per_pair)
#define CALL2_V(E, LV0)                                      \
    do {                                                     \
        if (v.nslot == 2) CALL2(E, LV0, 64, 2, 1, 4);        \
        else if (v.nslot == 8) CALL2(E, LV0, 32, 8, 5, 3);   \
        else if (v.nslot == 12) CALL2(E, LV0, 32, 12, 8, 3); \
        else CALL2(E, LV0, 32, 20, 8, 2);                    \
    } while (0)
    if (depth > 0) CALL2_V(ElemF32, false);
    else if (dtype == SVX_F32) CALL2_V(ElemF32, true);
    else if (dtype == SVX_F16) CALL2_V(ElemF16, true);
    else CALL2_V(ElemBF16, true);
#undef CALL2_V
#undef CALL2
    SVX_LAUNCH_CHECK(ctx, "k_band_costs2");
    return SVX_OK;
}
