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
constexpr int B2_MAXPASS = 16;
constexpr int B2_MAXSLOT = 20;     // overlap layers (both sides) one pass can stage
constexpr int B2_MAXTPP = 16;      // alignment types per pass

__device__ const uint4 band_zero16[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

// one pass = a run of consecutive alignment types and the overlap layers ("slots") they read
struct BandPass {
    int t0, nt, nslot;
    int8_t slot_side[B2_MAXSLOT], slot_layer[B2_MAXSLOT];
    int8_t type_xslot[B2_MAXTPP], type_yslot[B2_MAXTPP];
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
                            int* ltx, int* lty, int* tmask) {
    using C = Band2<E, ROWS, NSLOT, UPW, S>;
    using St = typename E::storage;
    constexpr int XT = C::XT, PW = C::PW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        const int xa = spx[tid], ya = spy[tid];
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
                const int side = ps.slot_side[slot], layer = ps.slot_layer[slot];
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
                const int side = ps.slot_side[slot], layer = ps.slot_layer[slot];
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
        // ---- units of this wave: (type of the pass, x tile)
        f32x4_t acc[UPW][XT];
        int aoff[UPW], boffb[UPW], umask[UPW];
        const int nunits = ntp * XT;
#pragma unroll
        for (int s = 0; s < UPW; s++) {
#pragma unroll
            for (int j = 0; j < XT; j++) acc[s][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int u = wave + B2_WAVES * s;
            const int tl = u < nunits ? u / XT : 0, xt = u % XT;
            aoff[s] = (ps.type_xslot[tl] * ROWS + xt * 16) * B2_SLAB + loff;
            boffb[s] = (ps.type_yslot[tl] * ROWS) * B2_SLAB + loff;
            umask[s] = 0;
        }
        auto mma = [&](const char* stage) {
#pragma unroll
            for (int s = 0; s < UPW; s++) {
                if (umask[s]) {  // wave-uniform
                    const uint4 fa = *reinterpret_cast<const uint4*>(stage + aoff[s]);
#pragma unroll
                    for (int j = 0; j < XT; j++)
                        if ((umask[s] >> j) & 1) {
                            const uint4 fb = *reinterpret_cast<const uint4*>(stage + boffb[s] + j * 16 * B2_SLAB);
                            mma_slab16<E>(acc[s][j], fa, fb);
                        }
                }
            }
        };
        __syncthreads();  // tmask complete (first pass) / the previous pass is done with the stages and Fs
#pragma unroll
        for (int s = 0; s < UPW; s++) {
            const int u = wave + B2_WAVES * s;
            umask[s] = u < nunits ? tmask[u % XT] : 0;
        }
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
            const int xs = ps.type_xslot[tl] * ROWS, ys = ps.type_yslot[tl] * ROWS;
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
    band2_block<E, ROWS, NSLOT, UPW, S>(g, ty, plan, a0, TAe, item % nchunk_b, st0, st1, st2, st3, Fs, spx, spy, snrm, sinv, ltx, lty, tmask);
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
            if (!xs[lx]) { ps.slot_side[ps.nslot] = 0; ps.slot_layer[ps.nslot] = (int8_t)lx; xs[lx] = ++ps.nslot; }
            if (!ys[ly]) { ps.slot_side[ps.nslot] = 1; ps.slot_layer[ps.nslot] = (int8_t)ly; ys[ly] = ++ps.nslot; }
            ps.type_xslot[ps.nt] = (int8_t)(xs[lx] - 1);
            ps.type_yslot[ps.nt] = (int8_t)(ys[ly] - 1);
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
    const Variant opts[3] = {{32, 8, 5, 3}, {32, 12, 8, 3}, {32, 20, 8, 2}};
    for (int i = 0; i < 3; i++) {
        const int tpp = B2_WAVES * opts[i].upw / (opts[i].rows / 16);
        const bool last = i == 2;
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
bool svxl_band2_limits(const SvxTypes& types, int W, int depth, int* lim, int* tamax) {
    const char* env = getenv("SVX_BAND_V1");
    if (env && atoi(env) != 0) return false;
    if (depth > 0) return false;  // (deeper levels: single-layer fp32 rows keep the first-generation kernel for now)
    Variant v;
    BandPlan plan;
    if (!choose_variant(types, depth, &v, &plan)) return false;
    const int B = 2 * W, tbe = B < B2_TB ? B : B2_TB;
    *lim = v.rows - tbe;
    *tamax = 2 * (v.rows - tbe) + 1;
    return true;
}

int svxl_band_costs2_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int max_A, const SvxTypes& types, int W,
                           int dtype, int d) {
    (void)d;
    if (n_pairs <= 0 || max_A <= 0) return SVX_OK;
    Variant v;
    BandPlan plan;
    if (!choose_variant(types, depth, &v, &plan)) return svx_fail(ctx, SVX_ERR_ARG, "band costs: no kernel shape for %d alignment types", types.n);
    const int B = 2 * W, tbe = B < B2_TB ? B : B2_TB;
    const int lim = v.rows - tbe;
    const int nca = (max_A + lim) / (lim + 1), ncb = (B + B2_TB - 1) / B2_TB;  // a chunk holds at least lim + 1 points
    const int per_pair = nca * ncb;
    dim3 grid((unsigned)per_pair * (unsigned)n_pairs);
    hipStream_t st = ctx->stream;
#define CALL2(E, LV0, NSLOT, UPW, S) \
    hipLaunchKernelGGL((k_band_costs2<E, LV0, 32, NSLOT, UPW, S>), grid, dim3(B2_THREADS), 0, st, pairs, depth, types, plan, W, ncb, per_pair)
#define CALL2_V(E, LV0)                                  \
    do {                                                 \
        if (v.nslot == 8) CALL2(E, LV0, 8, 5, 3);        \
        else if (v.nslot == 12) CALL2(E, LV0, 12, 8, 3); \
        else CALL2(E, LV0, 20, 8, 2);                    \
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
