// svx_tiles.hip -- wide bands (Sakoe-Chiba search around the straight diagonal, and the dense "every cell"
// mode): similarity tiles on the matrix cores feeding the dynamic programme directly, swept as a wavefront of
// tiles over all CUs.
//
// Reference semantics: make_sparse_costs + sparse_dp (svecalign/vecalign/dp_core.pyx:165-267, 269-404) on a
// straight search path (append_slant, dp_utils.py:177-196) with band half-width W -- what
// dp_utils.vecalign() evaluates when max_size_full_dp is large and width_over2 covers the documents
// (Modes B and C).  The first-generation path materialised the [T][A][B] cost tensor (10.7 GB
// at 32768 x 32768, band 2048) and ran the DP of a pair in one workgroup.
//
// Here the lattice of DP nodes (x, y) is cut into 32 x 32 tiles.  A tile's T cost planes are 16 x 16 MFMA blocks
// of the raw 16-bit / fp32 rows (32 rows x all overlap layers per side, LDS-DMA ring of swizzled 64-byte k-slabs:
// the machinery of svx_band.hip), scaled by the two inverse norms and turned into costs in double like the
// reference; they never leave LDS.  The tile's nodes are then relaxed along its 63 anti-diagonals by one wave
// (two half-waves split the moves and merge by (total, move) so that the reference's "first strictly smaller
// candidate" order is kept), reading a 4..8 node halo of the float64 sums of the tiles above and to the left.
// Tiles depend on their upper, left and upper-left neighbours only, so all tiles of a tile anti-diagonal run
// concurrently: persistent workgroups take tickets in (pair, tile anti-diagonal) order -- every dependency of a
// ticket has a smaller number, hence is already claimed by a running workgroup: no deadlock under any dispatch
// order -- compute the cost planes first (they need no neighbour), then wait for the neighbours' flags.
// Results go to the reference's node arrays (csum [A+2][B] float64, packed back-pointers), which the existing
// traceback kernel walks.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svx_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TL = 32;             // tile side in DP nodes
constexpr int TT_THREADS = 256, TT_WAVES = 4, TT_SLAB = 64;
constexpr int TT_MAXSLOT = 12, TT_MAXT = 16, TT_HALO = 8;
constexpr int CS_W = TL + TT_HALO;  // csum tile with halo

__device__ const uint4 tile_zero16[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

struct TilePlan {
    int nt, nslot, halo;
    int slot_info[TT_MAXSLOT];  // side << 8 | layer
    int type_slots[TT_MAXT];    // x slot | y slot << 8
};

__device__ __forceinline__ float cost_formula_t(float sumx, int p, int q, float n0, float n1) {
#pragma clang fp contract(off)
    return (float)((((2.0 * (double)p) * (double)q) * (1.0 - (double)sumx)) / ((1e-6 + (double)n0) + (double)n1));
}

template <typename E>
__device__ __forceinline__ void mma_t16(f32x4_t& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mma_t16<ElemBF16>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_t16<ElemF16>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_t16<ElemF32>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_t() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int swz_t(int row_in_tile) { return (0x1320 >> (4 * ((row_in_tile >> 2) & 3))) & 3; }

// b_offset_out of a pair, then per tile anti-diagonal s the run of tiles (I, s - I) that hold band nodes.
// Node (x, y) is in the band iff 0 <= y - bo[x + y] < B and 0 <= x <= xs, 0 <= y <= ys.
__global__ __launch_bounds__(256) void k_tile_ranges(const SvxPairDev* __restrict__ pairs, int W, int B) {
    const SvxPairDev& P = pairs[blockIdx.x];
    const SvxLevel& Lv = P.lev[0];
    const int A = *Lv.path_len;
    const int tid = threadIdx.x;
    for (int i = tid; i < P.t_cap; i += 256) P.t_flag[i] = 0;
    if (A <= 0 || *P.status != 0) {
        if (tid == 0) P.t_pref[P.t_nd] = 0;
        for (int s = tid; s < P.t_nd; s += 256) { P.t_cnt[s] = 0; P.t_pref[s] = 0; }
        return;
    }
    const int2* path = reinterpret_cast<const int2*>(Lv.path);
    for (int a = tid; a < A + 2; a += 256) Lv.boff_out[a] = (a < 2 ? path[0].y : path[a - 2].y + 1) - W;
    for (int a = tid; a < A; a += 256) Lv.boff[a] = path[a].y - W;
    __syncthreads();
    const int xs = Lv.n[0], ys = Lv.n[1];
    for (int s = tid; s < P.t_nd; s += 256) {
        int ilo = 1 << 30, ihi = -1;
        for (int r = 0; r <= 2 * (TL - 1); r++) {
            const int a = TL * s + r;
            if (a > xs + ys || a >= A + 2) break;
            const int bo = Lv.boff_out[a];
            int ylo = bo > 0 ? bo : 0, yhi = bo + B - 1;
            if (a - xs > ylo) ylo = a - xs;
            if (yhi > ys) yhi = ys;
            if (yhi > a) yhi = a;
            if (ylo > yhi) continue;
            const int jm_lo = r - (TL - 1) > 0 ? r - (TL - 1) : 0, jm_hi = r < TL - 1 ? r : TL - 1;
            // tiles J on this diagonal whose rows 32 J + [jm_lo, jm_hi] meet [ylo, yhi]
            int jmin = (ylo - jm_hi + TL - 1) / TL;
            if (ylo - jm_hi < 0) jmin = 0;
            const int jmax = (yhi - jm_lo) >= 0 ? (yhi - jm_lo) / TL : -1;
            if (jmin > jmax) continue;
            if (s - jmax < ilo) ilo = s - jmax;
            if (s - jmin > ihi) ihi = s - jmin;
        }
        if (ilo < 0) ilo = 0;
        P.t_lo[s] = ilo <= ihi ? ilo : 0;
        P.t_cnt[s] = ilo <= ihi ? ihi - ilo + 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int s = 0; s < P.t_nd; s++) {
            P.t_pref[s] = run;
            run += P.t_cnt[s];
        }
        P.t_pref[P.t_nd] = run > P.t_cap ? 0 : run;  // (cannot exceed the capacity the planner derived from the same geometry)
        if (run > P.t_cap) *P.status = SVX_ERR_ARG;
    }
}

// Ticket order: tile anti-diagonal s of every pair, then s + 1 of every pair, ...: the pairs of a batch advance
// together, so that the chip sees (pairs x tiles per diagonal) independent tiles instead of one pair's wavefront.
// gpref[s * n_pairs + p] = tickets before (s, p); gpref[max_nd * n_pairs] = total.  One workgroup, blocked scan.
__global__ __launch_bounds__(1024) void k_tile_prefix(const SvxPairDev* __restrict__ pairs, int n_pairs, int max_nd, int* gpref, int* ticket) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const long long n = (long long)max_nd * n_pairs;
    const long long per = (n + 1023) / 1024;
    const long long lo = tid * per, hi = (lo + per) < n ? (lo + per) : n;
    int sum = 0;
    for (long long e = lo; e < hi; e++) {
        const int s = (int)(e / n_pairs), p = (int)(e % n_pairs);
        sum += s < pairs[p].t_nd ? pairs[p].t_cnt[s] : 0;
    }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 1024; i++) { const int v = part[i]; part[i] = run; run += v; }
        gpref[n] = run;
        *ticket = 0;
    }
    __syncthreads();
    int run = part[tid];
    for (long long e = lo; e < hi; e++) {
        const int s = (int)(e / n_pairs), p = (int)(e % n_pairs);
        gpref[e] = run;
        run += s < pairs[p].t_nd ? pairs[p].t_cnt[s] : 0;
    }
}

template <typename E, int NSLOT, int UPW, int S>
struct TileCfg {
    static constexpr int STAGE = NSLOT * TL * TT_SLAB;
    static constexpr int NQ = NSLOT * TL / 16;
    static constexpr int PW = NQ / TT_WAVES;
    static constexpr int TPP = TT_WAVES * UPW / 2;
    static_assert(NQ % TT_WAVES == 0, "DMA pieces must divide over the waves");
    static_assert(TPP <= TT_MAXT && NSLOT <= TT_MAXSLOT, "plan limits");
};

struct DpMerge {
    double tot;
    int key;
};

template <typename E, int NSLOT, int UPW, int S>
__global__ __launch_bounds__(TT_THREADS) void k_band_tiles(const SvxPairDev* __restrict__ pairs, int n_pairs, SvxTypes ty, TilePlan plan,
                                                           int W, int max_nd, const int* __restrict__ gpref, int* ticket,
                                                           unsigned long long* prof) {
    using C = TileCfg<E, NSLOT, UPW, S>;
    using St = typename E::storage;
    constexpr int PW = C::PW;
    __shared__ __attribute__((aligned(1024))) char st0[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st1[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st2[S >= 3 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(1024))) char st3[S >= 4 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(1024))) char st4[S >= 5 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(16))) float planes[C::TPP * TL * TL];   // [type][x row][y row] costs of the tile
    __shared__ __attribute__((aligned(16))) double cs[CS_W * CS_W + 2];       // csum of the tile's nodes and halo; [CS_W^2] = +inf
    __shared__ unsigned char bpt[TL * TL];
    __shared__ float snrm[NSLOT * TL], sinv[NSLOT * TL];
    __shared__ int bo_l[2 * TL + 2 * TT_HALO + 2];
    __shared__ int tpk[TT_MAXT + 2];
    __shared__ int sh_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = 2 * W, T = ty.n, NTt = T + 2, H = plan.halo;
    const long long nent = (long long)max_nd * n_pairs;
    const int total = gpref[nent];
    for (int t = tid; t < NTt; t += TT_THREADS) tpk[t] = (int)ty.x[t] | ((int)ty.y[t] << 8);
    const int lrow = lane & 15, lkg = lane >> 4;
    const int loff = lrow * TT_SLAB + 16 * (lkg ^ swz_t(lrow));
    const double inf = __builtin_inf();
    long long ecur = 0;  // tickets grow: the (diagonal, pair) entry of a workgroup's ticket only moves forward

    // (diagnostic build-in: with SVX_TILE_PROF=1 thread 0 sums the 100 MHz ticks of every phase into prof[0..6])
    unsigned long long t_prev = prof ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int slot) {
        if (prof && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            atomicAdd(prof + slot, now - t_prev);
            t_prev = now;
        }
    };
    for (;;) {
        __syncthreads();  // the previous tile is finished with LDS
        stamp(6);
        if (tid == 0) sh_ticket = atomicAdd(ticket, 1);
        __syncthreads();
        const int tk = sh_ticket;
        if (tk >= total) break;
        // (gallop, then bisect: a workgroup's next ticket is usually a few entries ahead)
        if (gpref[ecur + 1] <= tk) {
            long long step = 1, lo = ecur + 1, hi = nent - 1;
            while (lo + step < nent && gpref[lo + step] <= tk) { lo += step; step <<= 1; }
            if (lo + step < hi) hi = lo + step;
            while (lo < hi) {
                const long long mid = (lo + hi + 1) >> 1;
                if (gpref[mid] <= tk) lo = mid; else hi = mid - 1;
            }
            ecur = lo;
        }
        const int s = (int)(ecur / n_pairs);
        const SvxPairDev& P = pairs[ecur % n_pairs];
        const SvxLevel& Lv = P.lev[0];
        const int I = P.t_lo[s] + (tk - gpref[ecur]), J = s - I;
        const int xs = Lv.n[0], ys = Lv.n[1], d = P.d;
        const int rowbytes = d * (int)sizeof(St);
        const int NK = (rowbytes + TT_SLAB - 1) / TT_SLAB;
        const int X0 = TL * I - 1, Y0 = TL * J - 1;  // cost cell of node (x, y) is (x - 1, y - 1)
        const int A = *Lv.path_len;
        const double pen = *Lv.pen;

        stamp(0);
        // ---- phase 1: cost planes (no neighbour needed)
        constexpr int SPT = (NSLOT * TL + TT_THREADS - 1) / TT_THREADS;
        float r_nrm[SPT], r_inv[SPT];
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int r = tid + i * TT_THREADS;
            r_nrm[i] = 0.f;
            r_inv[i] = 1.f;
            if (r < plan.nslot * TL) {
                const int slot = r / TL, loc = r % TL;
                const int side = plan.slot_info[slot] >> 8, layer = plan.slot_info[slot] & 255;
                const int gi = (side ? Y0 : X0) + loc, nn = side ? ys : xs;
                if (gi >= 0 && gi < nn) {
                    const size_t o = (size_t)layer * nn + gi;
                    r_nrm[i] = Lv.nrm[side][o];
                    if (Lv.inv[side]) r_inv[i] = Lv.inv[side][o];
                }
            }
        }
        const char* src[PW];
        unsigned live = 0;
#pragma unroll
        for (int i = 0; i < PW; i++) {
            const int q = wave + TT_WAVES * i;
            const int r = 16 * q + (lane >> 2);
            const int slot = r / TL, loc = r % TL;
            const int piece = (lane & 3) ^ swz_t(lane >> 2);
            src[i] = reinterpret_cast<const char*>(tile_zero16);
            if (slot < plan.nslot) {
                const int side = plan.slot_info[slot] >> 8, layer = plan.slot_info[slot] & 255;
                const int gi = (side ? Y0 : X0) + loc, nn = side ? ys : xs;
                if (gi >= 0 && gi < nn) {
                    src[i] = reinterpret_cast<const char*>(P.v[side]) + ((size_t)layer * nn + gi) * rowbytes + piece * 16;
                    live |= 1u << i;
                }
            }
        }
        const int piece_byte = ((lane & 3) ^ swz_t(lane >> 2)) * 16;
        auto issue = [&](int k, char* stage) {
#pragma unroll
            for (int i = 0; i < PW; i++) {
                const char* s2 = src[i] + (size_t)k * TT_SLAB;
                if (!((live >> i) & 1u) || k * TT_SLAB + piece_byte >= rowbytes) s2 = reinterpret_cast<const char*>(tile_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)s2, (lptr_t)(stage + (wave + TT_WAVES * i) * 1024), 16, 0, 0);
            }
        };
        f32x4_t acc[UPW][2];
        int aoff[UPW], boffb[UPW];
        const int nunits = T * 2;
#pragma unroll
        for (int u2 = 0; u2 < UPW; u2++) {
            acc[u2][0] = acc[u2][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int u = wave + TT_WAVES * u2;
            const int tl = u < nunits ? u / 2 : 0, xt = u % 2;
            const int tsl = plan.type_slots[tl];
            aoff[u2] = ((tsl & 255) * TL + xt * 16) * TT_SLAB + loff;
            boffb[u2] = ((tsl >> 8) * TL) * TT_SLAB + loff;
        }
        auto mma = [&](const char* stage) {
            uint4 fa[UPW], fb[UPW][2];
#pragma unroll
            for (int u2 = 0; u2 < UPW; u2++)
                if (wave + TT_WAVES * u2 < nunits) {
                    fa[u2] = *reinterpret_cast<const uint4*>(stage + aoff[u2]);
                    fb[u2][0] = *reinterpret_cast<const uint4*>(stage + boffb[u2]);
                    fb[u2][1] = *reinterpret_cast<const uint4*>(stage + boffb[u2] + 16 * TT_SLAB);
                }
#pragma unroll
            for (int u2 = 0; u2 < UPW; u2++)
                if (wave + TT_WAVES * u2 < nunits) {
                    mma_t16<E>(acc[u2][0], fa[u2], fb[u2][0]);
                    mma_t16<E>(acc[u2][1], fa[u2], fb[u2][1]);
                }
        };
        char* stages[5] = {st0, st1, st2, st3, st4};
        static_assert(S >= 2 && S <= 5 && 3 * PW <= 63, "ring depth / vmcnt range");
#pragma unroll
        for (int k = 0; k < S - 1; k++)
            if (k < NK) issue(k, stages[k]);
        auto step = [&](int k, const char* rd, char* wr) {
            if (k < NK) {
                const int younger = (NK - 1 - k) < (S - 2) ? (NK - 1 - k) : (S - 2);  // slabs issued after slab k
                if (younger >= 3) wait_vm_t<3 * PW>();
                else if (younger == 2) wait_vm_t<2 * PW>();
                else if (younger == 1) wait_vm_t<PW>();
                else wait_vm_t<0>();
                __builtin_amdgcn_s_barrier();
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
            } else if (S == 4) {
                step(k0, st0, st3);
                step(k0 + 1, st1, st0);
                step(k0 + 2, st2, st1);
                step(k0 + 3, st3, st2);
            } else {
                step(k0, st0, st4);
                step(k0 + 1, st1, st0);
                step(k0 + 2, st2, st1);
                step(k0 + 3, st3, st2);
                step(k0 + 4, st4, st3);
            }
        }
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int r = tid + i * TT_THREADS;
            if (r < NSLOT * TL) {
                snrm[r] = r_nrm[i];
                sinv[r] = r_inv[i];
            }
        }
        // band offsets of the node diagonals this tile and its halo touch: a in [32 s - 2 H, 32 s + 62]
        const int abase = TL * s - 2 * H;
        for (int i = tid; i < 2 * TL + 2 * H; i += TT_THREADS) {
            const int a = abase + i;
            bo_l[i] = (a >= 0 && a < A + 2) ? Lv.boff_out[a] : (1 << 28);  // (no such diagonal: nothing is in its band)
        }
        __syncthreads();
#pragma unroll
        for (int u2 = 0; u2 < UPW; u2++) {
            const int u = wave + TT_WAVES * u2;
            if (u >= nunits) continue;
            const int tl = u / 2, xt = u % 2;
            const int p = tpk[tl] & 255, q = tpk[tl] >> 8;
            const int xsl = (plan.type_slots[tl] & 255) * TL, ysl = (plan.type_slots[tl] >> 8) * TL;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int yloc = 16 * j + lrow;
                const float ny = snrm[ysl + yloc], iy = sinv[ysl + yloc];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int xloc = 16 * xt + 4 * lkg + r;
                    const float sumx = acc[u2][j][r] * sinv[xsl + xloc] * iy;
                    planes[(tl * TL + xloc) * TL + yloc] = cost_formula_t(sumx, p, q, snrm[xsl + xloc], ny);
                }
            }
        }
        stamp(1);
        // ---- phase 2: wait for the neighbours (all three, when they exist, have smaller tickets), then their halo
        if (tid == 0) {
            const int nb[3][2] = {{I - 1, J}, {I, J - 1}, {I - 1, J - 1}};
            for (int e = 0; e < 3; e++) {
                const int ni = nb[e][0], nj = nb[e][1], ns = ni + nj;
                if (ni < 0 || nj < 0) continue;
                const int off = ni - P.t_lo[ns];
                if (off < 0 || off >= P.t_cnt[ns]) continue;  // not a band tile: nothing to wait for
                const int* flag = P.t_flag + P.t_pref[ns] + off;
                // (bounded: a neighbour's ticket is smaller, so its workgroup is running; the bound only turns a
                //  programming error into an error code instead of a hung GPU)
                long spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < (1l << 25)) {
                    __builtin_amdgcn_s_sleep(2);
                    spins++;
                }
                if (spins >= (1l << 25)) *P.status = SVX_ERR_HIP;
            }
        }
        __syncthreads();
        stamp(2);
        // csum tile: position (i + 8, j + 8) <-> node (32 I + i, 32 J + j), i, j in [-H, 32).  Halo nodes come from the
        // neighbours' results; nodes that do not exist or lie outside the band are +inf (a move from them never wins).
        for (int e = tid; e < CS_W * CS_W + 2; e += TT_THREADS) cs[e] = inf;
        __syncthreads();
        {
            // halo positions: H rows above (H x (32 + H)), then H columns to the left (32 x H); <= 3 per thread, their
            // loads in flight together
            const int ntop = H * (TL + H), nhalo = ntop + TL * H;
            unsigned long long hv[3];
            int hp[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int h = tid + u * TT_THREADS;
                hp[u] = -1;
                hv[u] = 0;
                if (h < nhalo) {
                    int ii, jj;
                    if (h < ntop) { ii = h / (TL + H) - H; jj = h % (TL + H) - H; }
                    else { ii = (h - ntop) / H; jj = (h - ntop) % H - H; }
                    const int xx = TL * I + ii, yy = TL * J + jj;
                    if (xx >= 0 && yy >= 0 && xx <= xs && yy <= ys) {
                        const int a = xx + yy, b = yy - bo_l[a - abase];
                        if (b >= 0 && b < B) {
                            // (a neighbour's value: stored sc1 and drained before its flag, so an sc1 load behind the flag
                            //  poll sees it without an agent-scope acquire: MI355X_MICROARCH.md, hand-offs with sc1 loads)
                            hp[u] = (ii + TT_HALO) * CS_W + (jj + TT_HALO);
                            hv[u] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(Lv.csum) + ((size_t)a * B + b),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 3; u++)
                if (hp[u] >= 0) cs[hp[u]] = __longlong_as_double((long long)hv[u]);
        }
        __syncthreads();
        stamp(3);
        // ---- phase 3: the tile's 63 node anti-diagonals, all four waves: thread = (node j = tid >> 3, move slot = tid & 7).
        // A slot relaxes the moves t = slot, slot + 8, ... (kept in registers, unrolled) with every csum / cost read of the
        // diagonal issued up front and no branch; the eight slots of a node sit in eight adjacent lanes and merge by
        // (total, move index) with three DPP exchanges, so the reference's "first strictly smaller candidate" order holds.
        // Predecessors outside the lattice or the band, and the moves a slot does not have, read a +inf cell.
        {
            constexpr int MS = (C::TPP + 2 + 7) / 8;  // moves per slot: 2 for the 10-type shape (12 moves over 8 slots), 3 for the 16-type one
            const int j = tid >> 3, slot = tid & 7;
            int m_off[MS], m_key[MS], m_plb[MS];
            bool m_ok[MS], m_del[MS];
#pragma unroll
            for (int m = 0; m < MS; m++) {
                const int t = slot + 8 * m;
                const int pk = t < NTt ? tpk[t] : 0;
                m_ok[m] = t < NTt;
                m_off[m] = -((pk & 255) * CS_W + (pk >> 8));   // csum position of the predecessor, relative
                m_key[m] = (t << 16) | pk;
                m_del[m] = !(t < T);                           // a deletion costs `pen`
                m_plb[m] = t < T ? t * TL * TL : 0;            // cost plane of the move (any plane for the others: value unused)
            }
            // Loop invariants of this thread's node column j, and the band offsets of the tile's 63 diagonals (a = 32 (I + J) + dd
            // for every node of diagonal dd) one per lane, fetched by v_readlane in the loop: the loop body is instruction-bound
            // (~180 instructions per diagonal with the offsets read from LDS behind a branch and short-circuit merges; ~120 so).
            const int yy = TL * J + j;
            const bool y_in = yy <= ys, y_pos = yy >= 1;
            const int xrel_max = xs - TL * I;                  // node rows of the tile that exist: ic <= xrel_max
            const int a0 = TL * (I + J);
            const int bo_mine = bo_l[a0 + (tid & 63) - abase]; // (lane 63: one past the tile's last diagonal, never selected; inside bo_l)
            const bool edge_tile = I == 0 || J == 0;           // only these tiles hold nodes of row 0 / column 0
            double cconst[MS];
            bool use_cv[MS];
#pragma unroll
            for (int m = 0; m < MS; m++) {
                use_cv[m] = m_ok[m] && !m_del[m];
                cconst[m] = m_ok[m] ? pen : inf;               // a move this slot does not have can never win
                if (!m_ok[m]) m_off[m] = 0;                    // (its csum read is any valid cell)
            }
            for (int dd = 0; dd <= 2 * (TL - 1); dd++) {
                const int i = dd - j;
                const bool inside = (unsigned)i < (unsigned)TL;
                const int ic = inside ? i : 0;
                const int cpos = (ic + TT_HALO) * CS_W + (j + TT_HALO), ppos = ic * TL + j;
                double pv[MS];
                float cv[MS];
#pragma unroll
                for (int m = 0; m < MS; m++) {
                    pv[m] = cs[cpos + m_off[m]];
                    cv[m] = planes[m_plb[m] + ppos];
                    asm volatile("" : "+v"(cv[m]));   // (every read issued here, none sunk into a branch behind its own wait)
                }
                const int bo = __builtin_amdgcn_readlane(bo_mine, dd);
                const int b = yy - bo;
                const bool node = inside & (ic <= xrel_max) & y_in & ((unsigned)b < (unsigned)B);
                const bool general = node & (TL * I + ic >= 1) & y_pos & (a0 + dd - 2 < A);
                // (the slot's first move starts the minimum; "no move reaches this node" is read off the total at the end)
                DpMerge best{pv[0] + (use_cv[0] ? (double)cv[0] : cconst[0]), m_key[0]};
#pragma unroll
                for (int m = 1; m < MS; m++) {
                    const double tot = pv[m] + (use_cv[m] ? (double)cv[m] : cconst[m]);
                    const bool take = tot < best.tot;
                    best.tot = take ? tot : best.tot;
                    best.key = take ? m_key[m] : best.key;
                }
                // merge over the node's eight lanes: lane ^ 1, lane ^ 2 (quad permutes), then the other quad (half-row mirror);
                // (total, move index) order without a short-circuit branch
#define TILE_MERGE(CTRL)                                                                                                      \
    {                                                                                                                         \
        const unsigned long long u_ = __double_as_longlong(best.tot);                                                         \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u_, CTRL, 0xf, 0xf, true);                      \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u_ >> 32), CTRL, 0xf, 0xf, true);              \
        const int ok_ = __builtin_amdgcn_mov_dpp(best.key, CTRL, 0xf, 0xf, true);   /* (every lane has a source: no old value) */ \
        const double ot_ = __longlong_as_double(((unsigned long long)hi_ << 32) | lo_);                                       \
        const bool tk_ = (ot_ < best.tot) | ((ot_ == best.tot) & (ok_ < best.key));                                           \
        best.tot = tk_ ? ot_ : best.tot;                                                                                      \
        best.key = tk_ ? ok_ : best.key;                                                                                      \
    }
                TILE_MERGE(0xB1)   // quad_perm [1,0,3,2]
                TILE_MERGE(0x4E)   // quad_perm [2,3,0,1]
                TILE_MERGE(0x141)  // row_half_mirror: lane k of an 8-lane group <-> lane 7 - k (the other quad)
#undef TILE_MERGE
                const bool won = general & (best.tot < inf);
                double v = won ? best.tot : inf;
                int bpv = won ? (((best.key & 255) << 4) | ((best.key >> 8) & 255)) : 0xFF;
                if (edge_tile) {   // workgroup-uniform
                    const int xx = TL * I + ic;
                    if (node && xx == 0) { v = pen * (double)yy; bpv = (0 << 4) | 1; }
                    else if (node && yy == 0) { v = pen * (double)xx; bpv = (1 << 4) | 0; }
                }
                if (inside && slot == 0) {
                    cs[cpos] = node ? v : inf;
                    bpt[ppos] = (unsigned char)(node ? bpv : 0xFF);
                }
                __syncthreads();   // the diagonal is complete before any wave reads it
            }
        }
        __syncthreads();
        stamp(4);
        // ---- phase 4: results into the node arrays.  What the neighbours read (the last H rows and columns) goes first,
        // sc1 and drained, then the flag; the interior follows with plain stores (only the traceback kernel reads it).
        auto store_nodes = [&](bool edge) {
            for (int e = tid; e < TL * TL; e += TT_THREADS) {
                const int i = e / TL, j2 = e % TL;
                if ((i >= TL - H || j2 >= TL - H) != edge) continue;
                const int xx = TL * I + i, yy = TL * J + j2;
                if (xx > xs || yy > ys) continue;
                const int a = xx + yy, b = yy - bo_l[a - abase];
                if (b < 0 || b >= B) continue;
                const size_t o = (size_t)a * B + b;
                const double v = cs[(i + TT_HALO) * CS_W + (j2 + TT_HALO)];
                if (edge) __hip_atomic_store(reinterpret_cast<unsigned long long*>(Lv.csum) + o, (unsigned long long)__double_as_longlong(v),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else Lv.csum[o] = v;
                Lv.bpk[o] = bpt[e];
            }
        };
        store_nodes(true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(P.t_flag + P.t_pref[s] + (I - P.t_lo[s]), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(5);
        store_nodes(false);
    }
}

bool make_tile_plan(const SvxTypes& ty, int tpp, int nslot_max, TilePlan* plan) {
    memset(plan, 0, sizeof(*plan));
    if (ty.n < 1 || ty.n > tpp) return false;
    int xs[SVX_MAX_TYPES + 2], ys[SVX_MAX_TYPES + 2];
    memset(xs, 0, sizeof(xs));
    memset(ys, 0, sizeof(ys));
    int halo = 1;
    for (int t = 0; t < ty.n; t++) {
        const int lx = ty.x[t] - 1, ly = ty.y[t] - 1;
        if (ty.x[t] > 15 || ty.y[t] > 15) return false;  // packed back-pointers
        if (!xs[lx]) { if (plan->nslot >= nslot_max) return false; plan->slot_info[plan->nslot] = lx; xs[lx] = ++plan->nslot; }
        if (!ys[ly]) { if (plan->nslot >= nslot_max) return false; plan->slot_info[plan->nslot] = (1 << 8) | ly; ys[ly] = ++plan->nslot; }
        plan->type_slots[t] = (xs[lx] - 1) | ((ys[ly] - 1) << 8);
        if (ty.x[t] > halo) halo = ty.x[t];
        if (ty.y[t] > halo) halo = ty.y[t];
    }
    if (halo > TT_HALO) return false;
    plan->nt = ty.n;
    plan->halo = halo;
    return true;
}

}  // namespace

// Can the tile kernel take this type set?  (<= 16 types on <= 12 overlap layers, steps of at most 8 segments.)
bool svxl_band_tiles_ok(const SvxTypes& types) {
    TilePlan plan;
    return make_tile_plan(types, 10, 8, &plan) || make_tile_plan(types, 16, 12, &plan);
}

// b_offset_out, tile ranges (which also clear the pairs' tile flags), ticket table, then the persistent tile sweep.
// gpref [max_nd * n_pairs + 1] and ticket live in the arena.
int svxl_band_tiles_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, const SvxTypes& types, int W, int dtype, int max_nd,
                          int* gpref, int* ticket) {
    if (n_pairs <= 0) return SVX_OK;
    hipStream_t st = ctx->stream;
    TilePlan plan;
    const bool small = make_tile_plan(types, 10, 8, &plan);
    if (!small && !make_tile_plan(types, 16, 12, &plan))
        return svx_fail(ctx, SVX_ERR_ARG, "wide band: %d alignment types / their overlap layers exceed the tile kernel (16 types, 12 layers, steps <= 8)", types.n);
    hipLaunchKernelGGL(k_tile_ranges, dim3(n_pairs), dim3(256), 0, st, pairs, W, 2 * W);
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(1024), 0, st, pairs, n_pairs, max_nd, gpref, ticket);
    SVX_LAUNCH_CHECK(ctx, "k_tile_ranges");
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    // one persistent workgroup per CU (> 100 KB of LDS each): all of them are resident, so a workgroup that waits for
    // a neighbour's flag always waits for a running workgroup
    dim3 grid((unsigned)ncu);
    const char* penv = getenv("SVX_TILE_PROF");
    unsigned long long* prof = nullptr;
    if (penv && atoi(penv) != 0) {
        SVX_HIP(ctx, hipMalloc(&prof, 8 * sizeof(unsigned long long)));
        SVX_HIP(ctx, hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), st));
    }
#define TILES(E)                                                                                                                   \
    do {                                                                                                                           \
        if (small) hipLaunchKernelGGL((k_band_tiles<E, 8, 5, 5>), grid, dim3(TT_THREADS), 0, st, pairs, n_pairs, types, plan, W, max_nd, gpref, ticket, prof); \
        else hipLaunchKernelGGL((k_band_tiles<E, 12, 8, 2>), grid, dim3(TT_THREADS), 0, st, pairs, n_pairs, types, plan, W, max_nd, gpref, ticket, prof); \
    } while (0)
    if (dtype == SVX_F32) TILES(ElemF32);
    else if (dtype == SVX_F16) TILES(ElemF16);
    else TILES(ElemBF16);
#undef TILES
    SVX_LAUNCH_CHECK(ctx, "k_band_tiles");
    if (prof) {
        unsigned long long h[8];
        SVX_HIP(ctx, hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, st));
        SVX_HIP(ctx, hipStreamSynchronize(st));
        (void)hipFree(prof);
        static const char* nm[7] = {"ticket+lookup", "costs", "wait", "halo", "cs-init..dp-start", "store+flag", "interior-store+loop"};
        // (slot 3 = halo fill, slot 4 = the DP sweep; names follow the stamps' positions)
        fprintf(stderr, "[svx tile sweep, summed over %d workgroups, ms]", ncu);
        for (int i = 0; i < 7; i++) fprintf(stderr, " p%d(%s)=%.2f", i, nm[i], (double)h[i] * 1e-5);
        fprintf(stderr, "\n");
    }
    return SVX_OK;
}
