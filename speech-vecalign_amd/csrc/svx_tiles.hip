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

// pair_pref[p] = tickets before pair p; zero the ticket counter
__global__ void k_tile_prefix(const SvxPairDev* __restrict__ pairs, int n_pairs, int* pair_pref, int* ticket) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int run = 0;
    for (int p = 0; p < n_pairs; p++) {
        pair_pref[p] = run;
        run += pairs[p].t_pref[pairs[p].t_nd];
    }
    pair_pref[n_pairs] = run;
    *ticket = 0;
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
                                                           int W, const int* __restrict__ pair_pref, int* ticket) {
    using C = TileCfg<E, NSLOT, UPW, S>;
    using St = typename E::storage;
    constexpr int PW = C::PW;
    __shared__ __attribute__((aligned(1024))) char st0[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st1[C::STAGE];
    __shared__ __attribute__((aligned(1024))) char st2[S >= 3 ? C::STAGE : 16];
    __shared__ __attribute__((aligned(16))) float planes[C::TPP * TL * TL];   // [type][x row][y row] costs of the tile
    __shared__ __attribute__((aligned(16))) double cs[CS_W * CS_W];           // csum of the tile's nodes, halo first
    __shared__ unsigned char bpt[TL * TL];
    __shared__ float snrm[NSLOT * TL], sinv[NSLOT * TL];
    __shared__ int bo_l[2 * TL + 2 * TT_HALO + 2];
    __shared__ int tpk[TT_MAXT + 2];
    __shared__ int sh_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = 2 * W, T = ty.n, NTt = T + 2, H = plan.halo;
    const int total = pair_pref[n_pairs];
    for (int t = tid; t < NTt; t += TT_THREADS) tpk[t] = (int)ty.x[t] | ((int)ty.y[t] << 8);
    const int lrow = lane & 15, lkg = lane >> 4;
    const int loff = lrow * TT_SLAB + 16 * (lkg ^ swz_t(lrow));
    const double inf = __builtin_inf();
    int pcur = 0;  // tickets are handed out in pair order: the pair index only moves forward

    for (;;) {
        __syncthreads();  // the previous tile is finished with LDS
        if (tid == 0) sh_ticket = atomicAdd(ticket, 1);
        __syncthreads();
        const int tk = sh_ticket;
        if (tk >= total) break;
        while (pair_pref[pcur + 1] <= tk) pcur++;
        const SvxPairDev& P = pairs[pcur];
        const SvxLevel& Lv = P.lev[0];
        const int local = tk - pair_pref[pcur];
        // tile anti-diagonal of the ticket: largest s with t_pref[s] <= local
        int lo = 0, hi = P.t_nd - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (P.t_pref[mid] <= local) lo = mid; else hi = mid - 1;
        }
        const int s = lo, I = P.t_lo[s] + (local - P.t_pref[s]), J = s - I;
        const int xs = Lv.n[0], ys = Lv.n[1], d = P.d;
        const int rowbytes = d * (int)sizeof(St);
        const int NK = (rowbytes + TT_SLAB - 1) / TT_SLAB;
        const int X0 = TL * I - 1, Y0 = TL * J - 1;  // cost cell of node (x, y) is (x - 1, y - 1)
        const int A = *Lv.path_len;
        const double pen = *Lv.pen;

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
        char* stages[3] = {st0, st1, st2};
#pragma unroll
        for (int k = 0; k < S - 1; k++)
            if (k < NK) issue(k, stages[k]);
        auto step = [&](int k, const char* rd, char* wr) {
            if (k < NK) {
                const int younger = (NK - 1 - k) < (S - 2) ? (NK - 1 - k) : (S - 2);
                if (younger >= 1) wait_vm_t<PW>();
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
            } else {
                step(k0, st0, st2);
                step(k0 + 1, st1, st0);
                step(k0 + 2, st2, st1);
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
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // csum tile: position (i + H, j + H) <-> node (32 I + i, 32 J + j), i, j in [-H, 32).  Halo nodes come from the
        // neighbours' results; nodes that do not exist or lie outside the band are +inf (a move from them never wins).
        for (int e = tid; e < CS_W * CS_W; e += TT_THREADS) {
            const int ii = e / CS_W - TT_HALO, jj = e % CS_W - TT_HALO;
            double v = inf;
            if ((ii < 0 || jj < 0) && ii >= -H && jj >= -H) {
                const int xx = TL * I + ii, yy = TL * J + jj;
                if (xx >= 0 && yy >= 0 && xx <= xs && yy <= ys) {
                    const int a = xx + yy, b = yy - bo_l[a - abase];
                    if (b >= 0 && b < B) v = Lv.csum[(size_t)a * B + b];
                }
            }
            cs[e] = v;
        }
        __syncthreads();
        // ---- phase 3: the tile's 63 node anti-diagonals, one wave; half-wave h relaxes the moves t = h, h + 2, ...
        if (wave == 0) {
            const int j = lane & 31, half = lane >> 5;
            for (int dd = 0; dd <= 2 * (TL - 1); dd++) {
                const int i = dd - j;
                const bool inside = i >= 0 && i < TL;
                const int xx = TL * I + i, yy = TL * J + j;
                const int a = xx + yy;
                const int b = inside ? yy - bo_l[a - abase] : -1;
                const bool node = inside && xx <= xs && yy <= ys && b >= 0 && b < B;
                const bool general = node && xx >= 1 && yy >= 1 && (a - 2) < A;
                DpMerge best{inf, 0x7fffffff};
                if (general) {
                    for (int t = half; t < NTt; t += 2) {
                        const int xo = tpk[t] & 255, yo = tpk[t] >> 8;
                        if (xo > xx || yo > yy) continue;
                        const double prev = cs[(i - xo + TT_HALO) * CS_W + (j - yo + TT_HALO)];
                        const double c = t < T ? (double)planes[(t * TL + i) * TL + j] : pen;
                        const double tot = prev + c;
                        if (tot < best.tot) { best.tot = tot; best.key = (t << 16) | tpk[t]; }
                    }
                }
                {   // merge the two halves by (total, move index)
                    const unsigned long long u = __double_as_longlong(best.tot);
                    const unsigned ol = xchg32_u32((unsigned)u, lane), oh = xchg32_u32((unsigned)(u >> 32), lane);
                    const double ot = __longlong_as_double(((unsigned long long)oh << 32) | ol);
                    const int ok = (int)xchg32_u32((unsigned)best.key, lane);
                    if (ot < best.tot || (ot == best.tot && ok < best.key)) { best.tot = ot; best.key = ok; }
                }
                double v = best.key != 0x7fffffff ? best.tot : inf;
                int bx = best.key != 0x7fffffff ? (best.key & 255) : -1, by = best.key != 0x7fffffff ? ((best.key >> 8) & 255) : -1;
                if (node && xx == 0) { v = pen * (double)yy; bx = 0; by = 1; }
                else if (node && yy == 0) { v = pen * (double)xx; bx = 1; by = 0; }
                if (inside && half == 0) {
                    cs[(i + TT_HALO) * CS_W + (j + TT_HALO)] = node ? v : inf;
                    bpt[i * TL + j] = (!node || bx < 0) ? (unsigned char)0xFF : (unsigned char)((bx << 4) | by);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        // ---- phase 4: results into the node arrays, then the flag
        for (int e = tid; e < TL * TL; e += TT_THREADS) {
            const int i = e / TL, j2 = e % TL;
            const int xx = TL * I + i, yy = TL * J + j2;
            if (xx > xs || yy > ys) continue;
            const int a = xx + yy, b = yy - bo_l[a - abase];
            if (b < 0 || b >= B) continue;
            const size_t o = (size_t)a * B + b;
            Lv.csum[o] = cs[(i + TT_HALO) * CS_W + (j2 + TT_HALO)];
            Lv.bpk[o] = bpt[e];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(P.t_flag + P.t_pref[s] + (I - P.t_lo[s]), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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

// b_offset_out, tile ranges (which also clear the pairs' tile flags), ticket tables, then the persistent tile sweep.
// pair_pref [n_pairs + 1] and ticket live in the arena.
int svxl_band_tiles_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, const SvxTypes& types, int W, int dtype, int* pair_pref,
                          int* ticket) {
    if (n_pairs <= 0) return SVX_OK;
    hipStream_t st = ctx->stream;
    TilePlan plan;
    const bool small = make_tile_plan(types, 10, 8, &plan);
    if (!small && !make_tile_plan(types, 16, 12, &plan))
        return svx_fail(ctx, SVX_ERR_ARG, "wide band: %d alignment types / their overlap layers exceed the tile kernel (16 types, 12 layers, steps <= 8)", types.n);
    hipLaunchKernelGGL(k_tile_ranges, dim3(n_pairs), dim3(256), 0, st, pairs, W, 2 * W);
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(64), 0, st, pairs, n_pairs, pair_pref, ticket);
    SVX_LAUNCH_CHECK(ctx, "k_tile_ranges");
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    // one persistent workgroup per CU (> 100 KB of LDS each): all of them are resident, so a workgroup that waits for
    // a neighbour's flag always waits for a running workgroup
    dim3 grid((unsigned)ncu);
#define TILES(E)                                                                                                                   \
    do {                                                                                                                           \
        if (small) hipLaunchKernelGGL((k_band_tiles<E, 8, 5, 3>), grid, dim3(TT_THREADS), 0, st, pairs, n_pairs, types, plan, W, pair_pref, ticket); \
        else hipLaunchKernelGGL((k_band_tiles<E, 12, 8, 2>), grid, dim3(TT_THREADS), 0, st, pairs, n_pairs, types, plan, W, pair_pref, ticket); \
    } while (0)
    if (dtype == SVX_F32) TILES(ElemF32);
    else if (dtype == SVX_F16) TILES(ElemF16);
    else TILES(ElemBF16);
#undef TILES
    SVX_LAUNCH_CHECK(ctx, "k_band_tiles");
    return SVX_OK;
}
