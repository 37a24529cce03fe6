// bzx_bwt.hip -- Burrows-Wheeler transform of bzip2 blocks on gfx950.
//
// Contract (reference src/bwt_algorithms/bwt_sort.rs:27-58, bwt_encode): sort all cyclic
// rotations of the block, L[j] = byte preceding the j-th smallest rotation, orig_ptr = row
// of rotation 0.  The reference sorts with a comparator (bwt_sort.rs:61-86) or SA-IS
// (sais_fallback.rs:469-578); neither maps to a GPU.  Here one workgroup sorts one block by
// cyclic PREFIX DOUBLING on 64-bit records
//        [ group start g : 20 | rank of rotation sa+h : 20 | sa : 20 ]
// with an LSD radix sort (8-bit digits, wave64 match-any ranking, LDS digit offsets) over
// the still-unresolved rotations only:
//   I1  build records [first 4 bytes : 32 | i : 20], 4 digit histograms          (n)
//   I2  4 radix passes                                                            (n each)
//   R   re-rank: group boundaries, new ranks -> ISA, positions -> SA, compact the
//       unresolved rotations                                                      (m)
//   loop h = 4, 8, 16, ... while unresolved and h < n:
//       G  gather key2 = ISA[(sa+h) mod n], 5 digit histograms                    (m)
//       5 radix passes over (g, key2)                                             (m each)
//       R  re-rank + compact                                                      (m)
//   F   L[j] = T[SA[j]-1], orig_ptr
// Rotations still tied when h >= n are identical: the block is periodic and
// is flagged BZX_ST_PERIODIC; L is unaffected by the tie order.
//
// HBM-bound integer work: all traffic is 8-byte records streamed coalesced, plus one 4-byte
// random read (G) and one 4-byte random write (R) per unresolved rotation and round.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define SORT_NT 1024
#define SORT_NW (SORT_NT / 64)
#define SORT_E 4
#define SA_MASK 0xFFFFFull

__shared__ uint32_t s_hist[5][256];
__shared__ uint32_t s_base[256];
__shared__ uint32_t s_wcnt[2][SORT_NW * 256];
__shared__ uint32_t s_scratch[2 * SORT_NW];
__shared__ uint32_t s_bcast[4];   // [0] block index, [1] carry group start (+1), [2] carry count, [3] orig_ptr

// Add one to hist[d]; when the whole wave holds the same digit a single lane adds the count.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t d, bool valid)
{
    const uint64_t act = __ballot(valid);
    if (act == 0) return;
    const int first = __ffsll((unsigned long long)act) - 1;
    const uint32_t d0 = __shfl(d, first);
    const uint64_t same = __ballot(valid && d == d0);
    if (same == act) {
        if ((int)bzx_lane() == first) atomicAdd(&hist[d0], (uint32_t)__popcll(act));
    } else if (valid) {
        atomicAdd(&hist[d], 1u);
    }
}

// One stable LSD pass: src[0..m) -> dst by the 8-bit digit at `shift`; hist = digit histogram (LDS).
__device__ void radix_pass(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t m, int shift,
                           const uint32_t *hist)
{
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    uint32_t tot;
    const uint32_t v = tid < 256 ? hist[tid] : 0u;
    const uint32_t ex = bzx_block_excl_sum<SORT_NT>(v, s_scratch, tot);
    if (tid < 256) s_base[tid] = ex;
    for (uint32_t i = tid; i < SORT_NW * 256; i += SORT_NT) s_wcnt[0][i] = 0;
    __syncthreads();

    int cur = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t t0 = 0; t0 < m; t0 += SORT_NT * SORT_E) {
        const uint32_t wbase = t0 + wave * (64 * SORT_E);
        uint64_t rec[SORT_E];
        uint32_t off[SORT_E], dig[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = wbase + e * 64 + lane;
            rec[e] = idx < m ? src[idx] : 0ull;
        }
        uint32_t *wc = &s_wcnt[cur][wave * 256];
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = wbase + e * 64 + lane;
            const bool valid = idx < m;
            const uint32_t d = (uint32_t)(rec[e] >> shift) & 255u;
            const uint64_t peers = bzx_match_any(d, 8, valid);
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            const int leader = __ffsll((unsigned long long)peers) - 1;
            uint32_t old = 0;
            if (valid && (int)lane == leader) {
                old = wc[d];
                wc[d] = old + (uint32_t)__popcll(peers);
            }
            old = __shfl(old, leader & 63);
            off[e] = old + rank;
            dig[e] = d;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = s_base[tid];
#pragma unroll
            for (int w = 0; w < SORT_NW; w++) {
                const uint32_t t = s_wcnt[cur][w * 256 + tid];
                s_wcnt[cur][w * 256 + tid] = run;
                run += t;
            }
            s_base[tid] = run;
        }
        for (uint32_t i = tid; i < SORT_NW * 256; i += SORT_NT) s_wcnt[cur ^ 1][i] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = wbase + e * 64 + lane;
            if (idx < m) dst[wc[dig[e]] + off[e]] = rec[e];
        }
        cur ^= 1;
    }
    __syncthreads();
}

// Re-rank the sorted records U[0..m): a new group starts wherever (rec >> 20) changes.
// new rank = SA position of the group's first member; ISA[sa] = rank; SA[pos] = sa; the members of
// groups of size > 1 are compacted to Unew/Snew (record = rank << 40 | sa, slot = SA position).
// Returns the number of unresolved rotations.
template <bool INITIAL>
__device__ uint32_t rerank(const uint64_t *__restrict__ U, const uint32_t *__restrict__ S, uint32_t m,
                           uint64_t *__restrict__ Unew, uint32_t *__restrict__ Snew, uint32_t *__restrict__ ISA,
                           uint32_t *__restrict__ SA)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_bcast[1] = 0;
        s_bcast[2] = 0;
        s_bcast[3] = 0;
    }
    __syncthreads();
    uint32_t my_maxgrp = 0;
    for (uint32_t t0 = 0; t0 < m; t0 += SORT_NT * SORT_E) {
        const uint32_t k0 = t0 + tid * SORT_E;
        const uint32_t carry_ks = s_bcast[1], carry_cnt = s_bcast[2];
        uint64_t r[SORT_E + 2];
#pragma unroll
        for (int j = 0; j < SORT_E + 2; j++) {
            const uint32_t kk = k0 + j;   // r[j] holds U[kk - 1]
            r[j] = (kk >= 1 && kk - 1 < m) ? U[kk - 1] : 0ull;
        }
        bool f[SORT_E + 1];
#pragma unroll
        for (int j = 0; j <= SORT_E; j++) {
            const uint32_t k = k0 + j;
            f[j] = (k >= m) || (k == 0) || ((r[j + 1] >> 20) != (r[j] >> 20));
        }
        uint32_t my_ks = 0, my_cnt = 0;
#pragma unroll
        for (int j = 0; j < SORT_E; j++) {
            const uint32_t k = k0 + j;
            if (k < m) {
                if (f[j]) my_ks = k + 1;
                if (!(f[j] && f[j + 1])) my_cnt++;
            }
        }
        uint32_t cnt_excl, cnt_total, ks_excl, ks_total;
        bzx_block_scan_sum_max<SORT_NT>(my_cnt, my_ks, s_scratch, cnt_excl, cnt_total, ks_excl, ks_total);
        uint32_t ks = ks_excl ? ks_excl : carry_ks;
        uint32_t o = carry_cnt + cnt_excl;
#pragma unroll
        for (int j = 0; j < SORT_E; j++) {
            const uint32_t k = k0 + j;
            if (k < m) {
                if (f[j]) ks = k + 1;
                const uint32_t kstart = ks - 1;
                if (k - kstart + 1 > my_maxgrp) my_maxgrp = k - kstart + 1;
                const uint32_t newrank = INITIAL ? kstart : S[kstart];
                const uint32_t sa = (uint32_t)(r[j + 1] & SA_MASK);
                const uint32_t pos = INITIAL ? k : S[k];
                ISA[sa] = newrank;
                SA[pos] = sa;
                if (!(f[j] && f[j + 1])) {
                    Unew[o] = ((uint64_t)newrank << 40) | (uint64_t)sa;
                    Snew[o] = pos;
                    o++;
                }
            }
        }
        if (tid == 0) {
            if (ks_total) s_bcast[1] = ks_total;
            s_bcast[2] = carry_cnt + cnt_total;
        }
        __syncthreads();
    }
    atomicMax(&s_bcast[3], my_maxgrp);
    __syncthreads();
    const uint32_t res = s_bcast[2];
    __syncthreads();
    return res;
}

// key2 = rank of the rotation h positions further on; fills the five digit histograms.
__device__ void gather_keys(uint64_t *__restrict__ U, uint32_t m, const uint32_t *__restrict__ ISA, uint32_t n,
                            uint32_t h)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 5 * 256; i += SORT_NT) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < m; t0 += SORT_NT * SORT_E) {
        uint64_t rec[SORT_E];
        uint32_t key2[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t k = t0 + e * SORT_NT + tid;
            rec[e] = k < m ? U[k] : 0ull;
        }
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t k = t0 + e * SORT_NT + tid;
            uint32_t p = (uint32_t)(rec[e] & SA_MASK) + h;
            if (p >= n) p -= n;
            key2[e] = k < m ? ISA[p] : 0u;
        }
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t k = t0 + e * SORT_NT + tid;
            const bool valid = k < m;
            const uint64_t x = rec[e] | ((uint64_t)key2[e] << 20);
            if (valid) U[k] = x;
#pragma unroll
            for (int p = 0; p < 5; p++) hist_add(s_hist[p], (uint32_t)(x >> (20 + 8 * p)) & 255u, valid);
        }
    }
    __syncthreads();
}

// ---- fast refinement round: every group fits one wave tile --------------------------------------
// When the largest group has <= SEG_T members, a doubling round needs no global radix passes: each
// wave walks its share of the compacted records in tiles of SEG_T (8 per lane) cut at group
// boundaries, gathers key2 = ISA[(sa+h) mod n], sorts the tile by (g, key2) with a register/cross-lane
// bitonic network and writes it back in place.  No LDS, no workgroup barriers; HBM traffic is one
// 8-byte read + one 8-byte write per record plus the 4-byte gather.
#define SEG_T 512
#define SEG_PER_LANE 8

__device__ __forceinline__ void cmpx(uint64_t &a, uint64_t &b, bool asc)
{
    const bool sw = (a > b) == asc;
    const uint64_t t = a;
    a = sw ? b : a;
    b = sw ? t : b;
}

__device__ __forceinline__ void bitonic512(uint64_t (&v)[SEG_PER_LANE], uint32_t lane)
{
#pragma unroll
    for (uint32_t k = 2; k <= SEG_T; k <<= 1) {
#pragma unroll
        for (uint32_t st = k >> 1; st > 0; st >>= 1) {
            if (st >= SEG_PER_LANE) {
                const uint32_t lst = st / SEG_PER_LANE;          // lane stride
                const bool lower = (lane & lst) == 0;
#pragma unroll
                for (int j = 0; j < SEG_PER_LANE; j++) {
                    const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                    const bool asc = (e & k) == 0;
                    const uint64_t o = __shfl_xor(v[j], (int)lst);
                    const bool keep_min = lower == asc;
                    v[j] = keep_min ? (v[j] < o ? v[j] : o) : (v[j] > o ? v[j] : o);
                }
            } else {
#pragma unroll
                for (int j = 0; j < SEG_PER_LANE; j++) {
                    const int pj = j ^ (int)st;
                    if (pj > j) {
                        const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                        cmpx(v[j], v[pj], (e & k) == 0);
                    }
                }
            }
        }
    }
}

// first group start at or after r0 (groups have <= SEG_T members), or m
__device__ __forceinline__ uint32_t seg_align(const uint64_t *__restrict__ U, uint32_t m, uint32_t r0, uint32_t lane)
{
    if (r0 == 0) return 0;
    if (r0 >= m) return m;
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < SEG_PER_LANE; j++) {
        const uint32_t i = r0 + lane * SEG_PER_LANE + (uint32_t)j;
        if (i < m) {
            if ((U[i] >> 40) != (U[i - 1] >> 40) && i < best) best = i;
        } else if (m < best) {
            best = m;
        }
    }
    // wave minimum
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_xor(best, d);
        best = o < best ? o : best;
    }
    return best;
}

__device__ void seg_sort_round(uint64_t *__restrict__ U, uint32_t m, const uint32_t *__restrict__ ISA, uint32_t n,
                               uint32_t h)
{
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    const uint32_t share = (m + SORT_NW - 1) / SORT_NW;
    uint32_t a = seg_align(U, m, wave * share < m ? wave * share : m, lane);
    const uint32_t end = seg_align(U, m, (wave + 1) * share < m ? (wave + 1) * share : m, lane);
    while (a < end) {
        const uint32_t lim = (end - a < SEG_T) ? end : a + SEG_T;     // tile may reach at most lim
        uint64_t v[SEG_PER_LANE];
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            v[j] = i < lim ? U[i] : ~0ull;
        }
        // b = last group boundary in (a, lim]  (lim == end is a boundary)
        uint32_t b = lim;
        if (lim != end) {
            const uint64_t nxt0 = __shfl_down(v[0], 1);     // first record of the next lane
            uint64_t after = nxt0;
            if (lane == 63) after = U[lim];                 // lim < end <= m
            uint32_t best = 0;
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j + 1;       // candidate boundary index
                const uint64_t cur = v[j];
                const uint64_t nx = (j + 1 < SEG_PER_LANE) ? v[(j + 1) & (SEG_PER_LANE - 1)] : after;
                if (i <= lim && (nx >> 40) != (cur >> 40)) best = i;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const uint32_t o = __shfl_xor(best, d);
                best = o > best ? o : best;
            }
            b = best;       // > a because the group starting at a has <= SEG_T members
            if (b <= a) b = lim;   // defensive: never stall (cannot happen when the size bound holds)
        }
        // gather key2 for the records of this tile, mask the rest
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            if (i < b) {
                uint32_t p = (uint32_t)(v[j] & SA_MASK) + h;
                if (p >= n) p -= n;
                v[j] |= (uint64_t)ISA[p] << 20;
            } else {
                v[j] = ~0ull;
            }
        }
        bitonic512(v, lane);
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            if (i < b) U[i] = v[j];
        }
        a = b;
    }
    __syncthreads();
}

// diagnostic phase timers (B.dbg != null only in profiling runs): accumulate wall-clock ticks per phase
#define PHASE_STAMP(slot)                                                     \
    do {                                                                      \
        if (B.dbg && tid == 0) {                                              \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&B.dbg[slot], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)

__global__ __launch_bounds__(SORT_NT) void bzx_bwt_kernel(BzxBatch B)
{
    unsigned long long t_last = 0;
    const uint32_t tid = threadIdx.x;
    const BzxSortWs ws = B.sort_ws[blockIdx.x];

    for (;;) {
        if (tid == 0) s_bcast[0] = atomicAdd(&B.counters[0], 1u);
        __syncthreads();
        const uint32_t j_ = s_bcast[0];
        __syncthreads();
        if (j_ >= B.nblk) break;
        const uint32_t b = B.blk_first + j_ * B.blk_step;

        const uint32_t n = B.blk[b].n;
        const uint8_t *__restrict__ T = B.in + B.blk[b].in_off;
        uint8_t *__restrict__ L = B.bwt + (size_t)b * BZX_BLK_STRIDE;
        if (B.dbg && tid == 0) t_last = wall_clock64();

        // ---- I1: records [first 4 bytes | i] and their four digit histograms
        for (uint32_t i = tid; i < 4 * 256; i += SORT_NT) (&s_hist[0][0])[i] = 0;
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n; t0 += SORT_NT) {
            const uint32_t i = t0 + tid;
            const bool valid = i < n;
            uint32_t key = 0;
            if (valid) {
                uint32_t p = i;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    key = (key << 8) | T[p];
                    p++;
                    if (p >= n) p = 0;
                }
                ws.u0[i] = ((uint64_t)key << 20) | (uint64_t)i;
            }
#pragma unroll
            for (int p = 0; p < 4; p++) hist_add(s_hist[p], (key >> (8 * p)) & 255u, valid);
        }
        __syncthreads();

        PHASE_STAMP(0);
        // ---- I2: four LSD passes over the 32-bit key (record bits 20..51)
        radix_pass(ws.u0, ws.u1, n, 20, s_hist[0]);
        radix_pass(ws.u1, ws.u0, n, 28, s_hist[1]);
        radix_pass(ws.u0, ws.u1, n, 36, s_hist[2]);
        radix_pass(ws.u1, ws.u0, n, 44, s_hist[3]);

        PHASE_STAMP(1);
        // ---- R: ranks by the first four bytes
        uint64_t *ua = ws.u1, *ub = ws.u0;      // ua: current compacted records, ub: sort scratch
        uint32_t *sa_cur = ws.s0, *sa_alt = ws.s1;
        uint32_t m = rerank<true>(ws.u0, nullptr, n, ua, sa_cur, ws.isa, ws.sa);

        PHASE_STAMP(2);
        // ---- doubling rounds
        uint32_t h = 4;
        uint32_t round = 0;
        while (m > 0 && h < n) {
            const uint32_t maxgrp = s_bcast[3];
            __syncthreads();
            if (maxgrp <= SEG_T) {
                // fast round: in-place wave-tile sort of ua, then re-rank into ub
                seg_sort_round(ua, m, ws.isa, n, h);
                PHASE_STAMP(8 + (round < 7 ? round : 7) * 3);
                m = rerank<false>(ua, sa_cur, m, ub, sa_alt, ws.isa, ws.sa);
                PHASE_STAMP(9 + (round < 7 ? round : 7) * 3);
                uint64_t *tu = ua;
                ua = ub;
                ub = tu;
            } else {
                gather_keys(ua, m, ws.isa, n, h);
                // five passes: ua -> ub -> ua -> ub -> ua -> ub
                radix_pass(ua, ub, m, 20, s_hist[0]);
                radix_pass(ub, ua, m, 28, s_hist[1]);
                radix_pass(ua, ub, m, 36, s_hist[2]);
                radix_pass(ub, ua, m, 44, s_hist[3]);
                radix_pass(ua, ub, m, 52, s_hist[4]);
                PHASE_STAMP(10 + (round < 7 ? round : 7) * 3);
                m = rerank<false>(ub, sa_cur, m, ua, sa_alt, ws.isa, ws.sa);
                PHASE_STAMP(9 + (round < 7 ? round : 7) * 3);
            }
            uint32_t *ts = sa_cur;
            sa_cur = sa_alt;
            sa_alt = ts;
            h <<= 1;
            round++;
        }

        // ---- F: last column and orig_ptr
        for (uint32_t j = tid; j < n; j += SORT_NT) {
            const uint32_t sa = ws.sa[j];
            L[j] = T[sa ? sa - 1 : n - 1];
            if (sa == 0) B.blk[b].orig_ptr = j;
        }
        PHASE_STAMP(3);
        if (tid == 0) {
            B.blk[b].status = (m > 0) ? BZX_ST_PERIODIC : 0u;
            if (m > 0) B.plist[atomicAdd(&B.counters[5], 1u)] = b;   // tie order fixed up by bzx_periodic.hip
        }
        __syncthreads();
    }
}

void bzx_launch_bwt(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_bwt_kernel, dim3(grid), dim3(SORT_NT), 0, stream, B);
}

uint32_t bzx_bwt_max_blocks_per_cu()
{
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bzx_bwt_kernel, SORT_NT, 0) != hipSuccess || nb < 1) nb = 1;
    return nb > 2 ? 2u : (uint32_t)nb;
}
