// bzx_bwt.hip -- Burrows-Wheeler transform of bzip2 blocks on gfx950.
//
// Contract (reference src/bwt_algorithms/bwt_sort.rs:27-58, bwt_encode): sort all cyclic
// rotations of the block, L[j] = byte preceding the j-th smallest rotation, orig_ptr = row
// of rotation 0.  The reference sorts with a comparator (bwt_sort.rs:61-86) or SA-IS
// (sais_fallback.rs:469-578); neither maps to a GPU.  Here one workgroup (1024 lanes) sorts one
// block, keeping only the rotations that are still tied ("unresolved") from round to round:
//
//   A   bytes in use -> dense symbol ids of `bits` bits; the block is packed once into a bit string P, and every
//       key is one shifted 8-byte window of P (32-bit initial key = 32/bits symbols + the top of the next one,
//       round key = the next csym = 48/bits symbols)
//   I   records [key:32 | i:20 | preceding byte:8], built from the block by the first of four stable
//       8-bit LSD radix passes (per-wave LDS digit masks and counters, LDS-staged coalesced scatter)     (n each)
//   R   re-rank: group boundaries, SA[pos] = sa, L[pos] = preceding byte, compaction of the
//       rotations in groups of size > 1 to records [group start g:20 | .. | sa:20 | prev:8]             (m)
//   TEXT rounds (csym symbols deeper each):
//       groups larger than a tile are split by single symbols (counting sorts inside the group's range);
//       groups that resist are FROZEN for the RANK rounds; then each wave sorts tiles of <= 256 records,
//       cut at group boundaries, by (g, next csym symbols at sa+depth) with a register / cross-lane
//       bitonic network.  No ISA exists in this mode: the only random accesses are 8-byte reads of P.   (m)
//   RANK rounds (only for deep repeats; h doubles): ISA is built once from SA (radix pass + LDS windows),
//       key2 = ISA[(sa+h) mod n]; oversized groups are split by the digits of key2, wave tiles of <= 512
//       records finish the order; R writes ISA where a rank changed.                                    (m)
// Rotations still tied when depth >= n are identical: the block is periodic and
// is flagged BZX_ST_PERIODIC (tie order fixed by bzx_periodic.hip); L is unaffected.
//
// HBM-bound integer work: records stream coalesced as 8-byte words; see DESIGN.md section 6 for the
// measured traffic per phase.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"
#include "bzx_pack.h"

#define SORT_NT 1024
#define SORT_NW (SORT_NT / 64)
#define SORT_E 4
#define SA_MASK 0xFFFFFull
// record layouts (bit positions)
//   INIT : [key32 @28..59][i @8..27][prev @0..7]
//   TEXT : [g @44..63][loc @28..37][sa @8..27][prev @0..7]      group key = rec >> 28
//   RANK : [g @44..63][key2 @24..43][sa @4..23]                  group key = rec >> 24
#define G_SHIFT 44
#define TXT_KEY_SHIFT 28
#define TXT_SA_SHIFT 8
#define RNK_KEY_SHIFT 24
#define RNK_SA_SHIFT 4
#define MODE_INIT 0
#define MODE_TEXT 1
#define MODE_RANK 2
#define TEXT_ROUNDS 4
// TEXT records of a group that the symbol-wise splitter gave up on (deep repeats inside a large group): the
// group is left alone by the tile rounds and resolved by prefix doubling (RANK rounds) afterwards
#define TXT_FROZEN (1ull << 38)
// RANK records: same idea for one round (a large group whose members all have the same key2 cannot be refined
// in this round): bit 0 of the record (bits 0..3 are free)
#define RNK_FROZEN 1ull
// RANK records whose group start was changed by the digit splitter since the last rerank (their ISA entry is stale)
#define RNK_MOVED 2ull

__shared__ uint32_t s_hist[5][256];
__shared__ uint32_t s_base[256];
__shared__ uint32_t s_wcnt[2][SORT_NW * 256];
__shared__ uint32_t s_scratch[2 * SORT_NW];
#define RADIX_STAGED 1
#define RADIX_LDS_MATCH 1
#if RADIX_LDS_MATCH
__shared__ uint64_t s_wmask[SORT_NW][256];      // per wave, per digit: lanes holding that digit (transient)
#endif
#if RADIX_STAGED
__shared__ uint64_t s_stage[SORT_NT * SORT_E];   // one tile of records, grouped by digit, for coalesced write-out
__shared__ uint32_t s_lbase[256];                // tile-local start of every digit's run
__shared__ uint32_t s_gbase[256];                // global start of every digit's run of this tile
__shared__ uint32_t s_wpart[4];
#endif
__shared__ uint8_t s_seq[256];     // byte value -> dense symbol id (order preserving)
__shared__ uint32_t s_inuse[256];
__shared__ uint32_t s_bcast[4];   // [0] block index, [1] carry group start (+1), [2] carry count, [3] largest group

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

// ---- the packed block: see bzx_pack.h -----------------------------------------------------------------------
#define PK_OFFSET 65536u      // P lives in the ISA array (unused in TEXT mode) behind the splitter's scratch lists

__device__ __attribute__((noinline)) void pk_build(const uint8_t *__restrict__ T, uint32_t n, uint32_t bits,
                                                   uint8_t *__restrict__ P)
{
    pk_build_t<SORT_NT>(T, n, bits, P, s_seq);
}

// INIT record of rotation i: [first 32 bits of the packed rotation @28 | i:20 @8 | preceding byte:8 @0]
__device__ __forceinline__ uint64_t init_record(const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, uint32_t n,
                                                uint32_t i, uint32_t bits)
{
    const uint32_t key = (uint32_t)(pk_window(P, i, bits) >> 32);
    const uint32_t prev = T[i ? i - 1 : n - 1];
    return ((uint64_t)key << TXT_KEY_SHIFT) | ((uint64_t)i << TXT_SA_SHIFT) | (uint64_t)prev;
}

// One stable LSD pass: src[0..m) -> dst by the 8-bit digit at `shift`; hist = digit histogram (LDS).
// SRC_TEXT: the source records are generated on the fly from the block bytes (first pass of the initial sort).
// SRC_SA:   the source records are [SA[j]:20 @20 | j:20 @0] generated from the suffix array (ISA build).
#define SRC_REC 0
#define SRC_TEXT 1
#define SRC_SA 2
// Keys of the SORT_E = 4 consecutive rotations i0 .. i0+3 from ONE window of P: rotation i0+e starts e*bits further
// on, and 3*8 + 32 bits still lie inside the >= 57 valid bits of the window.
__device__ __forceinline__ void init_keys4(const uint8_t *__restrict__ P, uint32_t i0, uint32_t bits, uint32_t (&key)[SORT_E])
{
    const uint64_t x = pk_window(P, i0, bits);
#pragma unroll
    for (int e = 0; e < SORT_E; e++) key[e] = (uint32_t)((x << ((uint32_t)e * bits)) >> 32);
}

// INIT records of rotations i0 .. i0+3 (nvalid of them exist): one window of P, one 4-byte read of the block
__device__ __forceinline__ void init_records4(const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, uint32_t n,
                                              uint32_t i0, uint32_t nvalid, uint32_t bits, uint64_t (&out)[SORT_E])
{
#pragma unroll
    for (int e = 0; e < SORT_E; e++) out[e] = 0;
    if (nvalid == 0) return;
    uint32_t key[SORT_E];
    init_keys4(P, i0, bits, key);
    uint32_t pv = 0;                                    // bytes i0-1 .. i0+2 of the block (the preceding bytes)
    if (i0 >= 1 && i0 + 3 <= n) {
        __builtin_memcpy(&pv, T + i0 - 1, 4);
    } else {
        for (uint32_t e = 0; e < nvalid; e++) pv |= (uint32_t)T[i0 + e ? i0 + e - 1 : n - 1] << (8 * e);
    }
#pragma unroll
    for (int e = 0; e < SORT_E; e++)
        if ((uint32_t)e < nvalid)
            out[e] = ((uint64_t)key[e] << TXT_KEY_SHIFT) | ((uint64_t)(i0 + e) << TXT_SA_SHIFT) | (uint64_t)((pv >> (8 * e)) & 255u);
}

// record index of slot e of a lane: SRC_TEXT gives every lane SORT_E consecutive rotations (init_records4); the
// order in which a pass visits equal digits only permutes rotations that tie on the whole key, which is irrelevant
template <int SRC> __device__ __forceinline__ uint32_t radix_index(uint32_t wbase, int e, uint32_t lane)
{
    return SRC == SRC_TEXT ? wbase + lane * SORT_E + (uint32_t)e : wbase + (uint32_t)e * 64 + lane;
}

template <int SRC>
__device__ __forceinline__ uint64_t radix_source(const uint64_t *__restrict__ src, uint32_t idx, uint32_t m,
                                                 const uint8_t *__restrict__ T, uint32_t bits, const uint8_t *__restrict__ P)
{
    if (SRC == SRC_TEXT) return init_record(T, P, m, idx, bits);
    if (SRC == SRC_SA) return ((uint64_t)((const uint32_t *)T)[idx] << 20) | (uint64_t)idx;
    return src[idx];
}

template <int SRC>
__device__ __attribute__((noinline)) void radix_pass(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t m, int shift,
                           const uint32_t *hist, const uint8_t *__restrict__ T = nullptr, uint32_t bits = 8,
                           const uint8_t *__restrict__ P = nullptr)
{
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    uint32_t tot;
    const uint32_t v = tid < 256 ? hist[tid] : 0u;
    const uint32_t ex = bzx_block_excl_sum<SORT_NT>(v, s_scratch, tot);
    if (tid < 256) s_base[tid] = ex;
    for (uint32_t i = tid; i < SORT_NW * 256; i += SORT_NT) s_wcnt[0][i] = 0;
#if RADIX_LDS_MATCH
    for (uint32_t i = tid; i < SORT_NW * 256; i += SORT_NT) (&s_wmask[0][0])[i] = 0;
#endif
    __syncthreads();

    int cur = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // software pipeline: the records of tiles t+1 and t+2 are loaded while tile t is ranked and scattered; the
    // barriers inside the loop order LDS only, so those loads (and the scatter stores) stay in flight.
    uint64_t nxt[SORT_E], nxt2[SORT_E];         // one and two tiles ahead
    {
        const uint32_t wbase = wave * (64 * SORT_E);
        if (SRC == SRC_TEXT) {
            const uint32_t i0 = wbase + lane * SORT_E;
            init_records4(T, P, m, i0 < m ? i0 : 0u, i0 >= m ? 0u : (m - i0 < SORT_E ? m - i0 : (uint32_t)SORT_E), bits, nxt);
            const uint32_t i2 = i0 + SORT_NT * SORT_E;
            init_records4(T, P, m, i2 < m ? i2 : 0u, i2 >= m ? 0u : (m - i2 < SORT_E ? m - i2 : (uint32_t)SORT_E), bits, nxt2);
        } else {
#pragma unroll
            for (int e = 0; e < SORT_E; e++) {
                const uint32_t idx = wbase + e * 64 + lane;
                nxt[e] = idx < m ? radix_source<SRC>(src, idx, m, T, bits, P) : 0ull;
                const uint32_t id2 = idx + SORT_NT * SORT_E;
                nxt2[e] = id2 < m ? radix_source<SRC>(src, id2, m, T, bits, P) : 0ull;
            }
        }
    }
    for (uint32_t t0 = 0; t0 < m; t0 += SORT_NT * SORT_E) {
        const uint32_t wbase = t0 + wave * (64 * SORT_E);
        uint64_t rec[SORT_E];
        uint32_t off[SORT_E], dig[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            rec[e] = nxt[e];
            nxt[e] = nxt2[e];
        }
        {
            const uint32_t nbase = wbase + 2 * SORT_NT * SORT_E;
            if (SRC == SRC_TEXT) {
                const uint32_t i0 = nbase + lane * SORT_E;
                init_records4(T, P, m, i0 < m ? i0 : 0u, i0 >= m ? 0u : (m - i0 < SORT_E ? m - i0 : (uint32_t)SORT_E), bits, nxt2);
            } else {
#pragma unroll
                for (int e = 0; e < SORT_E; e++) {
                    const uint32_t idx = nbase + e * 64 + lane;
                    nxt2[e] = idx < m ? radix_source<SRC>(src, idx, m, T, bits, P) : 0ull;
                }
            }
        }
        uint32_t *wc = &s_wcnt[cur][wave * 256];
#if RADIX_LDS_MATCH
        uint64_t *wm = &s_wmask[wave][0];
#endif
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = radix_index<SRC>(wbase, e, lane);
            const bool valid = idx < m;
            const uint32_t d = (uint32_t)(rec[e] >> shift) & 255u;
#if RADIX_LDS_MATCH
            // peers = lanes of this wave with my digit, collected by OR-ing lane bits into a per-wave, per-digit
            // 64-bit LDS word (LDS executes a wave's instructions in order); the leader clears the word again
            if (valid) atomicOr((unsigned long long *)&wm[d], 1ull << lane);
            bzx_wave_sync();
            const uint64_t peers = valid ? ((volatile uint64_t *)wm)[d] : 0ull;
            uint32_t old = valid ? ((volatile uint32_t *)wc)[d] : 0u;
            bzx_wave_sync();
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            if (valid && (peers & lt_mask) == 0) {
                wc[d] = old + (uint32_t)__popcll(peers);
                wm[d] = 0;
            }
            bzx_wave_sync();
#else
            const uint64_t peers = bzx_match_any(d, 8, valid);
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            const int leader = __ffsll((unsigned long long)peers) - 1;
            uint32_t old = 0;
            if (valid && (int)lane == leader) {
                old = wc[d];
                wc[d] = old + (uint32_t)__popcll(peers);
            }
            old = __shfl(old, leader & 63);
#endif
            off[e] = old + rank;
            dig[e] = d;
        }
        bzx_lds_barrier();
#if RADIX_STAGED
        if (tid < 256) {
            // cross-wave exclusive counts of digit tid inside this tile, tile total, global start
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < SORT_NW; w++) {
                const uint32_t t = s_wcnt[cur][w * 256 + tid];
                s_wcnt[cur][w * 256 + tid] = run;
                run += t;
            }
            s_gbase[tid] = s_base[tid];
            s_base[tid] += run;
            const uint32_t incl = bzx_wave_incl_sum(run);     // exclusive prefix over digits: within my wave ...
            s_lbase[tid] = incl - run;
            if (lane == 63) s_wpart[wave] = incl;             // ... plus the totals of the waves before (added below)
        }
        for (uint32_t i = tid; i < SORT_NW * 256; i += SORT_NT) s_wcnt[cur ^ 1][i] = 0;
        bzx_lds_barrier();
        const uint32_t wp1 = s_wpart[0], wp2 = wp1 + s_wpart[1], wp3 = wp2 + s_wpart[2];
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = radix_index<SRC>(wbase, e, lane);
            if (idx < m) {
                const uint32_t d = dig[e], q = d >> 6;
                const uint32_t lb = s_lbase[d] + (q == 0 ? 0u : q == 1 ? wp1 : q == 2 ? wp2 : wp3);
                s_stage[lb + wc[d] + off[e]] = rec[e];
            }
        }
        bzx_lds_barrier();
        {
            const uint32_t tile_n = (m - t0 < SORT_NT * SORT_E) ? m - t0 : SORT_NT * SORT_E;
#pragma unroll
            for (int e = 0; e < SORT_E; e++) {
                const uint32_t kpos = e * SORT_NT + tid;
                if (kpos < tile_n) {
                    const uint64_t r = s_stage[kpos];
                    const uint32_t d = (uint32_t)(r >> shift) & 255u, q = d >> 6;
                    const uint32_t lb = s_lbase[d] + (q == 0 ? 0u : q == 1 ? wp1 : q == 2 ? wp2 : wp3);
                    dst[s_gbase[d] + (kpos - lb)] = r;
                }
            }
        }
#else
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
        bzx_lds_barrier();
#pragma unroll
        for (int e = 0; e < SORT_E; e++) {
            const uint32_t idx = radix_index<SRC>(wbase, e, lane);
            if (idx < m) dst[wc[dig[e]] + off[e]] = rec[e];
        }
#endif
        cur ^= 1;
    }
    __syncthreads();
}

// Re-rank the sorted records U[0..m): a new group starts wherever the group key (rec >> KEY_SHIFT) changes.
// new rank = SA position of the group's first member; SA[pos] = sa; L[pos] = preceding byte (from the record in
// INIT/TEXT mode, from the block in RANK mode); in RANK mode also ISA[sa] = rank.  Members of groups of size > 1 are compacted to Unew/Snew (slot = SA position).
// Returns the number of unresolved rotations; s_bcast[3] = size of the largest group.
template <int MODE>
__device__ __attribute__((noinline)) uint32_t rerank(const uint64_t *__restrict__ U, const uint32_t *__restrict__ S, uint32_t m,
                           uint64_t *__restrict__ Unew, uint32_t *__restrict__ Snew, uint32_t *__restrict__ ISA,
                           uint32_t *__restrict__ SA, uint8_t *__restrict__ L, uint32_t *__restrict__ orig_out,
                           const uint8_t *__restrict__ T, uint32_t n)
{
    constexpr int KEY_SHIFT = (MODE == MODE_RANK) ? RNK_KEY_SHIFT : TXT_KEY_SHIFT;
    constexpr int SA_SHIFT = (MODE == MODE_RANK) ? RNK_SA_SHIFT : TXT_SA_SHIFT;
    constexpr bool INITIAL = (MODE == MODE_INIT);
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_bcast[3] = 0;
    __syncthreads();
    uint32_t my_maxgrp = 0;
    // carries across tiles (last group start + 1, unresolved so far): every lane keeps them itself from the scan
    // totals, so a tile costs the two barriers of the scan and nothing more
    uint32_t carry_ks = 0, carry_cnt = 0;
    // software pipeline: the records of the next two tiles are loaded while this one is scanned and stored; the
    // barriers of the loop order LDS only
    uint64_t nx[SORT_E + 2], nx2[SORT_E + 2];   // one and two tiles ahead (two tiles of loads in flight per lane)
#pragma unroll
    for (int j = 0; j < SORT_E + 2; j++) {
        const uint32_t kk = tid * SORT_E + j;   // nx[j] holds U[kk - 1]
        nx[j] = (kk >= 1 && kk - 1 < m) ? U[kk - 1] : 0ull;
        const uint32_t k2 = kk + SORT_NT * SORT_E;
        nx2[j] = (k2 - 1 < m) ? U[k2 - 1] : 0ull;
    }
    for (uint32_t t0 = 0; t0 < m; t0 += SORT_NT * SORT_E) {
        const uint32_t k0 = t0 + tid * SORT_E;
        uint64_t r[SORT_E + 2];
#pragma unroll
        for (int j = 0; j < SORT_E + 2; j++) {
            r[j] = nx[j];
            nx[j] = nx2[j];
            const uint32_t kk = k0 + 2 * SORT_NT * SORT_E + j;
            nx2[j] = (kk - 1 < m) ? U[kk - 1] : 0ull;
        }
        bool f[SORT_E + 1];
#pragma unroll
        for (int j = 0; j <= SORT_E; j++) {
            const uint32_t k = k0 + j;
            f[j] = (k >= m) || (k == 0) || ((r[j + 1] >> KEY_SHIFT) != (r[j] >> KEY_SHIFT));
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
        bzx_block_scan_sum_max_lds<SORT_NT>(my_cnt, my_ks, s_scratch, cnt_excl, cnt_total, ks_excl, ks_total);
        uint32_t ks = ks_excl ? ks_excl : carry_ks;
        uint32_t o = carry_cnt + cnt_excl;
        // first re-rank: position = index, so the rows of a lane's four records are adjacent: SA and L take one
        // 16-byte and one 4-byte store instead of four 4-byte and four 1-byte stores
        const bool vec = INITIAL && k0 + SORT_E <= m;
#pragma unroll
        for (int j = 0; j < SORT_E; j++) {
            const uint32_t k = k0 + j;
            if (k < m) {
                if (f[j]) ks = k + 1;
                const uint32_t kstart = ks - 1;
                const uint64_t rec = r[j + 1];
                const bool frozen = (MODE == MODE_TEXT) && (rec & TXT_FROZEN);
                if (!frozen && k - kstart + 1 > my_maxgrp) my_maxgrp = k - kstart + 1;
                const uint32_t newrank = INITIAL ? kstart : S[kstart];
                const uint32_t sa = (uint32_t)(rec >> SA_SHIFT) & 0xFFFFFu;
                const uint32_t pos = INITIAL ? k : S[k];
                if (MODE == MODE_RANK) {
                    // (the suffix array itself is not needed any more: it only seeds the ISA when RANK mode starts;
                    // the rank of a rotation changes only when its group was split)
                    if (newrank != (uint32_t)(rec >> G_SHIFT) || (rec & RNK_MOVED)) ISA[sa] = newrank;
                    // RANK records carry no preceding byte: fetch it once, when the rotation's row is final
                    if (f[j] && f[j + 1]) L[pos] = T[sa ? sa - 1 : n - 1];
                } else if (!vec) {
                    SA[pos] = sa;
                    L[pos] = (uint8_t)rec;
                }
                if (sa == 0) *orig_out = pos;
                if (!(f[j] && f[j + 1])) {
                    Unew[o] = (MODE == MODE_RANK)
                                  ? (((uint64_t)newrank << G_SHIFT) | ((uint64_t)sa << RNK_SA_SHIFT))
                                  : (((uint64_t)newrank << G_SHIFT) | ((uint64_t)sa << TXT_SA_SHIFT) | (rec & 0xFFull) |
                                     (frozen ? TXT_FROZEN : 0ull));
                    Snew[o] = pos;
                    o++;
                }
            }
        }
        if (vec) {
            uint32_t lw = 0;
            uint4 sv;
            sv.x = (uint32_t)(r[1] >> SA_SHIFT) & 0xFFFFFu;
            sv.y = (uint32_t)(r[2] >> SA_SHIFT) & 0xFFFFFu;
            sv.z = (uint32_t)(r[3] >> SA_SHIFT) & 0xFFFFFu;
            sv.w = (uint32_t)(r[4] >> SA_SHIFT) & 0xFFFFFu;
#pragma unroll
            for (int j = 0; j < SORT_E; j++) lw |= ((uint32_t)r[j + 1] & 255u) << (8 * j);
            *reinterpret_cast<uint4 *>(SA + k0) = sv;        // k0 is a multiple of 4, the arrays are 256-byte aligned
            *reinterpret_cast<uint32_t *>(L + k0) = lw;
        }
        if (ks_total) carry_ks = ks_total;
        carry_cnt += cnt_total;
    }
    atomicMax(&s_bcast[3], my_maxgrp);
    __syncthreads();
    return carry_cnt;
}

// ISA[SA[j]] = j for every sorted position j (a permutation inversion).  A direct scatter writes 4 bytes into
// a different cache line per store; instead one stable radix pass groups the pairs (SA[j], j) by SA[j] >> 12
// (every value occurs once, so digit d owns exactly the output range [4096 d, 4096 (d+1)) and no histogram pass
// is needed), then each 8192-entry window of ISA is assembled in LDS and written out coalesced.
__device__ __attribute__((noinline)) void isa_build(const uint32_t *__restrict__ SA, uint32_t *__restrict__ ISA,
                                                    uint64_t *__restrict__ tmp, uint32_t n)
{
    const uint32_t tid = threadIdx.x;
    if (tid < 256) {
        const uint32_t lo = tid << 12;
        s_hist[0][tid] = lo >= n ? 0u : (n - lo < 4096u ? n - lo : 4096u);
    }
    __syncthreads();
    radix_pass<SRC_SA>(nullptr, tmp, n, 32, s_hist[0], (const uint8_t *)SA);
    uint32_t *win = (uint32_t *)s_stage;                  // 8192 ranks
    for (uint32_t w0 = 0; w0 < n; w0 += 8192) {
        const uint32_t cnt = n - w0 < 8192u ? n - w0 : 8192u;
        for (uint32_t t = tid; t < cnt; t += SORT_NT) {
            const uint64_t rec = tmp[w0 + t];
            win[(uint32_t)(rec >> 20) & 8191u] = (uint32_t)rec & 0xFFFFFu;
        }
        __syncthreads();
        for (uint32_t t = tid; t < cnt; t += SORT_NT) ISA[w0 + t] = win[t];
        __syncthreads();
    }
}

// ---- tile rounds: every group fits one wave tile ---------------------------------------------------
// When the largest group has <= SEG_T members a refinement round needs no global radix passes: each
// wave walks its share of the compacted records in tiles of SEG_T (8 per lane) cut at group boundaries,
// fetches the secondary key (TEXT: the next 8 block bytes, big-endian; RANK: ISA[(sa+h) mod n]), sorts
// the tile by (g, key) with a register / cross-lane bitonic network and writes it back in place.
// No LDS, no workgroup barriers; HBM traffic is one 8-byte read + one 8-byte write per record plus the
// key fetch.
#define SEG_T 512            // RANK tiles: 8 records per lane
#define SEG_T_TEXT 256       // TEXT tiles: 4 two-word records per lane (keeps the tile in registers)

struct SegRec {
    uint64_t rec;   // memory record
    uint64_t key;   // secondary key (TEXT: 8 text bytes; RANK: unused, key2 lives inside rec)
};

// Sorts the 64 * SEG_PER_LANE words v[].rec of a wave (lane-major order) ascending: compare-exchange inside a lane for
// strides below SEG_PER_LANE, across lanes (shuffles) above.  Values are distinct except for the all-ones padding.
template <int SEG_PER_LANE> __device__ __forceinline__ void bitonic_tile(SegRec (&v)[SEG_PER_LANE], uint32_t lane)
{
    constexpr uint32_t SEG_TILE = 64 * SEG_PER_LANE;
#pragma unroll
    for (uint32_t k = 2; k <= SEG_TILE; k <<= 1) {
#pragma unroll
        for (uint32_t st = k >> 1; st > 0; st >>= 1) {
            if (st >= SEG_PER_LANE) {
                const uint32_t lst = st / SEG_PER_LANE;          // lane stride
                const bool lower = (lane & lst) == 0;
#pragma unroll
                for (int j = 0; j < SEG_PER_LANE; j++) {
                    const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                    const bool asc = (e & k) == 0;
                    const uint64_t o = __shfl_xor(v[j].rec, (int)lst);
                    const bool keep_min = lower == asc;
                    if ((v[j].rec < o) != keep_min) v[j].rec = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SEG_PER_LANE; j++) {
                    const int pj = j ^ (int)st;
                    if (pj > j) {
                        const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                        const bool asc = (e & k) == 0;
                        if ((v[j].rec < v[pj].rec) != asc) {
                            const uint64_t t = v[j].rec;
                            v[j].rec = v[pj].rec;
                            v[pj].rec = t;
                        }
                    }
                }
            }
        }
    }
}

// first group start at or after r0 (groups have <= 64*SEG_PER_LANE members), or m
template <bool TEXT, int SEG_PER_LANE>
__device__ __forceinline__ uint32_t seg_align(const uint64_t *__restrict__ U, uint32_t m, uint32_t r0, uint32_t lane)
{
    if (r0 == 0) return 0;
    if (r0 >= m) return m;
    if (U[r0] & (TEXT ? TXT_FROZEN : RNK_FROZEN)) return r0;      // a frozen group may be cut anywhere: the round skips it
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < SEG_PER_LANE; j++) {
        const uint32_t i = r0 + lane * SEG_PER_LANE + (uint32_t)j;
        if (i < m) {
            if ((U[i] >> G_SHIFT) != (U[i - 1] >> G_SHIFT) && i < best) best = i;
        } else if (m < best) {
            best = m;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_xor(best, d);
        best = o < best ? o : best;
    }
    return best;
}

// TEXT: T = packed block, key = its next csym symbols at sa+h ; RANK: key2 = ISA[(sa+h) mod n] merged into the record.
template <bool TEXT, int SEG_PER_LANE>
__device__ __attribute__((noinline)) void seg_sort_round(uint64_t *__restrict__ U, uint32_t m, const uint32_t *__restrict__ ISA,
                               const uint8_t *__restrict__ T, uint32_t n, uint32_t h, uint32_t bits = 8,
                               uint32_t csym = 8)
{
    constexpr uint32_t SEG_TILE = 64 * SEG_PER_LANE;
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    const uint32_t share = (m + SORT_NW - 1) / SORT_NW;
    uint32_t a = seg_align<TEXT, SEG_PER_LANE>(U, m, wave * share < m ? wave * share : m, lane);
    const uint32_t end = seg_align<TEXT, SEG_PER_LANE>(U, m, (wave + 1) * share < m ? (wave + 1) * share : m, lane);
    const uint32_t hmod = h % n;
    while (a < end) {
        const uint32_t lim = (end - a < SEG_TILE) ? end : a + SEG_TILE;     // tile may reach at most lim
        SegRec v[SEG_PER_LANE];
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            v[j].rec = i < lim ? U[i] : ~0ull;
            v[j].key = 0;
        }
        constexpr uint64_t FROZEN = TEXT ? TXT_FROZEN : RNK_FROZEN;
        {
            // frozen group at the tile start: skip to its end (the first boundary in the tile) without sorting
            const uint64_t first = __shfl(v[0].rec, 0);
            if (first & FROZEN) {
                const uint64_t after = __shfl_down(v[0].rec, 1);
                uint32_t fb = lim;
#pragma unroll
                for (int j = SEG_PER_LANE - 1; j >= 0; j--) {
                    const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j + 1;
                    const uint64_t nx = (j + 1 < SEG_PER_LANE) ? v[(j + 1) & (SEG_PER_LANE - 1)].rec : after;
                    if (i < lim && (nx >> G_SHIFT) != (v[j].rec >> G_SHIFT)) fb = i;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const uint32_t o = __shfl_xor(fb, d);
                    fb = o < fb ? o : fb;
                }
                a = fb;
                continue;
            }
        }
        // b = last group boundary in (a, lim]  (lim == end is a boundary, unless a frozen group was cut there)
        const bool tail_frozen = lim == end && (U[lim - 1] & FROZEN);
        uint32_t b = lim;
        if (lim != end || tail_frozen) {
            uint64_t after = __shfl_down(v[0].rec, 1);      // first record of the next lane
            if (lane == 63) after = lim < m ? U[lim] : ~0ull;
            uint32_t best = 0;
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j + 1;       // candidate boundary index
                const uint64_t cur = v[j].rec;
                const uint64_t nx = (j + 1 < SEG_PER_LANE) ? v[(j + 1) & (SEG_PER_LANE - 1)].rec : after;
                if ((i < lim || (i == lim && !tail_frozen)) && (nx >> G_SHIFT) != (cur >> G_SHIFT)) best = i;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const uint32_t o = __shfl_xor(best, d);
                best = o > best ? o : best;
            }
            b = best;       // > a: the group starting at a is not frozen, so it has <= SEG_TILE members
            if (b <= a) b = lim;   // defensive: never stall
        }
        // secondary keys for the records of this tile, mask the rest
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            if (i < b) {
                if (TEXT) {
                    uint32_t p = ((uint32_t)(v[j].rec >> TXT_SA_SHIFT) & 0xFFFFFu) + hmod;
                    if (p >= n) p -= n;
                    // next csym symbols of the rotation: one 8-byte read of the packed block
                    const uint64_t key = pk_window(T, p, bits) >> (64u - csym * bits);
                    v[j].key = key;
                    v[j].rec &= ~(0x3FFull << TXT_KEY_SHIFT);     // clear the tile-local slot of the last round
                } else {
                    uint32_t p = ((uint32_t)(v[j].rec >> RNK_SA_SHIFT) & 0xFFFFFu) + hmod;
                    if (p >= n) p -= n;
                    v[j].rec |= (uint64_t)ISA[p] << RNK_KEY_SHIFT;
                }
            } else {
                v[j].rec = ~0ull;
                v[j].key = ~0ull;
            }
        }
        if (TEXT) {
            // ONE 64-bit word per rotation goes through the network,
            //   [tile-local group index:8 | round key:48 | position in the tile:8]
            // (the records are in g order already and a group never spans tiles), i.e. half the cross-lane traffic
            // of sorting (record, key) pairs; the records wait in LDS and are picked up by position afterwards.
            uint64_t *park = s_stage + wave * SEG_TILE;
            uint32_t cnt = 0, fl = 0;
            const uint64_t pg = __shfl_up(v[SEG_PER_LANE - 1].rec, 1) >> G_SHIFT;
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                const uint64_t before = j ? (v[j - 1].rec >> G_SHIFT) : pg;
                const bool nw = !(lane == 0 && j == 0) && (v[j].rec >> G_SHIFT) != before;
                cnt += nw;
                fl |= (uint32_t)nw << j;
            }
            uint32_t lg = bzx_wave_incl_sum(cnt) - cnt;
            SegRec c[SEG_PER_LANE];
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                lg += (fl >> j) & 1u;
                const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                park[e] = v[j].rec;
                c[j].rec = a + e < b ? ((uint64_t)lg << 56) | (v[j].key << 8) | (uint64_t)e : ~0ull;
                c[j].key = 0;
            }
            bitonic_tile<SEG_PER_LANE>(c, lane);
            bzx_wave_sync();
            // tile-local index of the first member of every new (g, key) group -> record bits 28..37
            const uint64_t pc = __shfl_up(c[SEG_PER_LANE - 1].rec, 1) >> 8;
            uint32_t last = 0, nfl = 0;
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                const uint64_t before = j ? (c[j - 1].rec >> 8) : pc;
                const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                if (e == 0 || (c[j].rec >> 8) != before) {
                    last = e + 1;
                    nfl |= 1u << j;
                }
            }
            const uint32_t incl = bzx_wave_incl_max(last);
            uint32_t run = __shfl_up(incl, 1);
            if (lane == 0) run = 0;
#pragma unroll
            for (int j = 0; j < SEG_PER_LANE; j++) {
                const uint32_t e = lane * SEG_PER_LANE + (uint32_t)j;
                if ((nfl >> j) & 1u) run = e + 1;
                const uint64_t rec = a + e < b ? park[(uint32_t)c[j].rec & 255u] : ~0ull;
                v[j].rec = rec | ((uint64_t)((run - 1) & 0x3FFu) << TXT_KEY_SHIFT);
            }
            bzx_wave_sync();
        } else {
            bitonic_tile<SEG_PER_LANE>(v, lane);
        }
#pragma unroll
        for (int j = 0; j < SEG_PER_LANE; j++) {
            const uint32_t i = a + lane * SEG_PER_LANE + (uint32_t)j;
            if (i < b) U[i] = v[j].rec;
        }
        a = b;
    }
    __syncthreads();
}

// ---- large groups in TEXT mode ---------------------------------------------------------------------------
// Real text leaves groups far larger than a wave tile after the initial key (tens of thousands of rotations
// starting "self._" ...).  Instead of falling back to global radix passes, every such group is split by its
// NEXT symbol, one wave per group: per-wave 256-bin histogram and cursors in LDS, a counting sort inside the
// group's own range (order inside a bucket is irrelevant: its members are still tied).  Sub-groups that are
// still larger than a tile are split again by the following symbol.  Splitting only strengthens the round
// invariant (all members of a group agree on at least `depth` symbols), so the tile round that follows is
// unchanged.
#define BIG_MAX 3600         // big groups per pass (lists live in the ISA array, which TEXT mode does not use)
#define BIG_PASSES 64
#define BIG_COOP 4096        // groups above this size are split by the whole workgroup, smaller ones by one wave
__shared__ uint32_t s_big[5];     // [0] groups in the current list, [1] in the next list, [2] overflow flag,
                                  // [3] rotations in the current list's groups, [4] in the next list's groups

template <bool TEXT>
__device__ __attribute__((noinline)) void big_find(const uint64_t *__restrict__ U, const uint32_t *__restrict__ S,
                                                    uint32_t m, uint32_t *__restrict__ list)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_big[0] = 0;
        s_big[1] = 0;
        s_big[2] = 0;
        s_big[3] = 0;
        s_big[4] = 0;
    }
    __syncthreads();
    for (uint32_t k = tid; k < m; k += SORT_NT) {
        const uint64_t rec = U[k];
        const uint32_t g = (uint32_t)(rec >> G_SHIFT);
        const bool is_end = (k + 1 == m) || ((uint32_t)(U[k + 1] >> G_SHIFT) != g);
        if (is_end) {
            const uint32_t size = S[k] - g + 1;        // slots of a group are consecutive SA positions
            if (size > (TEXT ? SEG_T_TEXT : SEG_T) && !(TEXT && (rec & TXT_FROZEN))) {
                const uint32_t idx = atomicAdd(&s_big[0], 1u);
                atomicAdd(&s_big[3], size);
                if (idx < BIG_MAX) {
                    list[2 * idx] = k + 1 - size;
                    list[2 * idx + 1] = size;
                } else {
                    s_big[2] = 1;
                }
            }
        }
    }
    __syncthreads();
}

// digit of a record for the split pass.  TEXT: dense id of the symbol d places into the rotation.  RANK: 8 bits
// (pass 0, 1) or 4 bits (pass 2) of key2 = rank of the rotation h places on, most significant first; key2 is
// fetched once, by the counting sweep of pass 0 (FETCH), and travels in the record's key2 field afterwards.
template <bool TEXT, bool FETCH>
__device__ __forceinline__ uint32_t big_digit(uint64_t &rec, const uint8_t *__restrict__ T,
                                              const uint32_t *__restrict__ ISA, uint32_t n, uint32_t dmod, uint32_t pass)
{
    if (TEXT || FETCH) {
        uint32_t p = ((uint32_t)(rec >> (TEXT ? TXT_SA_SHIFT : RNK_SA_SHIFT)) & 0xFFFFFu) + dmod;
        if (p >= n) p -= n;
        if (TEXT) return (uint32_t)(pk_window(T, p, pass) >> (64u - pass));      // TEXT: T = packed block, pass = bits per symbol
        rec |= (uint64_t)ISA[p] << RNK_KEY_SHIFT;
    }
    const uint32_t r = (uint32_t)(rec >> RNK_KEY_SHIFT) & 0xFFFFFu;
    return pass == 0 ? (r >> 12) & 255u : pass == 1 ? (r >> 4) & 255u : r & 15u;
}

// d: TEXT symbol index / RANK distance h.  pass: TEXT bits per symbol / RANK digit selector.  T: TEXT packed block.
template <bool TEXT>
__device__ __attribute__((noinline)) void big_split_pass(uint64_t *__restrict__ U, uint64_t *__restrict__ Utmp,
                                                          const uint32_t *__restrict__ list, uint32_t nlist,
                                                          uint32_t *__restrict__ next, const uint8_t *__restrict__ T,
                                                          const uint32_t *__restrict__ ISA, uint32_t n, uint32_t d,
                                                          uint32_t pass)
{
    constexpr uint32_t BIG = TEXT ? SEG_T_TEXT : SEG_T;
    constexpr uint64_t LOW = TEXT ? ((1ull << TXT_KEY_SHIFT) - 1ull)                  // sa | prev, tile slot cleared
                                  : ((1ull << G_SHIFT) - 16ull) | RNK_MOVED;           // key2 | sa, frozen flag cleared
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    uint32_t *hist = &s_wcnt[0][wave * 256];     // counts, then bucket bases
    uint32_t *cur = &s_wcnt[1][wave * 256];      // scatter cursors
    const uint32_t dmod = d % n;
    for (uint32_t li = wave; li < nlist; li += SORT_NW) {
        const uint32_t ks = list[2 * li], size = list[2 * li + 1];
        if (size > BIG_COOP) continue;          // handled by the whole workgroup below
        for (uint32_t i = lane; i < 256; i += 64) hist[i] = 0;
        bzx_wave_sync();
        const uint32_t g0 = (uint32_t)(U[ks] >> G_SHIFT);
        for (uint32_t o = lane; o < size; o += 64) {
            uint64_t rec = U[ks + o];
            if (!TEXT && pass == 0) {
                atomicAdd(&hist[big_digit<TEXT, true>(rec, T, ISA, n, dmod, pass)], 1u);
                U[ks + o] = rec;                  // key2 stays in the record for the sweeps that follow
            } else {
                atomicAdd(&hist[big_digit<TEXT, false>(rec, T, ISA, n, dmod, pass)], 1u);
            }
        }
        bzx_wave_sync();
        // exclusive scan of the 256 counts: 4 digits per lane
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[i] = hist[4 * lane + i];
            sum += c[i];
        }
        uint32_t run = bzx_wave_incl_sum(sum) - sum;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hist[4 * lane + i] = run;
            cur[4 * lane + i] = run;
            if (c[i] > BIG) {
                atomicAdd(&s_big[4], c[i]);
                const uint32_t idx = atomicAdd(&s_big[1], 1u);
                if (idx < BIG_MAX) {
                    next[2 * idx] = ks + run;
                    next[2 * idx + 1] = c[i];
                } else {
                    s_big[2] = 1;
                }
            }
            run += c[i];
        }
        bzx_wave_sync();
        for (uint32_t o = lane; o < size; o += 64) {
            uint64_t rec = U[ks + o];
            const uint32_t dg = big_digit<TEXT, false>(rec, T, ISA, n, dmod, pass);
            const uint32_t pos = atomicAdd(&cur[dg], 1u);
            Utmp[ks + pos] = ((uint64_t)(g0 + hist[dg]) << G_SHIFT) | (rec & LOW) | ((!TEXT && hist[dg]) ? RNK_MOVED : 0ull);
        }
        // all lanes' stores must have landed before the wave reads the range back
        bzx_drain_stores();
        bzx_wave_sync();
        for (uint32_t o = lane; o < size; o += 64) U[ks + o] = Utmp[ks + o];
        bzx_wave_sync();
    }
    __syncthreads();
    // very large groups: all lanes of the workgroup on one group at a time (same steps, block barriers)
    {
        const uint32_t tid = threadIdx.x;
        uint32_t *bh = s_hist[0], *bc = s_hist[1];      // counts -> bases, cursors
        for (uint32_t li = 0; li < nlist; li++) {
            const uint32_t ks = list[2 * li], size = list[2 * li + 1];
            if (size <= BIG_COOP) continue;
            if (tid < 256) bh[tid] = 0;
            __syncthreads();
            const uint32_t g0 = (uint32_t)(U[ks] >> G_SHIFT);
            for (uint32_t o = tid; o < size; o += SORT_NT) {
                uint64_t rec = U[ks + o];
                if (!TEXT && pass == 0) {
                    atomicAdd(&bh[big_digit<TEXT, true>(rec, T, ISA, n, dmod, pass)], 1u);
                    U[ks + o] = rec;
                } else {
                    atomicAdd(&bh[big_digit<TEXT, false>(rec, T, ISA, n, dmod, pass)], 1u);
                }
            }
            __syncthreads();
            uint32_t tot;
            const uint32_t cnt = tid < 256 ? bh[tid] : 0u;
            const uint32_t base = bzx_block_excl_sum<SORT_NT>(cnt, s_scratch, tot);
            if (tid < 256) {
                bh[tid] = base;
                bc[tid] = base;
                if (cnt > BIG) {
                    atomicAdd(&s_big[4], cnt);
                    const uint32_t idx = atomicAdd(&s_big[1], 1u);
                    if (idx < BIG_MAX) {
                        next[2 * idx] = ks + base;
                        next[2 * idx + 1] = cnt;
                    } else {
                        s_big[2] = 1;
                    }
                }
            }
            __syncthreads();
            for (uint32_t o = tid; o < size; o += SORT_NT) {
                uint64_t rec = U[ks + o];
                const uint32_t dg = big_digit<TEXT, false>(rec, T, ISA, n, dmod, pass);
                const uint32_t pos = atomicAdd(&bc[dg], 1u);
                Utmp[ks + pos] = ((uint64_t)(g0 + bh[dg]) << G_SHIFT) | (rec & LOW) | ((!TEXT && bh[dg]) ? RNK_MOVED : 0ull);
            }
            __syncthreads();
            for (uint32_t o = tid; o < size; o += SORT_NT) U[ks + o] = Utmp[ks + o];
            __syncthreads();
        }
    }
}

// Splits every group that does not fit a wave tile until it does.
// TEXT: by single symbols (symbol depth, depth+1, ...).  Groups that single symbols separate too slowly (deep
//   repeats: two passes in a row that leave >88 % of the listed rotations in oversized groups, or a work budget of
//   4 n rotation-passes) are FROZEN: flagged in their records, skipped by the tile rounds and resolved by prefix
//   doubling later.  Returns 0 when every group fits a tile now, else the number of symbols (>= depth) that all
//   frozen groups are known to agree on.
// RANK (d = h): by key2 = ISA[(sa + h) mod n], most significant digit first, at most three passes (8 + 8 + 4
//   bits); sub-groups come out ordered by key2 and the tile sort finishes them.  What is still oversized after
//   the last digit has one key2 throughout: this round cannot refine it, it is flagged RNK_FROZEN for the round.
template <bool TEXT>
__device__ __attribute__((noinline)) uint32_t big_split(uint64_t *__restrict__ U, uint64_t *__restrict__ Utmp,
                                                         const uint32_t *__restrict__ S, uint32_t m,
                                                         uint32_t *__restrict__ scratch, const uint8_t *__restrict__ T,
                                                         const uint32_t *__restrict__ ISA, uint32_t n, uint32_t d,
                                                         uint32_t bits = 8)
{
    constexpr uint32_t BIG = TEXT ? SEG_T_TEXT : SEG_T;
    constexpr uint64_t FROZEN = TEXT ? TXT_FROZEN : RNK_FROZEN;
    uint32_t *la = scratch, *lb = scratch + 2 * BIG_MAX;
    big_find<TEXT>(U, S, m, la);
    uint32_t stall = 0, pass = 0;
    uint64_t work = 0;
    for (;; pass++) {
        const uint32_t nlist = s_big[0], ovf = s_big[2], before = s_big[3];
        __syncthreads();
        if (nlist == 0 && !ovf) return 0;
        work += before;
        if (TEXT ? (ovf || pass >= BIG_PASSES || stall >= 2 || work > 4ull * n) : (ovf || pass >= 3)) break;
        big_split_pass<TEXT>(U, Utmp, la, nlist, lb, T, ISA, n, TEXT ? d + pass : d, TEXT ? bits : pass);
        const uint32_t after = s_big[4];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_big[0] = s_big[1] < BIG_MAX ? s_big[1] : BIG_MAX;
            s_big[1] = 0;
            s_big[3] = after;
            s_big[4] = 0;
        }
        __syncthreads();
        stall = ((uint64_t)after * 100 > (uint64_t)before * 88) ? stall + 1 : 0;
        uint32_t *t = la;
        la = lb;
        lb = t;
    }
    // freeze what is still oversized: the member at offset o of a group (slots are consecutive, so o = S[k] - g)
    // belongs to a group of more than BIG rotations iff the record BIG - o places on has the same g
    for (uint32_t k = threadIdx.x; k < m; k += SORT_NT) {
        const uint64_t rec = U[k];
        if (rec & FROZEN) continue;
        const uint32_t g = (uint32_t)(rec >> G_SHIFT), off = S[k] - g;
        bool big = off >= BIG;
        if (!big) {
            const uint32_t j = k + (BIG - off);
            big = j < m && (uint32_t)(U[j] >> G_SHIFT) == g;
        }
        if (big) U[k] = rec | FROZEN;
    }
    __syncthreads();
    return d + pass;        // TEXT: every listed group was split by the symbols depth .. depth+pass-1
}

// Resume mode: the bucket sorter (bzx_bsort.hip) left the block's order so far as SAX[r] = [group start:1 @32 |
// rotation:20] for every rank r.  Builds what the RANK rounds start from: SA[r], and for the ranks in groups of more
// than one the compacted records [g:20 @44 | sa:20 @8] with their slots.  Returns the number of such ranks;
// s_bcast[3] = size of the largest group.
__device__ __attribute__((noinline)) uint32_t resume_load(const uint64_t *__restrict__ SAX, uint32_t n, uint64_t *__restrict__ Unew,
                                                          uint32_t *__restrict__ Snew, uint32_t *__restrict__ SA)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_bcast[3] = 0;
    __syncthreads();
    uint32_t my_maxgrp = 0, carry_ks = 0, carry_cnt = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += SORT_NT * SORT_E) {
        const uint32_t k0 = t0 + tid * SORT_E;
        uint64_t r[SORT_E + 1];
#pragma unroll
        for (int j = 0; j <= SORT_E; j++) r[j] = k0 + j < n ? SAX[k0 + j] : (1ull << 32);
        uint32_t my_ks = 0, my_cnt = 0;
#pragma unroll
        for (int j = 0; j < SORT_E; j++) {
            const uint32_t k = k0 + j;
            if (k < n) {
                const bool f0 = k == 0 || ((r[j] >> 32) & 1u), f1 = (r[j + 1] >> 32) & 1u;
                if (f0) my_ks = k + 1;
                if (!(f0 && f1)) my_cnt++;
            }
        }
        uint32_t cnt_excl, cnt_total, ks_excl, ks_total;
        bzx_block_scan_sum_max_lds<SORT_NT>(my_cnt, my_ks, s_scratch, cnt_excl, cnt_total, ks_excl, ks_total);
        uint32_t ks = ks_excl ? ks_excl : carry_ks;
        uint32_t o = carry_cnt + cnt_excl;
#pragma unroll
        for (int j = 0; j < SORT_E; j++) {
            const uint32_t k = k0 + j;
            if (k < n) {
                const bool f0 = k == 0 || ((r[j] >> 32) & 1u), f1 = (r[j + 1] >> 32) & 1u;
                if (f0) ks = k + 1;
                const uint32_t kstart = ks - 1, sa = (uint32_t)r[j] & 0xFFFFFu;
                if (k - kstart + 1 > my_maxgrp) my_maxgrp = k - kstart + 1;
                SA[k] = sa;
                if (!(f0 && f1)) {
                    Unew[o] = ((uint64_t)kstart << G_SHIFT) | ((uint64_t)sa << TXT_SA_SHIFT);
                    Snew[o] = k;
                    o++;
                }
            }
        }
        if (ks_total) carry_ks = ks_total;
        carry_cnt += cnt_total;
    }
    atomicMax(&s_bcast[3], my_maxgrp);
    __syncthreads();
    return carry_cnt;
}

// diagnostic phase timers (B.dbg != null only in profiling runs): accumulate wall-clock ticks per phase
#ifdef BZX_DIAG
#define PHASE_STAMP(slot)                                                     \
    do {                                                                      \
        if (B.dbg && tid == 0) {                                              \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&B.dbg[slot], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)
#define DBG_STOP(k) (B.dbg_stop == (k))
#define DBG_ON (B.dbg != nullptr)
#define DBG_STOP_ANY (B.dbg_stop != 0)
#else
#define PHASE_STAMP(slot) do {} while (0)
#define DBG_STOP(k) false
#define DBG_ON false
#define DBG_STOP_ANY false
#endif

__device__ __forceinline__ void bwt_body(const BzxBatch &B)
{
    unsigned long long t_last = 0;
    (void)t_last;
    const uint32_t tid = threadIdx.x;
    const BzxSortWs ws = B.sort_ws[blockIdx.x + B.slot_base];

    for (;;) {
        if (tid == 0) s_bcast[0] = atomicAdd(&B.counters[B.ctr_bwt], 1u);
        __syncthreads();
        const uint32_t j_ = s_bcast[0];
        __syncthreads();
        if (j_ >= (B.redo == 2 ? B.counters[BZX_CTR_RESUME] : B.redo ? B.counters[BZX_CTR_REDO] : B.nblk)) break;
        const uint32_t b = B.redo == 2 ? B.resume_list[j_] : B.redo ? B.redo_list[j_] : B.blk_first + j_ * B.blk_step;
        if (B.redo == 2 && !(B.blk[b].status & BZX_ST_RESUME)) continue;      // finished by the bucket sorter's rank rounds
                                                                               // (or by this kernel's early launch)
        if (B.redo == 2 && tid == 0) atomicAdd(&B.counters[BZX_CTR_RESUME_LEFT], 1u);

        const uint32_t n = B.blk[b].n;
        const uint8_t *__restrict__ T = BZX_BLOCK_PTR(B, B.blk[b]);
        uint8_t *__restrict__ L = B.bwt + BZX_SLAB(B, b) * BZX_BLK_STRIDE;
        unsigned long long t_blk0 = 0;
        if (DBG_ON && tid == 0) t_last = t_blk0 = wall_clock64();

        uint64_t *ua = ws.u0, *ub = ws.u1;      // ua: current compacted records, ub: the other buffer
        uint32_t *sa_cur = ws.s0, *sa_alt = ws.s1;
        uint32_t *orig_out = &B.blk[b].orig_ptr;
        uint32_t m = 0, depth = 0, frozen_depth = 0xffffffffu;
        if (B.redo == 2) {
            // resume: the bucket sorter resolved most rotations; the RANK rounds below finish the leftover groups, all of
            // which agree on at least blk.n_mtf symbols (the smallest depth at which a bucket gave up)
            m = resume_load(B.rec_a + BZX_SLAB(B, b) * BZX_MAX_N, n, ua, sa_cur, ws.sa);
            depth = B.blk[b].n_mtf;
            frozen_depth = depth;              // (also marks the group sizes as not measured by a rerank)
        } else {
        // ---- A: bytes in use -> dense symbol ids; symbols per key
        if (tid < 256) s_inuse[tid] = 0;
        __syncthreads();
        {
            const uint32_t n16 = n & ~15u;
            for (uint32_t i = tid * 16; i < n16; i += SORT_NT * 16) {
                uint4 v;
                __builtin_memcpy(&v, T + i, 16);          // unaligned: zero-copy blocks start anywhere in the raw input
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) s_inuse[(w[q] >> (8 * k)) & 255u] = 1;
            }
            for (uint32_t i = n16 + tid; i < n; i += SORT_NT) s_inuse[T[i]] = 1;
        }
        __syncthreads();
        uint32_t n_in_use;
        {
            const uint32_t flag = tid < 256 ? s_inuse[tid] : 0u;
            const uint32_t ex = bzx_block_excl_sum<SORT_NT>(flag, s_scratch, n_in_use);
            if (tid < 256) s_seq[tid] = (uint8_t)ex;
        }
        __syncthreads();
        uint32_t bits = 1;
        while ((1u << bits) < n_in_use) bits++;
        const uint32_t ksym = 32 / bits;                              // whole symbols in the 32-bit initial key
        const uint32_t csym = 48u / bits;                             // symbols in a round key (one window of P; 48 bits
                                                                      // leave room for the tile bookkeeping, see seg_sort_round)
        uint8_t *__restrict__ P = reinterpret_cast<uint8_t *>(ws.isa) + PK_OFFSET;
        pk_build(T, n, bits, P);

        // ---- I1: four digit histograms of the INIT records (the records themselves are built by pass 1).
        // Eight copies of the histograms (chosen by lane) keep lanes with the same digit -- common in the top
        // digits of text -- from serialising on one LDS word; they are summed afterwards.
        uint32_t *sub = &s_wcnt[0][0];                   // [8][4][256], free until the first pass starts
        for (uint32_t i = tid; i < 8 * 4 * 256; i += SORT_NT) sub[i] = 0;
        __syncthreads();
        {
            uint32_t *mine = sub + (tid & 7u) * 1024u;
            for (uint32_t t0 = 0; t0 < n; t0 += SORT_NT * SORT_E) {
                const uint32_t i0 = t0 + tid * SORT_E;          // SORT_E consecutive rotations per lane, one window of P
                if (i0 < n) {
                    uint32_t key[SORT_E];
                    init_keys4(P, i0, bits, key);
#pragma unroll
                    for (int e = 0; e < SORT_E; e++) {
                        if (i0 + e < n) {
#pragma unroll
                            for (int p = 0; p < 4; p++) atomicAdd(&mine[p * 256 + ((key[e] >> (8 * p)) & 255u)], 1u);
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            uint32_t t = 0;                               // tid = digit position * 256 + digit value
#pragma unroll
            for (int c = 0; c < 8; c++) t += sub[c * 1024 + tid];
            (&s_hist[0][0])[tid] = t;
        }
        __syncthreads();
        PHASE_STAMP(0);
        if (DBG_STOP(1)) continue;

        // ---- I2: four LSD passes over the 32-bit key (record bits 28..59)
        radix_pass<SRC_TEXT>(nullptr, ws.u0, n, TXT_KEY_SHIFT, s_hist[0], T, bits, P);
        radix_pass<SRC_REC>(ws.u0, ws.u1, n, TXT_KEY_SHIFT + 8, s_hist[1]);
        radix_pass<SRC_REC>(ws.u1, ws.u0, n, TXT_KEY_SHIFT + 16, s_hist[2]);
        radix_pass<SRC_REC>(ws.u0, ws.u1, n, TXT_KEY_SHIFT + 24, s_hist[3]);
        PHASE_STAMP(1);
        if (DBG_STOP(2)) continue;

        // ---- R: ranks by the first four bytes
        m = rerank<MODE_INIT>(ws.u1, nullptr, n, ua, sa_cur, ws.isa, ws.sa, L, orig_out, T, n);
        PHASE_STAMP(2);
        if (DBG_STOP(3)) continue;

        // ---- TEXT rounds: csym more symbols per round; oversized groups are split by single symbols first
        depth = ksym;
        uint32_t round = 0;
        while (m > 0 && depth < n && round < TEXT_ROUNDS) {
            const uint32_t maxgrp = s_bcast[3];
            __syncthreads();
            if (maxgrp > SEG_T_TEXT) {
                const uint32_t fr = big_split<true>(ua, ub, sa_cur, m, ws.isa, P, nullptr, n, depth, bits);
                PHASE_STAMP(10 + (round < 7 ? round : 7) * 3);
                if (fr && fr < frozen_depth) frozen_depth = fr;           // frozen groups agree on >= fr symbols
            }
            if (DBG_STOP(4)) break;
            seg_sort_round<true, 4>(ua, m, nullptr, P, n, depth, bits, csym);
            PHASE_STAMP(8 + (round < 7 ? round : 7) * 3);
            if (DBG_STOP(5)) break;
            const uint32_t m_before = m;
            m = rerank<MODE_TEXT>(ua, sa_cur, m, ub, sa_alt, ws.isa, ws.sa, L, orig_out, T, n);
            PHASE_STAMP(9 + (round < 7 ? round : 7) * 3);
            uint64_t *tu = ua; ua = ub; ub = tu;
            uint32_t *ts = sa_cur; sa_cur = sa_alt; sa_alt = ts;
            depth += csym;
            round++;
            if (DBG_STOP(6)) break;
            // long repeats: when a round resolves less than 30 % of what it was given, doubling is cheaper
            if (round >= 2 && (uint64_t)m * 10 > (uint64_t)m_before * 7) break;
            if ((uint64_t)m * 10 > (uint64_t)m_before * 9) break;      // (almost) nothing but deep repeats / frozen groups left
        }
        __syncthreads();
        }
        if (DBG_STOP_ANY) continue;

        // ---- RANK rounds (deep repeats): build ISA once, then prefix doubling on ranks
        if (m > 0 && depth < n) {
            // resolved rotations: final rank.  A block that went through the bucket sorter's rank rounds has them in
            // its rank arrays already (the array the block's last round wrote is complete: a rank settled in a round
            // is copied to the other array in the next); a scatter of n ranks by one workgroup took 5 ms.
            const uint32_t *isa_src = nullptr;
            if (B.redo == 2 && B.blk[b].n_selectors < B.rk_blocks) {
                uint32_t r = 0;
                for (uint64_t hh = depth; r < RK_ROUNDS && hh < n; hh *= 3) r++;      // rounds the block took part in
                isa_src = B.isa2 + ((size_t)B.blk[b].n_selectors * 2 + (r & 1u)) * BZX_MAX_N;
            }
            if (isa_src) {
                for (uint32_t k = tid; k < n; k += SORT_NT) ws.isa[k] = isa_src[k] & ~RK_COARSE;
                __syncthreads();
            } else {
                isa_build(ws.sa, ws.isa, ub, n);
            }
            for (uint32_t k = tid; k < m; k += SORT_NT) {                           // unresolved: group start
                const uint64_t rec = ua[k];
                const uint32_t sa = (uint32_t)(rec >> TXT_SA_SHIFT) & 0xFFFFFu;
                const uint32_t g = (uint32_t)(rec >> G_SHIFT);
                ws.isa[sa] = g;
                ua[k] = ((uint64_t)g << G_SHIFT) | ((uint64_t)sa << RNK_SA_SHIFT);
            }
            __syncthreads();
            PHASE_STAMP(5);
            // every unresolved group agrees on at least h symbols (frozen groups stopped at an earlier depth)
            uint32_t h = depth < frozen_depth ? depth : frozen_depth, rround = 0;
            bool unmeasured = frozen_depth != 0xffffffffu;       // the last TEXT rerank did not measure frozen groups
            while (m > 0 && h < n) {
                const uint32_t maxgrp = unmeasured ? 0xFFFFFu : s_bcast[3];
                unmeasured = false;
                __syncthreads();
                if (DBG_ON && tid == 0) {       // diagnostics: unresolved rotations / largest group entering RANK round r
                    atomicAdd(&B.dbg[52 + (rround < 5 ? rround : 5) * 2], (unsigned long long)m);
                    atomicAdd(&B.dbg[53 + (rround < 5 ? rround : 5) * 2], (unsigned long long)maxgrp);
                }
                rround++;
                // groups larger than a tile are first split by the digits of key2 (counting sorts inside the
                // group's own range; the scratch lists live in the slot map that the next rerank overwrites)
                if (maxgrp > SEG_T) {
                    (void)big_split<false>(ua, ub, sa_cur, m, sa_alt, T, ws.isa, n, h % n);
                    PHASE_STAMP(7);
                }
                seg_sort_round<false, 8>(ua, m, ws.isa, T, n, h);
                PHASE_STAMP(6);
                m = rerank<MODE_RANK>(ua, sa_cur, m, ub, sa_alt, ws.isa, ws.sa, L, orig_out, T, n);
                {
                    uint64_t *tu = ua; ua = ub; ub = tu;
                }
                uint32_t *ts = sa_cur; sa_cur = sa_alt; sa_alt = ts;
                h <<= 1;
                PHASE_STAMP(4);
            }
            // rotations still tied (periodic block): their rows of L are tie-invariant, write them now
            for (uint32_t k = tid; k < m; k += SORT_NT) {
                const uint32_t sa = (uint32_t)(ua[k] >> RNK_SA_SHIFT) & 0xFFFFFu;
                L[sa_cur[k]] = T[sa ? sa - 1 : n - 1];
            }
            PHASE_STAMP(4);
        }
        PHASE_STAMP(3);
        if (DBG_ON && tid == 0) B.blk[b].pad_[1] = (uint32_t)((wall_clock64() - t_blk0) / 100);   // microseconds in this kernel
        if (tid == 0) {
            // (a block sorted from scratch keeps BZX_ST_REDO: the bucket sort kernels, possibly still running beside
            // this launch, skip whatever buckets the split kernel had emitted for it before it refused the block)
            B.blk[b].status = ((m > 0) ? BZX_ST_PERIODIC : 0u) | (B.redo == 1 ? BZX_ST_REDO : 0u);
            if (m > 0) {
                B.blk[b].pad_[0] = s_bcast[3];                       // copies of every rotation (k of u^k)
                B.plist[atomicAdd(&B.counters[5], 1u)] = b;          // tie order fixed up by bzx_periodic.hip
            }
        }
        __syncthreads();
        if (B.redo_once == 1) break;
    }
}

// (two names for one body: the early launches beside the bucket sorter wait for compute units while it runs, and a
// profile would book that wait as time of the sorter proper)
__global__ __launch_bounds__(SORT_NT) void bzx_bwt_kernel(BzxBatch B) { bwt_body(B); }
__global__ __launch_bounds__(SORT_NT) void bzx_bwt_side_kernel(BzxBatch B) { bwt_body(B); }

void bzx_launch_bwt(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    if (B.redo_once) hipLaunchKernelGGL(bzx_bwt_side_kernel, dim3(grid), dim3(SORT_NT), 0, stream, B);
    else hipLaunchKernelGGL(bzx_bwt_kernel, dim3(grid), dim3(SORT_NT), 0, stream, B);
}

uint32_t bzx_bwt_max_blocks_per_cu()
{
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bzx_bwt_kernel, SORT_NT, 0) != hipSuccess || nb < 1) nb = 1;
    return nb > 2 ? 2u : (uint32_t)nb;
}
