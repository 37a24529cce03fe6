// bzx_mtf.hip -- move-to-front + zero-run (RUNA/RUNB) coding of the BWT last column on gfx950.
//
// Contract (reference src/tools/rle2_mtf.rs:23-177, rle2_mtf_encode): bytes in use -> dense
// symbol ids; MTF rank of every byte; runs of rank 0 -> bijective base-2 digits RUNA(0)/RUNB(1);
// other ranks -> rank+1; EOB = nInUse+1 appended; histogram of the EMITTED symbols incl. EOB
// (libbz2 counting; the reference's freqs[] is mis-indexed).
//
// MTF is a serial recurrence over the block -- but only the HEADS of its runs take part in it: a byte equal to its
// predecessor has rank 0 and leaves the list as it is (the last column of text is mostly runs: 45 % of its bytes are
// heads).  So the block is first compacted to its run heads (byte + position, one parallel pass that also marks the
// bytes in use); the recurrence runs over the heads, cut into up to 1024 chunks (one lane each): the MTF list at a
// chunk start is "symbols by most recent occurrence before the chunk", which each lane rebuilds from the per-chunk
// recency lists of the chunks before it; then every lane runs the plain MTF over its own heads.  Zero-run coding is a
// parallel pass over the heads: a head emits its rank (+1) and the RUNA/RUNB digits of the zeros behind it, whose
// number is the distance to the next head (sum-scan for offsets).
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define MTF_NT 1024
#define MTF_LIST_BYTES (72 * 1024)
#define MTF_E 16

// one pool: per-chunk recency lists (most recent first), then the per-chunk working MTF lists (+ read slack of the
// 4-word walk); the zero-run pass parks a tile's symbols in all of it (up to ~63,000 of them: 8,192 heads of runs)
__shared__ __attribute__((aligned(16))) uint8_t m_pool[2 * MTF_LIST_BYTES + 32];
#define m_rec m_pool
#define m_list (m_pool + MTF_LIST_BYTES)
__shared__ uint16_t m_reccnt[MTF_NT];
#define MTF_GROUP 64                                   // chunks per group of the start lists: the chunks of one wave
__shared__ __attribute__((aligned(8))) uint8_t m_super[(MTF_NT / MTF_GROUP) * 264 + 8];   // recency list of every group of chunks
__shared__ uint16_t m_supercnt[MTF_NT / MTF_GROUP];
__shared__ uint64_t m_supermask[(MTF_NT / MTF_GROUP) * 4];        // the symbols of every group's list
__shared__ uint32_t m_inuse[256];
__shared__ uint8_t m_seq[256];
__shared__ uint32_t m_freq[BZX_MAX_ALPHA + 2];
__shared__ uint32_t m_scratch[2 * (MTF_NT / 64)];
__shared__ uint32_t m_bcast[4];   // [0] block, [1] carry last-nonzero+1, [2] carry output count

// Set of dense symbol ids < 64 * NW, one bit each, in registers (NW = 1, 2 or 4 words).
template <int NW> struct SeenSet {
    uint64_t w[NW];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < NW; i++) w[i] = 0;
    }
    // returns true if s was already present; marks it
    __device__ __forceinline__ bool test_set(uint32_t s)
    {
        const uint64_t bit = 1ull << (s & 63u);
        if (NW == 1) {
            const bool was = (w[0] & bit) != 0;
            w[0] |= bit;
            return was;
        }
        const uint32_t q = s >> 6;
        bool was = false;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const bool here = q == (uint32_t)i;
            was = was || (here && (w[i] & bit) != 0);
            w[i] |= here ? bit : 0ull;
        }
        return was;
    }
    __device__ __forceinline__ bool test(uint32_t s) const
    {
        uint64_t x = w[0];
#pragma unroll
        for (int i = 1; i < NW; i++) x = (s >> 6) == (uint32_t)i ? w[i] : x;
        return (x >> (s & 63u)) & 1ull;
    }
};

// MTF ranks of my chunk with the WHOLE list in registers (alphabets of <= 8*NWORD symbols, e.g. text):
// byte k of word k/8 = list entry k.  Branch-free: every lane finds the word and byte that hold the symbol
// with the zero-byte trick, then every word is either shifted by one entry (words before the hit), patched
// (the hit word) or left alone -- no LDS traffic and no divergence inside the loop.
template <int NWORD>
__device__ __forceinline__ void mtf_ranks_regs(const uint8_t *L, uint8_t *R, uint32_t c_lo,
                                               uint32_t c_hi, const uint64_t *lst64, const uint8_t *seq)
{
    const uint64_t ones = 0x0101010101010101ull, highs = 0x8080808080808080ull;
    uint64_t w[NWORD];
#pragma unroll
    for (int i = 0; i < NWORD; i++) w[i] = lst64[i];
    uint4 nxt = c_lo < c_hi ? *reinterpret_cast<const uint4 *>(L + c_lo) : make_uint4(0, 0, 0, 0);
    for (uint32_t i0 = c_lo; i0 < c_hi; i0 += 16) {
        const uint4 v = nxt;
        if (i0 + 16 < c_hi) nxt = *reinterpret_cast<const uint4 *>(L + i0 + 16);     // in flight during the 16 steps below
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + q * 4 + k;
                if (i < c_hi) {
                    const uint64_t s = seq[(wd[q] >> (8 * k)) & 255u];
                    const uint64_t sp = s * ones;
                    // hit word h and byte j
                    uint32_t h = NWORD, j = 0;
#pragma unroll
                    for (int t = NWORD - 1; t >= 0; t--) {
                        const uint64_t x = w[t] ^ sp;
                        const uint64_t z = (x - ones) & ~x & highs;
                        if (z) {
                            h = (uint32_t)t;
                            j = (uint32_t)(__ffsll((unsigned long long)z) - 1) >> 3;
                        }
                    }
                    const uint32_t rank = 8 * h + j;
                    if (rank) {
                        const uint64_t lowmask = j ? ((1ull << (8 * j)) - 1ull) : 0ull;
                        const uint64_t himask = j == 7 ? 0ull : (~0ull << (8 * (j + 1)));
                        uint64_t carry = s;
#pragma unroll
                        for (int t = 0; t < NWORD; t++) {
                            const uint64_t cur = w[t];
                            const uint64_t shifted = (cur << 8) | carry;
                            const uint64_t patched = (cur & himask) | ((cur & lowmask) << 8) | carry;
                            w[t] = (uint32_t)t < h ? shifted : ((uint32_t)t == h ? patched : cur);
                            carry = cur >> 56;
                        }
                    }
                    o[q] |= rank << (8 * k);
                }
            }
        }
        *reinterpret_cast<uint4 *>(R + i0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Recency list of chunk [c_lo, c_hi): distinct symbols by last occurrence, most recent first.
template <int NW>
__device__ __forceinline__ void mtf_recency(const uint8_t *__restrict__ L, uint32_t c_lo, uint32_t c_hi, uint32_t n_in_use,
                                            uint8_t *rec, uint16_t *cnt_out, uint64_t *mask_out)
{
    SeenSet<NW> seen;
    seen.clear();
    uint32_t cnt = 0;
    // 16 bytes per load, walking backwards (chunk starts are 16-byte aligned, slabs 256-byte aligned)
    uint4 nxt = *reinterpret_cast<const uint4 *>(L + ((c_hi - 1) & ~15u));
    for (uint32_t i0 = (c_hi - 1) & ~15u; cnt < n_in_use;) {
        const uint4 v = nxt;
        if (i0 != c_lo) nxt = *reinterpret_cast<const uint4 *>(L + i0 - 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 3; q >= 0; q--) {
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                const uint32_t i = i0 + (uint32_t)(q * 4 + k);
                if (i < c_hi) {
                    const uint32_t s = m_seq[(w[q] >> (8 * k)) & 255u];
                    if (!seen.test_set(s)) rec[cnt++] = (uint8_t)s;
                }
            }
        }
        if (i0 == c_lo) break;
        i0 -= 16;
    }
    *cnt_out = (uint16_t)cnt;
#pragma unroll
    for (int i = 0; i < NW; i++) mask_out[i] = seen.w[i];              // the symbols the chunk holds
}

// Is symbol `sym` in the set m (NW words)?
template <int NW>
__device__ __forceinline__ bool mtf_in_set(const uint64_t *m, uint32_t sym)
{
    uint64_t x = m[0];
#pragma unroll
    for (int i = 1; i < NW; i++) x = (sym >> 6) == (uint32_t)i ? m[i] : x;
    return (x >> (sym & 63u)) & 1ull;
}

// One merge step by a whole wave: the entries src[0 .. n) that are not in the set `skip` are appended to dst at `at`, in
// order (entry k belongs to lane k & 63: a ballot and a count of the lanes below place it).
template <int NW, typename SrcPtr>
__device__ __forceinline__ void mtf_append_new(SrcPtr src, uint32_t n, const uint64_t *skip, uint8_t *dst, uint32_t &at, uint32_t lane)
{
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)NW; j++) {
        const uint32_t k = lane + 64 * j;
        if (64 * j < n) {
            const uint32_t sym = k < n ? src[k] : 0u;
            const bool keep = k < n && !mtf_in_set<NW>(skip, sym);
            const uint64_t m = __ballot(keep);
            if (keep) dst[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)sym;
            at += (uint32_t)__popcll(m);
        }
    }
}

// The MTF list at the start of every chunk -- the symbols by their most recent use before the chunk, then the never-seen
// ones in id order -- from the chunks' recency lists (recs: LDS, or global memory for large alphabets) and symbol sets
// (masks, global memory) into the working lists (LDS).
// Rounds 1-3 let every lane walk back through the lists of the chunks before its own (two levels: 32 chunks, groups of
// 32) until it had seen every symbol.  On source code -- a hundred symbols, a few of them rare, so no walk ends early
// and every lane of a wave stops somewhere else -- that was 0.8 ms per block, as much as the ranking itself.  Now a
// WAVE owns 64 consecutive chunks and every step is one parallel pass over a list (<= 64 * NW entries, entry k on lane
// k & 63):
//   A. the recency list of its 64 chunks together: from the last chunk backwards, append what is new (chunks whose
//      set brings nothing new are skipped after one AND);
//   B. the list at its first chunk: the group lists of the waves before it, most recent first, the same way, then
//      the never-seen symbols;
//   C. chunk after chunk: list(c + 1) = recency list of chunk c, then list(c) without the symbols of chunk c.
template <int NW, typename RecPtr>
__device__ __forceinline__ void mtf_start_lists(uint32_t tid, uint32_t nch_used, uint32_t n_in_use, uint32_t stride,
                                                RecPtr recs, const uint64_t *masks, uint8_t *lists,
                                                unsigned long long *dbg, unsigned long long &t_last)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t lo = wave * 64;
    const uint32_t hi = lo + 64 < nch_used ? lo + 64 : nch_used;
    // lane l keeps what the steps need to know about chunk lo + l: its set and the length of its recency list
    uint64_t my_mask[NW];
    uint32_t my_rc = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) my_mask[i] = 0;
    if (lo + lane < nch_used) {
#pragma unroll
        for (int i = 0; i < NW; i++) my_mask[i] = masks[(size_t)(lo + lane) * NW + i];
        my_rc = m_reccnt[lo + lane];
    }
    if (lo < nch_used) {
        uint64_t seen[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) seen[i] = 0;
        uint32_t gc = 0;
        uint8_t *grp = m_super + wave * 264;
        for (uint32_t c = hi; c > lo && gc < n_in_use;) {
            c--;
            uint64_t mk[NW];
            bool any = false;
#pragma unroll
            for (int i = 0; i < NW; i++) {
                mk[i] = __shfl(my_mask[i], (int)(c - lo));
                any = any || (mk[i] & ~seen[i]) != 0;
            }
            const uint32_t rc = (uint32_t)__shfl((int)my_rc, (int)(c - lo));
            if (any) {
                mtf_append_new<NW>(recs + c * stride, rc, seen, grp, gc, lane);
#pragma unroll
                for (int i = 0; i < NW; i++) seen[i] |= mk[i];
            }
        }
        if (lane == 0) {
            m_supercnt[wave] = (uint16_t)gc;
#pragma unroll
            for (int i = 0; i < NW; i++) m_supermask[wave * 4 + i] = seen[i];
        }
    }
    __syncthreads();
    if (dbg && tid == 0) {                                      // (diagnostic phase timer: step A)
        const unsigned long long now_ = wall_clock64();
        atomicAdd(&dbg[37], now_ - t_last);
        t_last = now_;
    }
    if (lo < nch_used) {
        uint64_t seen[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) seen[i] = 0;
        uint32_t cnt = 0;
        uint8_t *lst = lists + lo * stride;
        for (uint32_t g = wave; g > 0 && cnt < n_in_use;) {
            g--;
            uint64_t mk[NW];
            bool any = false;
#pragma unroll
            for (int i = 0; i < NW; i++) {
                mk[i] = m_supermask[g * 4 + i];
                any = any || (mk[i] & ~seen[i]) != 0;
            }
            if (any) {
                mtf_append_new<NW>((const uint8_t *)m_super + g * 264, (uint32_t)m_supercnt[g], seen, lst, cnt, lane);
#pragma unroll
                for (int i = 0; i < NW; i++) seen[i] |= mk[i];
            }
        }
        // symbols never seen so far keep the initial (ascending) order
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)NW; j++) {
            const uint32_t sy = lane + 64 * j;
            const bool keep = sy < n_in_use && !mtf_in_set<NW>(seen, sy);
            const uint64_t m = __ballot(keep);
            if (keep) lst[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)sy;
            cnt += (uint32_t)__popcll(m);
        }
        for (uint32_t k = n_in_use + lane; k < stride; k += 64) lst[k] = 0xff;        // padding never matches before a real entry
        lds_order();
        // ---- C.  (the recency list of the next step is loaded while this one is merged)
        uint32_t rnext[NW];
        {
            const uint32_t rc = (uint32_t)__shfl((int)my_rc, 0);
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)NW; j++) rnext[j] = lane + 64 * j < rc ? recs[lo * stride + lane + 64 * j] : 0u;
        }
        for (uint32_t c = lo; c + 1 < hi; c++) {
            const uint8_t *cur = lists + c * stride;
            uint8_t *nxt = lists + (c + 1) * stride;
            uint64_t mk[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) mk[i] = __shfl(my_mask[i], (int)(c - lo));
            const uint32_t rc = (uint32_t)__shfl((int)my_rc, (int)(c - lo));
            const uint32_t rc1 = (uint32_t)__shfl((int)my_rc, (int)(c + 1 - lo));
            uint32_t rcur[NW];
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)NW; j++) {
                rcur[j] = rnext[j];
                rnext[j] = c + 2 < hi && lane + 64 * j < rc1 ? recs[(c + 1) * stride + lane + 64 * j] : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)NW; j++)
                if (lane + 64 * j < rc) nxt[lane + 64 * j] = (uint8_t)rcur[j];
            uint32_t at = rc;
            mtf_append_new<NW>(cur, n_in_use, mk, nxt, at, lane);
            for (uint32_t k = n_in_use + lane; k < stride; k += 64) nxt[k] = 0xff;
            lds_order();
        }
    }
}

#define MTF_STAMP(slot)                                                       \
    do {                                                                      \
        if (B.dbg && tid == 0) {                                              \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&B.dbg[slot], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)

__global__ __launch_bounds__(MTF_NT) void bzx_mtf_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x;
    unsigned long long t_last = 0;

    for (;;) {
        if (tid == 0) m_bcast[0] = atomicAdd(&B.counters[B.ctr_mtf], 1u);
        __syncthreads();
        const uint32_t j_ = m_bcast[0];
        __syncthreads();
        if (j_ >= B.nblk) break;
        const uint32_t b = B.blk_first + j_ * B.blk_step;

        const uint32_t n = B.blk[b].n;
        const uint8_t *__restrict__ L = B.bwt + BZX_SLAB(B, b) * BZX_BLK_STRIDE;
        uint8_t *H = B.rank + BZX_SLAB(B, b) * BZX_BLK_STRIDE;          // run heads (bytes), then their ranks, in place
        uint32_t *__restrict__ P = B.hpos + BZX_SLAB(B, b) * (size_t)B.hpos_stride;     // position of every head; P[nh] = n
        uint16_t *__restrict__ V = B.mtfv + BZX_SLAB(B, b) * BZX_BLK_STRIDE;

        if (B.dbg && tid == 0) t_last = wall_clock64();
        // ---- 1. bytes in use (rle2_mtf.rs:26-45) and the run heads: byte i is a head unless it equals byte i-1.  A tile
        // of 16 bytes per lane; the heads of a tile are parked in LDS (bytes and positions) and leave by consecutive lanes.
        if (tid < 256) m_inuse[tid] = 0;
        for (uint32_t i = tid; i < BZX_MAX_ALPHA + 2; i += MTF_NT) m_freq[i] = 0;
        if (tid == 0) m_bcast[1] = 0;
        __syncthreads();
        {
            uint8_t *sth = m_rec;                                       // [MTF_NT * 16] heads of the tile
            uint32_t *stp = reinterpret_cast<uint32_t *>(m_list);       // [MTF_NT * 16] their positions
            static_assert(MTF_LIST_BYTES >= MTF_NT * MTF_E * 4, "tile staging");
            uint32_t done = 0;                                          // heads of the earlier tiles
            // (the next tile's bytes are loaded while this one is compacted: a tile is four barriers and little else, and
            // the load at its top was a quarter of the pass)
            uint4 vn = make_uint4(0, 0, 0, 0);
            uint32_t pn = 0x100;                                        // (no byte: position 0 is a head)
            if (tid * MTF_E < n) {
                vn = *reinterpret_cast<const uint4 *>(L + tid * MTF_E); // bytes past n are ignored below
                if (tid) pn = L[tid * MTF_E - 1];
            }
            for (uint32_t t0 = 0; t0 < n; t0 += MTF_NT * MTF_E) {
                const uint32_t i0 = t0 + tid * MTF_E;
                const uint4 v = vn;
                uint32_t prev = pn;
                {
                    const uint32_t i1 = i0 + MTF_NT * MTF_E;
                    if (i1 < n) {
                        vn = *reinterpret_cast<const uint4 *>(L + i1);
                        pn = L[i1 - 1];
                    }
                }
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t hm = 0;
#pragma unroll
                for (int k = 0; k < MTF_E; k++) {
                    const uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 255u;
                    if (i0 + k < n) {
                        m_inuse[c] = 1;
                        hm |= (uint32_t)(c != prev) << k;
                    }
                    prev = c;
                }
                uint32_t tot;
                const uint32_t ex = bzx_block_excl_sum_lds<MTF_NT>((uint32_t)__popc(hm), m_scratch, tot);
                uint32_t o = ex;
#pragma unroll
                for (int k = 0; k < MTF_E; k++) {
                    if ((hm >> k) & 1u) {
                        sth[o] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 255u);
                        stp[o] = i0 + k;
                        o++;
                    }
                }
                bzx_lds_barrier();
                for (uint32_t q = tid; q < tot; q += MTF_NT) {
                    H[done + q] = sth[q];
                    P[done + q] = stp[q];
                }
                done += tot;
                bzx_lds_barrier();
            }
            if (tid == 0) {
                P[done] = n;
                m_bcast[1] = done;
            }
        }
        __syncthreads();
        const uint32_t nh = m_bcast[1];                                 // run heads (>= 1)
        uint32_t n_in_use;
        {
            const uint32_t flag = tid < 256 ? m_inuse[tid] : 0u;
            const uint32_t ex = bzx_block_excl_sum<MTF_NT>(flag, m_scratch, n_in_use);
            if (tid < 256) {
                m_seq[tid] = (uint8_t)ex;
                B.in_use[BZX_SLAB(B, b) * 256 + tid] = (uint8_t)flag;
            }
        }
        __syncthreads();                                                // (also: the heads written above are visible)

        MTF_STAMP(32);
        // ---- 2. chunking of the heads: a list is stride64 (odd) 8-byte words per chunk, lists fit 72 KiB
        const uint32_t stride64 = ((n_in_use + 7) / 8) | 1u;      // odd word stride: conflict-free 64-bit LDS access
        const uint32_t stride = stride64 * 8;
        // Large alphabets: 72 KB of working lists are 279 lanes' worth at 256 symbols -- one wave per SIMD, and the walk
        // through a list is a chain of dependent LDS round trips that nothing hides.  So the recency lists (written
        // once, read once, in between) move out to global memory -- the symbol slab of the block, idle until the
        // zero-run pass -- and the working lists get the whole pool: twice the lanes.
        const bool big = n_in_use > 64;
        uint8_t *grec = reinterpret_cast<uint8_t *>(V);            // [nch][stride] (<= 1,024 x 264 B of the 1.8 MB slab)
        uint64_t *gmask = reinterpret_cast<uint64_t *>(grec + MTF_NT * 264);        // [nch][NW] the symbols of every chunk
        uint8_t *lists = big ? m_pool : m_list;
        uint32_t nch = (big ? 2 * MTF_LIST_BYTES : MTF_LIST_BYTES) / stride;
        if (nch > MTF_NT) nch = MTF_NT;
        uint32_t csz = (nh + nch - 1) / nch;
        csz = (csz + 15u) & ~15u;
        const uint32_t nch_used = (nh + csz - 1) / csz;
        const uint32_t c_lo = tid * csz;
        const uint32_t c_hi = (c_lo + csz < nh) ? c_lo + csz : nh;
        const bool have_chunk = tid < nch_used;

        // ---- 3. recency list of my chunk: distinct symbols by last occurrence, most recent first
        if (have_chunk) {
            if (n_in_use <= 64) mtf_recency<1>(H, c_lo, c_hi, n_in_use, m_rec + tid * stride, &m_reccnt[tid], gmask + tid);
            else if (n_in_use <= 128) mtf_recency<2>(H, c_lo, c_hi, n_in_use, grec + tid * stride, &m_reccnt[tid], gmask + tid * 2);
            else mtf_recency<4>(H, c_lo, c_hi, n_in_use, grec + tid * stride, &m_reccnt[tid], gmask + tid * 4);
        }
        __syncthreads();                 // (also orders the global recency lists: all waves of a workgroup share the unit's L1)
        MTF_STAMP(33);

        // ---- 4. MTF list at every chunk start (two-level walk over the recency lists)
        if (n_in_use <= 64) mtf_start_lists<1>(tid, nch_used, n_in_use, stride, (const uint8_t *)m_rec, gmask, lists, B.dbg, t_last);
        else if (n_in_use <= 128) mtf_start_lists<2>(tid, nch_used, n_in_use, stride, (const uint8_t *)grec, gmask, lists, B.dbg, t_last);
        else mtf_start_lists<4>(tid, nch_used, n_in_use, stride, (const uint8_t *)grec, gmask, lists, B.dbg, t_last);
        __syncthreads();
        MTF_STAMP(34);

        // ---- 5. plain MTF over my heads (rle2_mtf.rs:61-138); a head's rank replaces its byte.
        // The first 8 list entries live in a register (byte 0 = front); deeper entries in LDS as 8-byte words,
        // searched and shifted one word at a time.
        if (have_chunk && n_in_use <= 32) {
            const uint64_t *lst64 = reinterpret_cast<const uint64_t *>(lists + tid * stride);
            if (n_in_use <= 8) mtf_ranks_regs<1>(H, H, c_lo, c_hi, lst64, m_seq);
            else if (n_in_use <= 16) mtf_ranks_regs<2>(H, H, c_lo, c_hi, lst64, m_seq);
            else if (n_in_use <= 24) mtf_ranks_regs<3>(H, H, c_lo, c_hi, lst64, m_seq);
            else mtf_ranks_regs<4>(H, H, c_lo, c_hi, lst64, m_seq);
        } else if (have_chunk) {
            uint64_t *lst64 = reinterpret_cast<uint64_t *>(lists + tid * stride);
            uint64_t w = lst64[0];
            const uint64_t ones = 0x0101010101010101ull, highs = 0x8080808080808080ull;
            uint4 nxt = *reinterpret_cast<const uint4 *>(H + c_lo);
            for (uint32_t i0 = c_lo; i0 < c_hi; i0 += 16) {
                const uint4 v = nxt;
                if (i0 + 16 < c_hi) nxt = *reinterpret_cast<const uint4 *>(H + i0 + 16);
                const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; q++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t i = i0 + q * 4 + k;
                        if (i < c_hi) {
                            const uint64_t s = m_seq[(wd[q] >> (8 * k)) & 255u];
                            const uint64_t sp = s * ones;
                            uint64_t x = w ^ sp;
                            uint64_t z = (x - ones) & ~x & highs;
                            uint32_t rank;
                            if (z) {
                                const uint32_t j = (uint32_t)(__ffsll((unsigned long long)z) - 1) >> 3;
                                rank = j;
                                if (j) {
                                    const uint64_t lowmask = (1ull << (8 * j)) - 1ull;
                                    const uint64_t himask = j == 7 ? 0ull : (~0ull << (8 * (j + 1)));
                                    w = (w & himask) | ((w & lowmask) << 8) | s;
                                }
                            } else {
                                // deeper than the front word: walk the LDS list four words (32 entries) per step so
                                // that a deep hit costs a quarter of the dependent LDS round trips; words in front
                                // of the hit are shifted by one entry on the way (reads past my list's last word
                                // see the neighbour's list or padding, but the symbol is always found before them)
                                uint64_t carry = w >> 56;
                                w = (w << 8) | s;
                                uint32_t qq = 1;
                                for (;;) {
                                    const uint64_t c0 = lst64[qq], c1 = lst64[qq + 1], c2 = lst64[qq + 2], c3 = lst64[qq + 3];
                                    const uint64_t x0 = c0 ^ sp, x1 = c1 ^ sp, x2 = c2 ^ sp, x3 = c3 ^ sp;
                                    const uint64_t z0 = (x0 - ones) & ~x0 & highs, z1 = (x1 - ones) & ~x1 & highs;
                                    const uint64_t z2 = (x2 - ones) & ~x2 & highs, z3 = (x3 - ones) & ~x3 & highs;
                                    const uint32_t hk = z0 ? 0u : z1 ? 1u : z2 ? 2u : z3 ? 3u : 4u;
                                    if (hk > 0) lst64[qq] = (c0 << 8) | carry;
                                    if (hk > 1) lst64[qq + 1] = (c1 << 8) | (c0 >> 56);
                                    if (hk > 2) lst64[qq + 2] = (c2 << 8) | (c1 >> 56);
                                    if (hk > 3) {
                                        lst64[qq + 3] = (c3 << 8) | (c2 >> 56);
                                        carry = c3 >> 56;
                                        qq += 4;
                                        continue;
                                    }
                                    const uint64_t ch = hk == 0 ? c0 : hk == 1 ? c1 : hk == 2 ? c2 : c3;
                                    const uint64_t zh = hk == 0 ? z0 : hk == 1 ? z1 : hk == 2 ? z2 : z3;
                                    const uint64_t cin = hk == 0 ? carry : (hk == 1 ? c0 : hk == 2 ? c1 : c2) >> 56;
                                    const uint32_t j = (uint32_t)(__ffsll((unsigned long long)zh) - 1) >> 3;
                                    const uint64_t lowmask = j ? ((1ull << (8 * j)) - 1ull) : 0ull;
                                    const uint64_t himask = j == 7 ? 0ull : (~0ull << (8 * (j + 1)));
                                    lst64[qq + hk] = (ch & himask) | ((ch & lowmask) << 8) | cin;
                                    rank = 8 * (qq + hk) + j;
                                    break;
                                }
                            }
                            o[q] |= rank << (8 * k);
                        }
                    }
                }
                *reinterpret_cast<uint4 *>(H + i0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        MTF_STAMP(35);
        // ---- 6. zero-run coding + symbol emission + histogram (rle2_mtf.rs:63-172), one head at a time: head k emits
        // rank + 1 (every head but possibly the first has a nonzero rank) and then the RUNA/RUNB digits of the zeros
        // that follow it up to the next head: P[k+1] - P[k] - 1 of them (one more for a first head of rank 0).
        if (tid == 0) m_bcast[2] = 0;
        __syncthreads();
        uint32_t hot0 = 0, hot1 = 0, hot2 = 0;      // RUNA, RUNB and symbol 2 (rank 1) counted in registers
        // The symbols of a tile are collected in LDS (the recency lists are dead by now) and written to V as aligned
        // 4-byte pairs by consecutive lanes, instead of one 2-byte store per symbol.  When the count so far is odd,
        // the last symbol waits in stg[0] for the next tile: pend = carry_out & 1.
        uint16_t *stg = reinterpret_cast<uint16_t *>(m_pool);
        constexpr uint32_t HE = 8;                  // heads per lane and tile
        // (a tile's symbols fit: H heads of runs l_i emit H + sum floor(log2 l_i) symbols, sum l_i <= 900,000; for H = 8,192
        // that is at most 8,192 * 7 + 5,875 = 63,219 symbols of 2 bytes)
        static_assert(2 * MTF_LIST_BYTES >= 2 * 64000, "symbol staging of a tile");
        // (the next tile's ranks and positions are loaded while this one is coded, as in pass 1)
        uint2 rvn = make_uint2(0, 0);
        uint32_t ppn[HE + 1];
#pragma unroll
        for (uint32_t k = 0; k <= HE; k++) ppn[k] = n;
        if (tid * HE < nh) {
            __builtin_memcpy(&rvn, H + tid * HE, 8);                    // (8-byte aligned: a multiple of 8)
#pragma unroll
            for (uint32_t k = 0; k <= HE; k++) ppn[k] = tid * HE + k <= nh ? P[tid * HE + k] : n;
        }
        for (uint32_t t0 = 0; t0 < nh; t0 += MTF_NT * HE) {
            const uint32_t k0 = t0 + tid * HE;
            const uint32_t carry_out = m_bcast[2];
            uint32_t rk[HE], zr[HE];
            uint32_t my_cnt = 0;
            const uint2 rv = rvn;
            uint32_t pp[HE + 1];
#pragma unroll
            for (uint32_t k = 0; k <= HE; k++) pp[k] = ppn[k];
            {
                const uint32_t k1 = k0 + MTF_NT * HE;
                if (k1 < nh) {
                    __builtin_memcpy(&rvn, H + k1, 8);
#pragma unroll
                    for (uint32_t k = 0; k <= HE; k++) ppn[k] = k1 + k <= nh ? P[k1 + k] : n;
                }
            }
            if (k0 < nh) {
#pragma unroll
                for (uint32_t k = 0; k < HE; k++) {
                    rk[k] = ((k < 4 ? rv.x : rv.y) >> (8 * (k & 3))) & 255u;
                    zr[k] = 0;
                    if (k0 + k < nh) {
                        zr[k] = pp[k + 1] - pp[k] - 1u + ((k0 + k == 0 && rk[k] == 0) ? 1u : 0u);
                        my_cnt += (rk[k] ? 1u : 0u) + (zr[k] ? 31u - (uint32_t)__clz(zr[k] + 1u) : 0u);
                    } else {
                        rk[k] = 0;
                    }
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < HE; k++) rk[k] = zr[k] = 0;
            }
            uint32_t cnt_total;
            const uint32_t cnt_excl = bzx_block_excl_sum_lds<MTF_NT>(my_cnt, m_scratch, cnt_total);
            const uint32_t pend = carry_out & 1u;
            uint32_t o = pend + cnt_excl;               // index in stg; V index = carry_out - pend + index
#pragma unroll
            for (uint32_t k = 0; k < HE; k++) {
                if (rk[k]) {
                    stg[o++] = (uint16_t)(rk[k] + 1);
                    if (rk[k] == 1) hot2++; else atomicAdd(&m_freq[rk[k] + 1], 1u);
                }
                uint32_t z = zr[k];
                if (z) {
                    z--;
                    for (;;) {
                        const uint32_t sym = z & 1u;
                        stg[o++] = (uint16_t)sym;
                        if (sym) hot1++; else hot0++;
                        if (z < 2) break;
                        z = (z - 2) >> 1;
                    }
                }
            }
            bzx_lds_barrier();
            {
                const uint32_t total = pend + cnt_total;
                uint32_t *dst = reinterpret_cast<uint32_t *>(V + (carry_out - pend));
                const uint32_t *src = reinterpret_cast<const uint32_t *>(stg);
                for (uint32_t q = tid; q < (total >> 1); q += MTF_NT) dst[q] = src[q];
                const uint16_t left = stg[total ? total - 1 : 0];
                if (tid == 0) m_bcast[2] = carry_out + cnt_total;
                bzx_lds_barrier();
                if (tid == 0 && (total & 1u)) stg[0] = left;
            }
        }
        __syncthreads();
        if (hot0) atomicAdd(&m_freq[0], hot0);
        if (hot1) atomicAdd(&m_freq[1], hot1);
        if (hot2) atomicAdd(&m_freq[2], hot2);
        __syncthreads();
        MTF_STAMP(36);
        // EOB (the zeros behind the last head were its own)
        if (tid == 0) {
            uint32_t o = m_bcast[2];
            if (o & 1u) V[o - 1] = stg[0];                              // the symbol still waiting for a partner
            const uint32_t eob = n_in_use + 1;
            V[o++] = (uint16_t)eob;
            m_freq[eob]++;
            B.blk[b].n_mtf = o;
            B.blk[b].n_in_use = n_in_use;
        }
        __syncthreads();
        for (uint32_t i = tid; i < 260; i += MTF_NT) B.freq[BZX_SLAB(B, b) * 260 + i] = i < BZX_MAX_ALPHA ? m_freq[i] : 0u;
        __syncthreads();
    }
}

void bzx_launch_mtf(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_mtf_kernel, dim3(grid), dim3(MTF_NT), 0, stream, B);
}
