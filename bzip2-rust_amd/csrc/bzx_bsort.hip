// bzx_bsort.hip -- bucket suffix sorter: the Burrows-Wheeler transform of bzip2 blocks, many workgroups per block.
//
// Contract (reference src/bwt_algorithms/bwt_sort.rs:27-58, bwt_encode): sort all cyclic rotations of the block,
// L[j] = byte preceding the j-th smallest rotation, orig_ptr = row of rotation 0.
//
// Design (MI355X: 160 KB LDS per CU, 256 CUs): an MSD partition followed by sorts that never leave LDS.
//   split kernel  (one workgroup per block): bytes in use -> dense ids -> packed block P (bzx_pack.h); histogram of
//                 the first 15 bits of every rotation in LDS; adjacent bins are merged into BUCKETS of at most BS_C
//                 rotations (bins above BS_C/3 stay alone); one pass writes a 64-bit record per rotation
//                 [next 32 key bits | rotation:20 | preceding byte:8] into its bucket's range of ranks.  A bin that
//                 alone exceeds BS_C is split again by its next 12 bits (records re-keyed from P), and so on.
//   sort kernel   (work items = buckets, any workgroup takes any bucket of any block): the bucket's records are
//                 loaded into LDS once and sorted there: stable 8-bit LSD passes over the 32 key bits, then
//                 refinement rounds in which every rotation that is still tied fetches its next 50 key bits from P
//                 (one 8-byte read; P of a block is shared by ~150 buckets and stays in L2) and each tied group is
//                 ordered by them: groups of <= 64 by counting (each lane ranks its own record), up to 512 by one
//                 wave (wave-level LSD passes, digits on which the group agrees are skipped), larger ones by the
//                 whole workgroup.  L and orig_ptr are written when no ties are left.
// HBM traffic per block: block read + P written/read + records written once and read once + L written: ~19 n.
// A bucket that is still tied after BS_ROUNDS rounds (repeats of hundreds of symbols; identical rotations of a
// periodic block) keeps what it has: it writes its order so far and its group starts, and the block goes on
// B.resume_list; a second, short launch of the sort kernel (the fill pass) writes the order of that block's finished
// buckets, the rank rounds (below) finish the leftover groups bucket by bucket, and the general sorter (bzx_bwt.hip)
// runs its prefix-doubling rounds on whatever they leave -- it also detects periodic blocks.  Blocks the split kernel cannot handle (more split levels than BS_MAX_BIG
// tracks) are sorted from scratch by the general sorter (B.redo_list).
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"
#include "bzx_pack.h"

#define BS_NT 1024                      // split kernel: lanes per workgroup
#define BS_NW (BS_NT / 64)
#ifndef BS_C
#define BS_C 2048                       // rotations per bucket (records in LDS)
#endif
#define BS_FW (BS_C / 64)               // 64-bit words of group-start flags
#define BS_ISO (BS_C / 3)               // bins above this always stay alone
#ifndef BS_ISO_SHIFT
#define BS_ISO_SHIFT 5                  // a smaller limit is taken while it isolates less than n >> this many rotations
#endif
#ifndef BS_BIN1
#define BS_BIN1 15                      // bits of the level-1 bins
#endif
#define BS_BIN2 12                      // bits of the deeper levels
#define BS_NBIN1 (1u << BS_BIN1)
#define BS_TAB_WORDS (BS_NBIN1 + (BS_NBIN1 >> 5))
#define BS_MAX_BK 1024                   // buckets one split (one level of one range) may produce
#ifndef BS_MAX_BIG
#define BS_MAX_BIG 512                  // oversized bins waiting for a deeper split at any one time (ring, per block)
#endif
#define SP_U 4                          // split kernel: groups of four rotations in flight per lane
#ifndef BS_MAX_DEPTH
#define BS_MAX_DEPTH 511                // deepest split: 15 + 41 x 12 key bits ...
#endif
#ifndef BS_MAX_SPLITS
#define BS_MAX_SPLITS 1024              // ... and at most this many deeper splits per block: beyond, the general sorter
#endif
#ifndef BS_ROUNDS
#define BS_ROUNDS 4                     // refinement rounds of 50 bits before a bucket gives up: what is still tied then is
                                        // mostly tied for hundreds of symbols, and the rank rounds get there faster
#endif
#ifndef BS_KEYBITS
#define BS_KEYBITS 32                   // leading bits of the 32-bit record key sorted by the initial LSD passes (a multiple of 8)
#endif
#define BS_TINY 64                      // groups up to this size are ranked by counting
#define BS_MED 512                      // ... up to this size by one wave
#define REC_IDX(r) ((uint32_t)((r) >> 12) & 0xFFFFFu)
#define REC_PREV(r) ((uint32_t)((r) >> 4) & 0xFFu)
#define W_POS_MASK 0x3FFFull            // low 14 bits of a round word: position of the record in the bucket

static_assert(BS_C <= 16384, "bucket capacity");

// Phase timers and event counts, diagnostic build only (-DBZX_DIAG, libbzx_diag.so): B.dbg[128], 100 MHz ticks.
#ifdef BZX_DIAG
// accumulated per workgroup in LDS by lane 0 and flushed once at kernel exit (a global atomic per stamp made the
// timers themselves the bottleneck once a launch had hundreds of thousands of buckets)
__shared__ unsigned long long s_diag[128];
#define DIAG_T0()                                                             \
    unsigned long long t_last_ = 0;                                           \
    if (B.dbg && threadIdx.x == 0) {                                          \
        for (int i_ = 0; i_ < 128; i_++) s_diag[i_] = 0;                      \
        t_last_ = wall_clock64();                                             \
    }                                                                         \
    (void)t_last_;
#define DIAG_STAMP(slot)                                                      \
    do {                                                                      \
        if (B.dbg && threadIdx.x == 0) {                                      \
            const unsigned long long now_ = wall_clock64();                   \
            s_diag[slot] += now_ - t_last_;                                   \
            t_last_ = now_;                                                   \
        }                                                                     \
    } while (0)
#define DIAG_COUNT(slot, v)                                                   \
    do {                                                                      \
        if (B.dbg && threadIdx.x == 0) s_diag[slot] += (unsigned long long)(v); \
    } while (0)
#define DIAG_FLUSH()                                                          \
    do {                                                                      \
        if (B.dbg && threadIdx.x == 0)                                        \
            for (int i_ = 0; i_ < 128; i_++)                                  \
                if (s_diag[i_]) atomicAdd(&B.dbg[i_], s_diag[i_]);            \
    } while (0)
#else
#define DIAG_T0() do {} while (0)
#define DIAG_STAMP(slot) do {} while (0)
#define DIAG_COUNT(slot, v) do {} while (0)
#define DIAG_FLUSH() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------ split kernel
__shared__ uint32_t b_tab[BS_TAB_WORDS + BS_MAX_BK];     // bin counts, then bucket ids (index padded: b + b/32); level-1 partition: see there
__shared__ uint32_t b_start[BS_MAX_BK + 1];  // first rank of every bucket (relative to the range being split)
__shared__ uint32_t b_first[BS_MAX_BK + 1];  // first bin of every bucket
__shared__ uint32_t b_cur[BS_MAX_BK];        // scatter cursors
__shared__ uint8_t b_b0[BS_MAX_BK];          // bits of the bin index that all bins of the bucket share
__shared__ uint32_t b_inuse[256];
__shared__ uint8_t b_seq[256];
__shared__ uint32_t b_scratch[2 * BS_NW];
__shared__ uint32_t b_lbase[8];              // first item of this block's buckets in each of the eight work lists
__shared__ uint32_t b_bcast[12];             // [0] block, [2] item base, [3] oversized bins pushed so far, [4] ring overflow,
                                             // [8..10] rotations in bins a smaller isolation limit would leave alone
__shared__ uint32_t b_big[BS_MAX_BIG][3];    // oversized bins: {start | buffer << 31, cnt, depth bits}

#define TAB(b) b_tab[(b) + ((b) >> 5)]

// Bins [0, BINS) counted in TAB -> buckets.  With an isolation limit ISO, a new bucket starts at bin b when b or b-1 is
// above ISO, or when the bin's first rank lies in another (BS_C - ISO)-window than its predecessor's: a merged bucket is
// at most one window plus one bin <= BS_C.  The smaller ISO, the wider the window and the fuller the buckets, but bins
// between ISO and BS_ISO then make small buckets of their own: the smallest of BS_C/8, /6, /4 that leaves fewer than
// total >> BS_ISO_SHIFT rotations in such bins is taken, else BS_ISO.  (b_bcast[8..10] are zero on entry.)
// Returns the number of buckets (0: more than BS_MAX_BK); TAB(b) = bucket of bin b afterwards.
template <int ISO, int PER>
__device__ __forceinline__ void bucket_flags(const uint32_t (&c)[PER], uint32_t bin0, uint32_t excl, uint32_t prevc, uint32_t &fl, uint32_t &nf)
{
    constexpr uint32_t CP = BS_C - ISO;
    uint32_t s = excl, ps = excl - prevc, pc = prevc;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const bool f = (bin0 + k == 0) || c[k] > (uint32_t)ISO || pc > (uint32_t)ISO || (s / CP != ps / CP);
        fl |= (uint32_t)f << k;
        nf += f;
        ps = s;
        s += c[k];
        pc = c[k];
    }
}

template <int BINS, bool TAB16 = false>
__device__ __forceinline__ uint32_t form_buckets(uint32_t total)
{
    constexpr int PER = BINS / BS_NT;
    const uint32_t tid = threadIdx.x, bin0 = tid * PER;
    uint32_t c[PER], sum = 0, e8 = 0, e6 = 0, e4 = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        c[k] = TAB(bin0 + k);
        sum += c[k];
        const uint32_t m = c[k] <= BS_ISO ? c[k] : 0u;
        e8 += m > BS_C / 8 ? m : 0u;
        e6 += m > BS_C / 6 ? m : 0u;
        e4 += m > BS_C / 4 ? m : 0u;
    }
    const uint32_t prevc = tid ? TAB(bin0 - 1) : 0u;
    if (__any(e8 != 0)) {
        e8 = bzx_wave_incl_sum(e8);
        e6 = bzx_wave_incl_sum(e6);
        e4 = bzx_wave_incl_sum(e4);
        if (bzx_lane() == 63) {
            atomicAdd(&b_bcast[8], e8);
            if (e6) atomicAdd(&b_bcast[9], e6);
            if (e4) atomicAdd(&b_bcast[10], e4);
        }
    }
    uint32_t tot;
    const uint32_t excl = bzx_block_excl_sum<BS_NT>(sum, b_scratch, tot);      // (its first barrier publishes b_bcast[8..10])
    const uint32_t allow = total >> BS_ISO_SHIFT;
    uint32_t fl = 0, nf = 0;
    if (b_bcast[8] <= allow) bucket_flags<BS_C / 8>(c, bin0, excl, prevc, fl, nf);
    else if (b_bcast[9] <= allow) bucket_flags<BS_C / 6>(c, bin0, excl, prevc, fl, nf);
    else if (b_bcast[10] <= allow) bucket_flags<BS_C / 4>(c, bin0, excl, prevc, fl, nf);
    else bucket_flags<BS_ISO>(c, bin0, excl, prevc, fl, nf);
    uint32_t nbk;
    const uint32_t fexcl = bzx_block_excl_sum<BS_NT>(nf, b_scratch, nbk);
    if (nbk > BS_MAX_BK) return 0;
    {
        uint32_t id = fexcl, s = excl;         // id = flags before this bin
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if ((fl >> k) & 1u) {
                b_start[id] = s;
                b_first[id] = bin0 + k;
                id++;
            }
            // (TAB16: the bucket ids go to a dense 16-bit table over the start of the count table -- every count was read
            // into registers before the first scan's barrier, so nobody still needs the words this overwrites)
            if (TAB16) reinterpret_cast<uint16_t *>(b_tab)[bin0 + k] = (uint16_t)(id - 1);
            else TAB(bin0 + k) = id - 1;
            s += c[k];
        }
    }
    if (tid == 0) {
        b_start[nbk] = total;
        b_first[nbk] = BINS;
    }
    __syncthreads();
    return nbk;
}

// Bucket descriptors after form_buckets: shared bits, cursors.
template <int BINW>
__device__ __forceinline__ void bucket_setup(uint32_t nbk)
{
    for (uint32_t k = threadIdx.x; k < nbk; k += BS_NT) {
        const uint32_t first = b_first[k], last = b_first[k + 1] - 1u;
        b_b0[k] = (uint8_t)(first == last ? BINW : (uint32_t)__builtin_clz(first ^ last) - (32u - BINW));
        b_cur[k] = 0;
    }
}

// Work items for the buckets formed last: ranks base+b_start[k].. in buffer `buf`, keys from bit depth+b_b0[k].
// Oversized buckets go to the big list instead (always a single bin: all its BINW bits are shared).
// Returns false when the big list is full.
template <int BINW>
__device__ __forceinline__ bool bucket_emit(const BzxBatch &B, uint32_t blk, uint32_t jj, uint32_t nbk, uint32_t base, uint32_t buf,
                                            uint32_t depth, uint32_t bits, uint32_t done, uint32_t salt)
{
    // jj: index of the block in the launch.  The nbk items go to list jj % 8 (bk_affine), or item k to list (jj + salt + k) % 8
    // (salt: the number of the split within the block -- a small batch of repetitive blocks emits hundreds of one-bucket
    // splits, and they must not all land in the same eighth of the work list).
    const uint32_t tid = threadIdx.x, cap8 = B.bk_cap >> 3;
    if (!B.bk_affine) jj += salt;
    if (tid < 8) {
        const uint32_t mine = B.bk_affine ? (tid == (jj & 7u) ? nbk : 0u) : (nbk + 7u - ((tid - jj) & 7u)) >> 3;
        uint32_t at = mine ? atomicAdd(&B.counters[BZX_CTR_BK_LIST0 + tid], mine) : 0u;
        if (at + mine > cap8) {                        // work list full (the lists are zeroed per batch: unwritten items are empty)
            b_bcast[4] = 1;
            at = 0;
        }
        b_lbase[tid] = tid * cap8 + at;
    }
    __syncthreads();
    if (b_bcast[4]) return false;
    for (uint32_t k = tid; k < nbk; k += BS_NT) {
        BzxBucket it;
        it.blk = blk;
        it.start = (base + b_start[k]) | (buf << 31);
        it.cnt = b_start[k + 1] - b_start[k];
        it.dbits = (depth + b_b0[k]) | (bits << 16);
        if (it.cnt > BS_C) {
            const uint32_t q = atomicAdd(&b_bcast[3], 1u);      // ring: entries [done, q] are pending
            if (q - done < BS_MAX_BIG) {
                b_big[q % BS_MAX_BIG][0] = it.start;
                b_big[q % BS_MAX_BIG][1] = it.cnt;
                b_big[q % BS_MAX_BIG][2] = depth + BINW;
            } else {
                b_bcast[4] = 1;                                   // ring full
            }
            it.cnt = 0;                          // the sort kernel skips empty items
        }
        B.bk_list[B.bk_affine ? b_lbase[jj & 7u] + k : b_lbase[(jj + k) & 7u] + (k >> 3)] = it;
    }
    __syncthreads();
    return b_bcast[4] == 0;
}

// Item of the lists of oversized bins (deeper split levels) and of oversized groups (regrouping pass).
struct BzxDeepItem {
    uint32_t blk, st, cnt, dbits;           // st: first rank | buffer << 31; dbits: depth in key bits | bits per symbol << 16
};

// Rank arrays (rank of every rotation, uint32[BZX_MAX_N]) of the block in slab k: TWO per block -- a rank round reads
// one and writes the other, so no rank changes under a reader (see the rank rounds).  nullptr: none (stress builds).
__device__ __forceinline__ uint32_t *rank_array(const BzxBatch &B, uint32_t k, uint32_t which)
{
    if (k >= B.rk_blocks) return nullptr;
    return B.isa2 + ((size_t)k * 2 + which) * BZX_MAX_N;
}

// An oversized bin the split gives up on (depth or split limit: thousands of rotations sharing a long prefix -- table
// borders, padding patterns) is left as ONE group of tied ranks in the resume format ([group start:1 @32 | rotation:20],
// see the sort kernel) and the block joins the resume blocks.  The rank rounds cannot refine such a group (it does not
// fit a workgroup), and multiplying the depth is only valid where every rank read is as deep as the round assumes: so the
// members' entries in both rank arrays carry RK_COARSE, a bucket that reads a coarse rank leaves the rounds with its own
// tied ranks marked coarse too (see the rank rounds), and the general sorter finishes whatever is left of the block --
// this group, those buckets -- afterwards.  (Until round 3 such a block went to the general sorter with ALL its tied
// ranks, half a block of them in real files: 15-50 ms by one workgroup, the tail of the whole batch.)
// rec_a / rec_b: the block's slabs.
__device__ __forceinline__ void emit_giant(const BzxBatch &B, uint32_t blk, uint64_t *rec_a, const uint64_t *rec_b,
                                           uint32_t st, uint32_t cnt, uint32_t depth, uint32_t bits)
{
    const uint32_t tid = threadIdx.x, base = st & 0x7fffffffu;
    const uint64_t *src = ((st >> 31) ? rec_b : rec_a) + base;      // (may be the very range written below)
    uint32_t *__restrict__ isa0 = rank_array(B, (uint32_t)BZX_SLAB(B, blk), 0), *__restrict__ isa1 = rank_array(B, (uint32_t)BZX_SLAB(B, blk), 1);
    // listed for the regrouping pass (bzx_brank_giant_kernel) when there is room and the block has rank arrays: then the
    // members' ranks are plain group-head ranks for now; otherwise they are coarse from the start
    if (tid == 0) b_bcast[7] = isa0 ? atomicAdd(&B.counters[BZX_CTR_GIANT_CNT], 1u) : 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t at = b_bcast[7];
    const bool listed = at < B.deep_cap;
    for (uint32_t i = tid; i < cnt; i += BS_NT) {
        const uint64_t r = src[i];
        rec_a[base + i] = (uint64_t)REC_IDX(r) | (i == 0 ? 1ull << 32 : 0ull);
        if (isa0) isa0[REC_IDX(r)] = isa1[REC_IDX(r)] = base | (listed ? 0u : RK_COARSE);
    }
    if (tid == 0) {
        atomicMin(&B.blk[blk].n_mtf, depth / bits);
        if (listed) {
            BzxDeepItem it;
            it.blk = blk;
            it.st = base;
            it.cnt = cnt;
            it.dbits = 0;
            reinterpret_cast<BzxDeepItem *>(B.deep_list)[(size_t)2 * B.deep_cap + at] = it;
        }
        atomicAdd(&B.blk[blk].n_groups, 1u);                      // (an open "bucket": closed by the regrouping pass if it dissolves the group)
        if ((atomicOr(&B.blk[blk].status, BZX_ST_RESUME) & BZX_ST_RESUME) == 0)
            B.resume_list[atomicAdd(&B.counters[BZX_CTR_RESUME], 1u)] = blk;
    }
    __syncthreads();
}

__device__ __forceinline__ void block_redo(const BzxBatch &B, uint32_t b)
{
    if ((atomicOr(&B.blk[b].status, BZX_ST_REDO) & BZX_ST_REDO) == 0)
        B.redo_list[atomicAdd(&B.counters[BZX_CTR_REDO], 1u)] = b;
}

// ---- deeper levels.  An oversized bin (more than BS_C rotations share its prefix) is split again by its next BS_BIN2
// bits into the other record buffer (its key field holds exactly those bits first); the records are re-keyed from P at
// the new depth.  The bins wait in a ring in LDS: b_bcast[3] counts the entries pushed (bucket_emit), `done` the entries
// taken.  Round 2 let the workgroup of a block work its ring off alone, and on real files a few blocks with hundreds of
// such bins set the time of the whole split kernel (python sources: 13 ms for 3.4 ms of average work per block).  Now
// the entries from `flush_from` on are handed to the list of level `push_lvl` (>= 0) as far as it has room, and a
// launch of bzx_bsplit_deep_kernel per level deals them to all compute units; past the last level (or a full list)
// a workgroup finishes its subtree itself.  Returns false when the block has to be sorted from scratch.

__device__ bool deep_process(const BzxBatch &B, uint32_t b, uint32_t j_, uint32_t n, uint32_t bits, const uint8_t *__restrict__ P,
                             uint64_t *__restrict__ rec_a, uint64_t *__restrict__ rec_b, uint32_t done, uint32_t flush_from,
                             int push_lvl)
{
    const uint32_t tid = threadIdx.x;
    bool ok = true;
    while (ok) {
        const uint32_t nbig = b_bcast[3];
        __syncthreads();
        if (done >= nbig) break;
        if (push_lvl >= 0 && done >= flush_from) {
            const uint32_t np = nbig - done;
            if (tid == 0) b_bcast[5] = atomicAdd(&B.counters[BZX_CTR_DEEP_CNT + push_lvl], np);
            __syncthreads();
            const uint32_t at = b_bcast[5];
            __syncthreads();
            if (at + np <= B.deep_cap) {
                BzxDeepItem *dl = reinterpret_cast<BzxDeepItem *>(B.deep_list) + (size_t)(push_lvl & 1) * B.deep_cap + at;
                for (uint32_t q = tid; q < np; q += BS_NT) {
                    const uint32_t r = (done + q) % BS_MAX_BIG;
                    BzxDeepItem it;
                    it.blk = b;
                    it.st = b_big[r][0];
                    it.cnt = b_big[r][1];
                    it.dbits = b_big[r][2] | (bits << 16);
                    dl[q] = it;
                }
                __syncthreads();
                break;
            }
            push_lvl = -1;                        // (no room: the reserved entries stay empty, this workgroup goes on)
        }
        const uint32_t st = b_big[done % BS_MAX_BIG][0], cnt = b_big[done % BS_MAX_BIG][1], depth = b_big[done % BS_MAX_BIG][2];
        done++;
        if (tid == 0) b_bcast[6] = atomicAdd(reinterpret_cast<uint32_t *>(&B.blk[b].pack_word), 1u);      // splits spent on the block
        __syncthreads();                                      // (the slot may be reused by this split's pushes)
        const uint32_t spent = b_bcast[6];
        const uint32_t base = st & 0x7fffffffu, buf = st >> 31;
        const uint64_t *__restrict__ src = (buf ? rec_b : rec_a) + base;
        uint64_t *__restrict__ dst = (buf ? rec_a : rec_b) + base;
        // a whole turn of the block is shared (periodic blocks): general sorter, from scratch
        if (depth + BS_BIN2 + 32 >= n * bits) {
            ok = false;
            break;
        }
        // still oversized after BS_MAX_DEPTH key bits, or BS_MAX_SPLITS splits spent on this block (near-identical
        // copies of content): the bin stays one group
        if (depth + BS_BIN2 > BS_MAX_DEPTH || spent >= BS_MAX_SPLITS) {
            emit_giant(B, b, rec_a, rec_b, st, cnt, depth, bits);
            continue;
        }
        constexpr uint32_t NB2 = 1u << BS_BIN2;
        for (uint32_t i = tid; i < NB2 + (NB2 >> 5); i += BS_NT) b_tab[i] = 0;
        if (tid == 0) b_bcast[1] = 0;
        if (tid < 3) b_bcast[8 + tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += BS_NT) atomicAdd(&TAB((uint32_t)(src[i] >> (64 - BS_BIN2))), 1u);
        __syncthreads();
        const uint32_t nbk = form_buckets<(int)NB2>(cnt);
        if (nbk == 0) {
            ok = false;
            break;
        }
        bucket_setup<BS_BIN2>(nbk);
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += BS_NT) {
            const uint64_t r = src[i];
            const uint32_t k = TAB((uint32_t)(r >> (64 - BS_BIN2)));
            uint32_t x = REC_IDX(r) * bits + depth + b_b0[k];           // bit offsets wrap at the block end
            if (x >= n * bits) x -= n * bits;
            const uint32_t key = (uint32_t)(pk_window_bit(P, x) >> 32);
            const uint32_t slot = atomicAdd(&b_cur[k], 1u);
            dst[b_start[k] + slot] = ((uint64_t)key << 32) | (r & 0xFFFFFFFFull);
        }
        __syncthreads();
        ok = bucket_emit<BS_BIN2>(B, b, j_, nbk, base, buf ^ 1u, depth, bits, done, spent + 1u);
        DIAG_COUNT(104, 1);
        DIAG_COUNT(105, cnt);
    }
    return ok;
}

__global__ __launch_bounds__(BS_NT) void bzx_bsplit_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x;
    DIAG_T0();
    for (;;) {
        if (tid == 0) b_bcast[0] = atomicAdd(&B.counters[BZX_CTR_SPLIT_FETCH], 1u);
        __syncthreads();
        const uint32_t j_ = b_bcast[0];
        __syncthreads();
        if (j_ >= B.nblk) break;
        const uint32_t b = B.blk_first + j_ * B.blk_step;
        const uint32_t n = B.blk[b].n;
        const uint8_t *__restrict__ T = BZX_BLOCK_PTR(B, B.blk[b]);
        const size_t sb = BZX_SLAB(B, b);
        uint8_t *__restrict__ P = B.pk + sb * BZX_PK_STRIDE;
        uint64_t *__restrict__ rec_a = B.rec_a + sb * BZX_MAX_N;
        uint64_t *__restrict__ rec_b = B.rec_b + sb * BZX_MAX_N;

        // ---- bytes in use -> dense symbol ids, bits per symbol, packed block
        DIAG_STAMP(96);
        if (tid < 256) b_inuse[tid] = 0;
        if (tid == 0) {
            b_bcast[1] = 0;
            b_bcast[3] = 0;
            b_bcast[4] = 0;
            b_bcast[8] = b_bcast[9] = b_bcast[10] = 0;
            B.blk[b].pack_word = 0;              // (until the stream is laid out: deeper splits spent on the block)
            B.blk[b].n_mtf = 0xFFFFFFFFu;        // (until the MTF stage: smallest depth, in symbols, at which a bucket gave up)
            B.blk[b].n_groups = 0;               // (until the Huffman stage: buckets that gave up and are still open)
            B.blk[b].n_selectors = (uint32_t)BZX_SLAB(B, b);      // (until then: the block's pair of rank arrays)
        }
        __syncthreads();
        {
            const uint32_t n16 = n & ~15u;
            for (uint32_t i = tid * 16; i < n16; i += BS_NT * 16) {
                uint4 v;
                __builtin_memcpy(&v, T + i, 16);          // unaligned: zero-copy blocks start anywhere in the raw input
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) b_inuse[(w[q] >> (8 * k)) & 255u] = 1;
            }
            for (uint32_t i = n16 + tid; i < n; i += BS_NT) b_inuse[T[i]] = 1;
        }
        __syncthreads();
        uint32_t n_in_use;
        {
            const uint32_t flag = tid < 256 ? b_inuse[tid] : 0u;
            const uint32_t ex = bzx_block_excl_sum<BS_NT>(flag, b_scratch, n_in_use);
            if (tid < 256) b_seq[tid] = (uint8_t)ex;
        }
        __syncthreads();
        uint32_t bits = 1;
        while ((1u << bits) < n_in_use) bits++;
        DIAG_STAMP(97);
        pk_build_t<BS_NT>(T, n, bits, P, b_seq);
        DIAG_STAMP(98);

        // ---- level 1: histogram of the first BS_BIN1 bits of every rotation
        for (uint32_t i = tid; i < BS_TAB_WORDS; i += BS_NT) b_tab[i] = 0;
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n; t0 += BS_NT * 4 * SP_U) {
            // four consecutive rotations per window of P (3*8+15 <= 57 bits), SP_U windows per lane in flight
            uint64_t x[SP_U];
#pragma unroll
            for (uint32_t u = 0; u < SP_U; u++) {
                const uint32_t i0 = t0 + (u * BS_NT + tid) * 4;
                x[u] = i0 < n ? pk_window(P, i0, bits) : 0ull;
            }
#pragma unroll
            for (uint32_t u = 0; u < SP_U; u++) {
                const uint32_t i0 = t0 + (u * BS_NT + tid) * 4;
#pragma unroll
                for (uint32_t e = 0; e < 4; e++)
                    if (i0 + e < n) atomicAdd(&TAB((uint32_t)((x[u] << (e * bits)) >> (64 - BS_BIN1))), 1u);
            }
        }
        __syncthreads();
        DIAG_STAMP(99);
        uint32_t nbk = form_buckets<(int)BS_NBIN1, true>(n);
        if (nbk) bucket_setup<BS_BIN1>(nbk);
        __syncthreads();
        DIAG_STAMP(100);
        if (nbk == 0) {
            DIAG_COUNT(106, 1);
            if (tid == 0) block_redo(B, b);
            __syncthreads();
            continue;
        }

        // ---- level 1: one record per rotation into its bucket's range, staged through LDS.  A scattered 8-byte store
        // per rotation is one L2 request per record, and those requests -- not the bytes -- bounded this kernel (2/3
        // of its time; tools/ubench/scatter.hip: runs of 16 records go 3.4x faster than single ones).  So the block is
        // cut into tiles of SP_TILE rotations: a returning LDS atomic per rotation gives its place among the tile's
        // records for the same bucket, a scan turns the tile's bucket counts into offsets, the records are parked in
        // LDS in bucket order and leave as runs (~12 records per bucket and tile on text), consecutive lanes writing
        // consecutive records.  The 16-bit bucket table (64 KB) and the parking area (64 KB) share the histogram's LDS.
        {
            constexpr uint32_t SP_TILE = 8192, SP_V = SP_TILE / (BS_NT * 4);       // groups of four rotations per lane and tile
            static_assert(BS_MAX_BK <= BS_NT && BS_TAB_WORDS >= 32768 + BS_MAX_BK, "tile scan: one bucket per lane");
            uint16_t *tab16 = reinterpret_cast<uint16_t *>(b_tab);                  // bin -> bucket | shared bits of the bucket << 10
            for (uint32_t i = tid; i < BS_NBIN1 / 2; i += BS_NT) {
                uint32_t v = reinterpret_cast<uint32_t *>(b_tab)[i];
                v |= ((uint32_t)b_b0[v & 1023u] << 10) | ((uint32_t)b_b0[(v >> 16) & 1023u] << 26);
                reinterpret_cast<uint32_t *>(b_tab)[i] = v;
            }
            uint64_t *stage = reinterpret_cast<uint64_t *>(b_tab + 16384);         // [SP_TILE]
            uint32_t *tcnt0 = b_tab + 32768;                                       // [BS_MAX_BK]
            uint32_t *delta = b_first;                                             // [BS_MAX_BK] (dead since bucket_setup)
            // tile counts, then offsets: two arrays used by alternate tiles (the one not in use is zeroed during the
            // other's scan, which saves a barrier per tile)
            uint32_t *tcnt1 = tcnt0 + BS_MAX_BK;
            if (tid < nbk) tcnt0[tid] = tcnt1[tid] = 0;
            uint64_t hi[SP_V], lo[SP_V];
            uint32_t pv[SP_V];                                                     // bytes i0-1 .. i0+2 of the block
            auto load_tile = [&](uint32_t t0) {
#pragma unroll
                for (uint32_t u = 0; u < SP_V; u++) {
                    const uint32_t i0 = t0 + (u * BS_NT + tid) * 4;
                    hi[u] = lo[u] = 0;
                    pv[u] = 0;
                    if (i0 < n) {
                        uint64_t w2[2];
                        __builtin_memcpy(w2, P + ((i0 * bits) >> 3), 16);          // 128-bit window: 64 valid bits for all four
                        hi[u] = w2[0];
                        lo[u] = w2[1];
                        if (i0 >= 1 && i0 + 3 <= n) {
                            __builtin_memcpy(&pv[u], T + i0 - 1, 4);
                        } else {
                            const uint32_t nvalid = n - i0 < 4u ? n - i0 : 4u;
                            for (uint32_t e = 0; e < nvalid; e++) pv[u] |= (uint32_t)T[i0 + e ? i0 + e - 1 : n - 1] << (8 * e);
                        }
                    }
                }
            };
            load_tile(0);
            bzx_lds_barrier();
            uint32_t tpar = 0;
            for (uint32_t t0 = 0; t0 < n; t0 += SP_TILE, tpar ^= 1u) {
                uint32_t *tcnt = tpar ? tcnt1 : tcnt0, *tother = tpar ? tcnt0 : tcnt1;
                uint32_t kr[SP_V * 4], key[SP_V * 4], pvc[SP_V];           // bucket | place in the tile's run << 16
#pragma unroll
                for (uint32_t u = 0; u < SP_V; u++) {
                    const uint32_t i0 = t0 + (u * BS_NT + tid) * 4;
                    const uint32_t nvalid = i0 < n ? (n - i0 < 4u ? n - i0 : 4u) : 0u;
                    const uint32_t sh0 = (i0 * bits) & 7u;
                    const uint64_t h = __builtin_bswap64(hi[u]), l = __builtin_bswap64(lo[u]);
                    pvc[u] = pv[u];
#pragma unroll
                    for (uint32_t e = 0; e < 4; e++) {
                        kr[u * 4 + e] = 0xFFFFFFFFu;
                        key[u * 4 + e] = 0;
                        if (e < nvalid) {
                            const uint32_t sh = sh0 + e * bits;               // <= 31
                            const uint64_t w = sh ? (h << sh) | (l >> (64u - sh)) : h;
                            const uint32_t kb = tab16[(uint32_t)(w >> (64 - BS_BIN1))], k = kb & 1023u;
                            key[u * 4 + e] = (uint32_t)((w << (kb >> 10)) >> 32);
                            kr[u * 4 + e] = k | (atomicAdd(&tcnt[k], 1u) << 16);
                        }
                    }
                }
                if (t0 + SP_TILE < n) load_tile(t0 + SP_TILE);             // (in flight across the barriers below)
                bzx_lds_barrier();
                {
                    // exclusive scan of the tile's bucket counts (one bucket per lane; the wave totals go through a
                    // scratch row of the tile's parity, so one barrier is enough)
                    const uint32_t c = tid < nbk ? tcnt[tid] : 0u;
                    const uint32_t incl = bzx_wave_incl_sum(c);
                    uint32_t *wtot = b_scratch + tpar * BS_NW;
                    if (bzx_lane() == 63) wtot[bzx_wave()] = incl;
                    bzx_lds_barrier();
                    uint32_t pre = 0;
                    const uint32_t wv = bzx_wave();
#pragma unroll
                    for (uint32_t i = 0; i < BS_NW; i++) pre += i < wv ? wtot[i] : 0u;
                    const uint32_t ex = pre + incl - c;
                    if (tid < nbk) {
                        tcnt[tid] = ex;
                        tother[tid] = 0;
                        delta[tid] = b_start[tid] + b_cur[tid] - ex;
                        b_cur[tid] += c;
                    }
                }
                bzx_lds_barrier();
#pragma unroll
                for (uint32_t u = 0; u < SP_V; u++)
#pragma unroll
                    for (uint32_t e = 0; e < 4; e++)
                        if (kr[u * 4 + e] != 0xFFFFFFFFu) {
                            const uint32_t k = kr[u * 4 + e] & 0xFFFFu, rel = (u * BS_NT + tid) * 4 + e;
                            stage[tcnt[k] + (kr[u * 4 + e] >> 16)] = ((uint64_t)key[u * 4 + e] << 32) | ((uint64_t)k << 21) | ((uint64_t)rel << 8) |
                                                                     (uint64_t)((pvc[u] >> (8 * e)) & 255u);
                        }
                bzx_lds_barrier();
                const uint32_t tn = n - t0 < SP_TILE ? n - t0 : SP_TILE;
#pragma unroll
                for (uint32_t u = 0; u < SP_TILE / BS_NT; u++) {
                    const uint32_t i = u * BS_NT + tid;
                    if (i < tn) {
                        const uint64_t r = stage[i];
                        const uint32_t k = (uint32_t)(r >> 21) & 1023u, rel = (uint32_t)(r >> 8) & 8191u;
                        rec_a[delta[k] + i] = (r & 0xFFFFFFFF00000000ull) | ((uint64_t)(t0 + rel) << 12) | ((r & 255ull) << 4);
                    }
                }
            }
        }
        __syncthreads();
        DIAG_STAMP(101);
        bool ok = bucket_emit<BS_BIN1>(B, b, j_, nbk, 0, 0, 0, bits, 0, 0);
        DIAG_STAMP(102);

        // ---- deeper levels: every oversized bin is split again by its next BS_BIN2 bits (deep_process): by the launches
        // of the deep-split kernel that follow, any workgroup on any bin of any block, as far as the levels' lists reach
        if (ok) ok = deep_process(B, b, j_, n, bits, P, rec_a, rec_b, 0, 0, BZX_DEEP_LEVELS > 0 ? 0 : -1);
        DIAG_STAMP(103);
        if (!ok) DIAG_COUNT(107, 1);
        if (!ok && tid == 0) block_redo(B, b);
        __syncthreads();
    }
    DIAG_FLUSH();
}

// One level of the deeper splits: the oversized bins the level before (or the split kernel) listed, any workgroup
// on any bin.  Children that are still oversized go to the next level's list; after the last level a workgroup
// works its bin's subtree off itself.  An empty list: every workgroup leaves at once.
__global__ __launch_bounds__(BS_NT) void bzx_bsplit_deep_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x, lvl = B.deep_lvl;
    const uint32_t listed = B.counters[BZX_CTR_DEEP_CNT + lvl];
    const uint32_t n_items = listed < B.deep_cap ? listed : B.deep_cap;
    if (n_items == 0) return;
    DIAG_T0();
    const BzxDeepItem *list = reinterpret_cast<const BzxDeepItem *>(B.deep_list) + (size_t)(lvl & 1u) * B.deep_cap;
    for (;;) {
        if (tid == 0) b_bcast[0] = atomicAdd(&B.counters[BZX_CTR_DEEP_FETCH + lvl], 1u);
        __syncthreads();
        const uint32_t i = b_bcast[0];
        __syncthreads();
        if (i >= n_items) break;
        const BzxDeepItem d = list[i];
        if (d.cnt == 0) continue;                                              // reserved, never written
        const uint32_t b = d.blk;
        if (__atomic_load_n(&B.blk[b].status, __ATOMIC_RELAXED) & BZX_ST_REDO) continue;   // sorted from scratch anyway
        const uint32_t n = B.blk[b].n, bits = d.dbits >> 16;
        const size_t sb = BZX_SLAB(B, b);
        const uint8_t *__restrict__ P = B.pk + sb * BZX_PK_STRIDE;
        uint64_t *__restrict__ rec_a = B.rec_a + sb * BZX_MAX_N;
        uint64_t *__restrict__ rec_b = B.rec_b + sb * BZX_MAX_N;
        if (tid == 0) {
            b_bcast[3] = 1;                                                    // the ring holds this bin
            b_bcast[4] = 0;
            b_big[0][0] = d.st;
            b_big[0][1] = d.cnt;
            b_big[0][2] = d.dbits & 0xFFFFu;
        }
        __syncthreads();
        const bool ok = deep_process(B, b, (b - B.blk_first) / B.blk_step, n, bits, P, rec_a, rec_b, 0, 1,
                                     lvl + 1 < BZX_DEEP_LEVELS ? (int)lvl + 1 : -1);
        if (!ok && tid == 0) block_redo(B, b);
        __syncthreads();
    }
    DIAG_FLUSH();
}

// -------------------------------------------------------------------------------------------------- sort kernel
// 512-lane workgroups, BS_E records per lane, ~40 KB of LDS each: three or four of them share a compute unit, so the
// LDS round trips and barriers of one overlap the work of the others (the kernel is latency-bound, not ALU-bound:
// rocprofv3 SQ_WAIT_ANY was 72 % of the wave cycles with one dependent LDS round trip per row).
#ifndef SK_NT
#define SK_NT 256
#endif
#define SK_NW (SK_NT / 64)
#define BS_E (BS_C / SK_NT)             // records per lane
#define SK_DB 8                         // digit bits of the LDS radix passes
#define SK_ND (1u << SK_DB)
static_assert(BS_C % SK_NT == 0 && BS_E >= 1 && BS_E <= 8, "bucket capacity");

__shared__ uint64_t s_x[BS_C];               // records, in rank order after the initial sort (never moved again)
__shared__ uint64_t s_w[BS_C + 4];           // rank p: [current 50 key bits | index into s_x of the record ranked p:14] (+ read slack)
__shared__ uint32_t s_cnt[SK_NW][SK_ND];     // per-wave digit counters
__shared__ uint32_t s_dbase[SK_ND];
__shared__ uint32_t s_part[4];
__shared__ uint64_t s_f[BS_FW + 1];          // bit p: a group starts at rank p (all set from cnt on)
__shared__ uint32_t s_med[BS_C / BS_TINY + 1];     // groups of 65..512: first rank | size << 16
__shared__ uint32_t s_large[BS_C / BS_MED + 1];    // larger groups
__shared__ uint32_t s_rc[2][4];              // per round (parity): [0] any rank tied, [1] #med, [2] #large, [3] med fetch
__shared__ uint32_t s_bc[8];                 // [2] vote, [5..6] diff
__shared__ uint32_t s_m[2];                  // lengths of the two lists of tied ranks
__shared__ uint32_t s_red[SK_NW][4];         // per wave: OR (lo, hi) and AND (lo, hi) of the bucket's records as loaded

template <int WHICH> __device__ __forceinline__ uint64_t *lds_arr() { return WHICH ? s_w : s_x; }

// (lds_order, bzx_uni, bzx_tid_here: bzx_wg.h)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return bzx_uni(v); }
__device__ __forceinline__ uint32_t tid_here() { return bzx_tid_here(); }
__device__ __forceinline__ BzxBucket uni(BzxBucket b)
{
    BzxBucket r;
    r.blk = uni(b.blk);
    r.start = uni(b.start);
    r.cnt = uni(b.cnt);
    if (r.cnt >> 31) r.cnt = 0;             // a bucket that gave up in the first launch: nothing for the fill pass
    r.dbits = uni(b.dbits);
    return r;
}

// Stable rank of digit d (valid lanes only) among everything this wave has counted in wc so far:
// old = entries with digit d in earlier rows, rank = lanes below with the same digit in this row.
// The lanes holding the same digit find each other with one ballot per digit bit (peers &= bit ? m : ~m, a single
// 3-input v_bitop3 per half on gfx950); no LDS atomics on the per-element path: LDS atomics retire about one lane
// per clock, which made OR-ing lane bits into LDS masks the bound of the whole kernel (measured).  Only the lowest
// lane of each digit adds the row's count to the wave's counter, after every lane has read it (LDS executes a
// wave's instructions in order, so nobody waits for the read before the add is issued).
__device__ __forceinline__ void wave_rank(uint32_t *wc, uint32_t d, bool valid, uint32_t lane, uint32_t &old, uint32_t &rank)
{
    uint64_t m = __ballot(valid);
    uint32_t plo = (uint32_t)m, phi = (uint32_t)(m >> 32);
#pragma unroll
    for (int b = 0; b < SK_DB; b++) {
        const uint32_t mask = 0u - ((d >> b) & 1u);               // all ones where my bit is set
        m = __ballot((d >> b) & 1u);
        plo &= ~((uint32_t)m ^ mask);
        phi &= ~((uint32_t)(m >> 32) ^ mask);
    }
    const uint64_t peers = ((uint64_t)phi << 32) | plo;
    rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    old = wc[d];
    lds_order();
    if (valid && rank == 0) atomicAdd(&wc[d], (uint32_t)__popcll(peers));
    lds_order();
}

// Stable LSD sort (8-bit digits) of A[base .. base+cnt) (A = s_x or s_w) by bits [lo, hi) with the whole workgroup;
// digits on which all agree are skipped.  Wave w owns the contiguous chunk [w*rows*64, (w+1)*rows*64), row j = 64
// consecutive elements.  Starts and ends with workgroup barriers.
template <int WHICH>
__device__ __attribute__((noinline)) void wg_radix_sort(uint32_t base, uint32_t cnt, int lo, int hi)
{
    uint64_t *A = lds_arr<WHICH>() + base;
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    const uint32_t rows = (cnt + SK_NT - 1) / SK_NT, chunk = rows * 64;
    if (tid == 0) {
        s_bc[5] = 0;
        s_bc[6] = 0;
    }
    __syncthreads();
    {
        const uint64_t a0 = A[0];
        uint64_t d = 0;
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            if (j < rows && e < cnt) d |= A[e] ^ a0;
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) d |= __shfl_xor(d, s);
        if (lane == 0 && d) {
            atomicOr(&s_bc[5], (uint32_t)d);
            atomicOr(&s_bc[6], (uint32_t)(d >> 32));
        }
    }
    __syncthreads();
    const uint64_t diff = ((uint64_t)s_bc[6] << 32) | s_bc[5];
    uint32_t *wc = s_cnt[wave];
    for (int shift = lo; shift < hi; shift += SK_DB) {
        if (((diff >> shift) & (uint64_t)(SK_ND - 1)) == 0) continue;
#pragma unroll
        for (uint32_t i = 0; i < SK_ND / 64; i++) wc[i * 64 + lane] = 0;
        lds_order();
        uint64_t v[BS_E];
        uint32_t old[BS_E], rk[BS_E];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            v[j] = j < rows && e < cnt ? A[e] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            wave_rank(wc, (uint32_t)(v[j] >> shift) & (SK_ND - 1), j < rows && e < cnt, lane, old[j], rk[j]);
        }
        __syncthreads();
        {
            // exclusive start of every digit: SK_DK consecutive digits per thread, over waves first, then over digits
            constexpr uint32_t SK_DK = SK_ND > SK_NT ? SK_ND / SK_NT : 1u;
            if (tid * SK_DK < SK_ND) {
                uint32_t tot[SK_DK], sum = 0;
#pragma unroll
                for (uint32_t i = 0; i < SK_DK; i++) {
                    uint32_t run = 0;
#pragma unroll
                    for (int w = 0; w < SK_NW; w++) {
                        const uint32_t t = s_cnt[w][tid * SK_DK + i];
                        s_cnt[w][tid * SK_DK + i] = run;
                        run += t;
                    }
                    tot[i] = run;
                    sum += run;
                }
                const uint32_t incl = bzx_wave_incl_sum(sum);
                uint32_t run = incl - sum;
#pragma unroll
                for (uint32_t i = 0; i < SK_DK; i++) {
                    s_dbase[tid * SK_DK + i] = run;
                    run += tot[i];
                }
                if (lane == 63) s_part[wave] = incl;
            }
        }
        __syncthreads();
        const uint32_t p1 = s_part[0], p2 = p1 + s_part[1], p3 = p2 + s_part[2];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            if (j < rows && e < cnt) {
                const uint32_t d = (uint32_t)(v[j] >> shift) & (SK_ND - 1);
                const uint32_t q = SK_ND > SK_NT ? 0u : d >> 6;           // wave of the thread that scanned digit d
                A[s_dbase[d] + (q == 0 ? 0u : q == 1 ? p1 : q == 2 ? p2 : p3) + wc[d] + old[j] + rk[j]] = v[j];
            }
        }
        __syncthreads();
    }
}

// Optimistic form of the pass above for the initial sort of s_x[0 .. cnt): a record's stable rank among the records of
// its wave with the same digit is the value ONE returning LDS atomic hands back -- no ballots.  That is the stable
// rank only if the LDS serves the lanes of a wave instruction that hit the same counter in lane order, which gfx950
// does in every trial (tools/ubench/lds_order.hip: 4.2e9 lane operations under load, none out of order) but which no
// manual promises.  So the caller does not trust it: it checks that the keys come out in non-decreasing order -- a
// pass that broke stability where it matters leaves an inversion of the full key, and an order that differs only among
// records with equal keys is as good as any other, since a tied group is refined as a set -- and sorts the bucket
// again with the ballot-ranked passes if they do not.  Random-address LDS atomics run at the rate of random LDS reads
// (~7 cycles per wave instruction), so a pass costs about three random LDS operations per row instead of five plus
// ~45 vector instructions of ballot ranking.
__device__ __attribute__((noinline)) void wg_radix_sort_opt(uint32_t cnt, int lo, int hi)
{
    uint64_t *A = s_x;
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    const uint32_t rows = (cnt + SK_NT - 1) / SK_NT, chunk = rows * 64;
    __syncthreads();                            // publishes s_x and the waves' OR / AND of their records (s_red)
    uint64_t diff;
    {
        uint32_t o0 = 0, o1 = 0, a0 = ~0u, a1 = ~0u;
#pragma unroll
        for (int w = 0; w < SK_NW; w++) {
            o0 |= s_red[w][0];
            o1 |= s_red[w][1];
            a0 &= s_red[w][2];
            a1 &= s_red[w][3];
        }
        diff = ((uint64_t)(o1 & ~a1) << 32) | (o0 & ~a0);        // bits on which the records differ
    }
    uint32_t *wc = s_cnt[wave];
    for (int shift = lo; shift < hi; shift += SK_DB) {
        if (((diff >> shift) & (uint64_t)(SK_ND - 1)) == 0) continue;
#pragma unroll
        for (uint32_t i = 0; i < SK_ND / 64; i++) wc[i * 64 + lane] = 0;
        lds_order();
        uint64_t v[BS_E];
        uint32_t old[BS_E];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            v[j] = j < rows && e < cnt ? A[e] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            // (rows in order: a wave's LDS instructions execute in issue order)
            old[j] = bzx_lds_ticket(&wc[(uint32_t)(v[j] >> shift) & (SK_ND - 1)], j < rows && e < cnt);
        }
        __syncthreads();
        {
            // exclusive start of every (digit, wave): one digit per thread, over waves first, then over digits; the
            // digit's start is folded into the per-wave counters
            static_assert(SK_ND <= SK_NT, "one digit per thread");
            if (tid < SK_ND) {
                uint32_t t[SK_NW], sum = 0;
#pragma unroll
                for (int w = 0; w < SK_NW; w++) {
                    t[w] = s_cnt[w][tid];
                    sum += t[w];
                }
                const uint32_t incl = bzx_wave_incl_sum(sum);
                uint32_t run = incl - sum;
#pragma unroll
                for (int w = 0; w < SK_NW; w++) {
                    s_cnt[w][tid] = run;
                    run += t[w];
                }
                if (lane == 63) s_part[wave] = incl;
            }
        }
        __syncthreads();
        const uint32_t p1 = s_part[0], p2 = p1 + s_part[1], p3 = p2 + s_part[2];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            const uint32_t e = wave * chunk + j * 64 + lane;
            if (j < rows && e < cnt) {
                const uint32_t d = (uint32_t)(v[j] >> shift) & (SK_ND - 1);
                const uint32_t q = d >> 6;                   // wave of the thread that scanned digit d
                A[wc[d] + (q == 0 ? 0u : q == 1 ? p1 : q == 2 ? p2 : p3) + old[j]] = v[j];
            }
        }
        __syncthreads();
    }
}

// The same for s_w[base .. base+s), s <= BS_MED, by ONE wave (all 64 lanes call it together; no workgroup barriers).
// TICKETS: ranks from returning LDS atomics (see wg_radix_sort_opt; the caller checks the order) instead of ballots.
template <bool TICKETS>
__device__ __attribute__((noinline)) void wave_radix_sort(uint32_t base, uint32_t s, int lo, int hi)
{
    constexpr uint32_t ROWS = BS_MED / 64;
    uint64_t *A = s_w + base;
    uint32_t *wc = s_cnt[bzx_wave()];
    const uint32_t lane = bzx_lane();
    const uint32_t rows = (s + 63u) / 64u;
    uint64_t v[ROWS];
    uint64_t diff = 0;
    {
        const uint64_t a0 = A[0];
#pragma unroll
        for (uint32_t j = 0; j < ROWS; j++) {
            const uint32_t e = j * 64 + lane;
            v[j] = e < s ? A[e] : 0ull;
            if (e < s) diff |= v[j] ^ a0;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) diff |= __shfl_xor(diff, sh);
    }
    for (int shift = lo; shift < hi; shift += SK_DB) {
        if (((diff >> shift) & (uint64_t)(SK_ND - 1)) == 0) continue;
#pragma unroll
        for (uint32_t i = 0; i < SK_ND / 64; i++) wc[i * 64 + lane] = 0;
        lds_order();
        uint32_t old[ROWS], rk[ROWS];
#pragma unroll
        for (uint32_t j = 0; j < ROWS; j++) {
            old[j] = rk[j] = 0;
            if (j < rows) {
                if (TICKETS) old[j] = bzx_lds_ticket(&wc[(uint32_t)(v[j] >> shift) & (SK_ND - 1)], j * 64 + lane < s);
                else wave_rank(wc, (uint32_t)(v[j] >> shift) & (SK_ND - 1), j * 64 + lane < s, lane, old[j], rk[j]);
            }
        }
        {
            // exclusive scan of the SK_ND counts: SK_ND/64 digits per lane
            uint32_t c[SK_ND / 64], sum = 0;
#pragma unroll
            for (uint32_t i = 0; i < SK_ND / 64; i++) {
                c[i] = wc[(SK_ND / 64) * lane + i];
                sum += c[i];
            }
            uint32_t run = bzx_wave_incl_sum(sum) - sum;
            lds_order();
#pragma unroll
            for (uint32_t i = 0; i < SK_ND / 64; i++) {
                wc[(SK_ND / 64) * lane + i] = run;
                run += c[i];
            }
        }
        lds_order();
#pragma unroll
        for (uint32_t j = 0; j < ROWS; j++)
            if (j * 64 + lane < s) A[wc[(uint32_t)(v[j] >> shift) & (SK_ND - 1)] + old[j] + rk[j]] = v[j];
        lds_order();
#pragma unroll
        for (uint32_t j = 0; j < ROWS; j++) {
            const uint32_t e = j * 64 + lane;
            v[j] = e < s ? A[e] : 0ull;
        }
        lds_order();
    }
}

__device__ __forceinline__ uint32_t fbit(uint32_t p) { return (uint32_t)(s_f[p >> 6] >> (p & 63u)) & 1u; }
__device__ __forceinline__ void fset(uint32_t p) { atomicOr((unsigned long long *)&s_f[p >> 6], 1ull << (p & 63u)); }

// Group [gs, ge) of rank p when it has at most BS_TINY members.
__device__ __forceinline__ bool tiny_bounds(uint32_t p, uint32_t &gs, uint32_t &ge)
{
    const uint32_t wi = p >> 6, bi = p & 63u;
    uint64_t w = s_f[wi] & (~0ull >> (63u - bi));
    if (w) {
        gs = wi * 64 + 63u - (uint32_t)__builtin_clzll(w);
    } else {
        if (wi == 0) return false;
        w = s_f[wi - 1];
        if (!w) return false;
        gs = (wi - 1) * 64 + 63u - (uint32_t)__builtin_clzll(w);
    }
    w = bi == 63u ? 0ull : s_f[wi] & (~0ull << (bi + 1u));
    if (w) {
        ge = wi * 64 + (uint32_t)__builtin_ctzll(w);
    } else {
        w = s_f[wi + 1];
        if (!w) return false;
        ge = (wi + 1) * 64 + (uint32_t)__builtin_ctzll(w);
    }
    return ge - gs <= BS_TINY;
}

// After a sort of s_w[base .. base+s) by bits [lo, 64): a group starts wherever those bits change.  Lanes [t, t+step, ..).
__device__ __forceinline__ void mark_changes(uint32_t base, uint32_t s, int lo, uint32_t t, uint32_t step)
{
    for (uint32_t e = 1 + t; e < s; e += step)
        if ((s_w[base + e] >> lo) != (s_w[base + e - 1] >> lo)) fset(base + e);
}

// true when a member of s_w[base .. base+s) still sits in a group of more than BS_TINY ranks
__device__ __forceinline__ bool any_big(uint32_t base, uint32_t s, uint32_t t, uint32_t step)
{
    bool big = false;
    for (uint32_t e = t; e < s; e += step) {
        uint32_t gs, ge;
        big |= !tiny_bounds(base + e, gs, ge);
    }
    return big;
}

// One wave orders s_w[base .. base+s) by bits [lo, 64) and flags the new group starts.  Ticket-ranked passes first;
// their result is used only if the keys come out non-decreasing (else: never seen; the ballot-ranked passes redo it).
__device__ __forceinline__ void wave_sort_mark(uint32_t base, uint32_t s, int lo, uint32_t lane)
{
    wave_radix_sort<true>(base, s, lo, 64);
    lds_order();
    uint32_t diff = 0;
    bool bad = false;
    for (uint32_t e = 1 + lane, k = 0; e < s; e += 64, k++) {
        const uint64_t a = s_w[base + e] >> lo, b = s_w[base + e - 1] >> lo;
        bad |= a < b;
        diff |= (uint32_t)(a != b) << k;
    }
    if (__ballot(bad)) {
        wave_radix_sort<false>(base, s, lo, 64);
        mark_changes(base, s, lo, lane, 64);
        return;
    }
    for (uint32_t e = 1 + lane, k = 0; e < s; e += 64, k++)
        if ((diff >> k) & 1u) fset(base + e);
}

// The list of tied ranks lives in the key halves of s_x, which are dead once the initial sort has set the group-start
// flags: two lists of 16-bit ranks (one read, one written), entry i in the upper dword of s_x[i].
#define LIST(which, i) (reinterpret_cast<uint16_t *>(s_x)[4u * (i) + 2u + (which)])

// Ranks that are tied (not alone in their group) -> list `to`, whose length s_m[to] must be zero: all ranks of the
// bucket (ALL, n = its size), or the entries of list `from` (n = its length).  One LDS atomic per wave and row of 64
// candidates reserves the entries; the list is a concatenation of ascending pieces in no particular order -- every
// step of a round works on the ranks, not on their order in the list.
template <bool ALL>
__device__ __attribute__((noinline)) void list_tied(uint32_t n, uint32_t from, uint32_t to)
{
    const uint32_t tid = tid_here(), lane = tid & 63u;
    const uint32_t nrow = (n + SK_NT - 1) / SK_NT;
#pragma unroll
    for (uint32_t j = 0; j < BS_E; j++) {
        if (j < nrow) {
            const uint32_t i = j * SK_NT + tid;
            uint32_t p = 0;
            bool t = false;
            if (i < n) {
                p = ALL ? i : (uint32_t)LIST(from, i);
                t = !(fbit(p) && fbit(p + 1));
            }
            const uint64_t mk = __ballot(t);
            if (mk) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_m[to], (uint32_t)__popcll(mk));
                base = bzx_bcast0(base);
                if (t) LIST(to, base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))) = (uint16_t)p;
            }
        }
    }
}

// Initial sort of the bucket's records s_x[0 .. cnt) by their 32 key bits; afterwards rank p holds record p
// (s_w[p] = p), the group-start flags say where the key differs from the predecessor's and list 0 holds the tied
// ranks.  The passes are the optimistic ones (see wg_radix_sort_opt); false: an inversion among the sorted keys --
// never seen -- and the caller hands the block to the general sorter.
__device__ __forceinline__ bool initial_sort(uint32_t cnt)
{
    const uint32_t tid = tid_here(), lane = tid & 63u, wave = tid >> 6;
    uint64_t fm[BS_E];                               // group-start flags of my wave's row j (wave-uniform)
    wg_radix_sort_opt(cnt, 64 - BS_KEYBITS, 64);
    int bad = 0;
#pragma unroll
    for (uint32_t j = 0; j < BS_E; j++) {
        const uint32_t p = j * SK_NT + tid;
        s_w[p] = p;
        const uint64_t kp = s_x[p] >> (64 - BS_KEYBITS), kq = p ? s_x[p - 1] >> (64 - BS_KEYBITS) : 0ull;
        const bool f = p >= cnt || p == 0 || kp != kq;
        bad |= p < cnt && p > 0 && kp < kq;
        fm[j] = __ballot(f);
        if (lane == 0) s_f[j * SK_NW + wave] = fm[j];
    }
    if (tid == 0) s_f[BS_FW] = ~0ull;
    if (__syncthreads_or(bad)) return false;
    // The list of tied ranks (see list_tied), straight from the flag words: rank p is tied unless a group starts at p
    // and at p + 1; the flag after a row's last rank is bit 0 of the next flag word (ranks from cnt on are all flagged).
#pragma unroll
    for (uint32_t j = 0; j < BS_E; j++) {
        const uint64_t nx = s_f[j * SK_NW + wave + 1];
        const uint64_t mk = ~(fm[j] & ((fm[j] >> 1) | (nx << 63)));
        if (mk) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_m[0], (uint32_t)__popcll(mk));
            base = bzx_bcast0(base);
            if ((mk >> lane) & 1ull) LIST(0, base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))) = (uint16_t)(j * SK_NT + tid);
        }
    }
    return true;
}

#ifndef SK_WAVES_PER_SIMD
#define SK_WAVES_PER_SIMD 4             // 128 VGPRs: no spills (at 80 the round loop spills ~40 registers to scratch)
#endif
// A bucket's records are read once: loaded with the streaming hint they do not push the block's packed text -- which
// every bucket of the block gathers from -- out of the XCD's L2.
#ifdef BZX_HIP_EMU
#define BS_LOAD_REC(p) (*(p))
#else
#define BS_LOAD_REC(p) __builtin_nontemporal_load(p)
#endif
__device__ __forceinline__ void bsort_body(const BzxBatch &B)
{
    const uint32_t tid0 = threadIdx.x;
    if (B.bsort_mode == 1 && B.counters[BZX_CTR_RESUME] == 0) return;          // fill pass: no block needs it
    DIAG_T0();
    // Work items: eight lists; workgroup g takes the items g/8, g/8 + G/8, .. of list g % 8 (~250 buckets each, so the
    // load evens out without an atomic fetch on the critical path).  The workgroups with equal g % 8 share an XCD, and
    // in a batch of many blocks a list holds whole blocks: every gather from a block's packed text then goes through
    // the same L2 (dealt over all XCDs, half of the gathers missed L2: rocprofv3 TCC_MISS).  The fixed order also makes
    // the NEXT item known early: its descriptor is loaded two iterations ahead and its records travel in registers
    // while the current bucket is sorted.
    const uint32_t G = gridDim.x >> 3;                                         // (the grid is a multiple of 8)
    const uint32_t cap8 = B.bk_cap >> 3, lx = blockIdx.x & 7u;
    const uint32_t n_lx = B.counters[BZX_CTR_BK_LIST0 + lx] < cap8 ? B.counters[BZX_CTR_BK_LIST0 + lx] : cap8;
    const uint32_t n_items = lx * cap8 + n_lx;                                 // end of my list
    uint32_t idx = lx * cap8 + (blockIdx.x >> 3);
    BzxBucket it = {0, 0, 0, 0}, nit = {0, 0, 0, 0};
    if (idx < n_items) it = uni(B.bk_list[idx]);
    if (idx + G < n_items) nit = uni(B.bk_list[idx + G]);
    uint64_t nxt[BS_E];
    uint32_t n_cur = 0, st_cur = 0;
    if (it.cnt) {
        const uint64_t *__restrict__ src = ((it.start >> 31) ? B.rec_b : B.rec_a) + BZX_SLAB(B, it.blk) * BZX_MAX_N +
                                           (it.start & 0x7fffffffu);
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) nxt[j] = j * SK_NT + tid0 < it.cnt ? BS_LOAD_REC(src + j * SK_NT + tid0) : ~0ull;
        n_cur = uni(B.blk[it.blk].n);
        st_cur = uni(__atomic_load_n(&B.blk[it.blk].status, __ATOMIC_RELAXED));
    }
    for (; idx < n_items; idx += G) {
        BzxBucket nit2 = {0, 0, 0, 0};
        if (idx + 2 * G < n_items) nit2 = B.bk_list[idx + 2 * G];      // (made scalar when it becomes `nit`)
        uint32_t tid = tid_here(), lane = tid & 63u, wave = tid >> 6;
        const uint32_t b = it.blk, cnt = it.cnt;
        const uint32_t start = it.start & 0x7fffffffu, bits = it.dbits >> 16, depth0 = it.dbits & 0xFFFFu;
        const uint8_t *__restrict__ P = B.pk + BZX_SLAB(B, b) * BZX_PK_STRIDE;
        const uint32_t nbits = n_cur * bits;
        // skipped: empty items; blocks that are sorted from scratch anyway; in the fill pass everything but the finished
        // buckets of blocks in which some other bucket gave up
        const bool skip = cnt == 0 || (cnt >> 31) || (st_cur & BZX_ST_REDO) || (B.bsort_mode == 1 && !(st_cur & BZX_ST_RESUME));
        DIAG_STAMP(64);
        if (!skip) {
            // (the bits on which the records differ, for the radix passes to skip digits all agree on: OR and AND of the
            // records of every wave, combined by the readers after the first barrier of the sort)
            uint32_t o0 = 0, o1 = 0, a0 = ~0u, a1 = ~0u;
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                s_x[j * SK_NT + tid] = nxt[j];
                if (j * SK_NT + tid < cnt) {
                    o0 |= (uint32_t)nxt[j];
                    o1 |= (uint32_t)(nxt[j] >> 32);
                    a0 &= (uint32_t)nxt[j];
                    a1 &= (uint32_t)(nxt[j] >> 32);
                }
            }
            o0 = bzx_wave_incl_or(o0);
            o1 = bzx_wave_incl_or(o1);
            a0 = bzx_wave_incl_and(a0);
            a1 = bzx_wave_incl_and(a1);
            if (lane == 63) {
                s_red[wave][0] = o0;
                s_red[wave][1] = o1;
                s_red[wave][2] = a0;
                s_red[wave][3] = a1;
            }
        }
        // The next bucket's records travel in registers while this one is finished: the loads are issued after the
        // refinement rounds (whose register needs leave no room for them: issued at the top of the iteration they were
        // spilled to scratch and reloaded), and land during the write-out and the next bucket's first barrier.
        uint32_t n_nx = 0, st_nx = 0;
#define PREFETCH_NEXT()                                                                                                   \
        if (nit.cnt) {                                                                                                    \
            const uint64_t *__restrict__ src = ((nit.start >> 31) ? B.rec_b : B.rec_a) + BZX_SLAB(B, nit.blk) * BZX_MAX_N + \
                                               (nit.start & 0x7fffffffu);                                                 \
            const uint32_t tp_ = tid_here();                                                                              \
            _Pragma("unroll") for (uint32_t j = 0; j < BS_E; j++) nxt[j] = j * SK_NT + tp_ < nit.cnt ? BS_LOAD_REC(src + j * SK_NT + tp_) : ~0ull; \
            n_nx = B.blk[nit.blk].n;                                   /* (made scalar when it becomes n_cur) */          \
            st_nx = __atomic_load_n(&B.blk[nit.blk].status, __ATOMIC_RELAXED);                                            \
        }
        if (skip) {
            PREFETCH_NEXT();
            it = nit;
            nit = uni(nit2);
            n_cur = uni(n_nx);
            st_cur = uni(st_nx);
            continue;
        }
        if (tid == 0) s_rc[0][0] = s_rc[0][1] = s_rc[0][2] = s_rc[0][3] = s_m[0] = s_m[1] = 0;
        DIAG_STAMP(65);
        if (!initial_sort(cnt)) {                               // (its first barrier also publishes s_x)
            // the optimistic passes did not deliver a sorted bucket (see wg_radix_sort_opt): the whole block goes to
            // the general sorter, which needs nothing from here
            if (tid == 0) {
                atomicAdd(&B.counters[BZX_CTR_STAT0 + 15], 1u);
                block_redo(B, b);
            }
            PREFETCH_NEXT();
            __syncthreads();
            it = nit;
            nit = uni(nit2);
            n_cur = uni(n_nx);
            st_cur = uni(st_nx);
            continue;
        }
        DIAG_STAMP(67);

        // ---- refinement rounds over the LIST of tied ranks.  After 47 key bits most ranks are alone in their group (83 %
        // on text), so the rounds do not walk the bucket: the tied ranks are compacted into a list and every step of a
        // round -- gather, group tiers, write-back -- runs over list entries only, with all lanes busy; the list is
        // filtered again after every round, and an empty list ends the bucket without a checking round.
        uint32_t dcur = depth0 + BS_KEYBITS;
        bool fail = false;
        uint32_t lpar = 0;
        __syncthreads();                                          // (list 0: written by initial_sort)
        DIAG_STAMP(73);
        for (uint32_t round = 0;; round++) {
            tid = tid_here();
            lane = tid & 63u;
            wave = tid >> 6;
            const uint32_t par = round & 1u;
            const uint32_t m = uni(s_m[lpar]);                       // tied ranks
            if (m == 0) break;
            if (!(round < BS_ROUNDS && dcur < nbits)) {                // deep repeats / identical rotations: rank rounds
                DIAG_COUNT(round >= BS_ROUNDS ? 84 : 85, 1);
                fail = true;
                break;
            }
            DIAG_COUNT(80, 1);
            DIAG_COUNT(88, m);
            const uint32_t nrow = (m + SK_NT - 1) / SK_NT;
            // tied ranks fetch their next 50 key bits (bit offsets wrap at the block end): all gathers of a lane
            // are issued together; groups above BS_TINY are listed while they are in flight
            uint32_t amask = 0, hmask = 0;                          // bit j: entry j of this lane exists / heads its group
            uint32_t pq[BS_E], xa[BS_E];                            // rank | position of its record in s_x << 16
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                pq[j] = 0;
                xa[j] = 0;
                if (j < nrow) {
                    const uint32_t i = j * SK_NT + tid;
                    if (i < m) {
                        const uint32_t p = LIST(lpar, i);
                        amask |= 1u << j;
                        hmask |= fbit(p) << j;
                        pq[j] = p | ((uint32_t)(s_w[p] & W_POS_MASK) << 16);
                    }
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                if ((amask >> j) & 1u) {
                    uint32_t x = REC_IDX(s_x[pq[j] >> 16]) * bits + dcur;
                    if (x >= nbits) x -= nbits;
                    xa[j] = x;
                }
            }
            uint64_t g[BS_E];
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) g[j] = ((amask >> j) & 1u) ? pk_window_bit(P, xa[j]) : 0ull;
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                if ((hmask >> j) & 1u) {                                    // first rank of its group: measure it
                    const uint32_t p = pq[j] & 0xFFFFu;
                    uint32_t wi = p >> 6;
                    uint64_t w = (s_f[wi] >> (p & 63u)) >> 1;               // flags after p
                    uint32_t e;
                    if (w) {
                        e = p + 1u + (uint32_t)__builtin_ctzll(w);
                    } else {
                        do w = s_f[++wi]; while (!w);                       // ends at the sentinel word at the latest
                        e = wi * 64 + (uint32_t)__builtin_ctzll(w);
                    }
                    const uint32_t size = e - p;
                    if (size > BS_MED) s_large[atomicAdd(&s_rc[par][2], 1u)] = p | (size << 16);
                    else if (size > BS_TINY) s_med[atomicAdd(&s_rc[par][1], 1u)] = p | (size << 16);
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++)
                if ((amask >> j) & 1u) s_w[pq[j] & 0xFFFFu] = (g[j] & ~W_POS_MASK) | (uint64_t)(pq[j] >> 16);
            if (tid == 0) {
                s_rc[par ^ 1u][0] = s_rc[par ^ 1u][1] = s_rc[par ^ 1u][2] = s_rc[par ^ 1u][3] = 0;
                s_m[lpar ^ 1u] = 0;
            }
            __syncthreads();
            DIAG_STAMP(68);
            const uint32_t nmed = s_rc[par][1], nlarge = s_rc[par][2];
            if (nmed | nlarge) {
                DIAG_COUNT(86, nmed);
                DIAG_COUNT(87, nlarge);
                // larger groups: sorted by the top 16 bits of their words only; the sub-groups that leaves are almost
                // always small enough for the counting tier below, which compares whole words.  Otherwise: full sort.
                for (;;) {
                    uint32_t k = 0;
                    if (lane == 0) k = atomicAdd(&s_rc[par][3], 1u);
                    k = __shfl(k, 0);
                    if (k >= nmed) break;
                    const uint32_t e = s_med[k], gs = e & 0xFFFFu, sz = e >> 16;
                    wave_sort_mark(gs, sz, 48, lane);
                    lds_order();
                    if (__ballot(any_big(gs, sz, lane, 64))) wave_sort_mark(gs, sz, 14, lane);
                }
                for (uint32_t k = 0; k < nlarge; k++) {
                    const uint32_t e = s_large[k], gs = e & 0xFFFFu, sz = e >> 16;
                    wg_radix_sort<1>(gs, sz, 48, 64);
                    mark_changes(gs, sz, 48, tid, SK_NT);
                    if (tid == 0) s_bc[2] = 0;
                    __syncthreads();
                    if (any_big(gs, sz, tid, SK_NT)) s_bc[2] = 1;
                    __syncthreads();
                    if (s_bc[2]) {
                        wg_radix_sort<1>(gs, sz, 14, 64);
                        mark_changes(gs, sz, 14, tid, SK_NT);
                    }
                }
                __syncthreads();
                DIAG_STAMP(69);
            }
            tid = tid_here();
            lane = tid & 63u;
            wave = tid >> 6;
            // groups of up to BS_TINY ranks: every lane ranks the word at its list entry among its group's.  One entry
            // at a time, four members per step (four LDS reads in flight): an entry costs as many steps as the largest
            // group among the wave's 64 entries needs, and most hold only groups of a few ranks.
            uint32_t dst_[BS_E];                                  // new rank | first-of-sub-group flag << 31; ~0: untouched
            uint64_t my[BS_E];
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                dst_[j] = 0xFFFFFFFFu;
                my[j] = 0;
                uint32_t a_ = 0, sz = 0;
                if ((amask >> j) & 1u) {
                    const uint32_t p = pq[j] & 0xFFFFu;
                    uint32_t e_;
                    if (tiny_bounds(p, a_, e_) && e_ - a_ > 1) {
                        sz = e_ - a_;
                        my[j] = s_w[p];
                    }
                }
                if (j < nrow) {
                    // r: members below my word (key, then position: a total order); rk: members below my KEY alone -- any word
                    // with a smaller key is below my word with its position bits cleared.  r == rk: nobody with my key
                    // precedes me, I start a (sub)group.
                    uint32_t r = 0, rk = 0;
                    const uint64_t myc = my[j] & ~W_POS_MASK;
                    for (uint32_t i = 0; i < sz; i += 4) {
                        uint64_t wq[4];
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) wq[k] = s_w[a_ + i + k];       // (past the group: read, not counted)
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const bool in = i + k < sz;
                            r += in && wq[k] < my[j];
                            rk += in && wq[k] < myc;
                        }
                    }
                    if (sz) dst_[j] = (a_ + r) | (r != rk ? 0u : 0x80000000u);
                }
            }
            __syncthreads();
            DIAG_STAMP(70);
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                if (dst_[j] != 0xFFFFFFFFu) {
                    const uint32_t q = dst_[j] & 0x7FFFFFFFu;
                    s_w[q] = my[j];
                    if ((dst_[j] >> 31) && !fbit(q)) fset(q);      // first of its sub-group
                }
            }
            __syncthreads();
            DIAG_STAMP(71);
            list_tied<false>(m, lpar, lpar ^ 1u);                  // the ranks of the list that are still tied
            __syncthreads();
            DIAG_STAMP(73);
            lpar ^= 1u;
            dcur += 50;
        }
        PREFETCH_NEXT();
        tid = tid_here();
        if (fail || B.bsort_mode == 1) {
            // A bucket that gave up keeps what it has: the order so far and the group starts go to the (dead) record
            // range of the bucket as [group start:1 @32 | rotation:20] -- what the general sorter resumes from if the
            // rank rounds do not finish the block -- and the block is queued for the rank rounds.  The fill pass
            // writes the same for the buckets of such a block that did finish (every rank its own group) and enters
            // their -- final -- ranks into both rank arrays of the block.
            // A bucket that gives up enters the rank of its group's first member for every rank into the first rank
            // array -- the one round 0 reads -- and for the ranks that are alone in their group (final) into the second
            // as well: the tied ones are written there by round 0.
            uint64_t *__restrict__ sax = B.rec_a + BZX_SLAB(B, b) * BZX_MAX_N + start;
            const uint32_t rk = uni(B.blk[b].n_selectors);
            uint32_t *__restrict__ isa0 = rank_array(B, rk, 0), *__restrict__ isa1 = rank_array(B, rk, 1);
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                const uint32_t p = j * SK_NT + tid;
                if (p < cnt) {
                    const uint32_t rot = REC_IDX(s_x[(uint32_t)(s_w[p] & W_POS_MASK)]);
                    const uint32_t f0 = fbit(p);
                    sax[p] = (uint64_t)rot | ((uint64_t)f0 << 32);
                    if (isa0) {
                        uint32_t head = p;
                        if (!f0) {
                            uint32_t wi = p >> 6;
                            uint64_t w = s_f[wi] & (~0ull >> (63u - (p & 63u)));
                            while (!w) w = s_f[--wi];              // (rank 0 of the bucket starts a group)
                            head = wi * 64 + 63u - (uint32_t)__builtin_clzll(w);
                        }
                        isa0[rot] = start + head;
                        if (f0 && fbit(p + 1)) isa1[rot] = start + head;
                    }
                }
            }
            if (fail) {
                // The rank rounds work on the TIED ranks only: they are written once more, compacted in rank order, as
                // 4-byte entries [rotation:20 | group start:1 | rank in the bucket:11] into the bucket's range of the
                // other record buffer (dead: records live in one buffer, their parent bin's in the other).  A round then
                // moves 4 bytes per tied rank instead of 8 bytes per rank of the bucket.
                uint32_t *__restrict__ cl = reinterpret_cast<uint32_t *>(B.rec_b + BZX_SLAB(B, b) * BZX_MAX_N + start);
                uint64_t mk[BS_E];
                lane = tid & 63u;
                wave = tid >> 6;
#pragma unroll
                for (uint32_t j = 0; j < BS_E; j++) {
                    const uint64_t f0 = s_f[j * SK_NW + wave], nx = s_f[j * SK_NW + wave + 1];
                    mk[j] = ~(f0 & ((f0 >> 1) | (nx << 63)));
                    if (lane == 0) s_cnt[0][j * SK_NW + wave] = (uint32_t)__popcll(mk[j]);
                }
                __syncthreads();
                uint32_t total = 0, before[BS_E];
#pragma unroll
                for (uint32_t j = 0; j < BS_E; j++) {
                    before[j] = 0;
#pragma unroll
                    for (uint32_t w = 0; w < SK_NW; w++) {
                        const uint32_t c = s_cnt[0][j * SK_NW + w];
                        if (w == wave) before[j] = total;
                        total += c;
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < BS_E; j++) {
                    if ((mk[j] >> lane) & 1ull) {
                        const uint32_t p = j * SK_NT + tid;
                        const uint32_t rot = REC_IDX(s_x[(uint32_t)(s_w[p] & W_POS_MASK)]);
                        cl[before[j] + (uint32_t)__popcll(mk[j] & ((1ull << lane) - 1ull))] = p | (fbit(p) << 11) | (rot << 12);
                    }
                }
                if (tid == 0) {
                    atomicMin(&B.blk[b].n_mtf, dcur / bits);
                    B.bk_list[idx].cnt = cnt | 0x80000000u;          // (the fill pass leaves this item alone)
                    const uint32_t ri = atomicAdd(&B.counters[BZX_CTR_RK_ITEMS], 1u);
                    B.rk_list[ri] = idx;
                    B.rk_list[B.bk_cap + ri] = total;                // tied ranks of the bucket
                    atomicAdd(&B.counters[BZX_CTR_RK_OPEN], 1u);
                    atomicAdd(&B.blk[b].n_groups, 1u);
                    if ((atomicOr(&B.blk[b].status, BZX_ST_RESUME) & BZX_ST_RESUME) == 0)
                        B.resume_list[atomicAdd(&B.counters[BZX_CTR_RESUME], 1u)] = b;
                }
            }
        }
        {
            // the bucket's rows of the last column: a lane takes eight consecutive ranks and writes their bytes with one
            // store (one byte per lane and store made the write path move 64 B per wave instruction: WRITE_SIZE was six
            // times the bytes written)
            constexpr uint32_t OUT_E = 8;                          // ranks per lane (lanes beyond BS_C / 8 have none)
            uint8_t *__restrict__ L = B.bwt + BZX_SLAB(B, b) * BZX_BLK_STRIDE + start;
            const uint32_t p0 = tid * OUT_E;
            if (p0 < cnt) {
                uint64_t w8[OUT_E];
#pragma unroll
                for (uint32_t k = 0; k < OUT_E; k++) w8[k] = s_w[p0 + k];
                uint64_t bytes = 0;
#pragma unroll
                for (uint32_t k = 0; k < OUT_E; k++) {
                    const uint32_t r = reinterpret_cast<const uint32_t *>(s_x)[2u * (uint32_t)(w8[k] & W_POS_MASK)];   // (low half of the record)
                    bytes |= (uint64_t)REC_PREV(r) << (8 * k);
                    if (p0 + k < cnt && REC_IDX(r) == 0) B.blk[b].orig_ptr = start + p0 + k;
                }
                if (p0 + OUT_E <= cnt) {
                    __builtin_memcpy(L + p0, &bytes, 8);
                } else {
                    for (uint32_t k = 0; p0 + k < cnt; k++) L[p0 + k] = (uint8_t)(bytes >> (8 * k));
                }
            }
            DIAG_COUNT(83, 1);
        }
        __syncthreads();
        DIAG_STAMP(72);
        it = nit;
        nit = uni(nit2);
        n_cur = uni(n_nx);
        st_cur = uni(st_nx);
    }
    DIAG_FLUSH();
}

// (two names for one body: profiles list the sort proper and the fill pass -- a no-op launch unless a block has a
// bucket that gave up -- separately)
__global__ __launch_bounds__(SK_NT, SK_WAVES_PER_SIMD) void bzx_bsort_kernel(BzxBatch B) { bsort_body(B); }
__global__ __launch_bounds__(SK_NT, SK_WAVES_PER_SIMD) void bzx_bfill_kernel(BzxBatch B) { bsort_body(B); }

// ---- rank rounds: the buckets that gave up, finished by prefix tripling ---------------------------------------
// A bucket gives up when some of its rotations still agree after BS_ROUNDS refinement rounds (deep repeats: duplicated
// files, licence headers, tables in binaries).  Its order and group starts were written back for ALL its ranks (the
// general sorter's fall-back) and once more, for the TIED ranks only, as a compact list of 4-byte entries; the fill
// pass entered the ranks of the finished buckets of the block into the block's two rank arrays.  From there on the
// leftover groups -- all inside one bucket -- are refined by the ranks of the rotations h and 2h symbols ahead (h = the
// block's smallest give-up depth, times three every round: two gathers per tied rank buy a third fewer visits of it,
// and the fixed cost of a bucket's round -- list in, barriers, list out -- is what the rounds are made of): the
// refinement round of the sort kernel with [ISA[(rotation + h) mod n], ISA[(rotation + 2h) mod n]] in place of the next
// 50 key bits, on the compact list.
//   before      : a bucket that gives up (sort kernel) enters the group-head rank of each of its ranks into rank array
//                 0, and the ranks that are alone in their group into array 1 too.
//   round r     : ONE launch.  A bucket loads its list, gathers from rank array r & 1, orders its groups, enters the
//                 new group-head ranks into rank array (r + 1) & 1 and stores the list back without the entries that
//                 left.  Nobody reads the array that is being written, so a reader never mixes ranks of two depths
//                 inside one comparison.  A rank that has become the only one of its group is final: it writes its row
//                 of the last column and its entry for the general sorter, and stays listed for ONE more round, in
//                 which it only copies its final rank into the other array (the one that was being read when it was
//                 settled) -- singletons of the list are exactly those.
// Any workgroup takes any open bucket (16 per fetch; finished ones are skipped by their flags).
#define RK_CHUNK 16
// The 16 lanes that test the items also fetch what a round needs to know about each -- bucket, block, list length:
// five dependent global loads, once for the chunk instead of one chain per bucket.
struct RkMeta {
    uint32_t bi, T, blk, start, cnt, n, rk, h0;     // item, tied ranks, block, first rank, bucket size, block size, rank arrays, give-up depth
};
__shared__ RkMeta s_meta[RK_CHUNK];

__device__ __forceinline__ uint32_t rk_fetch_chunk(const BzxBatch &B, uint32_t n_items, uint32_t &open)
{
    __shared__ uint32_t s_fetch[2];
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(&B.counters[B.rk_fetch], (uint32_t)RK_CHUNK);
        base = __shfl(base, 0);
        const uint32_t i = base + threadIdx.x;
        bool op = false;
        if (threadIdx.x < RK_CHUNK && i < n_items) {
            RkMeta m;
            m.bi = B.rk_list[i];
            m.T = B.rk_list[B.bk_cap + i];
            const BzxBucket it = B.bk_list[m.bi];
            op = !(it.dbits >> 31);
            m.blk = it.blk;
            m.start = it.start & 0x7fffffffu;
            m.cnt = it.cnt & 0x7fffffffu;
            m.n = B.blk[it.blk].n;
            m.rk = B.blk[it.blk].n_selectors;
            m.h0 = B.blk[it.blk].n_mtf;
            s_meta[threadIdx.x] = m;
        }
        const uint64_t m = __ballot(op);
        if (threadIdx.x == 0) {
            s_fetch[0] = base;
            s_fetch[1] = (uint32_t)m;
        }
    }
    __syncthreads();
    open = s_fetch[1];
    return s_fetch[0];
}

// Exclusive prefix of `v` over the slots c = j * SK_NT + tid of the workgroup in slot order (one bit per slot):
// keep[j] = bit mask of my wave's row j.  Returns the total; before[j] = kept slots before my row j.  Two barriers.
__device__ __forceinline__ uint32_t slot_prefix(const uint64_t *keep, uint32_t nrow, uint32_t *before)
{
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < BS_E; j++)
        if (j < nrow && lane == 0) s_cnt[0][j * SK_NW + wave] = (uint32_t)__popcll(keep[j]);
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (uint32_t j = 0; j < BS_E; j++) {
        before[j] = 0;
        if (j < nrow) {
#pragma unroll
            for (uint32_t w = 0; w < SK_NW; w++) {
                const uint32_t c = s_cnt[0][j * SK_NW + w];
                if (w == wave) before[j] = total;
                total += c;
            }
        }
    }
    return total;
}

__shared__ uint32_t s_stall[2 * BS_FW];       // rank rounds: group starts whose group leaves the rounds (one bit per slot)

// Slot at which the group of slot c starts (slot 0 starts a group).
__device__ __forceinline__ uint32_t group_start(uint32_t c)
{
    uint32_t wi = c >> 6;
    uint64_t w = s_f[wi] & (~0ull >> (63u - (c & 63u)));
    while (!w) w = s_f[--wi];
    return wi * 64 + 63u - (uint32_t)__builtin_clzll(w);
}

// One rank round of a bucket whose list has at most 64 entries, by ONE wave (slot = lane; no workgroup barrier, no
// flag words: the group starts are a ballot).  Most buckets of real data give up over a handful of pairs, and a
// workgroup per such bucket spent its time in barriers: four waves now take four buckets.  Same steps as the
// workgroup form below.
__device__ __forceinline__ void rank_round_wave(const BzxBatch &B, uint32_t i)
{
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    const RkMeta &mt = s_meta[i & (RK_CHUNK - 1)];          // (chunks start at multiples of RK_CHUNK)
    const uint32_t bi = bzx_bcast0(mt.bi);
    const uint32_t T = bzx_bcast0(mt.T);
    const uint32_t b = bzx_bcast0(mt.blk), start = bzx_bcast0(mt.start);
    const uint32_t rk = bzx_bcast0(mt.rk);
    uint32_t *__restrict__ isa_r = rank_array(B, rk, B.rk_h_shift & 1u);
    uint32_t *__restrict__ isa_w = rank_array(B, rk, (B.rk_h_shift & 1u) ^ 1u);
    const uint32_t n = bzx_bcast0(mt.n);
    const uint64_t h64 = (uint64_t)bzx_bcast0(mt.h0) * B.rk_h_mul;
    if (!isa_r || h64 >= n) return;
    const uint32_t h = (uint32_t)h64;
    uint32_t *__restrict__ cl = reinterpret_cast<uint32_t *>(B.rec_b + BZX_SLAB(B, b) * BZX_MAX_N + start);
    uint64_t *wk = s_w + wave * 64;                         // this wave's 64 words / rotations
    uint32_t *wr = reinterpret_cast<uint32_t *>(s_x + wave * 64);
    const uint32_t ent = lane < T ? cl[lane] : (1u << 11);
    const uint32_t rot = ent >> 12, pc = ent & 2047u;
    const uint64_t F = __ballot(lane >= T || ((ent >> 11) & 1u));          // group starts (all set from T on)
    const bool f1 = lane == 63 ? true : (F >> (lane + 1)) & 1ull;
    const bool single = lane < T && ((F >> lane) & 1ull) && f1;
    const bool tied = lane < T && !single;
    if (single) isa_w[rot] = start + pc;                                   // settled last round: the other array's copy
    uint32_t x = rot + h;
    if (x >= n) x -= n;
    uint32_t x2 = x + h;
    if (x2 >= n) x2 -= n;
    const uint32_t ahead = tied ? isa_r[x] : 0u, ahead2 = tied ? isa_r[x2] : 0u;      // ranks h and 2h symbols ahead
    const uint32_t gs = 63u - (uint32_t)__builtin_clzll((F & (~0ull >> (63u - lane))) | 1ull);
    const uint64_t above = lane == 63 ? 0ull : F >> (lane + 1);
    const uint32_t ge = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 64u;
    // A group in which a rank read is coarse cannot be ordered by this round, nor by a later one (its ranks would be
    // shallower than the round assumes): it leaves the rounds as it is.  Its members' entries in BOTH rank arrays get
    // RK_COARSE (same rank value as before: a reader of the array being read sees the old entry or the marked one, and
    // either is right for it), their rows go out in the resume format, they drop out of the list, and the block keeps
    // one more open "bucket" than it will ever close: the general sorter finishes it.
    const uint64_t C = __ballot((ahead | ahead2) >> 31);
    const bool stall = tied && (C & (~0ull >> (64u - ge)) & (~0ull << gs)) != 0;
    const uint64_t key = tied ? (stall ? 0ull : ((uint64_t)ahead << 28) | ((uint64_t)ahead2 << 8)) | lane : 0ull;   // (ranks, then slot: a stable order)
    lds_order();
    wk[lane] = key;
    lds_order();
    uint32_t r = 0, eq = 0;
    const uint32_t sz = tied ? ge - gs : 0u;
    for (uint32_t k = 0; __ballot(k < sz); k++) {
        if (k < sz) {
            const uint64_t o = wk[gs + k];
            const bool lt = o < key;
            r += lt;
            eq += lt && (o >> 8) == (key >> 8);
        }
    }
    // the element of slot `lane` moves to slot gs + r; it starts a (sub)group there unless an equal rank precedes it
    lds_order();
    if (tied) wr[2 * (gs + r)] = rot | (eq ? 0u : 1u << 20);
    lds_order();
    const uint32_t nv = tied ? wr[2 * lane] : 0u;
    const uint32_t rot_n = nv & 0xFFFFFu;
    const uint64_t Fn = __ballot(!tied || ((F >> lane) & 1ull) || ((nv >> 20) & 1u));
    const uint32_t hc = 63u - (uint32_t)__builtin_clzll((Fn & (~0ull >> (63u - lane))) | 1ull);
    const uint32_t p_head = (uint32_t)__shfl((int)pc, (int)hc);
    const bool f0n = (Fn >> lane) & 1ull, f1n = lane == 63 ? true : (Fn >> (lane + 1)) & 1ull;
    uint32_t out = 0;
    if (stall) {
        isa_r[rot_n] = isa_w[rot_n] = (start + p_head) | RK_COARSE;
        uint64_t *__restrict__ sax = B.rec_a + BZX_SLAB(B, b) * BZX_MAX_N + start;
        sax[pc] = (uint64_t)rot_n | ((uint64_t)f0n << 32);
    } else if (tied) {
        isa_w[rot_n] = start + p_head;
        out = pc | ((uint32_t)f0n << 11) | (rot_n << 12);
        uint64_t *__restrict__ sax = B.rec_a + BZX_SLAB(B, b) * BZX_MAX_N + start;
        if (f0n && f1n) {                                                  // settled
            const uint8_t *__restrict__ Tx = BZX_BLOCK_PTR(B, B.blk[b]);
            uint8_t *__restrict__ L = B.bwt + BZX_SLAB(B, b) * BZX_BLK_STRIDE + start;
            L[pc] = Tx[rot_n ? rot_n - 1 : n - 1];
            sax[pc] = (uint64_t)rot_n | (1ull << 32);
            if (rot_n == 0) B.blk[b].orig_ptr = start + pc;
        } else if (B.rk_last) {
            sax[pc] = (uint64_t)rot_n | ((uint64_t)f0n << 32);
        }
    }
    const uint64_t keep = __ballot(tied && !stall);
    if (tied && !stall) cl[(uint32_t)__popcll(keep & ((1ull << lane) - 1ull))] = out;
    if (lane == 0) {
        const uint32_t Tn = (uint32_t)__popcll(keep);
        B.rk_list[B.bk_cap + i] = Tn;
        if (C) atomicAdd(&B.blk[b].n_groups, 1u);
        if (Tn == 0) {
            B.bk_list[bi].dbits |= 0x80000000u;
            atomicSub(&B.counters[BZX_CTR_RK_OPEN], 1u);
            if (atomicSub(&B.blk[b].n_groups, 1u) == 1u) atomicAnd(&B.blk[b].status, ~BZX_ST_RESUME);
        }
    }
    lds_order();
}

__global__ __launch_bounds__(SK_NT, SK_WAVES_PER_SIMD) void bzx_brank_round_kernel(BzxBatch B)
{
    if (B.counters[BZX_CTR_RK_OPEN] == 0) return;
    const uint32_t n_items = B.counters[BZX_CTR_RK_ITEMS];
    uint32_t chunk0 = 0, open_items = 0;
    DIAG_T0();
    for (;;) {
        if (open_items == 0) {
            chunk0 = uni(rk_fetch_chunk(B, n_items, open_items));
            open_items = uni(open_items);
            if (chunk0 >= n_items) break;
            if (open_items) open_items |= 0x80000000u;               // (RK_CHUNK <= 16: bit 31 marks a fresh chunk)
            continue;
        }
        if (open_items & 0x80000000u) {
            // a fresh chunk: the lists of at most 64 entries go to single waves (wave w takes the w-th, (w+4)-th, ..
            // of them), what is left to the whole workgroup below
            open_items &= 0x7FFFFFFFu;
            uint32_t small = 0, seen = 0;
            for (uint32_t m = open_items; m; m &= m - 1u) {
                const uint32_t k = (uint32_t)__builtin_ctz(m);
                if (uni(s_meta[k].T) <= 64u) {
                    small |= 1u << k;
                    if ((seen++ & (SK_NW - 1)) == (threadIdx.x >> 6)) rank_round_wave(B, chunk0 + k);
                }
            }
            open_items &= ~small;
            DIAG_COUNT(111, __builtin_popcount(small));
            DIAG_COUNT(113 + (B.rk_h_shift < 14 ? B.rk_h_shift : 14), __builtin_popcount(small | open_items));
            __syncthreads();
            continue;
        }
        const uint32_t i = chunk0 + (uint32_t)__builtin_ctz(open_items);
        open_items &= open_items - 1u;
        const RkMeta &mt = s_meta[i - chunk0];
        const uint32_t bi = uni(mt.bi);
        const uint32_t T = uni(mt.T);
        const uint32_t b = uni(mt.blk), start = uni(mt.start);
        const uint32_t rk = uni(mt.rk);
        uint32_t *__restrict__ isa_r = rank_array(B, rk, B.rk_h_shift & 1u);
        uint32_t *__restrict__ isa_w = rank_array(B, rk, (B.rk_h_shift & 1u) ^ 1u);
        const uint32_t n = uni(mt.n);
        const uint64_t h64 = (uint64_t)uni(mt.h0) * B.rk_h_mul;
        if (!isa_r || h64 >= n) continue;                           // (left to the general sorter: see bzx_launch_brank)
        const uint32_t h = (uint32_t)h64;
        uint32_t *__restrict__ cl = reinterpret_cast<uint32_t *>(B.rec_b + BZX_SLAB(B, b) * BZX_MAX_N + start);
        uint32_t tid = tid_here(), lane = tid & 63u, wave = tid >> 6;
        const uint32_t nrow = (T + SK_NT - 1) / SK_NT;
        DIAG_COUNT(110, 1);
        DIAG_COUNT(112, T);
        // ---- load the list: slot c holds [rotation | group start | rank p]; p stays with the slot, rotations move
        uint32_t ent[BS_E];
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            ent[j] = 1u << 11;
            const uint32_t c = j * SK_NT + tid;
            if (j < nrow) {
                if (c < T) ent[j] = cl[c];
                s_x[c] = ent[j] >> 12;
                LIST(0, c) = (uint16_t)(ent[j] & 2047u);
                s_w[c] = c;
            }
            const uint64_t m = __ballot(c >= T || ((ent[j] >> 11) & 1u));
            if (lane == 0) s_f[j * SK_NW + wave] = m;                // (rows beyond the list: all flagged)
        }
        if (tid == 0) {
            s_f[BS_FW] = ~0ull;
            s_rc[0][0] = s_rc[0][1] = s_rc[0][2] = s_rc[0][3] = 0;
        }
        __syncthreads();
        // ---- tied slots fetch the rank of the rotation h ahead; single slots (settled last round) copy their final rank
        uint32_t tmask = 0, hmask = 0, zmask = 0;
        uint32_t xa[BS_E];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            xa[j] = 0;
            const uint32_t c = j * SK_NT + tid;
            if (j < nrow && c < T) {
                const uint32_t f0 = fbit(c), f1 = fbit(c + 1);
                if (f0 && f1) {
                    zmask |= 1u << j;
                } else {
                    tmask |= 1u << j;
                    hmask |= f0 << j;
                    uint32_t x = (ent[j] >> 12) + h;
                    if (x >= n) x -= n;
                    xa[j] = x;
                }
            }
        }
        uint64_t g[BS_E];                                            // [rank h ahead:20 @44 | rank 2h ahead:20 @24]
        uint32_t cmask = 0;                                          // slots that read a coarse rank
        {
            uint32_t g1[BS_E], g2[BS_E];
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                uint32_t x2 = xa[j] + h;
                if (x2 >= n) x2 -= n;
                g1[j] = ((tmask >> j) & 1u) ? isa_r[xa[j]] : 0u;
                g2[j] = ((tmask >> j) & 1u) ? isa_r[x2] : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < BS_E; j++) {
                cmask |= ((g1[j] | g2[j]) >> 31) << j;
                g[j] = ((uint64_t)(g1[j] & ~RK_COARSE) << 44) | ((uint64_t)(g2[j] & ~RK_COARSE) << 24);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++)
            if ((zmask >> j) & 1u) isa_w[ent[j] >> 12] = start + (ent[j] & 2047u);
        // groups in which a rank read is coarse leave the rounds (see rank_round_wave): the group start of every such
        // read is marked, then every tied slot looks its own group start up
        uint32_t smask = 0;
        {
            if (__syncthreads_or((int)cmask)) {
                if (tid < 2 * BS_FW) s_stall[tid] = 0;
                __syncthreads();
#pragma unroll
                for (uint32_t j = 0; j < BS_E; j++) {
                    if ((cmask >> j) & 1u) {
                        const uint32_t hc = group_start(j * SK_NT + tid);
                        atomicOr(&s_stall[hc >> 5], 1u << (hc & 31u));
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t j = 0; j < BS_E; j++) {
                    if ((tmask >> j) & 1u) {
                        const uint32_t hc = group_start(j * SK_NT + tid);
                        if ((s_stall[hc >> 5] >> (hc & 31u)) & 1u) {
                            smask |= 1u << j;
                            g[j] = 0;                                  // (equal keys: the group stays as it is)
                        }
                    }
                }
                if (tid == 0) atomicAdd(&B.blk[b].n_groups, 1u);       // (never closed: the block stays in resume state)
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            if ((hmask >> j) & 1u) {
                const uint32_t c = j * SK_NT + tid;
                uint32_t wi = c >> 6;
                uint64_t w = (s_f[wi] >> (c & 63u)) >> 1;
                uint32_t e;
                if (w) {
                    e = c + 1u + (uint32_t)__builtin_ctzll(w);
                } else {
                    do w = s_f[++wi]; while (!w);
                    e = wi * 64 + (uint32_t)__builtin_ctzll(w);
                }
                const uint32_t size = e - c;
                if (size > BS_MED) s_large[atomicAdd(&s_rc[0][2], 1u)] = c | (size << 16);
                else if (size > BS_TINY) s_med[atomicAdd(&s_rc[0][1], 1u)] = c | (size << 16);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++)
            if ((tmask >> j) & 1u) s_w[j * SK_NT + tid] = g[j] | (uint64_t)(j * SK_NT + tid);
        __syncthreads();
        const uint32_t nmed = s_rc[0][1], nlarge = s_rc[0][2];
        if (nmed | nlarge) {
            for (;;) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&s_rc[0][3], 1u);
                k = __shfl(k, 0);
                if (k >= nmed) break;
                const uint32_t e = s_med[k], gs = e & 0xFFFFu, sz = e >> 16;
                wave_sort_mark(gs, sz, 48, lane);
                lds_order();
                if (__ballot(any_big(gs, sz, lane, 64))) wave_sort_mark(gs, sz, 14, lane);
            }
            for (uint32_t k = 0; k < nlarge; k++) {
                const uint32_t e = s_large[k], gs = e & 0xFFFFu, sz = e >> 16;
                wg_radix_sort<1>(gs, sz, 48, 64);
                mark_changes(gs, sz, 48, tid, SK_NT);
                if (tid == 0) s_bc[2] = 0;
                __syncthreads();
                if (any_big(gs, sz, tid, SK_NT)) s_bc[2] = 1;
                __syncthreads();
                if (s_bc[2]) {
                    wg_radix_sort<1>(gs, sz, 14, 64);
                    mark_changes(gs, sz, 14, tid, SK_NT);
                }
            }
            __syncthreads();
        }
        tid = tid_here();
        lane = tid & 63u;
        wave = tid >> 6;
        // groups of up to BS_TINY slots: every lane ranks its own word among its group's
        uint32_t dst_[BS_E];
        uint64_t my[BS_E];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            dst_[j] = 0xFFFFFFFFu;
            my[j] = 0;
            uint32_t a_ = 0, sz = 0;
            if ((tmask >> j) & 1u) {
                const uint32_t c = j * SK_NT + tid;
                uint32_t e_;
                if (tiny_bounds(c, a_, e_) && e_ - a_ > 1) {
                    sz = e_ - a_;
                    my[j] = s_w[c];
                }
            }
            if (j < nrow) {
                uint32_t r = 0, rk = 0;                   // (see the sort kernel's counting tier)
                const uint64_t myc = my[j] & ~W_POS_MASK;
                for (uint32_t i2 = 0; i2 < sz; i2 += 4) {
                    uint64_t wq[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) wq[k] = s_w[a_ + i2 + k];          // (past the group: read, not counted)
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const bool in = i2 + k < sz;
                        r += in && wq[k] < my[j];
                        rk += in && wq[k] < myc;
                    }
                }
                if (sz) dst_[j] = (a_ + r) | (r != rk ? 0u : 0x80000000u);
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            if (dst_[j] != 0xFFFFFFFFu) {
                const uint32_t q = dst_[j] & 0x7FFFFFFFu;
                s_w[q] = my[j];
                if ((dst_[j] >> 31) && !fbit(q)) fset(q);
            }
        }
        __syncthreads();
        // ---- the new state of every slot that was tied: its rotation, its group's head rank into the array being
        // written; a slot that is alone now is settled (last column, general sorter's entry, row of rotation 0)
        const uint8_t *__restrict__ Tx = BZX_BLOCK_PTR(B, B.blk[b]);
        uint8_t *__restrict__ L = B.bwt + BZX_SLAB(B, b) * BZX_BLK_STRIDE + start;
        uint64_t *__restrict__ sax = B.rec_a + BZX_SLAB(B, b) * BZX_MAX_N + start;
        uint64_t keep[BS_E];
        uint32_t out[BS_E];
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++) {
            out[j] = 0;
            bool kp = false;
            const uint32_t c = j * SK_NT + tid;
            if ((tmask >> j) & 1u) {
                const uint32_t rot = (uint32_t)s_x[(uint32_t)(s_w[c] & W_POS_MASK)] & 0xFFFFFu;
                const uint32_t f0 = fbit(c), f1 = fbit(c + 1), pc = LIST(0, c);
                uint32_t wi = c >> 6;
                uint64_t w = s_f[wi] & (~0ull >> (63u - (c & 63u)));
                while (!w) w = s_f[--wi];                          // (slot 0 starts a group)
                const uint32_t hc = wi * 64 + 63u - (uint32_t)__builtin_clzll(w);
                if ((smask >> j) & 1u) {                           // leaves the rounds, coarse
                    isa_r[rot] = isa_w[rot] = (start + LIST(0, hc)) | RK_COARSE;
                    sax[pc] = (uint64_t)rot | ((uint64_t)f0 << 32);
                } else {
                    isa_w[rot] = start + LIST(0, hc);
                    out[j] = pc | (f0 << 11) | (rot << 12);
                    kp = true;
                    if (f0 && f1) {                                // settled
                        L[pc] = Tx[rot ? rot - 1 : n - 1];
                        sax[pc] = (uint64_t)rot | (1ull << 32);
                        if (rot == 0) B.blk[b].orig_ptr = start + pc;
                    } else if (B.rk_last) {
                        sax[pc] = (uint64_t)rot | ((uint64_t)f0 << 32);   // (still tied after the last round: general sorter)
                    }
                }
            }
            keep[j] = __ballot(kp);
        }
        uint32_t before[BS_E];
        const uint32_t Tn = slot_prefix(keep, nrow, before);
#pragma unroll
        for (uint32_t j = 0; j < BS_E; j++)
            if (j < nrow && ((keep[j] >> lane) & 1ull)) cl[before[j] + (uint32_t)__popcll(keep[j] & ((1ull << lane) - 1ull))] = out[j];
        if (tid == 0) {
            B.rk_list[B.bk_cap + i] = Tn;
            if (Tn == 0) {
                B.bk_list[bi].dbits |= 0x80000000u;
                atomicSub(&B.counters[BZX_CTR_RK_OPEN], 1u);
                if (atomicSub(&B.blk[b].n_groups, 1u) == 1u) atomicAnd(&B.blk[b].status, ~BZX_ST_RESUME);
            }
        }
        __syncthreads();
    }
    DIAG_FLUSH();
}

// ---- regrouping pass: the oversized groups the split gave up on, before the rank rounds ------------------------------
// Such a group (G > BS_C rotations that agree on the split's whole depth) does not fit a workgroup's round, and left
// alone it poisons its block: its coarse ranks stall every group that reads them, those stall their readers, and in a
// block that repeats itself the general sorter ends up with half a million tied ranks (python sources: 15 ms for one
// block).  But round 0 would only SORT the group by the ranks h0 and 2 h0 symbols ahead, and a sort can be started by a
// partition: the members are dealt into sub-buckets by the rank h0 ahead (4,096 bins over the range the ranks of the
// members span, adjacent bins merged into buckets of at most BS_C as in the split) -- sub-bucket k holds smaller ranks
// than sub-bucket k + 1, inside a sub-bucket the order is open.  A sub-bucket that is still too big is dealt again: by
// the same rank over its narrower span, and once all its members share that rank, by the rank 2 h0 ahead, then 3 h0,
// .. (every rank in the array is at least h0 deep wherever it is read, so the sequence of ranks h0, 2 h0, 3 h0, ..
// ahead orders the members like their text; round 0 looks at the first two and finds them equal, which is right).  Every
// sub-bucket of 2 .. BS_C members becomes an ordinary item of the rank rounds whose ranks are all tied in ONE group,
// and round 0 does the rest.  The rank arrays are not touched (round 0 readers still see the old group-head rank of
// the members, which is as deep as round 0 assumes -- and other workgroups of this launch may be reading it: a reader
// must not see some members of a group refined and others not; round 0 itself enters the new ranks into the other
// array).  What cannot be dealt -- more than BS_C members with the same BS_GIANT_KEYS ranks ahead: padding patterns -- stays one
// group with coarse ranks as before, and so does a range in which a member would end up alone between two full
// sub-buckets (its rank would be final, and a final rank has to enter both rank arrays, which only a round can do).
#ifndef BS_GIANT_KEYS
#define BS_GIANT_KEYS 16
#endif
__global__ __launch_bounds__(BS_NT) void bzx_brank_giant_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t listed = B.counters[BZX_CTR_GIANT_CNT];
    const uint32_t n_items = listed < B.deep_cap ? listed : B.deep_cap;
    if (n_items == 0) return;
    const BzxDeepItem *list = reinterpret_cast<const BzxDeepItem *>(B.deep_list) + (size_t)2 * B.deep_cap;
    constexpr uint32_t NB = 1u << BS_BIN2;
    for (;;) {
        if (tid == 0) b_bcast[0] = atomicAdd(&B.counters[BZX_CTR_GIANT_FETCH], 1u);
        __syncthreads();
        const uint32_t gi = b_bcast[0];
        __syncthreads();
        if (gi >= n_items) break;
        const BzxDeepItem d = list[gi];
        const uint32_t b = d.blk, base = d.st;
        if (__atomic_load_n(&B.blk[b].status, __ATOMIC_RELAXED) & BZX_ST_REDO) continue;   // sorted from scratch anyway
        const uint32_t n = B.blk[b].n, h0 = B.blk[b].n_mtf;
        const size_t sb = BZX_SLAB(B, b);
        uint32_t *__restrict__ isa0 = rank_array(B, (uint32_t)sb, 0), *__restrict__ isa1 = rank_array(B, (uint32_t)sb, 1);
        const uint32_t jj = (b - B.blk_first) / B.blk_step, cap8 = B.bk_cap >> 3, lx = B.bk_affine ? (jj & 7u) : (gi & 7u);
        // ranges waiting to be dealt: {first rank relative to the group, members, s: by the rank (s + 1) h0 ahead}
        if (tid == 0) {
            b_big[0][0] = 0;
            b_big[0][1] = d.cnt;
            b_big[0][2] = 0;
            b_bcast[3] = 1;
            b_bcast[11] = 0;                                                   // a part of the group stays coarse
        }
        __syncthreads();
        for (;;) {
            const uint32_t depth_ = b_bcast[3];
            __syncthreads();
            if (depth_ == 0) break;
            const uint32_t r0 = b_big[depth_ - 1][0], G = b_big[depth_ - 1][1], sel = b_big[depth_ - 1][2];
            __syncthreads();
            uint64_t *__restrict__ sax = B.rec_a + sb * BZX_MAX_N + base + r0;
            uint64_t *__restrict__ tmp = B.rec_b + sb * BZX_MAX_N + base + r0;      // (dead: the range held the parent bin's records)
            // ---- the rank ahead of every member; the span of these ranks
            for (uint32_t i = tid; i < NB + (NB >> 5); i += BS_NT) b_tab[i] = 0;
            if (tid == 0) {
                b_bcast[3] = depth_ - 1;                                       // popped
                b_bcast[1] = 0xFFFFFFFFu;
                b_bcast[2] = 0;
                b_bcast[4] = 0;
            }
            if (tid < 3) b_bcast[8 + tid] = 0;
            __syncthreads();
            {
                uint32_t mn = 0xFFFFFFFFu, mx = 0;
                for (uint32_t i = tid; i < G; i += BS_NT) {
                    const uint32_t rot = (uint32_t)sax[i] & 0xFFFFFu;
                    const uint32_t x = (uint32_t)((rot + (uint64_t)(sel + 1u) * h0) % n);
                    const uint32_t k1 = isa0[x] & ~RK_COARSE;
                    tmp[i] = ((uint64_t)k1 << 32) | rot;
                    mn = k1 < mn ? k1 : mn;
                    mx = k1 > mx ? k1 : mx;
                }
                atomicMin(&b_bcast[1], mn);
                atomicMax(&b_bcast[2], mx);
            }
            __syncthreads();
            const uint32_t lo = b_bcast[1], span = b_bcast[2] - lo;
            uint32_t nbk = 0;
            if (span) {
                const uint32_t sh = span < NB ? 0u : 32u - (uint32_t)__builtin_clz(span) - BS_BIN2;      // (span >> sh) < NB
                for (uint32_t i = tid; i < G; i += BS_NT) atomicAdd(&TAB(((uint32_t)(tmp[i] >> 32) - lo) >> sh), 1u);
                __syncthreads();
                nbk = form_buckets<(int)NB>(G);
                if (nbk) {
                    // sub-buckets of one member join a neighbour; b_first[k] becomes the sub-bucket of bucket k, b_start is
                    // compacted in place; b_cur[k']: members dealt so far, later the item's place, ~0 (coarse) or ~1 (dealt again)
                    if (tid == 0) {
                        uint32_t cur = 0, cur_cnt = 0, join_next = 0;
                        for (uint32_t k = 0; k < nbk; k++) {
                            const uint32_t st_k = b_start[k], c = b_start[k + 1] - st_k;
                            const bool fresh = k == 0 || !(join_next || (c == 1 && cur_cnt + 1 <= BS_C));
                            join_next = 0;
                            if (fresh) {
                                if (k) cur++;
                                b_start[cur] = st_k;
                                cur_cnt = 0;
                                if (c == 1 && k + 1 < nbk && 1 + (b_start[k + 2] - b_start[k + 1]) <= BS_C) join_next = 1;
                            }
                            cur_cnt += c;
                            b_first[k] = cur;
                        }
                        b_start[cur + 1] = G;
                        b_bcast[5] = cur + 1;
                    }
                    __syncthreads();
                    const uint32_t nsub_ = b_bcast[5];
                    for (uint32_t k = tid; k < nsub_; k += BS_NT) b_cur[k] = 0;
                    __syncthreads();
                    for (uint32_t i = tid; i < G; i += BS_NT) {
                        const uint64_t v = tmp[i];
                        const uint32_t k = b_first[TAB(((uint32_t)(v >> 32) - lo) >> sh)];
                        const uint32_t slot = atomicAdd(&b_cur[k], 1u);
                        sax[b_start[k] + slot] = (v & 0xFFFFFull) | (slot == 0 ? 1ull << 32 : 0ull);
                    }
                    __syncthreads();
                }
            }
            if (nbk == 0) {
                // all members share this rank: dealt by the next one, or -- that one shared too, or more than BS_MAX_BK
                // buckets -- the range stays one group, coarse
                if (tid == 0) {
                    if (span == 0 && sel + 1u < BS_GIANT_KEYS) {
                        const uint32_t q = b_bcast[3];
                        b_big[q][0] = r0;
                        b_big[q][1] = G;
                        b_big[q][2] = sel + 1u;
                        b_bcast[3] = q + 1;
                    } else {
                        b_bcast[4] = 1;
                        b_bcast[11] = 1;
                    }
                }
                __syncthreads();
                if (b_bcast[4]) {
                    for (uint32_t i = tid; i < G; i += BS_NT) {
                        const uint32_t rot = (uint32_t)sax[i] & 0xFFFFFu;
                        isa0[rot] = isa1[rot] = base | RK_COARSE;
                        sax[i] = (uint64_t)rot | (i == 0 ? 1ull << 32 : 0ull);
                    }
                }
                __syncthreads();
                continue;
            }
            const uint32_t nsub = b_bcast[5];
            // ---- items for the sub-buckets of 2 .. BS_C members, in the block's work list; larger ones are dealt again
            if (tid == 0) {
                uint32_t nit = 0, lone = 0, nbig = 0;
                for (uint32_t k = 0; k < nsub; k++) {
                    const uint32_t c = b_start[k + 1] - b_start[k];
                    if (c >= 2 && c <= BS_C) b_cur[k] = nit++;
                    else if (c > BS_C) {
                        b_cur[k] = 0xFFFFFFFEu;
                        nbig++;
                    } else {
                        b_cur[k] = 0xFFFFFFFFu;
                        lone = 1;
                    }
                }
                uint32_t at = nit && !lone ? atomicAdd(&B.counters[BZX_CTR_BK_LIST0 + lx], nit) : 0u;
                if (lone || at + nit > cap8 || b_bcast[3] + nbig > BS_MAX_BIG) {
                    // a member alone between two full sub-buckets, or no room in a list: the range stays one group, coarse
                    for (uint32_t k = 0; k < nsub; k++) b_cur[k] = 0xFFFFFFFFu;
                    nit = 0;
                    b_bcast[4] = 1;
                    b_bcast[11] = 1;
                } else {
                    for (uint32_t k = 0; k < nsub; k++) {
                        if (b_cur[k] == 0xFFFFFFFEu) {
                            const uint32_t q = b_bcast[3];
                            b_big[q][0] = r0 + b_start[k];
                            b_big[q][1] = b_start[k + 1] - b_start[k];
                            b_big[q][2] = sel;
                            b_bcast[3] = q + 1;
                        }
                    }
                }
                b_bcast[6] = lx * cap8 + at;
                b_bcast[7] = nit ? atomicAdd(&B.counters[BZX_CTR_RK_ITEMS], nit) : 0u;
                if (nit) {
                    atomicAdd(&B.counters[BZX_CTR_RK_OPEN], nit);
                    atomicAdd(&B.blk[b].n_groups, nit);
                }
            }
            __syncthreads();
            const uint32_t item0 = b_bcast[6], ri0 = b_bcast[7], all_coarse = b_bcast[4];
            for (uint32_t k = tid; k < nsub; k += BS_NT) {
                const uint32_t ord = b_cur[k];
                if (ord < 0xFFFFFFFEu) {
                    BzxBucket it;
                    it.blk = b;
                    it.start = base + r0 + b_start[k];
                    it.cnt = (b_start[k + 1] - b_start[k]) | 0x80000000u;      // (a bucket that "gave up": the sort kernels leave it alone)
                    it.dbits = 0;
                    B.bk_list[item0 + ord] = it;
                    B.rk_list[ri0 + ord] = item0 + ord;
                    B.rk_list[B.bk_cap + ri0 + ord] = b_start[k + 1] - b_start[k];
                }
            }
            // ---- the lists of the new items: every member tied, one group
            for (uint32_t p = tid; p < G; p += BS_NT) {
                const uint32_t rot = (uint32_t)sax[p] & 0xFFFFFu;
                if (all_coarse) {
                    isa0[rot] = isa1[rot] = base | RK_COARSE;
                    sax[p] = (uint64_t)rot | (p == 0 ? 1ull << 32 : 0ull);
                    continue;
                }
                uint32_t lo_k = 0, hi_k = nsub;                                // sub-bucket of rank p: last k with b_start[k] <= p
                while (hi_k - lo_k > 1) {
                    const uint32_t mid = (lo_k + hi_k) >> 1;
                    if (b_start[mid] <= p) lo_k = mid;
                    else hi_k = mid;
                }
                const uint32_t k = lo_k, j = p - b_start[k];
                if (b_cur[k] < 0xFFFFFFFEu) {
                    uint32_t *cl = reinterpret_cast<uint32_t *>(tmp + b_start[k]);
                    cl[j] = (rot << 12) | ((j == 0 ? 1u : 0u) << 11) | j;
                }
            }
            __syncthreads();
        }
        // the open "bucket" emit_giant counted for the group: closed unless a part of it stays coarse
        if (tid == 0 && b_bcast[11] == 0 && atomicSub(&B.blk[b].n_groups, 1u) == 1u) atomicAnd(&B.blk[b].status, ~BZX_ST_RESUME);
        __syncthreads();
    }
}

void bzx_launch_bgiant(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_brank_giant_kernel, dim3(grid), dim3(BS_NT), 0, stream, B);
}

void bzx_launch_brank(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    BzxBatch R = B;
    uint32_t launch = 0;
    auto go = [&](void (*k)(BzxBatch)) {
        R.rk_fetch = BZX_CTR_RK_FETCH + launch++;
        hipLaunchKernelGGL(k, dim3(grid), dim3(SK_NT), 0, stream, R);
    };
    R.rk_h_shift = 0;
    uint32_t mul = 1;
    R.rk_last = 0;
    for (uint32_t r = 0; r < RK_ROUNDS; r++) {
        R.rk_h_shift = r;
        R.rk_h_mul = mul;
        mul = mul < 0x10000000u ? mul * 3u : mul;                    // (saturated: the product with any depth is past every block)
        R.rk_last = r + 1 == RK_ROUNDS;
        go(bzx_brank_round_kernel);
    }
    static_assert(1 + RK_ROUNDS <= BZX_N_COUNTERS - BZX_CTR_RK_FETCH, "one fetch counter per launch");
}

void bzx_launch_bsplit(const BzxBatch &B, uint32_t grid, uint32_t grid_deep, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_bsplit_kernel, dim3(grid), dim3(BS_NT), 0, stream, B);
    BzxBatch D = B;
    for (uint32_t lvl = 0; lvl < BZX_DEEP_LEVELS; lvl++) {          // (a launch with an empty list is a no-op of a few microseconds)
        D.deep_lvl = lvl;
        hipLaunchKernelGGL(bzx_bsplit_deep_kernel, dim3(grid_deep), dim3(BS_NT), 0, stream, D);
    }
}

void bzx_launch_bsort(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    grid = grid < 8 ? 8 : grid & ~7u;              // eight work lists, workgroup g on list g % 8
    if (B.bsort_mode == 1) hipLaunchKernelGGL(bzx_bfill_kernel, dim3(grid), dim3(SK_NT), 0, stream, B);
    else hipLaunchKernelGGL(bzx_bsort_kernel, dim3(grid), dim3(SK_NT), 0, stream, B);
}

uint32_t bzx_bsort_blocks_per_cu()
{
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bzx_bsort_kernel, SK_NT, 0) != hipSuccess || nb < 1) nb = 1;
    return (uint32_t)nb;
}
