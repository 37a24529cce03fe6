// bzx_device.h -- shared device/host definitions for the MI355X bzip2 block pipeline.
//
// Execution model: ONE workgroup owns ONE bzip2 block for the whole of a stage kernel and
// pulls the next block from an atomic work counter when it is done.  Blocks are independent
// (reference src/compression/compress_block.rs:3-8), so no stage needs inter-workgroup
// communication; a batch of >= 256 blocks fills the 256 CUs.  All intermediates of a block
// stay in HBM slabs addressed from the block descriptor below.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define BZX_MAX_N 900000u          // libbz2 block arrays hold 100000*level bytes
#define BZX_BLK_STRIDE 900096u     // per-block byte slab stride (multiple of 256)
#define BZX_MAX_ALPHA 258
#define BZX_G_SIZE 50
#define BZX_N_ITERS 4
#define BZX_MAX_SEL 18002          // ceil(900001 / 50)
#define BZX_SEL_STRIDE 18048

#define BZX_IN_RAW (1ull << 63)
#define BZX_ST_PERIODIC 1u         // block is u^k, k>1: identical rotations exist
#define BZX_ST_REDO 2u             // the split kernel handed the block to the general sorter, to be sorted from scratch
#define BZX_ST_RESUME 4u           // a bucket gave up (deep repeats): the general sorter finishes the leftover groups
// Rank rounds of the bucket sorter (bzx_bsort.hip): round r compares the ranks h and 2h symbols ahead, h = (give-up
// depth of the block) * 3^r, and leaves the depth at 3h.
#ifndef RK_ROUNDS
#define RK_ROUNDS 11               // depths h0 .. h0 * 3^10 (past every block for give-up depths from 16 symbols on) + one round
#endif                             // in which the ranks settled last only leave their lists
#define RK_COARSE 0x80000000u      // rank array entry: this rank is NOT refined by the rank rounds (member of an oversized
                                   // group, or of a bucket that had to read such a rank): valid at the give-up depth only
#define BZX_PK_STRIDE 900352u      // per-block stride of the packed blocks (bzx_pack.h): (n + 207 symbols) * 8 bits max
#ifndef BZX_BK_PER_BLOCK
#define BZX_BK_PER_BLOCK 2048u     // bucket work items reserved per block (average over the blocks of a batch)
#endif

// counters[] slots (BzxBatch.counters, zeroed per batch)
#define BZX_CTR_PERIODIC 5         // blocks flagged periodic
#define BZX_CTR_BK_ITEMS 8         // (unused since the work list became eight lists)
#define BZX_CTR_BK_LIST0 40        // [40..47] bucket work items in each of the eight lists (BzxBatch.bk_list)
#ifndef BZX_DEEP_LEVELS
#define BZX_DEEP_LEVELS 2          // launches of the deep-split kernel (levels of oversized bins dealt over the whole chip), <= 8:
                                   // two levels take the tail off real files (6 were no better), an empty launch costs ~30 us
#endif
#define BZX_DEEP_PER_BLOCK 128u    // items of a level's list per block of the batch (a full list: the workgroup splits on by itself)
#define BZX_CTR_DEEP_CNT 48        // [48..55] oversized bins listed for each level
#define BZX_CTR_DEEP_FETCH 56      // [56..63] ... fetched
#define BZX_CTR_GIANT_CNT (BZX_CTR_DEEP_CNT + BZX_DEEP_LEVELS)       // oversized groups listed for the regrouping pass
#define BZX_CTR_GIANT_FETCH (BZX_CTR_DEEP_FETCH + BZX_DEEP_LEVELS)   // ... fetched
#define BZX_CTR_BK_FETCH 9         // ... fetched by the bucket sort kernel
#define BZX_CTR_REDO 10            // blocks handed to the general sorter
#define BZX_CTR_SPLIT_FETCH 11     // blocks fetched by the split kernel
#define BZX_CTR_REDO_FETCH 12      // blocks fetched by the general sorter in redo mode
#define BZX_CTR_RESUME 13          // blocks whose leftover groups the general sorter finishes (resume_list)
#define BZX_CTR_RESUME_FETCH 14
#define BZX_CTR_RK_ITEMS 32         // buckets that gave up (rk_list), finished by the rank rounds
#define BZX_CTR_RK_OPEN 33          // ... of which still open
#define BZX_CTR_RK_FETCH 64         // [64..191] one work-fetch counter per launch of the rank-round kernels
#define BZX_N_COUNTERS 192
#define BZX_CTR_RESUME_LEFT 34      // resume blocks the general sorter still had to finish
#define BZX_CTR_STAT0 16           // [16..31] diagnostics of the bucket sorter (rounds, leftovers, ...)

// One bucket of rotations: ranks [start, start+cnt) of block blk, all sharing the first `dbits` bits; records
// [next 32 key bits:32 | rotation:20 | preceding byte:8 | 0:4] at rec{A,B}[blk * BZX_MAX_N + start ..].
struct BzxBucket {
    uint32_t blk;
    uint32_t start;         // bit 31: records live in recB
    uint32_t cnt;
    uint32_t dbits;         // low 16: key depth in bits; bits 16..19: bits per symbol of the block
};

// Per-block descriptor, device resident; filled by the splitter (or the host for the
// per-block entry points), completed by each stage.
struct BzxBlock {
    uint64_t in_off;        // byte offset of the RLE1'd block in the block slab buffer (BzxBatch.in); with
                            // BZX_IN_RAW set: offset into the raw input (BzxBatch.raw) -- RLE1 is the identity there
    uint32_t n;             // RLE1'd length (1..BZX_MAX_N)
    uint32_t crc;           // CRC-32/BZIP2 of the raw bytes the block covers
    uint32_t orig_ptr;      // BWT: row of rotation 0
    uint32_t status;        // BZX_ST_*
    uint32_t n_mtf;         // MTF/RLE2: number of symbols incl. EOB
    uint32_t n_in_use;      // distinct byte values in the block
    uint32_t n_groups;      // Huffman: coding tables 2..6   (bucket sorter, until then: buckets of the block still open)
    uint32_t n_selectors;   // ceil(n_mtf / 50)              (bucket sorter, until then: the block's pair of rank arrays)
    uint64_t bits;          // size of the block image in bits (header .. last payload bit)
    uint64_t out_bit;       // bit position of the block image in the output buffer
    uint32_t sec_bits[4];   // [0] selectors, [1] coding tables, [2] payload, [3] symbol map
    uint32_t pad_[2];       // [0] copies of every rotation in a periodic block, [1] debug: microseconds in the BWT kernel
    uint64_t pack_word;     // sharded runs: first 32-bit word of the block image in this rank's packed buffer
};

// Per-workgroup-slot scratch of the suffix sorter (one slot per resident workgroup).
struct BzxSortWs {
    uint64_t *u0;           // record ping  [BZX_MAX_N]
    uint64_t *u1;           // record pong  [BZX_MAX_N]
    uint32_t *s0;           // slot map ping [BZX_MAX_N]
    uint32_t *s1;           // slot map pong [BZX_MAX_N]
    uint32_t *isa;          // rank of every rotation [BZX_MAX_N]
    uint32_t *sa;           // rotation index at every sorted position [BZX_MAX_N]
};

// Everything a stage kernel needs for a batch of blocks.
struct BzxBatch {
    BzxBlock *blk;          // [nblk]
    uint32_t nblk;          // blocks this launch works on: logical j in [0,nblk) -> block blk_first + j*blk_step
    uint32_t blk_first;     // round-robin sharding over GPUs: first = rank, step = world size
    uint32_t blk_step;
    uint32_t ctr_bwt;       // index of the block-fetch counter this BWT launch uses (0, or 6 for a concurrent second launch)
    uint32_t ctr_mtf;       // same for the MTF kernel (1 or 7)
    uint32_t packed;        // emit at blk.pack_word (packed per-rank buffer) instead of blk.out_bit (final stream)
    uint32_t *counters;     // [BZX_N_COUNTERS] atomic work counters, one per stage kernel (zeroed per batch); [5] = #periodic
    uint32_t *plist;        // [nblk] indices of the blocks flagged periodic by the BWT kernel
    const uint8_t *in;      // block slab buffer (RLE1'd bytes), block b at blk[b].in_off
    const uint8_t *raw;     // raw input (blocks on which RLE1 is the identity are read in place)
    uint8_t *bwt;           // [nblk][BZX_BLK_STRIDE]  last column L
    uint8_t *rank;          // [nblk][BZX_BLK_STRIDE]  MTF stage: the heads of the runs of L (bytes), then their ranks
    uint32_t *hpos;         // [nblk][hpos_stride]     MTF stage: position of every run head in L (+ the block length behind the last)
    uint32_t hpos_stride;   // in words (the positions live in the block's record slab, which is dead after the BWT)
    uint16_t *mtfv;         // [nblk][BZX_BLK_STRIDE]  symbols (RUNA/RUNB/rank+1/EOB)
    uint32_t *freq;         // [nblk][260]             symbol histogram
    uint8_t *in_use;        // [nblk][256]
    uint8_t *len;           // [nblk][6][260]          code lengths
    uint32_t *code;         // [nblk][6][260]          canonical codes
    uint8_t *selector;      // [nblk][BZX_SEL_STRIDE]
    uint8_t *selector_mtf;  // [nblk][BZX_SEL_STRIDE]
    uint16_t *gbits;        // [nblk][BZX_SEL_STRIDE]  payload bits of every 50-symbol group
    uint32_t *out;          // output bit buffer (zeroed), big-endian bit order
    BzxSortWs *sort_ws;     // [n_slots]
    uint8_t *pk;            // [nblk][BZX_PK_STRIDE]   packed blocks (bucket sorter)
    uint64_t *rec_a;        // [nblk][BZX_MAX_N]       bucket records
    uint64_t *rec_b;        // [nblk][BZX_MAX_N]       ... of the deeper split levels
    BzxBucket *bk_list;     // [bk_cap] bucket work items (zeroed per batch): EIGHT lists of bk_cap / 8 items.  Workgroup g of the
                            // sort kernel works through list g % 8 -- the workgroups that share an XCD (MI355X deals workgroups
                            // round-robin over its 8 XCDs) -- and with bk_affine every block keeps all its buckets in one list,
                            // so the block's packed text, which its ~660 buckets gather from, is fetched into ONE L2
    uint32_t bk_cap;        // capacity of this launch (a multiple of 8)
    uint32_t bk_affine;     // 1: block j -> list j % 8 (batches of many blocks); 0: the buckets of a block are dealt over all
                            // eight lists (few blocks: every compute unit must get work)
    uint32_t *redo_list;    // [nblk] blocks for the general sorter
    uint32_t *resume_list;  // [nblk] blocks the general sorter finishes (BZX_ST_RESUME)
    uint32_t *rk_list;      // [2 * allocated items] indices into bk_list of the buckets that gave up; [bk_cap + i]: tied ranks of
                            // bucket rk_list[i] (entries of its compact list, see the rank rounds)
    uint32_t *deep_list;    // [3][deep_cap] items of 16 B: oversized bins of the level being split / of the next one (zeroed per
                            // batch); [2]: the oversized groups the split gave up on (regrouping pass before the rank rounds)
    uint32_t deep_cap;
    uint32_t deep_lvl;      // level of this launch of the deep-split kernel
    uint32_t *isa2;         // [blocks][2][BZX_MAX_N] two rank arrays per block in resume state (index: its place in resume_list)
    uint32_t rk_blocks;     // blocks that get rank arrays (all; stress builds: a few)
    uint32_t rk_fetch;      // rank rounds: this launch's work-fetch counter (index into counters)
    uint32_t rk_last;       // rank rounds: this is the last update launch
    uint32_t rk_h_shift;    // rank rounds: number of this round (it reads rank array rk_h_shift & 1 and writes the other)
    uint32_t rk_h_mul;      // rank rounds: 3 ^ round; the round compares the ranks h and 2h symbols ahead, h = (give-up depth of the block) * rk_h_mul
    uint32_t redo;          // general sorter: 1 = sort the blocks of redo_list from scratch; 2 = finish the blocks of resume_list
    uint32_t bsort_mode;    // bucket sort kernel: 0 = sort; 1 = fill pass (write the order of the finished buckets of BZX_ST_RESUME blocks)
    uint32_t n_slots;
    uint32_t slot_base;     // general sorter: workgroup g works in sort slot slot_base + g
    uint32_t redo_once;     // general sorter, early launch beside the bucket sorter: 1 = redo mode, every workgroup takes at most one block
    uint32_t rk_slot0;      // rank rounds: first sort slot that holds rank arrays
    uint32_t dbg_stop;       // diagnostics only: leave the BWT kernel after phase k (0 = run everything)
    unsigned long long *dbg; // optional [64] phase timers (100 MHz ticks), null in production
};

#define BZX_OUT_STRIDE 921600u      // per-block output slab for the per-block entry points (bytes)

// Scratch of the RLE1 / block splitter (bzx_rle1.hip).
struct BzxSplitWs {
    uint64_t *tile_rs;     // [ntiles+1] A: last run start (+1) inside the tile / after S1: carry-in run start (+1)
    uint64_t *tile_off;    // [ntiles+1] B: emitted bytes / after S2: exclusive prefix F(tile start); [ntiles] = F(len)
    uint64_t *tile_np;     // [ntiles+1] B: 1 if the tile has a run position with k >= 3 / after S3: prefix count
    uint64_t *blk_raw;     // [max_blocks+1] raw start of every block; [nblk] = len
    uint64_t *blk_f;       // [max_blocks+1] F at the block start
    uint32_t *blk_plain;   // [max_blocks+1] 1 = RLE1 is the identity on the whole block (zero-copy)
    uint32_t *nblk;        // [1]
    uint32_t max_blocks;
};


// Slab of block b.  Block DESCRIPTORS (B.blk, plist, redo_list) are indexed by the global block number; the per-block
// slabs (bwt, rank, mtfv, tables, packed block, records, RLE1 bytes) exist only for the blocks a launch owns
// (round-robin sharding over GPUs: b = blk_first + j * blk_step owns slab j), so a rank of an 8-GPU job holds 1/8
// of them.
#define BZX_SLAB(B, b) ((size_t)(((b) - (B).blk_first) / (B).blk_step))

// Block bytes of a descriptor.
#define BZX_BLOCK_PTR(B, d) (((d).in_off & BZX_IN_RAW) ? ((B).raw + ((d).in_off & ~BZX_IN_RAW)) : ((B).in + (d).in_off))
