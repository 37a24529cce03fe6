/*
 * bzx.h -- C ABI of the MI355X-native bzip2 block-compression core (libbzx.so).
 *
 * This is the drop-in boundary for the per-block hot path of ohsnyt/bzip2-rust
 * (RLE1 -> BWT -> MTF -> RLE2 -> multi-table Huffman -> bit packing).  The reference has
 * no FFI of its own; the seam is the Rust function
 *     pub fn compress_block(block: &[u8], block_crc: u32) -> (Vec<u8>, u8)
 *                                               src/compression/compress_block.rs:24
 * its producer RLE1Block (src/tools/rle1.rs:33-263) and its consumer BitWriter
 * (src/bitstream/bitwriter.rs:42-172).  Every entry point below names the reference
 * interface it replaces.  INTEGRATION.md shows the Rust `extern "C"` block a maintainer
 * would add.  Plain pointers and sizes only; all functions return 0 or a negative
 * BZX_E_* code and never unwind.
 *
 * Output bits are those of C bzip2 1.0.8 (libbz2), which BASELINE.json's metric names;
 * where the Rust reference diverges from libbz2 libbz2 wins.
 *
 * There is NO CPU implementation behind this ABI: every compute entry point needs a HIP
 * device and fails with BZX_E_NODEVICE without one.
 */
#ifndef BZX_H
#define BZX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZX_OK 0
#define BZX_E_NODEVICE (-1)   /* no HIP device / HIP runtime error at init */
#define BZX_E_PARAM (-2)      /* bad argument (null pointer, n == 0, n > 900000, level not 1..9) */
#define BZX_E_NOMEM (-3)      /* host or device allocation failed */
#define BZX_E_OUTBUF (-4)     /* output buffer too small */
#define BZX_E_HIP (-5)        /* HIP runtime error during a call (see bzx_last_error) */
#define BZX_E_STATE (-6)      /* call sequence error (stream API) */
#define BZX_E_DATA (-7)       /* decompression: not a bzip2 stream, damaged data, or a CRC mismatch (see bzx_last_error) */

#define BZX_MAX_BLOCK 900000u

typedef struct bzx_ctx bzx_ctx;

/* Library / device management (replaces the rayon global pool, compress.rs:128). */
const char *bzx_version(void);
const char *bzx_strerror(int code);
/* HIP error text of the last failing call on this context ("" if none). */
const char *bzx_last_error(const bzx_ctx *ctx);
/* device: HIP device ordinal.  max_blocks: expected batch size (slabs grow on demand). */
int bzx_ctx_create(int device, uint32_t max_blocks, bzx_ctx **out);
void bzx_ctx_destroy(bzx_ctx *ctx);
/* Run all work of this context on an existing HIP stream (hipStream_t passed as void*). */
int bzx_ctx_set_stream(bzx_ctx *ctx, void *hip_stream);

/*
 * compress_block (compress_block.rs:24-67).  blk = one RLE1'd block, crc = CRC of the raw
 * bytes it covers.  out receives the byte-aligned block image (48-bit magic, crc, randomised
 * bit, origPtr, symbol map, selectors, coding tables, payload), last byte zero padded;
 * *pad_bits = number of pad bits (0..7), i.e. BitPacker::padding (bitpacker.rs:20-21).
 * Host pointers.  cap >= n + n/50 + 1024 is always enough.
 * Thread-safe and re-entrant like the Rust function: the reference calls it from every rayon worker at once
 * (compress.rs:125-132).  Calls that arrive together on one context are collected into one device batch (the
 * first caller leads it, the others block until their block is done); a lone caller pays a 0.3 ms window.
 * All other entry points of a context are serialised against each other by an internal lock.
 */
int bzx_compress_block(bzx_ctx *ctx, const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap,
                       size_t *out_len, uint8_t *pad_bits);

/*
 * Batched form: what the rayon fan-out over blocks (compress.rs:125-132) becomes.  All
 * nblk blocks are resident on the device at once and every stage kernel runs over the
 * whole batch.  Host pointers.
 */
int bzx_compress_blocks(bzx_ctx *ctx, uint32_t nblk, const uint8_t *const *blks, const size_t *ns,
                        const uint32_t *crcs, uint8_t *const *outs, const size_t *caps, size_t *out_lens,
                        uint8_t *pads);

/*
 * Stage entry points (host pointers), one per stage function of the reference, used by the
 * parity tests to compare each device stage with the oracle:
 *   bzx_stage_bwt       bwt_encode            src/bwt_algorithms/bwt_sort.rs:27
 *   bzx_stage_mtf       rle2_mtf_encode       src/tools/rle2_mtf.rs:23
 *   bzx_stage_huffman   huf_encode tables     src/huffman_coding/huffman.rs:87-374
 */
int bzx_stage_bwt(bzx_ctx *ctx, const uint8_t *blk, size_t n, uint8_t *bwt_out, uint32_t *orig_ptr,
                  uint32_t *status);
int bzx_stage_mtf(bzx_ctx *ctx, const uint8_t *bwt, size_t n, uint16_t *mtfv_out, uint32_t *n_mtf,
                  uint32_t freq_out[258], uint8_t in_use_out[256]);
int bzx_stage_huffman(bzx_ctx *ctx, const uint16_t *mtfv, uint32_t n_mtf, const uint32_t freq[258],
                      uint32_t alpha_size, uint32_t *n_groups, uint32_t *n_selectors, uint8_t *selectors,
                      uint8_t len_out[6][258], uint32_t code_out[6][258]);

/*
 * RLE1 + block split + per-block CRC (replaces RLE1Block, rle1.rs:33-263, and do_crc,
 * crc.rs:15-22) with libbz2's split rule.  Host pointers; whole input at once.
 * blocks_out: nblk_cap slabs of BZX_MAX_BLOCK bytes; ns/crcs: per block.  *nblk = blocks made.
 */
int bzx_split_rle1(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, uint8_t *blocks_out,
                   uint32_t nblk_cap, uint32_t *ns, uint32_t *crcs, uint32_t *nblk);

/*
 * Whole buffer -> .bz2 with every stage on the device (replaces compress(), compress.rs:40-136,
 * minus file I/O).  d_raw / d_out are DEVICE pointers (HBM); nothing but the final length
 * crosses PCIe.  cap >= len + len/50 + 4096.  Alignment: d_raw 16 bytes, d_out 4 bytes (BZX_E_PARAM otherwise,
 * with the reason in bzx_last_error): a view into a larger device tensor must start on such a boundary.
 * Cost: blocks that are an exact power u^k (inputs made of one repeated byte and the like) need
 * libbz2's tie order among identical rotations: ~0.1 s for an all-zero block instead of milliseconds, and up to
 * seconds for many copies of a long unit (measured worst case over the committed sweep of units of 1..30,011 bytes:
 * 29 copies of a 30,011-byte unit, 4.1 s for that one block).  Such blocks of one call are handled side by side; the
 * context serialises its entry points, so other callers of the SAME context wait that long (use one context per
 * caller where that matters).
 */
int bzx_compress_device(bzx_ctx *ctx, const void *d_raw, size_t len, int level, void *d_out, size_t cap,
                        size_t *out_len);
/* Same with host buffers (H2D, device pipeline, D2H). */
int bzx_compress_buffer(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, uint8_t *out, size_t cap,
                        size_t *out_len);

/*
 * Multi-GPU sharding (replaces the rayon fan-out over blocks, compress.rs:125-132, across
 * devices): bzip2 block i belongs to rank i mod world.  No collective happens inside the library; the
 * caller (one process per GPU) exchanges 8 bytes per block between the two calls:
 *   1. bzx_shard_prepare: split the whole input (block boundaries are a serial dependency over the
 *      stream, so every rank derives them from its copy of the raw bytes), run BWT/MTF/Huffman on this
 *      rank's blocks, write size-in-bits | crc << 32 of each to d_bits[i] (int64, device; other entries untouched).
 *   2. caller: all-reduce(sum) d_bits over the ranks.
 *   3. bzx_shard_emit_packed: lay out the whole stream from all sizes, emit this rank's block images back to
 *      back into d_packed (each on a 32-bit word boundary, with the bit phase it has in the final stream).
 *   4. caller: gather the packed buffers to one rank (each compressed byte crosses xGMI once).
 *   5. on that rank: bzx_shard_assemble_begin (zeroed stream + "BZh<level>" + footer + combined CRC), then
 *      bzx_shard_assemble_rank once per rank: word-wise OR of the images into their final positions.
 */
int bzx_shard_prepare(bzx_ctx *ctx, const void *d_raw, size_t len, int level, uint32_t rank, uint32_t world,
                      uint32_t *nblk_total, long long *d_bits, size_t bits_cap);
/*
 * The same with the split ANALYSIS sharded too (the reference's producer touches every byte once,
 * rle1.rs:89-223): the two per-byte passes of the block splitter -- run starts, and RLE1 byte counts of the 8 KiB tiles
 * that hold runs -- run on this rank's 1/world share of the tiles only; the ranks exchange 24 bytes per tile, and only
 * the chain of block boundaries (a serial dependency over the stream, libbz2's split rule) is walked by every rank.
 * d_tiles: int64[3 * P * world] on every rank, P = bzx_shard_scan_entries(len, world); array a (a = 0, 1, 2) starts at
 * d_tiles + a * P * world and rank r owns its entries [r * P, (r + 1) * P).
 *   1a. bzx_shard_scan_runs      then caller: all-gather, in place, of the rank's P entries of array 0
 *   1b. bzx_shard_scan_counts    then caller: all-gather, in place, of the rank's P entries of arrays 1 and 2
 *   1c. bzx_shard_prepare_scanned = bzx_shard_prepare without the per-byte passes; steps 2..5 as above.
 * A rank reads the raw bytes of its own tiles (plus the 4 bytes before them) in 1a/1b and of its own blocks afterwards.
 */
size_t bzx_shard_scan_entries(size_t len, uint32_t world);
int bzx_shard_scan_runs(bzx_ctx *ctx, const void *d_raw, size_t len, uint32_t rank, uint32_t world, long long *d_tiles);
int bzx_shard_scan_counts(bzx_ctx *ctx, const void *d_raw, size_t len, uint32_t rank, uint32_t world, long long *d_tiles);
int bzx_shard_prepare_scanned(bzx_ctx *ctx, const void *d_raw, size_t len, int level, uint32_t rank, uint32_t world,
                              long long *d_tiles, uint32_t *nblk_total, long long *d_bits, size_t bits_cap);
int bzx_shard_emit_packed(bzx_ctx *ctx, const long long *d_bits_all, void *d_packed, size_t cap, size_t *packed_len,
                          size_t *stream_len);
/* After bzx_shard_emit_packed: bytes of the longest packed buffer of any rank -- the common length a gather needs --
 * computed from the sizes every rank already holds: no further collective, no extra host round trip (world <= 64). */
int bzx_shard_packed_max(bzx_ctx *ctx, size_t *max_len);
int bzx_shard_assemble_begin(bzx_ctx *ctx, void *d_out, size_t cap, size_t *stream_len);
int bzx_shard_assemble_rank(bzx_ctx *ctx, const void *d_packed_r, uint32_t r, void *d_out);
/* bzx_shard_assemble_* only enqueue work on the context's stream; wait for it here (or on the caller's stream). */
int bzx_ctx_sync(bzx_ctx *ctx);

/*
 * Streaming forms.  The reference reads its input incrementally (RLE1Block::new(source: R, ...), rle1.rs:49-85;
 * Iterator::next :245-263) and overlaps production, compression and an ordered writer (compress.rs:66-132,
 * bitwriter.rs:77-132).  Block boundaries depend on everything before them, so a chunked caller
 * cannot split chunks independently; these entry points keep the state between calls.
 *
 * bzx_split_rle1_chunk: RLE1Block over a source that arrives in pieces.  Returns the blocks that are complete with
 * the bytes seen so far (layout as bzx_split_rle1); the last, unfinished block is withheld inside the context (its
 * pending run and partial block) and comes out of a later call or of the call with final != 0.  One stream per
 * context at a time.
 */
int bzx_split_rle1_chunk(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, int final, uint8_t *blocks_out,
                         uint32_t nblk_cap, uint32_t *ns, uint32_t *crcs, uint32_t *nblk_out);

/*
 * bzx_cstream_*: whole-stream compressor for host buffers fed in chunks of at most max_chunk bytes (0 = 256 MiB);
 * device memory is bounded by the chunk size, not by the input.  Copy of chunk k+1 to the device, compression of
 * chunk k and copy-back of chunk k-1 overlap (three HIP streams, double buffers); pass page-locked buffers
 * (bzx_host_alloc, or memory the caller registered with HIP) for truly asynchronous copies.
 *   feed: consumes raw[0..len); `out`/`cap` is the WHOLE output buffer, the same on every call; *produced = bytes
 *   of it that are final so far (a caller may write out[flushed..*produced) to its file after every call).
 *   The call with final != 0 (len may be 0) completes the stream: *produced = length of the .bz2.
 * bzx_compress_buffer is this over a whole buffer.
 */
typedef struct bzx_cstream bzx_cstream;
int bzx_cstream_begin(bzx_ctx *ctx, int level, size_t max_chunk, bzx_cstream **out);
int bzx_cstream_feed(bzx_cstream *s, const uint8_t *raw, size_t len, int final, uint8_t *out, size_t cap,
                     size_t *produced);
void bzx_cstream_end(bzx_cstream *s);
void *bzx_host_alloc(size_t bytes);     /* page-locked host memory (NULL on failure) */
void bzx_host_free(void *p);

/*
 * Decompression on the device (replaces decompress(), decompress.rs:38-404 minus file I/O; bwt_decode
 * bwt_sort.rs:91-130, rle2_mtf_decode_fast rle2_mtf.rs:191-287, rle1_decode rle1.rs:267-316): one .bz2 stream ->
 * raw bytes; every block CRC and the combined CRC are verified (a mismatch is BZX_E_DATA, unlike the reference, which
 * logs it and continues, decompress.rs:379-386).  The blocks of the stream are decoded side by side.
 * _device: d_bz2 / d_out are DEVICE pointers (d_out 16-byte aligned); _buffer: host pointers.
 * BZX_E_OUTBUF: *out_len = bytes needed.  Bytes after the end-of-stream marker are ignored.
 */
int bzx_decompress_device(bzx_ctx *ctx, const void *d_bz2, size_t len, void *d_out, size_t cap, size_t *out_len);
int bzx_decompress_buffer(bzx_ctx *ctx, const uint8_t *bz2, size_t len, uint8_t *out, size_t cap, size_t *out_len);

/* Per-call telemetry of the last bzx_compress_device/_buffer/_blocks call. */
typedef struct {
    uint32_t nblk;
    uint32_t n_periodic;        /* blocks flagged periodic */
    uint64_t raw_bytes;         /* N_b summed */
    uint64_t rle1_bytes;        /* n summed */
    uint64_t mtf_symbols;       /* nMTF summed */
    uint64_t out_bits;          /* compressed bits incl. stream header/footer when present */
    float ms_split, ms_bwt, ms_mtf, ms_huffman, ms_emit, ms_total;   /* HIP-event times of the stage kernels */
    uint32_t bwt_launches;      /* launches of the BWT kernel in the last run (2 when its partial last round
                                   ran beside the MTF stage on a second stream); ms_bwt covers all of them */
    uint32_t n_redo;            /* blocks the bucket sorter handed to the general sorter (deep repeats, periodic) */
    uint32_t n_buckets;         /* bucket work items of the bucket sorter */
    float ms_bwt_split, ms_bwt_sort, ms_bwt_general;   /* parts of ms_bwt: split kernel, bucket sort kernel, everything after it */
    uint32_t n_open_buckets;      /* buckets that gave up after the refinement rounds (deep repeats) */
    uint32_t n_open_left;         /* ... whose lists of tied ranks the rank rounds did not empty (left to the general sorter) */
    uint32_t n_resume_left;       /* blocks the general sorter had to finish: such buckets, oversized groups and the groups
                                     that read their ranks, periodic blocks */
    float ms_bwt_rank;            /* part of ms_bwt_general: rank rounds over the open buckets */
    uint32_t n_from_scratch;      /* blocks the split kernel refused (sorted from scratch by the general sorter) */
    uint32_t n_unsorted;          /* buckets whose optimistic initial sort failed its check (their blocks went to the general sorter): 0 */
} bzx_stats;
int bzx_get_stats(const bzx_ctx *ctx, bzx_stats *out);

/*
 * Per-block figures of the last bzx_compress_device / bzx_compress_buffer / bzx_cstream_* / bzx_compress_block(s)
 * call, blocks in stream order (a chunked stream: as many as the context's descriptor table holds, i.e. at least
 * the max_blocks of bzx_ctx_create; BZX_E_PARAM beyond) -- what the reference logs per block at -vvv
 * (src/compression/compress_block.rs:58-63, src/huffman_coding/huffman.rs:176-181).
 */
typedef struct bzx_block_info {
    uint32_t n;               /* RLE1'd bytes in the block */
    uint32_t crc;             /* CRC-32/BZIP2 of the raw bytes it covers */
    uint32_t orig_ptr;        /* BWT: row of rotation 0 */
    uint32_t periodic;        /* 1: the block is u^k (tie order from the libbz2 replay) */
    uint32_t n_in_use;        /* distinct byte values */
    uint32_t n_mtf;           /* MTF/RLE2 symbols incl. EOB */
    uint32_t n_tables;        /* Huffman coding tables, 2..6 */
    uint32_t n_selectors;
    uint32_t bits_symbol_map, bits_selectors, bits_tables, bits_payload;
    uint64_t bits;            /* size of the block image (header .. last payload bit) */
} bzx_block_info;
int bzx_get_block_info(const bzx_ctx *ctx, uint32_t block, bzx_block_info *out);

/*
 * Stream assembler (replaces BitWriter, bitwriter.rs:42-172): header "BZh<level>", bit-granular
 * append of block images minus their padding, footer magic + combined CRC (crc.rs:25-27).
 * Host-side; used with bzx_compress_block(s) when the caller keeps the reference's structure.
 */
typedef struct bzx_stream bzx_stream;
int bzx_stream_begin(int level, bzx_stream **out);
int bzx_stream_append_block(bzx_stream *s, const uint8_t *data, size_t len, uint8_t pad_bits);
/* Finishes the stream; *data stays valid until bzx_stream_free. */
int bzx_stream_finish(bzx_stream *s, const uint8_t **data, size_t *len);
void bzx_stream_free(bzx_stream *s);

/* Deterministic synthetic inputs (include/bzx_synth.h), host buffers. seed 0 = default. */
void bzx_synth_text(uint64_t seed, uint8_t *out, size_t nbytes);
void bzx_synth_random(uint64_t seed, uint8_t *out, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
