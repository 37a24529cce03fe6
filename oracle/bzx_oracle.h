/*
 * bzx_oracle.h -- CPU oracle for the bzip2 block-compression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * there only as the checker (or as the reported-only CPU baseline).
 *
 * What it restates: the per-block path of ohsnyt/bzip2-rust
 *   RLE1 + block split + CRC   src/tools/rle1.rs:33-263, src/tools/crc.rs:15-27
 *   BWT                        src/bwt_algorithms/bwt_sort.rs:27-58
 *   MTF + RLE2 + symbol map    src/tools/rle2_mtf.rs:23-177,293-322
 *   multi-table Huffman        src/huffman_coding/huffman.rs:79-468,
 *                              src/huffman_coding/huffman_code_from_weights.rs:17-109
 *   per-block bit packing      src/bitstream/bitpacker.rs:17-112
 *   stream assembly            src/bitstream/bitwriter.rs:42-172
 *   orchestration              src/compression/compress_block.rs:24-67, compress.rs:40-136
 * in plain C, with the bit-level decisions of C bzip2 1.0.8 (libbz2) wherever the Rust
 * reference diverges from it (D1 block split, D3/D4 initial tables, D5 heap
 * tie-breaks, D6 origPtr on periodic blocks, D7 empty input), because BASELINE.json's
 * metric is "bit-exact vs C bzip2".
 *
 * Pinning: the Rust reference cannot be built here (no cargo/rustc) and its own tests pin
 * only bit packing and the symbol map.  Those vectors plus whole-stream byte equality
 * against libbz2 1.0.8 (python `bz2`, present in this image) pin this oracle; see
 * tests/test_oracle.py and tests/golden/.
 */
#ifndef BZX_ORACLE_H
#define BZX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZO_MAX_ALPHA 258
#define BZO_G_SIZE 50
#define BZO_N_ITERS 4
#define BZO_MAX_SELECTORS (2 + (900000 / BZO_G_SIZE))

/* ---- stage functions (each is the checker for one device stage) ---- */

/* CRC-32/BZIP2 of buf (crc.rs:15-22). */
uint32_t bzo_crc32(const uint8_t *buf, size_t len);
/* combined = rotl1(combined) ^ block_crc (crc.rs:25-27). */
uint32_t bzo_stream_crc(uint32_t combined, uint32_t block_crc);

/*
 * RLE1 + block split with libbz2 semantics (replaces rle1.rs:89-223).
 * Consumes raw[*pos..len) until the block is full (nblock >= 100000*level-19 tested before
 * each input byte) or the input ends (pending run flushed into this block unless it is full:
 * BZ_RUN-mode feeding as done by the bzip2 CLI and python's bz2).
 * Writes the RLE1 bytes to blk (capacity >= 100000*level), returns nblock; *crc gets the CRC
 * of the raw bytes this block covers; *pos is advanced past the consumed raw bytes.
 * The pending run (state_ch,state_len) is carried between calls in st[2]; initialise st to
 * {256, 0}.
 */
size_t bzo_rle1_block(const uint8_t *raw, size_t len, size_t *pos, int level,
                      uint32_t st[2], uint8_t *blk, uint32_t *crc);

/*
 * BWT of a block over cyclic rotations (bwt_sort.rs:27-58): bwt[i] = blk[(sa[i]-1) mod n],
 * returns origPtr = index i with sa[i]==0.  Tie order on periodic blocks follows libbz2's
 * blockSort (mainSort with work budget, else fallbackSort) -- SURVEY.md D6.
 * sa_out may be NULL.
 */
int32_t bzo_bwt(const uint8_t *blk, int32_t n, uint8_t *bwt, uint32_t *sa_out);

/*
 * MTF + RLE2 (rle2_mtf.rs:23-177): input the BWT bytes, output symbols (RUNA=0, RUNB=1,
 * rank+1, EOB=nInUse+1 appended), mtf_freq[258] counted on the EMITTED symbol incl. EOB
 * (libbz2), in_use[256].  Returns nMTF.
 */
int32_t bzo_mtf_rle2(const uint8_t *bwt, int32_t n, uint16_t *mtfv, int32_t mtf_freq[BZO_MAX_ALPHA],
                     uint8_t in_use[256], int32_t *n_in_use);

/* Symbol map words (rle2_mtf.rs:293-322): words[0] = L1 map, then one word per set L1 bit. Returns count. */
int bzo_symbol_map(const uint8_t in_use[256], uint16_t words[17]);

/* libbz2 hbMakeCodeLengths (replaces huffman_code_from_weights.rs:17-84). */
void bzo_make_code_lengths(uint8_t *len, const int32_t *freq, int32_t alpha_size, int32_t max_len);
/* canonical codes (huffman.rs:361-374). */
void bzo_assign_codes(int32_t *code, const uint8_t *len, int32_t min_len, int32_t max_len, int32_t alpha_size);

/* Result of the table-selection stage, exposed for stage-level parity. */
typedef struct {
    int32_t n_groups;
    int32_t n_selectors;
    uint8_t selector[BZO_MAX_SELECTORS];
    uint8_t selector_mtf[BZO_MAX_SELECTORS];
    uint8_t len[6][BZO_MAX_ALPHA];
    int32_t code[6][BZO_MAX_ALPHA];
} bzo_huff_tables;

/* Table count, initial partition, 4 refinement passes, selector MTF, codes (huffman.rs:87-374 with D4/D5). */
void bzo_huff_optimise(const uint16_t *mtfv, int32_t n_mtf, const int32_t *mtf_freq, int32_t alpha_size,
                       bzo_huff_tables *t);

/* ---- bit packer (bitpacker.rs:17-112) ---- */
typedef struct {
    uint8_t *out;
    size_t cap;
    size_t len;      /* whole bytes written */
    uint64_t queue;
    int q_bits;
    int overflow;
} bzo_bitpacker;

void bzo_bp_init(bzo_bitpacker *bp, uint8_t *out, size_t cap);
void bzo_bp_put(bzo_bitpacker *bp, int nbits, uint32_t value);   /* nbits 0..32, MSB first */
void bzo_bp_out24(bzo_bitpacker *bp, uint32_t data);              /* length in top byte (bitpacker.rs:62-68) */
void bzo_bp_out32(bzo_bitpacker *bp, uint32_t data);
void bzo_bp_out16(bzo_bitpacker *bp, uint16_t data);
int bzo_bp_flush(bzo_bitpacker *bp);                              /* returns pad bits 0..7 */

/*
 * compress_block (compress_block.rs:24-67): blk = RLE1'd block bytes, crc = CRC of the raw
 * bytes it covers.  out receives the byte-aligned block image (magic, crc, 0, origPtr, maps,
 * selectors, tables, payload), last byte zero padded; *pad_bits = number of pad bits.
 * Returns 0, or -1 if cap is too small / n out of range.
 */
int bzo_compress_block(const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap,
                       size_t *out_len, uint8_t *pad_bits);

/* Per-block telemetry mirroring `bzip2 -vvv` / compress_block.rs:58-63. */
typedef struct {
    int32_t nblock, orig_ptr, n_mtf, n_in_use, n_groups, n_selectors;
    uint32_t crc;
    uint64_t bits;
} bzo_block_info;
int bzo_compress_block_info(const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap,
                            size_t *out_len, uint8_t *pad_bits, bzo_block_info *info);

/* ---- stream assembler (bitwriter.rs:42-172) ---- */
typedef struct {
    uint8_t *out;
    size_t cap;
    size_t len;
    uint64_t queue;
    int q_bits;
    uint32_t stream_crc;
    int level;
    int started;
    int overflow;
} bzo_stream;

void bzo_stream_begin(bzo_stream *s, uint8_t *out, size_t cap, int level);
void bzo_stream_add_block(bzo_stream *s, const uint8_t *data, size_t len, int pad_bits);
size_t bzo_stream_finish(bzo_stream *s);  /* footer + combined CRC + pad; returns total bytes (0 on overflow) */

/*
 * Whole buffer -> .bz2 (compress.rs:40-136 without the file I/O), single thread.
 * Returns compressed length, or 0 on overflow.  nblocks_out may be NULL.
 */
size_t bzo_compress_buffer(const uint8_t *raw, size_t len, int level, uint8_t *out, size_t cap,
                           int32_t *nblocks_out);
/* Same with nthreads worker threads, one block per worker, ordered assembly (compress.rs:125-132). */
size_t bzo_compress_buffer_mt(const uint8_t *raw, size_t len, int level, int nthreads, uint8_t *out,
                              size_t cap, int32_t *nblocks_out);

/* Synthetic inputs of SURVEY.md section 8(d); shared by tests and bench. */
void bzo_synthtext(uint64_t seed, uint8_t *out, size_t nbytes);
void bzo_xorshift_bytes(uint64_t seed, uint8_t *out, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
