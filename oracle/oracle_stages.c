/*
 * oracle_stages.c -- TEST INFRASTRUCTURE (see bzx_oracle.h).
 *
 * CRC, RLE1 + block split, MTF + RLE2, symbol map, Huffman table optimisation, bit packer,
 * compress_block, stream assembler.  Each function cites the reference location whose
 * contract it follows; where SURVEY.md F2 lists a divergence from C bzip2 1.0.8 the libbz2
 * behaviour is the one restated (noted per function).
 */
#include <stdlib.h>
#include <string.h>
#include "bzx_oracle.h"

/* ------------------------------------------------------------------ CRC (crc.rs:15-27) */

static uint32_t crc_table[256];
static int crc_ready = 0;

static void crc_init(void)
{
    /* CRC-32/BZIP2: poly 0x04C11DB7, MSB first, no reflection (table at crc.rs:41-298). */
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
        crc_table[i] = c;
    }
    crc_ready = 1;
}

static inline uint32_t crc_step(uint32_t crc, uint8_t b)
{
    return (crc << 8) ^ crc_table[(crc >> 24) ^ b];
}

uint32_t bzo_crc32(const uint8_t *buf, size_t len)
{
    if (!crc_ready) crc_init();
    uint32_t crc = 0xffffffffu;
    for (size_t i = 0; i < len; i++) crc = crc_step(crc, buf[i]);
    return ~crc;
}

uint32_t bzo_stream_crc(uint32_t combined, uint32_t block_crc)
{
    return ((combined << 1) | (combined >> 31)) ^ block_crc;
}

/* ------------------------------------------------------------------ RLE1 + split
 * Contract of RLE1Block (rle1.rs:33-263) with libbz2's split rule:
 * the block-full test happens before each input byte, the pending run is carried to the
 * next block, and only EOF flushes it into the current one. */

static size_t rle1_flush_run(uint32_t st[2], uint8_t *blk, size_t nblock, uint32_t *crc)
{
    uint8_t ch = (uint8_t)st[0];
    uint32_t len = st[1];
    for (uint32_t i = 0; i < len; i++) *crc = crc_step(*crc, ch);
    switch (len) {
    case 1:
        blk[nblock++] = ch;
        break;
    case 2:
        blk[nblock++] = ch; blk[nblock++] = ch;
        break;
    case 3:
        blk[nblock++] = ch; blk[nblock++] = ch; blk[nblock++] = ch;
        break;
    default:
        blk[nblock++] = ch; blk[nblock++] = ch; blk[nblock++] = ch; blk[nblock++] = ch;
        blk[nblock++] = (uint8_t)(len - 4);
        break;
    }
    return nblock;
}

size_t bzo_rle1_block(const uint8_t *raw, size_t len, size_t *pos, int level, uint32_t st[2],
                      uint8_t *blk, uint32_t *crc_out)
{
    if (!crc_ready) crc_init();
    size_t nblock = 0, p = *pos;
    const size_t nmax = (size_t)100000 * (size_t)level - 19;
    uint32_t crc = 0xffffffffu;

    while (nblock < nmax && p < len) {
        uint32_t b = raw[p++];
        if (b != st[0] && st[1] == 1) {
            uint8_t ch = (uint8_t)st[0];
            crc = crc_step(crc, ch);
            blk[nblock++] = ch;
            st[0] = b;
        } else if (b != st[0] || st[1] == 255) {
            if (st[0] < 256) nblock = rle1_flush_run(st, blk, nblock, &crc);
            st[0] = b;
            st[1] = 1;
        } else {
            st[1]++;
        }
    }
    if (p >= len && nblock < nmax && st[0] < 256) {
        /* end of input: flush the pending run into this block -- unless the block is already full:
         * the bzip2 CLI and python's bz2 feed libbz2 in BZ_RUN mode, where a full block is cut the
         * moment nblock >= nblockMAX is seen, before the compressor learns that the input has ended;
         * the pending run then forms a last block of its own. */
        nblock = rle1_flush_run(st, blk, nblock, &crc);
        st[0] = 256;
        st[1] = 0;
    }
    *pos = p;
    *crc_out = ~crc;
    return nblock;
}

/* ------------------------------------------------------------------ MTF + RLE2 (rle2_mtf.rs:23-177) */

int32_t bzo_mtf_rle2(const uint8_t *bwt, int32_t n, uint16_t *mtfv, int32_t mtf_freq[BZO_MAX_ALPHA],
                     uint8_t in_use[256], int32_t *n_in_use_out)
{
    uint8_t unseq_to_seq[256], yy[256];
    int32_t n_in_use = 0, eob, wr = 0, z_pend = 0, i;

    memset(in_use, 0, 256);
    for (i = 0; i < n; i++) in_use[bwt[i]] = 1;
    for (i = 0; i < 256; i++)
        if (in_use[i]) unseq_to_seq[i] = (uint8_t)n_in_use++;
    eob = n_in_use + 1;
    for (i = 0; i < BZO_MAX_ALPHA; i++) mtf_freq[i] = 0;
    for (i = 0; i < n_in_use; i++) yy[i] = (uint8_t)i;

#define FLUSH_ZRUN()                                           \
    if (z_pend > 0) {                                          \
        z_pend--;                                              \
        for (;;) {                                             \
            uint16_t sym = (uint16_t)(z_pend & 1); /* RUNA=0, RUNB=1 */ \
            mtfv[wr++] = sym;                                  \
            mtf_freq[sym]++;                                   \
            if (z_pend < 2) break;                             \
            z_pend = (z_pend - 2) / 2;                         \
        }                                                      \
        z_pend = 0;                                            \
    }

    for (i = 0; i < n; i++) {
        uint8_t ll = unseq_to_seq[bwt[i]];
        if (yy[0] == ll) {
            z_pend++;
        } else {
            FLUSH_ZRUN();
            int32_t j = 0;
            uint8_t tmp = yy[0];
            while (tmp != ll) {
                j++;
                uint8_t t2 = tmp;
                tmp = yy[j];
                yy[j] = t2;
            }
            yy[0] = tmp;
            mtfv[wr++] = (uint16_t)(j + 1);
            mtf_freq[j + 1]++;
        }
    }
    FLUSH_ZRUN();
#undef FLUSH_ZRUN
    mtfv[wr++] = (uint16_t)eob;
    mtf_freq[eob]++;
    *n_in_use_out = n_in_use;
    return wr;
}

/* symbol map (rle2_mtf.rs:293-322; KAT symbol_map.rs:45-59) */
int bzo_symbol_map(const uint8_t in_use[256], uint16_t words[17])
{
    int cnt = 1;
    uint16_t l1 = 0;
    for (int i = 0; i < 16; i++) {
        uint16_t w = 0;
        for (int j = 0; j < 16; j++)
            if (in_use[i * 16 + j]) w |= (uint16_t)(0x8000u >> j);
        if (w) {
            l1 |= (uint16_t)(0x8000u >> i);
            words[cnt++] = w;
        }
    }
    words[0] = l1;
    return cnt;
}

/* ------------------------------------------------------------------ Huffman code lengths
 * Contract of improve_code_len_from_weights (huffman_code_from_weights.rs:17-84): lengths <= 17
 * from frequencies, weight = max(f,1)<<8, ADDWEIGHTS (ibid. :105-109), halve-and-retry (:76-80).
 * Tie-breaking is libbz2's binary heap. */

void bzo_make_code_lengths(uint8_t *len, const int32_t *freq, int32_t alpha_size, int32_t max_len)
{
    int32_t n_nodes, n_heap, n1 = 0, n2 = 0, i, j, k;
    int too_long;
    int32_t heap[BZO_MAX_ALPHA + 2];
    int32_t weight[BZO_MAX_ALPHA * 2];
    int32_t parent[BZO_MAX_ALPHA * 2];

    for (i = 0; i < alpha_size; i++) weight[i + 1] = (freq[i] == 0 ? 1 : freq[i]) << 8;

    for (;;) {
        n_nodes = alpha_size;
        n_heap = 0;
        heap[0] = 0;
        weight[0] = 0;
        parent[0] = -2;

        for (i = 1; i <= alpha_size; i++) {
            parent[i] = -1;
            n_heap++;
            heap[n_heap] = i;
            { /* sift up */
                int32_t zz = n_heap, tmp = heap[zz];
                while (weight[tmp] < weight[heap[zz >> 1]]) {
                    heap[zz] = heap[zz >> 1];
                    zz >>= 1;
                }
                heap[zz] = tmp;
            }
        }
        while (n_heap > 1) {
            for (int rep = 0; rep < 2; rep++) {
                int32_t top = heap[1];
                heap[1] = heap[n_heap];
                n_heap--;
                { /* sift down */
                    int32_t zz = 1, yy, tmp = heap[zz];
                    for (;;) {
                        yy = zz << 1;
                        if (yy > n_heap) break;
                        if (yy < n_heap && weight[heap[yy + 1]] < weight[heap[yy]]) yy++;
                        if (weight[tmp] < weight[heap[yy]]) break;
                        heap[zz] = heap[yy];
                        zz = yy;
                    }
                    heap[zz] = tmp;
                }
                if (rep == 0) n1 = top; else n2 = top;
            }
            n_nodes++;
            parent[n1] = parent[n2] = n_nodes;
            {
                uint32_t w1 = (uint32_t)weight[n1], w2 = (uint32_t)weight[n2];
                uint32_t d1 = w1 & 0xff, d2 = w2 & 0xff;
                weight[n_nodes] = (int32_t)(((w1 & 0xffffff00u) + (w2 & 0xffffff00u)) | (1 + (d1 > d2 ? d1 : d2)));
            }
            parent[n_nodes] = -1;
            n_heap++;
            heap[n_heap] = n_nodes;
            {
                int32_t zz = n_heap, tmp = heap[zz];
                while (weight[tmp] < weight[heap[zz >> 1]]) {
                    heap[zz] = heap[zz >> 1];
                    zz >>= 1;
                }
                heap[zz] = tmp;
            }
        }

        too_long = 0;
        for (i = 1; i <= alpha_size; i++) {
            j = 0;
            k = i;
            while (parent[k] >= 0) {
                k = parent[k];
                j++;
            }
            len[i - 1] = (uint8_t)j;
            if (j > max_len) too_long = 1;
        }
        if (!too_long) break;

        for (i = 1; i <= alpha_size; i++) {
            j = weight[i] >> 8;
            j = 1 + (j / 2);
            weight[i] = j << 8;
        }
    }
}

void bzo_assign_codes(int32_t *code, const uint8_t *len, int32_t min_len, int32_t max_len, int32_t alpha_size)
{
    int32_t vec = 0;
    for (int32_t n = min_len; n <= max_len; n++) {
        for (int32_t i = 0; i < alpha_size; i++)
            if (len[i] == n) code[i] = vec++;
        vec <<= 1;
    }
}

/* Table count (huffman.rs:87-93), initial partition (huffman.rs:472-532 contract, libbz2 rule
 * per SURVEY.md D4), 4 passes of group cost -> first-minimum table -> rfreq -> new lengths
 * (huffman.rs:114-200), selector MTF (huffman.rs:237-292), canonical codes (huffman.rs:361-374). */
void bzo_huff_optimise(const uint16_t *mtfv, int32_t n_mtf, const int32_t *mtf_freq, int32_t alpha_size,
                       bzo_huff_tables *T)
{
    int32_t n_groups, t, v, iter, gs, ge, i;
    static __thread int32_t rfreq[6][BZO_MAX_ALPHA];

    for (t = 0; t < 6; t++)
        for (v = 0; v < alpha_size; v++) T->len[t][v] = 15;

    if (n_mtf < 200) n_groups = 2;
    else if (n_mtf < 600) n_groups = 3;
    else if (n_mtf < 1200) n_groups = 4;
    else if (n_mtf < 2400) n_groups = 5;
    else n_groups = 6;

    {
        int32_t n_part = n_groups, rem_f = n_mtf, t_freq, a_freq;
        gs = 0;
        while (n_part > 0) {
            t_freq = rem_f / n_part;
            ge = gs - 1;
            a_freq = 0;
            while (a_freq < t_freq && ge < alpha_size - 1) {
                ge++;
                a_freq += mtf_freq[ge];
            }
            if (ge > gs && n_part != n_groups && n_part != 1 && ((n_groups - n_part) % 2 == 1)) {
                a_freq -= mtf_freq[ge];
                ge--;
            }
            for (v = 0; v < alpha_size; v++) T->len[n_part - 1][v] = (v >= gs && v <= ge) ? 0 : 15;
            n_part--;
            gs = ge + 1;
            rem_f -= a_freq;
        }
    }

    int32_t n_sel = 0;
    for (iter = 0; iter < BZO_N_ITERS; iter++) {
        for (t = 0; t < n_groups; t++)
            for (v = 0; v < alpha_size; v++) rfreq[t][v] = 0;
        n_sel = 0;
        gs = 0;
        while (gs < n_mtf) {
            int32_t cost[6] = {0, 0, 0, 0, 0, 0};
            ge = gs + BZO_G_SIZE - 1;
            if (ge >= n_mtf) ge = n_mtf - 1;
            for (i = gs; i <= ge; i++) {
                uint16_t icv = mtfv[i];
                for (t = 0; t < n_groups; t++) cost[t] += T->len[t][icv];
            }
            int32_t bc = 999999999, bt = -1;
            for (t = 0; t < n_groups; t++)
                if (cost[t] < bc) {
                    bc = cost[t];
                    bt = t;
                }
            T->selector[n_sel++] = (uint8_t)bt;
            for (i = gs; i <= ge; i++) rfreq[bt][mtfv[i]]++;
            gs = ge + 1;
        }
        for (t = 0; t < n_groups; t++) bzo_make_code_lengths(T->len[t], rfreq[t], alpha_size, 17);
    }
    T->n_groups = n_groups;
    T->n_selectors = n_sel;

    {
        uint8_t pos[6];
        for (i = 0; i < n_groups; i++) pos[i] = (uint8_t)i;
        for (i = 0; i < n_sel; i++) {
            uint8_t ll = T->selector[i], tmp = pos[0];
            int32_t j = 0;
            while (ll != tmp) {
                j++;
                uint8_t t2 = tmp;
                tmp = pos[j];
                pos[j] = t2;
            }
            pos[0] = tmp;
            T->selector_mtf[i] = (uint8_t)j;
        }
    }
    for (t = 0; t < n_groups; t++) {
        int32_t min_len = 32, max_len = 0;
        for (i = 0; i < alpha_size; i++) {
            if (T->len[t][i] > max_len) max_len = T->len[t][i];
            if (T->len[t][i] < min_len) min_len = T->len[t][i];
        }
        bzo_assign_codes(T->code[t], T->len[t], min_len, max_len, alpha_size);
    }
}

/* ------------------------------------------------------------------ bit packer (bitpacker.rs:17-112) */

void bzo_bp_init(bzo_bitpacker *bp, uint8_t *out, size_t cap)
{
    bp->out = out;
    bp->cap = cap;
    bp->len = 0;
    bp->queue = 0;
    bp->q_bits = 0;
    bp->overflow = 0;
}

static inline void bp_drain(bzo_bitpacker *bp)
{
    while (bp->q_bits > 7) {
        uint8_t byte = (uint8_t)(bp->queue >> (bp->q_bits - 8));
        if (bp->len < bp->cap) bp->out[bp->len] = byte; else bp->overflow = 1;
        bp->len++;
        bp->q_bits -= 8;
    }
}

void bzo_bp_put(bzo_bitpacker *bp, int nbits, uint32_t value)
{
    if (nbits <= 0) return;
    uint64_t mask = nbits >= 32 ? 0xffffffffull : ((1ull << nbits) - 1);
    bp->queue = (bp->queue << nbits) | ((uint64_t)value & mask);
    bp->q_bits += nbits;
    bp_drain(bp);
}

void bzo_bp_out24(bzo_bitpacker *bp, uint32_t data) { bzo_bp_put(bp, (int)(data >> 24), data & 0x00ffffffu); }
void bzo_bp_out32(bzo_bitpacker *bp, uint32_t data) { bzo_bp_put(bp, 32, data); }
void bzo_bp_out16(bzo_bitpacker *bp, uint16_t data) { bzo_bp_put(bp, 16, data); }

int bzo_bp_flush(bzo_bitpacker *bp)
{
    int pad = 0;
    if (bp->q_bits > 0) {
        pad = 8 - bp->q_bits % 8;
        bp->queue <<= pad;
        bp->q_bits += pad;
        bp_drain(bp);
    }
    return pad;
}

/* ------------------------------------------------------------------ compress_block (compress_block.rs:24-67) */

int bzo_compress_block_info(const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap,
                            size_t *out_len, uint8_t *pad_bits, bzo_block_info *info)
{
    if (n == 0 || n > 900000) return -1;
    uint8_t *bwt = (uint8_t *)malloc(n);
    uint16_t *mtfv = (uint16_t *)malloc(sizeof(uint16_t) * (n + 2));
    bzo_huff_tables *T = (bzo_huff_tables *)malloc(sizeof(bzo_huff_tables));
    int32_t mtf_freq[BZO_MAX_ALPHA], n_in_use, n_mtf, orig, i, t;
    uint8_t in_use[256];
    uint16_t map[17];
    bzo_bitpacker bp;

    bzo_bp_init(&bp, out, cap);
    bzo_bp_out24(&bp, 0x18314159u);
    bzo_bp_out24(&bp, 0x18265359u);
    bzo_bp_out32(&bp, crc);
    bzo_bp_out24(&bp, 0x01000000u);

    orig = bzo_bwt(blk, (int32_t)n, bwt, NULL);
    bzo_bp_out24(&bp, 0x18000000u | (uint32_t)orig);

    n_mtf = bzo_mtf_rle2(bwt, (int32_t)n, mtfv, mtf_freq, in_use, &n_in_use);
    int32_t alpha = n_in_use + 2;
    bzo_huff_optimise(mtfv, n_mtf, mtf_freq, alpha, T);

    int nmap = bzo_symbol_map(in_use, map);
    for (i = 0; i < nmap; i++) bzo_bp_out16(&bp, map[i]);
    bzo_bp_put(&bp, 3, (uint32_t)T->n_groups);
    bzo_bp_put(&bp, 15, (uint32_t)T->n_selectors);
    for (i = 0; i < T->n_selectors; i++) {
        for (int j = 0; j < T->selector_mtf[i]; j++) bzo_bp_put(&bp, 1, 1);
        bzo_bp_put(&bp, 1, 0);
    }
    for (t = 0; t < T->n_groups; t++) {
        int32_t curr = T->len[t][0];
        bzo_bp_put(&bp, 5, (uint32_t)curr);
        for (i = 0; i < alpha; i++) {
            while (curr < T->len[t][i]) { bzo_bp_put(&bp, 2, 2); curr++; }
            while (curr > T->len[t][i]) { bzo_bp_put(&bp, 2, 3); curr--; }
            bzo_bp_put(&bp, 1, 0);
        }
    }
    {
        int32_t sel = 0, gs = 0, ge;
        while (gs < n_mtf) {
            ge = gs + BZO_G_SIZE - 1;
            if (ge >= n_mtf) ge = n_mtf - 1;
            const uint8_t *L = T->len[T->selector[sel]];
            const int32_t *C = T->code[T->selector[sel]];
            for (i = gs; i <= ge; i++) bzo_bp_put(&bp, L[mtfv[i]], (uint32_t)C[mtfv[i]]);
            gs = ge + 1;
            sel++;
        }
    }
    uint64_t bits = (uint64_t)bp.len * 8 + (uint64_t)bp.q_bits;
    int pad = bzo_bp_flush(&bp);
    if (info) {
        info->nblock = (int32_t)n;
        info->orig_ptr = orig;
        info->n_mtf = n_mtf;
        info->n_in_use = n_in_use;
        info->n_groups = T->n_groups;
        info->n_selectors = T->n_selectors;
        info->crc = crc;
        info->bits = bits;
    }
    free(bwt);
    free(mtfv);
    free(T);
    if (bp.overflow) return -1;
    *out_len = bp.len;
    *pad_bits = (uint8_t)pad;
    return 0;
}

int bzo_compress_block(const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap, size_t *out_len,
                       uint8_t *pad_bits)
{
    return bzo_compress_block_info(blk, n, crc, out, cap, out_len, pad_bits, NULL);
}

/* ------------------------------------------------------------------ stream assembler (bitwriter.rs:42-172) */

static inline void st_put8(bzo_stream *s, uint8_t b)
{
    s->queue = (s->queue << 8) | b;
    s->q_bits += 8;
    while (s->q_bits > 7) {
        uint8_t byte = (uint8_t)(s->queue >> (s->q_bits - 8));
        if (s->len < s->cap) s->out[s->len] = byte; else s->overflow = 1;
        s->len++;
        s->q_bits -= 8;
    }
}

void bzo_stream_begin(bzo_stream *s, uint8_t *out, size_t cap, int level)
{
    memset(s, 0, sizeof(*s));
    s->out = out;
    s->cap = cap;
    s->level = level;
}

static void st_header(bzo_stream *s)
{
    /* bitwriter.rs:67-72; written once for the first block (libbz2) */
    st_put8(s, 'B');
    st_put8(s, 'Z');
    st_put8(s, 'h');
    st_put8(s, (uint8_t)('0' + s->level));
    s->started = 1;
}

void bzo_stream_add_block(bzo_stream *s, const uint8_t *data, size_t len, int pad_bits)
{
    if (!s->started) st_header(s);
    uint32_t block_crc = ((uint32_t)data[6] << 24) | ((uint32_t)data[7] << 16) | ((uint32_t)data[8] << 8) | data[9];
    s->stream_crc = bzo_stream_crc(s->stream_crc, block_crc);
    /* every byte but the last whole; the last without its pad bits (bitwriter.rs:94-100) */
    for (size_t i = 0; i + 1 < len; i++) st_put8(s, data[i]);
    if (len > 0) {
        int keep = 8 - pad_bits;
        s->queue = (s->queue << keep) | (uint64_t)(data[len - 1] >> pad_bits);
        s->q_bits += keep;
        while (s->q_bits > 7) {
            uint8_t byte = (uint8_t)(s->queue >> (s->q_bits - 8));
            if (s->len < s->cap) s->out[s->len] = byte; else s->overflow = 1;
            s->len++;
            s->q_bits -= 8;
        }
    }
}

size_t bzo_stream_finish(bzo_stream *s)
{
    static const uint8_t magic[6] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90};
    if (!s->started) st_header(s);
    for (int i = 0; i < 6; i++) st_put8(s, magic[i]);
    st_put8(s, (uint8_t)(s->stream_crc >> 24));
    st_put8(s, (uint8_t)(s->stream_crc >> 16));
    st_put8(s, (uint8_t)(s->stream_crc >> 8));
    st_put8(s, (uint8_t)(s->stream_crc));
    if (s->q_bits > 0) {
        uint8_t byte = (uint8_t)((s->queue & ((1ull << s->q_bits) - 1)) << (8 - s->q_bits));
        if (s->len < s->cap) s->out[s->len] = byte; else s->overflow = 1;
        s->len++;
        s->q_bits = 0;
    }
    return s->overflow ? 0 : s->len;
}

/* ------------------------------------------------------------------ whole buffer (compress.rs:40-136) */

size_t bzo_compress_buffer(const uint8_t *raw, size_t len, int level, uint8_t *out, size_t cap,
                           int32_t *nblocks_out)
{
    bzo_stream s;
    uint32_t st[2] = {256, 0};
    size_t pos = 0, bcap = (size_t)100000 * level;
    uint8_t *blk = (uint8_t *)malloc(bcap + 8);
    size_t zcap = bcap + bcap / 50 + 1024;
    uint8_t *z = (uint8_t *)malloc(zcap);
    int32_t nb = 0;

    bzo_stream_begin(&s, out, cap, level);
    while (pos < len || st[0] < 256) {
        uint32_t crc;
        size_t n = bzo_rle1_block(raw, len, &pos, level, st, blk, &crc);
        if (n == 0) break;
        size_t zl;
        uint8_t pad;
        if (bzo_compress_block(blk, n, crc, z, zcap, &zl, &pad) != 0) {
            s.overflow = 1;
            break;
        }
        bzo_stream_add_block(&s, z, zl, pad);
        nb++;
    }
    size_t total = bzo_stream_finish(&s);
    free(blk);
    free(z);
    if (nblocks_out) *nblocks_out = nb;
    return total;
}
