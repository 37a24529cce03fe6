// bzx_emit.hip -- bit emission of a block image, stream layout and stream framing on gfx950.
//
// Contract: the bits huf_encode writes after table optimisation (reference
// src/huffman_coding/huffman.rs:209-466) behind the block header compress_block writes
// (src/compression/compress_block.rs:34-48), MSB first like BitPacker
// (src/bitstream/bitpacker.rs:43-84); then what BitWriter does on the host in the reference
// (src/bitstream/bitwriter.rs:67-132): blocks appended bit-granularly, "BZh<level>" in front,
// footer magic + combined CRC (src/tools/crc.rs:25-27) behind.
//
// Every variable-length item gets its bit offset from an exclusive scan of item sizes, so all
// items are written independently: one lane per 50-symbol group for the payload, one lane per
// selector chunk, one lane per coding table.  A lane packs its bits in a 64-bit register and
// stores whole 32-bit words; only the first and last word of a lane's range can be shared with
// a neighbour and are merged with atomicOr into the zeroed output buffer.  Because a block's
// out_bit is its final position in the stream (scan over block sizes, bzx_layout_kernel), no
// host-side or second-pass bit shifting is needed: the .bz2 is complete in HBM.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define EMIT_NT 512
#define EMIT_NW (EMIT_NT / 64)

struct BitW {
    uint32_t *out;
    uint64_t wi;      // next 32-bit word index
    uint64_t acc;
    uint32_t nacc;    // bits held in acc (< 32 between calls)
    bool first;
    __device__ __forceinline__ void init(uint32_t *o, uint64_t bitpos)
    {
        out = o;
        wi = bitpos >> 5;
        nacc = (uint32_t)(bitpos & 31u);   // leading bits belong to the predecessor: zeros here, OR-merged
        acc = 0;
        first = true;
    }
    __device__ __forceinline__ void flush_word(uint32_t w)
    {
        const uint32_t be = __builtin_bswap32(w);
        if (first) {
            if (be) atomicOr(&out[wi], be);
            first = false;
        } else {
            out[wi] = be;
        }
        wi++;
    }
    __device__ __forceinline__ void put(uint32_t nbits, uint32_t val)
    {
        if (nbits == 0) return;
        acc = (acc << nbits) | (uint64_t)val;
        nacc += nbits;
        if (nacc >= 32) {
            nacc -= 32;
            flush_word((uint32_t)(acc >> nacc));
            acc &= (1ull << nacc) - 1ull;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc > 0) {
            const uint32_t be = __builtin_bswap32((uint32_t)(acc << (32 - nacc)));
            if (be) atomicOr(&out[wi], be);
        }
    }
};

__shared__ uint8_t e_len[6][BZX_MAX_ALPHA + 2];
__shared__ uint32_t e_code[6][BZX_MAX_ALPHA + 2];
__shared__ uint32_t e_cl[6][BZX_MAX_ALPHA + 2];     // code | length << 24: one lookup per payload symbol
// one tile of payload symbols: EMIT_NT groups x 25 words, loaded coalesced, read back by group (stride 25 words: odd,
// so the lanes of a wave hit different banks)
__shared__ __attribute__((aligned(16))) uint32_t e_tile[EMIT_NT * (BZX_G_SIZE / 2)];
__shared__ uint32_t e_tabbits[8];
__shared__ uint32_t e_scratch[2 * EMIT_NW];
__shared__ uint32_t e_bcast[4];

#define EMIT_STAMP(slot)                                                      \
    do {                                                                      \
        if (B.dbg && tid == 0) {                                              \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&B.dbg[slot], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)

__global__ __launch_bounds__(EMIT_NT) void bzx_emit_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    unsigned long long t_last = 0;

    for (;;) {
        if (tid == 0) e_bcast[0] = atomicAdd(&B.counters[3], 1u);
        __syncthreads();
        const uint32_t j_ = e_bcast[0];
        __syncthreads();
        if (j_ >= B.nblk) break;
        const uint32_t b = B.blk_first + j_ * B.blk_step;

        if (B.dbg && tid == 0) t_last = wall_clock64();
        const BzxBlock d = B.blk[b];
        const uint32_t alpha = d.n_in_use + 2, n_mtf = d.n_mtf, n_sel = d.n_selectors, n_groups = d.n_groups;
        const uint16_t *__restrict__ V = B.mtfv + BZX_SLAB(B, b) * BZX_BLK_STRIDE;
        const uint8_t *__restrict__ SEL = B.selector + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        const uint8_t *__restrict__ SELM = B.selector_mtf + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        const uint16_t *__restrict__ GB = B.gbits + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        uint32_t *out = B.out;

        // final stream position, or (sharded runs) the same bit phase inside this rank's packed buffer
        const uint64_t ob = B.packed ? d.pack_word * 32 + (d.out_bit & 31u) : d.out_bit;
        const uint64_t o_map = ob + 105;
        const uint64_t o_ng = o_map + d.sec_bits[3];
        const uint64_t o_sel = o_ng + 18;
        const uint64_t o_tab = o_sel + d.sec_bits[0];
        const uint64_t o_pay = o_tab + d.sec_bits[1];

        for (uint32_t i = tid; i < 6 * 260; i += EMIT_NT) {
            const uint32_t t = i / 260, v = i % 260;
            if (v < BZX_MAX_ALPHA + 2) {
                e_len[t][v] = B.len[BZX_SLAB(B, b) * 6 * 260 + i];
                e_code[t][v] = B.code[BZX_SLAB(B, b) * 6 * 260 + i];
                e_cl[t][v] = B.code[BZX_SLAB(B, b) * 6 * 260 + i] | ((uint32_t)B.len[BZX_SLAB(B, b) * 6 * 260 + i] << 24);
            }
        }
        __syncthreads();

        // ---- (a) block header, symbol map, nGroups, nSelectors (compress_block.rs:34-48, huffman.rs:209-224)
        if (tid == 0) {
            BitW w;
            w.init(out, ob);
            w.put(24, 0x314159u);
            w.put(24, 0x265359u);
            w.put(32, d.crc);
            w.put(1, 0);
            w.put(24, d.orig_ptr);
            uint32_t l1 = 0, words[16];
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t wv = 0;
                for (uint32_t j = 0; j < 16; j++)
                    if (B.in_use[BZX_SLAB(B, b) * 256 + i * 16 + j]) wv |= 0x8000u >> j;
                words[i] = wv;
                if (wv) l1 |= 0x8000u >> i;
            }
            w.put(16, l1);
            for (uint32_t i = 0; i < 16; i++)
                if (words[i]) w.put(16, words[i]);
            w.put(3, n_groups);
            w.put(15, n_sel);
            w.finish();
        }

        EMIT_STAMP(48);
        // ---- (b) selectors, unary (huffman.rs:237-292)
        {
            const uint32_t per = (n_sel + EMIT_NT - 1) / EMIT_NT;
            const uint32_t lo = tid * per < n_sel ? tid * per : n_sel;
            const uint32_t hi = lo + per < n_sel ? lo + per : n_sel;
            uint32_t my_bits = 0;
            for (uint32_t i = lo; i < hi; i++) my_bits += SELM[i] + 1u;
            uint32_t tot;
            const uint32_t ex = bzx_block_excl_sum<EMIT_NT>(my_bits, e_scratch, tot);
            if (hi > lo) {
                BitW w;
                w.init(out, o_sel + ex);
                for (uint32_t i = lo; i < hi; i++) {
                    const uint32_t j = SELM[i];
                    w.put(j + 1, ((1u << j) - 1u) << 1);
                }
                w.finish();
            }
        }

        EMIT_STAMP(49);
        // ---- (c) coding tables, delta coded (huffman.rs:391-438), one lane per table
        if (lane == 0 && wave < n_groups) {
            uint32_t bits = 5;
            int32_t curr = e_len[wave][0];
            for (uint32_t i = 0; i < alpha; i++) {
                const int32_t l = e_len[wave][i];
                bits += 2u * (uint32_t)(l > curr ? l - curr : curr - l) + 1u;
                curr = l;
            }
            e_tabbits[wave] = bits;
        }
        __syncthreads();
        if (lane == 0 && wave < n_groups) {
            uint64_t pos = o_tab;
            for (uint32_t t = 0; t < wave; t++) pos += e_tabbits[t];
            BitW w;
            w.init(out, pos);
            int32_t curr = e_len[wave][0];
            w.put(5, (uint32_t)curr);
            for (uint32_t i = 0; i < alpha; i++) {
                const int32_t l = e_len[wave][i];
                while (curr < l) {
                    w.put(2, 2);
                    curr++;
                }
                while (curr > l) {
                    w.put(2, 3);
                    curr--;
                }
                w.put(1, 0);
            }
            w.finish();
        }

        __syncthreads();
        EMIT_STAMP(50);
        // ---- (d) payload (huffman.rs:452-466): one lane per group, offsets from a scan of group sizes
        {
            if (tid == 0) e_bcast[1] = 0;
            __syncthreads();
            for (uint32_t g0 = 0; g0 < n_sel; g0 += EMIT_NT) {
                const uint32_t g = g0 + tid;
                const uint32_t carry = e_bcast[1];
                const uint32_t gb = g < n_sel ? (uint32_t)GB[g] : 0u;
                uint32_t tot;
                const uint32_t ex = bzx_block_excl_sum<EMIT_NT>(gb, e_scratch, tot);
                // the tile's symbols: coalesced 16-byte loads into LDS (the slab is longer than any block's symbols,
                // so the last tile may read past n_mtf; such symbols are never coded)
                {
                    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(V + (size_t)g0 * BZX_G_SIZE);
                    uint4 *dst = reinterpret_cast<uint4 *>(e_tile);
                    const uint32_t ng = n_sel - g0 < EMIT_NT ? n_sel - g0 : EMIT_NT;
                    const uint32_t nq = (ng * (BZX_G_SIZE / 2) + 3u) / 4u;          // 16-byte pieces
                    for (uint32_t i = tid; i < nq; i += EMIT_NT) dst[i] = src[i];
                }
                __syncthreads();
                if (g < n_sel) {
                    const uint32_t gs = g * BZX_G_SIZE;
                    const uint32_t cnt = (n_mtf - gs < BZX_G_SIZE) ? n_mtf - gs : BZX_G_SIZE;
                    const uint32_t *vp = e_tile + tid * (BZX_G_SIZE / 2);
                    const uint32_t *__restrict__ cl = e_cl[SEL[g]];
                    uint32_t vw[BZX_G_SIZE / 2];
#pragma unroll
                    for (uint32_t k = 0; k < BZX_G_SIZE / 2; k++) vw[k] = 2 * k < cnt ? vp[k] : 0u;
                    BitW w;
                    w.init(out, o_pay + carry + ex);
#pragma unroll
                    for (uint32_t k = 0; k < BZX_G_SIZE / 2; k++) {
                        const uint32_t c0 = 2 * k < cnt ? cl[vw[k] & 0xffffu] : 0u;
                        const uint32_t c1 = 2 * k + 1 < cnt ? cl[vw[k] >> 16] : 0u;
                        w.put(c0 >> 24, c0 & 0xffffffu);
                        w.put(c1 >> 24, c1 & 0xffffffu);
                    }
                    w.finish();
                }
                if (tid == 0) e_bcast[1] = carry + tot;
                __syncthreads();
            }
        }
        __syncthreads();
        EMIT_STAMP(51);
    }
}

// out_bit of every block: first_bit + exclusive scan of block sizes (stream mode, stride_bits == 0)
// or b * stride_bits (per-block mode).  total[0] = first_bit + sum of sizes.
// phase (optional, chunked streams): the first bit comes from phase[0] on the device (bit phase 0..31 left by the
// previous chunk of the stream); afterwards phase[0] = phase of the next chunk, phase[1] = bits of this one.
__global__ __launch_bounds__(EMIT_NT) void bzx_layout_kernel(BzxBatch B, uint64_t first_bit, uint64_t stride_bits,
                                                            uint64_t *total, uint64_t *phase)
{
    __shared__ uint64_t l_wsum[EMIT_NW];
    __shared__ uint64_t l_carry;
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    if (phase) first_bit = phase[0];
    if (tid == 0) l_carry = first_bit;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < B.nblk; b0 += EMIT_NT) {
        const uint32_t b = b0 + tid;
        const uint64_t v = b < B.nblk ? B.blk[b].bits : 0ull;
        uint64_t x = v;
        for (uint32_t dd = 1; dd < 64; dd <<= 1) {
            const uint64_t y = __shfl_up(x, dd);
            if (lane >= dd) x += y;
        }
        if (lane == 63) l_wsum[wave] = x;
        const uint64_t carry = l_carry;
        __syncthreads();
        uint64_t pre = 0, tot = 0;
        for (uint32_t i = 0; i < EMIT_NW; i++) {
            if (i < wave) pre += l_wsum[i];
            tot += l_wsum[i];
        }
        if (b < B.nblk) B.blk[b].out_bit = stride_bits ? (uint64_t)b * stride_bits : carry + pre + x - v;
        __syncthreads();
        if (tid == 0) l_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) {
        total[0] = l_carry;
        if (phase) {
            phase[1] = l_carry - first_bit;
            phase[0] = l_carry & 31u;
        }
    }
}

// "BZh<level>" at bit 0, footer magic + combined CRC behind the last block; out_bytes[0] = stream length.
__global__ __launch_bounds__(256) void bzx_stream_frame_kernel(BzxBatch B, int level, const uint64_t *total,
                                                               uint64_t *out_bytes)
{
    // combined CRC (crc.rs:25-27): c = rotl(c, 1) ^ crc_b over the blocks in order.  Rotation is linear over
    // XOR, so c = XOR_b rotl(crc_b, (nblk - 1 - b) mod 32): one pass over the descriptors by the whole workgroup.
    __shared__ uint32_t part[4];
    if (blockIdx.x != 0) return;
    uint32_t x = 0;
    for (uint32_t b = threadIdx.x; b < B.nblk; b += blockDim.x) {
        const uint32_t c = B.blk[b].crc, r = (B.nblk - 1u - b) & 31u;
        x ^= r ? ((c << r) | (c >> (32u - r))) : c;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x ^= __shfl_xor(x, d);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint32_t combined = part[0] ^ part[1] ^ part[2] ^ part[3];
    BitW w;
    w.init(B.out, 0);
    w.put(8, 'B');
    w.put(8, 'Z');
    w.put(8, 'h');
    w.put(8, (uint32_t)('0' + level));
    w.finish();
    const uint64_t end = total[0];
    w.init(B.out, end);
    w.put(24, 0x177245u);
    w.put(24, 0x385090u);
    w.put(32, combined);
    w.finish();
    out_bytes[0] = (end + 80 + 7) >> 3;
}

// Sharded runs (round-robin blocks over GPUs): export my blocks' sizes / import everybody's sizes.
__global__ void bzx_bits_export_kernel(BzxBatch B, long long *bits)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < B.nblk; j += gridDim.x * blockDim.x) {
        const uint32_t b = B.blk_first + j * B.blk_step;
        bits[b] = (long long)(B.blk[b].bits | ((uint64_t)B.blk[b].crc << 32));   // size (< 2^32 bits) and CRC of my block
    }
}
__global__ void bzx_bits_import_kernel(BzxBatch B, const long long *bits)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B.nblk; b += gridDim.x * blockDim.x)
    {
        B.blk[b].bits = (uint64_t)bits[b] & 0xffffffffull;
        B.blk[b].crc = (uint32_t)((uint64_t)bits[b] >> 32);
    }
}
// Packed layout of one rank's blocks (b = first + j*step): every image starts on a word boundary of the packed
// buffer with the bit phase it has in the final stream, so merging is a word-wise OR.  total[0] = words used.
__global__ __launch_bounds__(EMIT_NT) void bzx_pack_layout_kernel(BzxBatch B, uint32_t first, uint32_t step,
                                                                 uint32_t nown, uint64_t *total)
{
    __shared__ uint64_t l_wsum[EMIT_NW];
    __shared__ uint64_t l_carry;
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    if (tid == 0) l_carry = 0;
    __syncthreads();
    for (uint32_t j0 = 0; j0 < nown; j0 += EMIT_NT) {
        const uint32_t j = j0 + tid;
        const uint32_t b = first + j * step;
        const uint64_t v = j < nown ? (((B.blk[b].out_bit & 31u) + B.blk[b].bits + 31u) >> 5) : 0ull;
        uint64_t x = v;
        for (uint32_t dd = 1; dd < 64; dd <<= 1) {
            const uint64_t y = __shfl_up(x, dd);
            if (lane >= dd) x += y;
        }
        if (lane == 63) l_wsum[wave] = x;
        const uint64_t carry = l_carry;
        __syncthreads();
        uint64_t pre = 0, tot = 0;
        for (uint32_t i = 0; i < EMIT_NW; i++) {
            if (i < wave) pre += l_wsum[i];
            tot += l_wsum[i];
        }
        if (j < nown) B.blk[b].pack_word = carry + pre + x - v;
        __syncthreads();
        if (tid == 0) l_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) total[0] = l_carry;
}

// Largest packed buffer of any rank, in words (every rank knows every block's size and position after the exchange
// of sizes, so the common length of the gather needs no further collective and no extra host round trip).
__global__ __launch_bounds__(EMIT_NT) void bzx_pack_max_kernel(BzxBatch B, uint32_t world, uint64_t *out)
{
    __shared__ unsigned long long sums[64];
    const uint32_t tid = threadIdx.x;
    if (tid < 64) sums[tid] = 0;
    __syncthreads();
    for (uint32_t b = tid; b < B.nblk; b += EMIT_NT)
        atomicAdd(&sums[b % world], (unsigned long long)(((B.blk[b].out_bit & 31u) + B.blk[b].bits + 31u) >> 5));
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = 0;
        for (uint32_t r = 0; r < world && r < 64; r++) m = sums[r] > m ? sums[r] : m;
        out[0] = m;
    }
}

// Merge one rank's packed buffer into the final stream: interior words are owned by one block, the first and
// last word of an image may be shared with its neighbours and are OR-ed into the zeroed buffer.
__global__ __launch_bounds__(256) void bzx_unpack_kernel(BzxBatch B, const uint32_t *__restrict__ packed, uint32_t first,
                                                        uint32_t step, uint32_t nown)
{
    for (uint32_t j = blockIdx.x; j < nown; j += gridDim.x) {
        const BzxBlock d = B.blk[first + j * step];
        const uint64_t nw = ((d.out_bit & 31u) + d.bits + 31u) >> 5;
        const uint32_t *src = packed + d.pack_word;
        uint32_t *dst = B.out + (d.out_bit >> 5);
        for (uint64_t w = threadIdx.x; w < nw; w += blockDim.x) {
            const uint32_t v = src[w];
            if (w == 0 || w + 1 == nw) {
                if (v) atomicOr(&dst[w], v);
            } else {
                dst[w] = v;
            }
        }
    }
}

// Sharded assembly writes every interior word of a block image with a plain store; only the words that are merged
// with atomicOr need to start from zero: the first and last word of every image, the header word and the first
// and last word of the footer.  (Zeroing the whole stream instead costs a pass over gigabytes at N = 8.)
__global__ __launch_bounds__(256) void bzx_zero_edges_kernel(BzxBatch B, const uint64_t *total)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B.nblk; b += gridDim.x * blockDim.x) {
        const uint64_t ob = B.blk[b].out_bit, nb = B.blk[b].bits;
        B.out[ob >> 5] = 0;
        if (nb) B.out[(ob + nb - 1) >> 5] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t end = total[0];
        B.out[0] = 0;
        for (uint64_t w = end >> 5; w <= (end + 79) >> 5; w++) B.out[w] = 0;
    }
}

void bzx_launch_zero_edges(const BzxBatch &B, const uint64_t *d_total, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_zero_edges_kernel, dim3(64), dim3(256), 0, stream, B, d_total);
}

void bzx_launch_pack_layout(const BzxBatch &B, uint32_t first, uint32_t step, uint32_t nown, uint64_t *d_total,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_pack_layout_kernel, dim3(1), dim3(EMIT_NT), 0, stream, B, first, step, nown, d_total);
}
void bzx_launch_pack_max(const BzxBatch &B, uint32_t world, uint64_t *d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_pack_max_kernel, dim3(1), dim3(EMIT_NT), 0, stream, B, world, d_out);
}
void bzx_launch_unpack(const BzxBatch &B, const uint32_t *packed, uint32_t first, uint32_t step, uint32_t nown,
                       uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_unpack_kernel, dim3(grid), dim3(256), 0, stream, B, packed, first, step, nown);
}

void bzx_launch_bits_export(const BzxBatch &B, long long *bits, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_bits_export_kernel, dim3(64), dim3(256), 0, stream, B, bits);
}
void bzx_launch_bits_import(const BzxBatch &B, const long long *bits, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_bits_import_kernel, dim3(64), dim3(256), 0, stream, B, bits);
}

void bzx_launch_emit(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_emit_kernel, dim3(grid), dim3(EMIT_NT), 0, stream, B);
}

void bzx_launch_layout(const BzxBatch &B, uint64_t first_bit, uint64_t stride_bits, uint64_t *d_total_bits,
                       hipStream_t stream, uint64_t *d_phase)
{
    hipLaunchKernelGGL(bzx_layout_kernel, dim3(1), dim3(EMIT_NT), 0, stream, B, first_bit, stride_bits, d_total_bits,
                       d_phase);
}

void bzx_launch_stream_frame(const BzxBatch &B, int level, const uint64_t *d_total_bits, uint64_t *d_out_bytes,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_stream_frame_kernel, dim3(1), dim3(256), 0, stream, B, level, d_total_bits, d_out_bytes);
}
