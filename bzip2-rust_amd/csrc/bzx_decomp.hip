// bzx_decomp.hip -- bzip2 decompression on gfx950 (round-trip verification on the device).
//
// Contract (reference src/compression/decompress.rs:38-404; bwt_decode bwt_sort.rs:91-130; rle2_mtf_decode_fast
// rle2_mtf.rs:191-287; rle1_decode rle1.rs:267-316; decode_sym_map symbol_map.rs:20-42): .bz2 stream -> raw bytes,
// every block CRC and the combined CRC checked.  The reference decodes one block after the other on one thread; here
// the blocks of a stream are decoded side by side:
//   scan      every bit offset is tested for the 48-bit block magic / end-of-stream magic (blocks start at arbitrary
//             bits); the host orders the candidates, a candidate that does not continue the chain of decoded blocks
//             is a chance match inside compressed data and is dropped
//   decode    one wave per block: header, symbol map, selectors, code lengths (decompress.rs:98-260), then the
//             Huffman / MTF / RUNA-RUNB loop (decompress.rs:293-358).  The bit reader and the Huffman tables are
//             wave-uniform; the 256-entry MTF list lives in four registers per lane and a move-to-front is a few
//             cross-lane shifts for any rank; decoded bytes (and, for the inverse BWT, the number of earlier
//             occurrences of the same byte) leave through 64-entry coalesced stores
//   ibwt      T[C[L[i]] + occ[i]] = i is a plain scatter (the decoder already counted occ), then ONE pointer walk
//             per block, 64 blocks per wave (bwt_sort.rs:91-130); the walk also measures the RLE1 expansion and
//             leaves a checkpoint every 4096 bytes
//   expand    RLE1 runs are expanded from the checkpoints in parallel (rle1.rs:267-316) at the block's final
//             offset; block CRCs come from the compressor's CRC kernel (bzx_rle1.hip) over the output
// Integer/byte work, latency-bound (a serial bit stream and a serial pointer chase per block); all blocks of a
// stream are in flight at once.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define DC_MAGIC_BLOCK 0x314159265359ull
#define DC_MAGIC_EOS 0x177245385090ull
#define DC_ERR_HEADER 0x100u       // BzxBlock.status bits set by the decoder
#define DC_ERR_DATA 0x200u
#define DC_CK_SHIFT 12             // a checkpoint every 4096 bytes of the RLE1 image
#define DC_CK_STRIDE 224           // checkpoints per block slab (900000 / 4096 + 1 = 220)

// ---- scan: candidate block starts ---------------------------------------------------------------------------
__global__ void bzx_dc_scan_kernel(const uint8_t *__restrict__ z, uint64_t nbytes, uint64_t *__restrict__ found,
                                   uint32_t *__restrict__ n_found, uint32_t cap)
{
    const uint64_t nwords = (nbytes + 3) / 4;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t byte0 = w * 4;
        uint64_t hi = 0, lo = 0;                            // bytes byte0 .. byte0+15, big-endian
#pragma unroll
        for (int i = 0; i < 8; i++) hi = (hi << 8) | (byte0 + i < nbytes ? z[byte0 + i] : 0u);
#pragma unroll
        for (int i = 8; i < 16; i++) lo = (lo << 8) | (byte0 + i < nbytes ? z[byte0 + i] : 0u);
#pragma unroll
        for (uint32_t s = 0; s < 32; s++) {
            const uint64_t x = s ? (hi << s) | (lo >> (64 - s)) : hi;
            const uint64_t v = x >> 16;
            const uint64_t bit = byte0 * 8 + s;
            if ((v == DC_MAGIC_BLOCK || v == DC_MAGIC_EOS) && bit >= 32 && bit + 48 <= nbytes * 8) {
                const uint32_t k = atomicAdd(n_found, 1u);
                if (k < cap) found[k] = (bit << 1) | (v == DC_MAGIC_EOS ? 1u : 0u);
            }
        }
    }
}

// ---- decode ---------------------------------------------------------------------------------------------------
// Wave-uniform bit reader over big-endian 32-bit words: every lane keeps one word of a 64-word window.
struct DcBits {
    const uint8_t *z;
    uint64_t nbytes;
    uint64_t win_word;      // index of the window's first 32-bit word
    uint32_t mine;          // my word of the window
    uint64_t pos;           // next bit
    __device__ __forceinline__ void load_window(uint64_t w0)
    {
        win_word = w0;
        const uint64_t b = (w0 + bzx_lane()) * 4;
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) v = (v << 8) | (b + i < nbytes ? z[b + i] : 0u);
        mine = v;
    }
    __device__ __forceinline__ void init(const uint8_t *z_, uint64_t nbytes_, uint64_t bit)
    {
        z = z_;
        nbytes = nbytes_;
        pos = bit;
        load_window(bit >> 5);
    }
    // the next n (<= 32) bits, not consumed
    __device__ __forceinline__ uint32_t peek(uint32_t n)
    {
        uint64_t w = pos >> 5;
        if (w + 1 >= win_word + 64) load_window(w);          // both words inside the window
        const uint32_t k = (uint32_t)(w - win_word);
        const uint64_t a = __shfl(mine, (int)k), b = __shfl(mine, (int)(k + 1));
        const uint64_t x = (a << 32) | b;
        return (uint32_t)((x << (pos & 31u)) >> (64u - n));
    }
    __device__ __forceinline__ uint32_t get(uint32_t n)
    {
        const uint32_t v = peek(n);
        pos += n;
        return v;
    }
    __device__ __forceinline__ bool past_end() const { return pos > nbytes * 8; }
};

__shared__ int32_t d_limit[6][24];
__shared__ int32_t d_base[6][24];
__shared__ uint16_t d_perm[6][260];
__shared__ uint8_t d_len[6][260];
__shared__ uint32_t d_minlen[6];
__shared__ uint32_t d_count[256];      // occurrences of every byte so far (unzftab)
__shared__ uint8_t d_seq2byte[256];

// hbCreateDecodeTables of libbz2 / huf_decode_map (decompress.rs:426-486) for table t.
__device__ static void dc_make_tables(uint32_t t, uint32_t alpha)
{
    uint32_t minl = 32, maxl = 0;
    for (uint32_t i = 0; i < alpha; i++) {
        const uint32_t l = d_len[t][i];
        maxl = l > maxl ? l : maxl;
        minl = l < minl ? l : minl;
    }
    uint32_t pp = 0;
    for (uint32_t l = minl; l <= maxl; l++)
        for (uint32_t j = 0; j < alpha; j++)
            if (d_len[t][j] == l) d_perm[t][pp++] = (uint16_t)j;
    for (uint32_t i = 0; i < 24; i++) {
        d_base[t][i] = 0;
        d_limit[t][i] = 0;
    }
    for (uint32_t i = 0; i < alpha; i++) d_base[t][d_len[t][i] + 1]++;
    for (uint32_t i = 1; i < 23; i++) d_base[t][i] += d_base[t][i - 1];
    int32_t vec = 0;
    for (uint32_t l = minl; l <= maxl; l++) {
        vec += d_base[t][l + 1] - d_base[t][l];
        d_limit[t][l] = vec - 1;
        vec <<= 1;
    }
    for (uint32_t l = minl + 1; l <= maxl; l++) d_base[t][l] = ((d_limit[t][l - 1] + 1) << 1) - d_base[t][l];
    d_minlen[t] = minl;
}

// One wave per block.  starts[j] = bit offset of the block magic.  Outputs: L bytes (B.bwt slab), occ ranks (u32,
// B.rec_a slab), byte counts (B.freq slab), descriptor: n, crc (stored), orig_ptr, bits = bit after the block's last
// symbol, status (DC_ERR_* on malformed data).
__global__ __launch_bounds__(64) void bzx_dc_decode_kernel(BzxBatch B, const uint8_t *__restrict__ z, uint64_t nbytes,
                                                          const uint64_t *__restrict__ starts, uint32_t max_n)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= B.nblk) return;
    BzxBlock &D = B.blk[b];
    uint8_t *__restrict__ L = B.bwt + (size_t)b * BZX_BLK_STRIDE;
    uint32_t *__restrict__ OCC = reinterpret_cast<uint32_t *>(B.rec_a + (size_t)b * BZX_MAX_N);
    uint8_t *__restrict__ SEL = B.selector + (size_t)b * BZX_SEL_STRIDE;
    DcBits br;
    br.init(z, nbytes, starts[b] + 48);
    uint32_t err = 0;
    const uint32_t crc = br.get(32);
    const uint32_t randomised = br.get(1);
    const uint32_t orig = br.get(24);
    if (randomised) err |= DC_ERR_HEADER;          // never written by bzip2 >= 0.9.5 nor by the reference (compress_block.rs:41)
    // symbol map (symbol_map.rs:20-42)
    const uint32_t l1 = br.get(16);
    uint32_t n_in_use = 0;
    for (uint32_t i = 0; i < 16; i++) {
        if ((l1 >> (15 - i)) & 1u) {
            const uint32_t w = br.get(16);
            for (uint32_t j = 0; j < 16; j++)
                if ((w >> (15 - j)) & 1u) {
                    if (lane == 0) d_seq2byte[n_in_use] = (uint8_t)(i * 16 + j);
                    n_in_use++;
                }
        }
    }
    const uint32_t alpha = n_in_use + 2;
    const uint32_t n_groups = br.get(3);
    const uint32_t n_sel = br.get(15);
    if (n_in_use == 0 || n_groups < 2 || n_groups > 6 || n_sel < 1 || n_sel > BZX_MAX_SEL) err |= DC_ERR_HEADER;
    if (!err) {
        // selectors: unary MTF indices (decompress.rs:159-203)
        uint32_t pos[6] = {0, 1, 2, 3, 4, 5};
        for (uint32_t i = 0; i < n_sel && !err; i++) {
            uint32_t j = 0;
            while (br.get(1)) {
                j++;
                if (j >= n_groups || br.past_end()) {
                    err |= DC_ERR_HEADER;
                    break;
                }
            }
            if (err) break;
            const uint32_t v = pos[j];
            for (uint32_t k = j; k > 0; k--) pos[k] = pos[k - 1];
            pos[0] = v;
            if (lane == 0) SEL[i] = (uint8_t)v;
        }
        // code lengths, delta coded (decompress.rs:216-260)
        for (uint32_t t = 0; t < n_groups && !err; t++) {
            int32_t cur = (int32_t)br.get(5);
            for (uint32_t i = 0; i < alpha; i++) {
                for (;;) {
                    if (cur < 1 || cur > 20 || br.past_end()) {
                        err |= DC_ERR_HEADER;
                        break;
                    }
                    if (!br.get(1)) break;
                    cur += br.get(1) ? -1 : 1;
                }
                if (err) break;
                if (lane == 0) d_len[t][i] = (uint8_t)cur;
            }
        }
    }
    __syncthreads();
    if (!err && lane < n_groups) dc_make_tables(lane, alpha);
    for (uint32_t i = lane; i < 256; i += 64) d_count[i] = 0;
    __syncthreads();

    // ---- Huffman + MTF + RUNA/RUNB (decompress.rs:293-358, rle2_mtf.rs:191-287)
    uint32_t v0 = lane, v1 = lane + 64, v2 = lane + 128, v3 = lane + 192;      // MTF list: position p = lane + 64*slot
    uint32_t n = 0;                           // bytes decoded
    uint32_t pend_b = 0, pend_o = 0;          // my slot of the 64-entry output stage
    uint32_t run = 0, run_w = 1;              // pending zero run (bijective base 2)
    const uint32_t eob = n_in_use + 1;
    uint32_t g = 0, g_left = 0, tsel = 0;
    auto emit = [&](uint32_t byte, uint32_t count) {
        // `count` copies of `byte` at L[n ..]; occ = d_count[byte] ..; all lanes take part (wave-uniform arguments)
        const uint32_t c0 = d_count[byte];
        __syncthreads();
        if (lane == 0) d_count[byte] = c0 + count;
        uint32_t done = 0;
        while (done < count) {
            const uint32_t fill = n & 63u;
            const uint32_t k = count - done < 64u - fill ? count - done : 64u - fill;
            if (lane >= fill && lane < fill + k) {
                pend_b = byte;
                pend_o = c0 + done + (lane - fill);
            }
            n += k;
            done += k;
            if ((n & 63u) == 0) {
                L[n - 64 + lane] = (uint8_t)pend_b;
                OCC[n - 64 + lane] = pend_o;
            }
        }
    };
    while (!err) {
        if (g_left == 0) {
            if (g >= n_sel) {
                err |= DC_ERR_DATA;
                break;
            }
            tsel = SEL[g++];
            g_left = 50;
        }
        g_left--;
        // canonical code: shortest length whose limit admits the prefix
        const uint32_t v20 = br.peek(20);
        uint32_t zn = d_minlen[tsel];
        int32_t zvec = (int32_t)(v20 >> (20 - zn));
        while (zn <= 20 && zvec > d_limit[tsel][zn]) {
            zn++;
            zvec = (int32_t)(v20 >> (20 - zn));
        }
        const int32_t idx = zvec - d_base[tsel][zn];
        if (zn > 20 || idx < 0 || idx >= (int32_t)alpha || br.past_end()) {
            err |= DC_ERR_DATA;
            break;
        }
        br.pos += zn;
        const uint32_t sym = d_perm[tsel][idx];
        if (sym <= 1) {                       // RUNA / RUNB
            run += (sym + 1) * run_w;
            run_w <<= 1;
            if (run > max_n) {
                err |= DC_ERR_DATA;
                break;
            }
            continue;
        }
        if (run) {
            if (n + run > max_n) {
                err |= DC_ERR_DATA;
                break;
            }
            emit(d_seq2byte[__shfl(v0, 0)], run);
            run = 0;
            run_w = 1;
        }
        if (sym == eob) break;
        if (n + 1 > max_n) {
            err |= DC_ERR_DATA;
            break;
        }
        // move the entry at position r = sym - 1 (>= 1) to the front: new[p] = p == 0 ? x : p <= r ? old[p-1] : old[p]
        // with p = lane + 64 * slot; old[p-1] comes from the lane below, or from lane 63 of the slot before
        const uint32_t r = sym - 1;
        const uint32_t sr = r >> 6, lr = r & 63u;
        const uint32_t o0 = v0, o1 = v1, o2 = v2, o3 = v3;
        const uint32_t src = sr == 0 ? o0 : sr == 1 ? o1 : sr == 2 ? o2 : o3;
        const uint32_t x = __shfl(src, (int)lr);
        {
            const uint32_t up = __shfl_up(o0, 1);
            v0 = lane == 0 ? x : (lane <= r ? up : o0);
        }
        if (r >= 64) {                                   // (text rarely gets here: ranks are small)
            const uint32_t up = __shfl_up(o1, 1), c = __shfl(o0, 63);
            v1 = lane + 64 <= r ? (lane == 0 ? c : up) : o1;
        }
        if (r >= 128) {
            const uint32_t up = __shfl_up(o2, 1), c = __shfl(o1, 63);
            v2 = lane + 128 <= r ? (lane == 0 ? c : up) : o2;
        }
        if (r >= 192) {
            const uint32_t up = __shfl_up(o3, 1), c = __shfl(o2, 63);
            v3 = lane + 192 <= r ? (lane == 0 ? c : up) : o3;
        }
        emit(d_seq2byte[x], 1);
    }
    // flush the partial stage
    if ((n & 63u) && lane < (n & 63u)) {
        L[(n & ~63u) + lane] = (uint8_t)pend_b;
        OCC[(n & ~63u) + lane] = pend_o;
    }
    __syncthreads();
    for (uint32_t i = lane; i < 256; i += 64) B.freq[(size_t)b * 260 + i] = d_count[i];
    if (lane == 0) {
        if (!err && (n == 0 || orig >= n)) err |= DC_ERR_DATA;
        D.n = n;
        D.crc = crc;
        D.orig_ptr = orig;
        D.out_bit = starts[b];
        D.bits = br.pos;
        D.status = err;
        D.n_in_use = n_in_use;
    }
}

// ---- inverse BWT (bwt_sort.rs:91-130) ---------------------------------------------------------------------------
// grid (tiles, blocks): TT[C[L[i]] + occ[i]] = i, C = exclusive prefix of the byte counts.
__global__ __launch_bounds__(256) void bzx_dc_scatter_kernel(BzxBatch B)
{
    __shared__ uint32_t c_start[256];
    __shared__ uint32_t c_scan[4];
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    if (B.blk[b].status) return;
    const uint32_t n = B.blk[b].n;
    {
        const uint32_t v = B.freq[(size_t)b * 260 + tid];
        const uint32_t incl = bzx_wave_incl_sum(v);
        if (bzx_lane() == 63) c_scan[bzx_wave()] = incl;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t w = 0; w < bzx_wave(); w++) pre += c_scan[w];
        c_start[tid] = pre + incl - v;
    }
    __syncthreads();
    const uint8_t *__restrict__ L = B.bwt + (size_t)b * BZX_BLK_STRIDE;
    const uint32_t *__restrict__ OCC = reinterpret_cast<const uint32_t *>(B.rec_a + (size_t)b * BZX_MAX_N);
    uint32_t *__restrict__ TT = reinterpret_cast<uint32_t *>(B.rec_b + (size_t)b * BZX_MAX_N);
    for (uint32_t i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const uint32_t j = c_start[L[i]] + OCC[i];
        if (j < n) TT[j] = i;
    }
}

// tt[j] = TT[j] << 8 | L[j]: one word per step of the walk (overwrites the occ array, no longer needed)
__global__ __launch_bounds__(256) void bzx_dc_pack_kernel(BzxBatch B)
{
    const uint32_t b = blockIdx.y;
    if (B.blk[b].status) return;
    const uint32_t n = B.blk[b].n;
    const uint8_t *__restrict__ L = B.bwt + (size_t)b * BZX_BLK_STRIDE;
    const uint32_t *__restrict__ TT = reinterpret_cast<const uint32_t *>(B.rec_b + (size_t)b * BZX_MAX_N);
    uint32_t *__restrict__ W = reinterpret_cast<uint32_t *>(B.rec_a + (size_t)b * BZX_MAX_N);
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) W[j] = (TT[j] << 8) | L[j];
}

struct DcCheck {
    uint32_t out_pos;       // expanded bytes before this image position
    uint32_t state;         // last byte | equal-run length so far (0..4) << 8
};

// One LANE per block: the pointer walk (a serial chain of dependent loads: ~n memory latencies per block, all blocks
// of the stream in flight).  Writes the RLE1 image, its expanded length and a checkpoint every 4096 image bytes.
__global__ __launch_bounds__(64) void bzx_dc_walk_kernel(BzxBatch B, uint8_t *__restrict__ img_slabs)
{
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B.nblk || B.blk[b].status) return;
    const uint32_t n = B.blk[b].n;
    const uint32_t *__restrict__ W = reinterpret_cast<const uint32_t *>(B.rec_a + (size_t)b * BZX_MAX_N);
    uint8_t *__restrict__ IMG = img_slabs + (size_t)b * BZX_BLK_STRIDE;
    DcCheck *__restrict__ CK = reinterpret_cast<DcCheck *>(B.gbits + (size_t)b * BZX_SEL_STRIDE);
    uint32_t tpos = W[B.blk[b].orig_ptr] >> 8;
    uint32_t last = 256, cnt = 0;
    uint64_t out = 0;
    for (uint32_t k = 0; k < n; k++) {
        if ((k & ((1u << DC_CK_SHIFT) - 1u)) == 0) {
            DcCheck c;
            c.out_pos = (uint32_t)out;
            c.state = (last & 0x1FFu) | (cnt << 9);
            CK[k >> DC_CK_SHIFT] = c;
        }
        if (tpos >= n) {                       // corrupt permutation: stop (the CRC check reports the block)
            B.blk[b].status = DC_ERR_DATA;
            break;
        }
        const uint32_t e = W[tpos];
        const uint32_t ch = e & 255u;
        tpos = e >> 8;
        IMG[k] = (uint8_t)ch;
        // RLE1 (rle1.rs:267-316): after four equal bytes the next one is a repeat count; then a fresh run starts
        if (cnt == 4) {
            out += ch;
            cnt = 0;
            last = 256;
        } else {
            cnt = ch == last ? cnt + 1 : 1;
            last = ch;
            out++;
        }
    }
    B.blk[b].pack_word = out;                  // expanded length
}

// grid (segments, blocks): expands one 4096-byte segment of the RLE1 image from its checkpoint to out + off[b].
__global__ __launch_bounds__(64) void bzx_dc_expand_kernel(BzxBatch B, const uint8_t *__restrict__ img_slabs,
                                                          const uint64_t *__restrict__ off, uint8_t *__restrict__ out,
                                                          uint64_t cap)
{
    const uint32_t b = blockIdx.y;
    if (B.blk[b].status) return;
    const uint32_t n = B.blk[b].n;
    const uint32_t seg = blockIdx.x * 64 + threadIdx.x;
    const uint32_t k0 = seg << DC_CK_SHIFT;
    if (k0 >= n) return;
    const uint32_t k1 = k0 + (1u << DC_CK_SHIFT) < n ? k0 + (1u << DC_CK_SHIFT) : n;
    const uint8_t *__restrict__ IMG = img_slabs + (size_t)b * BZX_BLK_STRIDE;
    const DcCheck c = reinterpret_cast<const DcCheck *>(B.gbits + (size_t)b * BZX_SEL_STRIDE)[seg];
    uint64_t o = off[b] + c.out_pos;
    uint32_t last = c.state & 0x1FFu, cnt = c.state >> 9;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t ch = IMG[k];
        if (cnt == 4) {
            for (uint32_t r = 0; r < ch; r++)
                if (o + r < cap) out[o + r] = (uint8_t)last;
            o += ch;
            cnt = 0;
            last = 256;
        } else {
            cnt = ch == last ? cnt + 1 : 1;
            last = ch;
            if (o < cap) out[o] = (uint8_t)ch;
            o++;
        }
    }
}

void bzx_launch_dc_scan(const uint8_t *z, uint64_t nbytes, uint64_t *found, uint32_t *n_found, uint32_t cap, uint32_t grid,
                        hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_dc_scan_kernel, dim3(grid), dim3(256), 0, stream, z, nbytes, found, n_found, cap);
}
void bzx_launch_dc_decode(const BzxBatch &B, const uint8_t *z, uint64_t nbytes, const uint64_t *starts, uint32_t max_n,
                          hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_dc_decode_kernel, dim3(B.nblk), dim3(64), 0, stream, B, z, nbytes, starts, max_n);
}
void bzx_launch_dc_ibwt(const BzxBatch &B, uint8_t *img_slabs, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_dc_scatter_kernel, dim3(32, B.nblk), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(bzx_dc_pack_kernel, dim3(32, B.nblk), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(bzx_dc_walk_kernel, dim3((B.nblk + 63) / 64), dim3(64), 0, stream, B, img_slabs);
}
void bzx_launch_dc_expand(const BzxBatch &B, const uint8_t *img_slabs, const uint64_t *off, uint8_t *out, uint64_t cap,
                          hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_dc_expand_kernel, dim3((DC_CK_STRIDE + 63) / 64, B.nblk), dim3(64), 0, stream, B, img_slabs, off, out, cap);
}
