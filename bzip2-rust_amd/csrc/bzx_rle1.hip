// bzx_rle1.hip -- RLE1, block splitting and block CRCs of a whole raw buffer on gfx950.
//
// Contract (reference src/tools/rle1.rs:33-263 RLE1Block + src/tools/crc.rs:15-22 do_crc):
// raw stream -> blocks of RLE1'd bytes (runs of 4..255 equal bytes -> 4 bytes + count) of at
// most 100000*level-19(+4+5) bytes, each with the CRC-32/BZIP2 of the raw bytes it covers.
// The split rule is libbz2's: the raw stream is cut into PIECES (maximal runs,
// cut every 255 bytes); a block is a whole number of pieces and ends with the first piece that
// brings its RLE1 length to >= nblockMAX; the pending run moves to the next block whole.
//
// The reference does this byte-serially under a lock (compress.rs:125-128).  Here:
//   A  per 8 KiB tile: last run start                      -> scan S1 (tile run-start carry-in)
//   B  per tile: RLE1 bytes emitted by the tile            -> scan S2 (tile RLE1 offsets F)
//      position p with k = (p - runstart(p)) mod 255 emits  k<3: 1 byte, k==3: 2 bytes (the 4th
//      copy + the count), k>3: nothing -- purely local once the run start is known.
//   C  block boundaries: a short serial chain over blocks (one workgroup): smallest position x
//      with F(x) >= F(start)+nblockMAX by search over the tile offsets and one tile scan, then
//      the end of the piece containing x.
//   D  per tile: scatter the emitted bytes into the block slabs (count byte = piece length - 4
//      by a <= 251 byte look-ahead)
//   E  per block: CRC of its raw range: per-lane table CRC of a chunk from a zero register,
//      combined with x^(8*len) mod P multiplications (GF(2) polynomial arithmetic).
#include <hip/hip_runtime.h>
#include <string.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define RL_NT 256
#define RL_BYTES 32
#define RL_TILE (RL_NT * RL_BYTES)   // 8192

// Per-lane view of 32 consecutive raw bytes of a tile.
struct TileLane {
    uint32_t w[8];         // the bytes
    uint64_t p0;           // raw position of byte 0
    uint32_t nvalid;       // bytes inside the input
    uint32_t prev;         // byte before p0 (256 if p0 == 0)
};

__device__ __forceinline__ void tile_load(const uint8_t *__restrict__ raw, uint64_t len, uint64_t tile, TileLane &t)
{
    t.p0 = tile * RL_TILE + (uint64_t)threadIdx.x * RL_BYTES;
    t.nvalid = t.p0 >= len ? 0u : (len - t.p0 < RL_BYTES ? (uint32_t)(len - t.p0) : (uint32_t)RL_BYTES);
#pragma unroll
    for (int i = 0; i < 8; i++) t.w[i] = 0;
    if (t.nvalid == RL_BYTES) {
        const uint4 a = *reinterpret_cast<const uint4 *>(raw + t.p0);
        const uint4 b = *reinterpret_cast<const uint4 *>(raw + t.p0 + 16);
        t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
        t.w[4] = b.x; t.w[5] = b.y; t.w[6] = b.z; t.w[7] = b.w;
    } else {
        for (uint32_t i = 0; i < t.nvalid; i++) t.w[i >> 2] |= (uint32_t)raw[t.p0 + i] << (8 * (i & 3));
    }
    t.prev = (t.p0 == 0 || t.nvalid == 0) ? 256u : raw[t.p0 - 1];
}

__device__ __forceinline__ uint32_t tile_byte(const TileLane &t, int i) { return (t.w[i >> 2] >> (8 * (i & 3))) & 255u; }

// last run start (+1) among my bytes, 0 if none
__device__ __forceinline__ uint64_t lane_last_rs(const TileLane &t)
{
    uint64_t rs = 0;
    uint32_t prev = t.prev;
#pragma unroll
    for (int i = 0; i < RL_BYTES; i++) {
        const uint32_t c = tile_byte(t, i);
        if ((uint32_t)i < t.nvalid && c != prev) rs = t.p0 + i + 1;
        prev = c;
    }
    return rs;
}

// Block-wide exclusive max scan of 64-bit values (0 = identity).  scratch: RL_NT/64 words.
__device__ __forceinline__ uint64_t block_excl_max64(uint64_t v, uint64_t *scratch, uint64_t &total)
{
    const uint32_t lane = bzx_lane(), wave = bzx_wave();
    uint64_t x = v;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d);
        if (lane >= d && y > x) x = y;
    }
    uint64_t ex = __shfl_up(x, 1);
    if (lane == 0) ex = 0;
    if (lane == 63) scratch[wave] = x;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
    for (uint32_t i = 0; i < RL_NT / 64; i++) {
        const uint64_t s = scratch[i];
        if (i < wave && s > pre) pre = s;
        if (s > tot) tot = s;
    }
    __syncthreads();
    total = tot;
    return ex > pre ? ex : pre;
}

// ---- A: last run start per tile
__global__ __launch_bounds__(RL_NT) void bzx_rl_runstart_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                                uint64_t t_lo, uint64_t ntiles, BzxSplitWs ws)
{
    // tiles [t_lo, ntiles): all of them, or one rank's share of a sharded analysis (bzx_shard_scan_runs)
    __shared__ uint64_t scratch[RL_NT / 64];
    for (uint64_t tile = t_lo + blockIdx.x; tile < ntiles; tile += gridDim.x) {
        TileLane t;
        bool run4 = true;                   // "this tile may hold a run position k >= 3": decided exactly by kernel B
        if (tile > 0 && (tile + 1) * RL_TILE <= len) {
            // Full inner tile: one load serves the run starts and a cheap test for B: a tile in which no four
            // consecutive equal bytes end (looking 3 bytes back into the previous tile) has no run position
            // k >= 3, so RLE1 copies it.
            t.p0 = tile * RL_TILE + (uint64_t)threadIdx.x * RL_BYTES;
            t.nvalid = RL_BYTES;
            uint32_t w0;
            __builtin_memcpy(&w0, raw + t.p0 - 4, 4);
            const uint4 a = *reinterpret_cast<const uint4 *>(raw + t.p0);
            const uint4 b = *reinterpret_cast<const uint4 *>(raw + t.p0 + 16);
            t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
            t.w[4] = b.x; t.w[5] = b.y; t.w[6] = b.z; t.w[7] = b.w;
            t.prev = w0 >> 24;
            uint64_t m = 0;                      // bit i: byte i of the 36 (from p0 - 4) equals byte i+1
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const uint32_t cur = i ? t.w[i - 1] : w0;
                const uint32_t nxt = i < 8 ? t.w[i] : ~t.w[7];
                const uint32_t z = cur ^ ((cur >> 8) | (nxt << 24));
                const uint32_t f = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu) >> 7;   // 0x01 in exactly the zero bytes
                m |= (uint64_t)((f * 0x01020408u) >> 24 & 0xFu) << (4 * i);
            }
            m >>= 1;                             // pairs starting at byte p0-3 and later
            run4 = (m & (m >> 1) & (m >> 2)) != 0;
        } else {
            tile_load(raw, len, tile, t);
        }
        uint64_t tot;
        (void)block_excl_max64(lane_last_rs(t), scratch, tot);
        const bool any4 = __syncthreads_or(run4);
        if (threadIdx.x == 0) {
            ws.tile_rs[tile] = tot;
            ws.tile_np[tile] = any4 ? 1 : 0;     // provisional: 0 = plain for sure, B skips the tile
            ws.tile_off[tile] = RL_TILE;
        }
    }
}

// ---- S1 / S2: single-workgroup exclusive scans over the tile summaries
__global__ __launch_bounds__(1024) void bzx_rl_scan_kernel(uint64_t *v_all, uint64_t n_all, int is_max, uint64_t seg,
                                                           uint64_t *tot_out)
{
    // Workgroup g: v_all[g seg .. min(n_all, (g+1) seg)) -> exclusive scan in place; tot_out[g] = its total.
    // (One workgroup with seg >= n_all and tot_out = v_all + n_all scans a whole array.)  8 consecutive elements
    // per lane, the next iteration's elements already in flight; one barrier per iteration (wave totals
    // double-buffered, every lane keeps the running carry itself).
    const uint64_t lo_ = (uint64_t)blockIdx.x * seg;
    if (lo_ >= n_all && blockIdx.x) return;
    uint64_t *v = v_all + lo_;
    const uint64_t n = lo_ >= n_all ? 0 : (n_all - lo_ < seg ? n_all - lo_ : seg);
    __shared__ uint64_t wsum[2][16];
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    constexpr int E = 8;
    uint64_t carry = 0;
    uint64_t nx[E];
#pragma unroll
    for (int j = 0; j < E; j++) nx[j] = (uint64_t)tid * E + j < n ? v[(uint64_t)tid * E + j] : 0ull;
    int buf = 0;
    for (uint64_t i0 = 0; i0 < n; i0 += 1024 * E, buf ^= 1) {
        const uint64_t ib = i0 + (uint64_t)tid * E;
        uint64_t a[E];
#pragma unroll
        for (int j = 0; j < E; j++) {
            a[j] = nx[j];
            const uint64_t in = ib + 1024 * E + j;
            nx[j] = in < n ? v[in] : 0ull;
        }
        uint64_t mine = 0;                 // combination of my E elements
#pragma unroll
        for (int j = 0; j < E; j++) mine = is_max ? (a[j] > mine ? a[j] : mine) : mine + a[j];
        uint64_t x = mine;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d);
            if (lane >= d) x = is_max ? (y > x ? y : x) : x + y;
        }
        uint64_t ex = __shfl_up(x, 1);
        if (lane == 0) ex = 0;
        if (lane == 63) wsum[buf][wave] = x;
        __syncthreads();
        uint64_t pre = 0, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            const uint64_t s = wsum[buf][w];
            if (is_max) {
                if (w < wave && s > pre) pre = s;
                if (s > tot) tot = s;
            } else {
                if (w < wave) pre += s;
                tot += s;
            }
        }
        uint64_t r;
        if (is_max) {
            r = ex > pre ? ex : pre;
            if (carry > r) r = carry;
        } else {
            r = carry + pre + ex;
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            if (ib + j < n) v[ib + j] = r;
            r = is_max ? (a[j] > r ? a[j] : r) : r + a[j];
        }
        carry = is_max ? (tot > carry ? tot : carry) : carry + tot;
    }
    if (tid == 0) tot_out[blockIdx.x] = carry;
}

// second level of a segmented scan: element i of segment g gets the scanned segment total off[g] combined in
__global__ __launch_bounds__(1024) void bzx_rl_scan_add_kernel(uint64_t *v, uint64_t n, int is_max, uint64_t seg,
                                                               const uint64_t *off, uint64_t nseg)
{
    const uint64_t o = off[blockIdx.x];
    const uint64_t lo = (uint64_t)blockIdx.x * seg, hi = lo + seg < n ? lo + seg : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const uint64_t x = v[i];
        v[i] = is_max ? (x > o ? x : o) : x + o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) v[n] = off[nseg];
}

#ifndef SCAN_SEG
#define SCAN_SEG 8192        // the emulator build uses a tiny segment so that small inputs take the two-level path
#endif
// v[0..n) -> exclusive scan in place, v[n] = total.  segtot: scratch of n / SCAN_SEG + 2 words.
static void launch_scan(hipStream_t st, uint64_t *v, uint64_t n, int is_max, uint64_t *segtot)
{
    if (n <= 2 * SCAN_SEG) {
        hipLaunchKernelGGL(bzx_rl_scan_kernel, dim3(1), dim3(1024), 0, st, v, n, is_max, n ? n : 1, v + n);
        return;
    }
    const uint64_t nseg = (n + SCAN_SEG - 1) / SCAN_SEG;
    hipLaunchKernelGGL(bzx_rl_scan_kernel, dim3((uint32_t)nseg), dim3(1024), 0, st, v, n, is_max, (uint64_t)SCAN_SEG, segtot);
    hipLaunchKernelGGL(bzx_rl_scan_kernel, dim3(1), dim3(1024), 0, st, segtot, nseg, is_max, nseg, segtot + nseg);
    hipLaunchKernelGGL(bzx_rl_scan_add_kernel, dim3((uint32_t)nseg), dim3(1024), 0, st, v, n, is_max, (uint64_t)SCAN_SEG,
                       segtot, nseg);
}

// Emission analysis of my 32 bytes: e[i] in {0,1,2} packed 2 bits each; returns my emitted byte count.
// rs_in = run start (+1) carried into the tile (0 = none, only possible at p == 0).
__device__ __forceinline__ uint32_t lane_emission(const TileLane &t, uint64_t rs_in_plus1, uint64_t &e_bits,
                                                  uint32_t &k_first, bool &any_long)
{
    any_long = false;
    // run start for my first byte: either inside earlier lanes of the tile / earlier tiles (rs_in) or my own byte
    uint32_t k = 0;
    if (t.nvalid) {
        const uint64_t rs = rs_in_plus1 ? rs_in_plus1 - 1 : 0;   // p0 == 0 has rs_in == 0 and starts a run itself
        k = (uint32_t)((t.p0 - rs) % 255u);
    }
    uint32_t prev = t.prev, cnt = 0;
    uint64_t bits = 0;
    k_first = k;
#pragma unroll
    for (int i = 0; i < RL_BYTES; i++) {
        const uint32_t c = tile_byte(t, i);
        if ((uint32_t)i < t.nvalid) {
            if (c != prev) k = 0;
            if (i == 0) k_first = k;
            const uint32_t e = k < 3 ? 1u : (k == 3 ? 2u : 0u);
            if (k >= 3) any_long = true;
            bits |= (uint64_t)e << (2 * i);
            cnt += e;
            k = (k + 1 == 255) ? 0 : k + 1;
        }
        prev = c;
    }
    e_bits = bits;
    return cnt;
}

// Shared tile analysis: every lane gets its bytes, emission bits and F (RLE1 offset, tile relative) of its first byte.
struct TileInfo {
    TileLane t;
    uint64_t e_bits;
    uint32_t k_first;
    uint32_t f_excl;     // emitted bytes of the tile before my first byte
    uint32_t f_total;    // emitted bytes of the whole tile
    bool any_long;       // my bytes contain a run position with k >= 3 (RLE1 is not the identity here)
};

__device__ __forceinline__ void tile_analyse(const uint8_t *__restrict__ raw, uint64_t len, uint64_t tile,
                                             const BzxSplitWs &ws, uint64_t *scratch64, uint32_t *scratch32, TileInfo &ti)
{
    tile_load(raw, len, tile, ti.t);
    uint64_t tot;
    const uint64_t rs_prev = block_excl_max64(lane_last_rs(ti.t), scratch64, tot);
    const uint64_t carry = ws.tile_rs[tile];
    const uint64_t rs_in = rs_prev ? rs_prev : carry;
    const uint32_t cnt = lane_emission(ti.t, rs_in, ti.e_bits, ti.k_first, ti.any_long);
    ti.f_excl = bzx_block_excl_sum<RL_NT>(cnt, scratch32, ti.f_total);
}

// ---- B: emitted bytes per tile
__global__ __launch_bounds__(RL_NT) void bzx_rl_count_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                             uint64_t t_lo, uint64_t ntiles, BzxSplitWs ws)
{
    __shared__ uint64_t s64[RL_NT / 64];
    __shared__ uint32_t s32[RL_NT / 64];
    // kernel A flagged the tiles that may hold a run position k >= 3; only those are analysed here.  A workgroup
    // takes 256 tiles at a time (one flag per lane, coalesced) and walks the flagged ones.
    __shared__ uint32_t s_list[RL_NT];
    __shared__ uint32_t s_cnt;
    for (uint64_t base = t_lo + (uint64_t)blockIdx.x * RL_NT; base < ntiles; base += (uint64_t)gridDim.x * RL_NT) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        {
            const uint64_t tile = base + threadIdx.x;
            if (tile < ntiles && !(tile > 0 && (tile + 1) * RL_TILE <= len && ws.tile_np[tile] == 0))
                s_list[atomicAdd(&s_cnt, 1u)] = threadIdx.x;
        }
        __syncthreads();
        const uint32_t nlist = s_cnt;
        for (uint32_t k = 0; k < nlist; k++) {
            const uint64_t tile = base + s_list[k];
            TileInfo ti;
            tile_analyse(raw, len, tile, ws, s64, s32, ti);
            const bool np = __syncthreads_or(ti.any_long);
            if (threadIdx.x == 0) {
                ws.tile_off[tile] = ti.f_total;
                ws.tile_np[tile] = np ? 1 : 0;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// end of the piece that contains position x (x < len): first piece start after x
__device__ uint64_t piece_end(const uint8_t *__restrict__ raw, uint64_t len, uint64_t x, uint32_t k_at_x)
{
    const uint8_t c = raw[x];
    uint64_t q = x + 1;
    uint32_t k = k_at_x + 1;
    while (q < len && k < 255 && raw[q] == c) {
        q++;
        k++;
    }
    return q;
}

#define BND_J 64                       // blocks per batch of the fast path
#define BND_W (4 * BND_J + 16)         // window bytes per predicted boundary (drift <= 4 per block), multiple of 16
// ---- C: block boundaries (single workgroup, serial over blocks)
__global__ __launch_bounds__(RL_NT) void bzx_rl_boundaries_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                                  uint64_t ntiles, uint32_t nmax, BzxSplitWs ws)
{
    __shared__ uint64_t s64[RL_NT / 64];
    __shared__ uint32_t s32[RL_NT / 64];
    __shared__ uint64_t s_x;      // candidate position
    __shared__ uint32_t s_k;      // k at the candidate
    __shared__ uint64_t s_next, s_fnext;
    // windows of raw bytes around the next BND_J predicted boundaries (see the batched fast path below)
    __shared__ __attribute__((aligned(16))) uint8_t s_win[BND_J][BND_W];
    const uint32_t tid = threadIdx.x;
    uint64_t start = 0, f_start = 0;
    uint32_t nb = 0;
    const uint64_t f_len = ws.tile_off[ntiles];
    while (start < len && nb < ws.max_blocks) {
        // ---- batched fast path: when no tile that the next J blocks can touch has a run position k >= 3, RLE1 is
        // the identity on all of them and block j ends within 4 bytes of start + (j+1) nmax + (0..4j).  The raw
        // bytes around all J predicted boundaries are fetched in one round trip into LDS, then the serial chain
        // (every lane runs it redundantly, lane 0 stores) costs LDS latency per block instead of HBM latency.
        {
            uint32_t J = BND_J;
            const uint64_t room = len > start + BND_W + 16 ? (len - start - BND_W - 16) / nmax : 0;   // blocks that end well before len
            if (room < J + 1) J = room > 1 ? (uint32_t)room - 1 : 0;
            if (nb + J > ws.max_blocks) J = ws.max_blocks - nb;
            if (J >= 2) {
                const uint64_t last = start + (uint64_t)(J + 1) * nmax + 4 * J + 8;
                uint64_t t1 = last / RL_TILE + 2;
                if (t1 > ntiles) t1 = ntiles;
                if (ws.tile_np[t1] != ws.tile_np[start / RL_TILE]) J = 0;
            }
            if (J >= 2) {
                for (uint32_t i = tid; i < J * (BND_W / 16); i += RL_NT) {
                    const uint32_t j = i / (BND_W / 16), c = i % (BND_W / 16);
                    uint4 v;
                    __builtin_memcpy(&v, raw + start + (uint64_t)(j + 1) * nmax - 4 + 16 * c, 16);
                    *reinterpret_cast<uint4 *>(&s_win[j][16 * c]) = v;
                }
                __syncthreads();
                const uint64_t start0 = start;
                for (uint32_t j = 0; j < J; j++) {
                    if (tid == 0) {
                        ws.blk_raw[nb] = start;
                        ws.blk_f[nb] = f_start;
                        ws.blk_plain[nb] = 1;
                    }
                    nb++;
                    const uint64_t x = start + nmax;
                    const uint8_t *w = &s_win[j][(uint32_t)(start - (start0 + (uint64_t)j * nmax))];   // bytes x-4 .. x+3
                    uint32_t k = 0;
                    while (k < 3 && w[3 - k] == w[4]) k++;
                    uint64_t q = x;
                    if (k) {
                        q = x + 1;
                        uint32_t kk = k + 1;
                        while (kk < 4 && q < x + 4 && w[4 + (q - x)] == w[4]) {
                            q++;
                            kk++;
                        }
                    }
                    f_start += q - start;
                    start = q;
                }
                __syncthreads();
                continue;
            }
        }
        if (tid == 0) {
            ws.blk_raw[nb] = start;
            ws.blk_f[nb] = f_start;
            ws.blk_plain[nb] = 0;
        }
        nb++;
        const uint64_t target = f_start + nmax;
        if (f_len < target) {
            // last block: plain if no tile from here to the end has a run position k >= 3
            if (tid == 0 && ws.tile_np[ntiles] == ws.tile_np[start / RL_TILE]) ws.blk_plain[nb - 1] = 1;
            start = len;
            f_start = f_len;
            break;
        }
        // ---- fast path: RLE1 is the identity on every tile this block can touch (no run position k >= 3),
        // so F(x) - F(start) = x - start and the boundary is the end of the piece around start + nmax.
        {
            const uint64_t x = start + nmax;
            const uint64_t t0 = start / RL_TILE;
            uint64_t t1 = x / RL_TILE + 2;
            if (t1 > ntiles) t1 = ntiles;
            if (x + 4 < len && x >= 4 && ws.tile_np[t1] == ws.tile_np[t0]) {
                uint8_t w[8];
                __builtin_memcpy(w, raw + x - 4, 8);        // bytes x-4 .. x+3
                // k(x) = equal bytes immediately before x (runs are <= 3 long here)
                uint32_t k = 0;
                while (k < 3 && w[3 - k] == w[4]) k++;
                uint64_t q = x;
                if (k) {
                    q = x + 1;
                    uint32_t kk = k + 1;
                    while (kk < 4 && q < x + 4 && w[4 + (q - x)] == w[4]) {
                        q++;
                        kk++;
                    }
                }
                if (tid == 0) ws.blk_plain[nb - 1] = 1;
                start = q;
                f_start = f_start + (q - (x - nmax));
                continue;
            }
        }
        // largest tile with F(tile start) < target   (tile_off is non-decreasing; tile_off[0] = 0 < target)
        uint64_t lo = 0, hi = ntiles;   // invariant: tile_off[lo] < target; hi = first tile with tile_off >= target or ntiles
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ws.tile_off[mid] < target) lo = mid; else hi = mid;
        }
        const uint64_t tile = lo;
        TileInfo ti;
        tile_analyse(raw, len, tile, ws, s64, s32, ti);
        // smallest x with F(x) >= target inside this tile, or the start of the next tile
        if (tid == 0) {
            s_x = (tile + 1) * RL_TILE < len ? (tile + 1) * RL_TILE : len;
            s_k = 0xffffffffu;
        }
        __syncthreads();
        {
            const uint64_t f0 = ws.tile_off[tile] + ti.f_excl;
            uint64_t f = f0;
            uint32_t k = ti.k_first;
            uint32_t prev = ti.t.prev;
            bool found = false;
            uint64_t fx = 0;
            uint32_t kx = 0;
            for (int i = 0; i < RL_BYTES; i++) {
                if ((uint32_t)i < ti.t.nvalid) {
                    const uint32_t c = tile_byte(ti.t, i);
                    if (i > 0) k = (c != prev) ? 0u : (k + 1 == 255 ? 0u : k + 1);
                    if (!found && f >= target) {
                        found = true;
                        fx = ti.t.p0 + i;
                        kx = k;
                    }
                    f += (ti.e_bits >> (2 * i)) & 3u;
                    prev = c;
                }
            }
            // lanes are ordered by position: the first lane that found one owns the minimum
            const uint64_t any = __ballot(found);
            if (any && (int)bzx_lane() == __ffsll((unsigned long long)any) - 1) {
                // lowest wave wins: atomicMin on position
                atomicMin((unsigned long long *)&s_x, (unsigned long long)fx);
            }
            __syncthreads();
            if (found && fx == s_x) s_k = kx;
            __syncthreads();
        }
        if (tid == 0) {
            uint64_t x = s_x, q;
            if (x >= len) {
                q = len;
            } else if (s_k == 0xffffffffu) {
                // x is the first byte of the next tile: its k is not known here; derive it from its run start
                uint64_t rs1 = ws.tile_rs[tile + 1];        // run start (+1) carried into that tile
                const bool newrun = raw[x] != raw[x - 1];
                uint32_t k = newrun ? 0u : (uint32_t)((x - (rs1 ? rs1 - 1 : 0)) % 255u);
                q = (k == 0) ? x : piece_end(raw, len, x, k);
            } else {
                q = (s_k == 0) ? x : piece_end(raw, len, x, s_k);
            }
            s_next = q;
        }
        __syncthreads();
        const uint64_t q = s_next;
        // F(q): emitted bytes before q
        if (q >= len) {
            if (tid == 0) s_fnext = f_len;
        } else {
            const uint64_t qt = q / RL_TILE;
            TileInfo tq;
            tile_analyse(raw, len, qt, ws, s64, s32, tq);
            uint64_t f = ws.tile_off[qt] + tq.f_excl;
            for (int i = 0; i < RL_BYTES; i++) {
                if (tq.t.p0 + i == q) s_fnext = f;
                f += (tq.e_bits >> (2 * i)) & 3u;
            }
        }
        __syncthreads();
        start = q;
        f_start = s_fnext;
        __syncthreads();
    }
    if (tid == 0) {
        ws.blk_raw[nb] = len;
        ws.blk_f[nb] = f_len;
        ws.nblk[0] = (start < len) ? 0xffffffffu : nb;   // more blocks than max_blocks: error marker
    }
}

// ---- D: scatter the emitted bytes into the block slabs + fill the block descriptors' in_off / n
// own_first/own_step: only blocks b = own_first (mod own_step) are materialised (round-robin sharding over GPUs).
__global__ __launch_bounds__(RL_NT) void bzx_rl_scatter_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                               uint64_t ntiles, BzxSplitWs ws, uint8_t *__restrict__ slabs,
                                                               BzxBlock *__restrict__ blk, uint32_t own_first,
                                                               uint32_t own_step)
{
    __shared__ uint64_t s64[RL_NT / 64];
    __shared__ uint32_t s32[RL_NT / 64];
    const uint32_t nblk = ws.nblk[0];
    const bool all_plain = ws.tile_np[ntiles] == 0;      // no run position k >= 3 anywhere: every block is zero-copy
    for (uint64_t tile = blockIdx.x; tile < ntiles && !all_plain; tile += gridDim.x) {
        {
            // skip tiles that lie entirely in blocks of other ranks or in zero-copy blocks (uniform decision)
            const uint64_t pa = tile * RL_TILE;
            const uint64_t pb = (pa + RL_TILE < len ? pa + RL_TILE : len) - 1;
            uint32_t lo = 0, hi = nblk;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ws.blk_raw[mid] <= pa) lo = mid; else hi = mid;
            }
            const uint32_t ka = lo;
            const uint32_t kb2 = (ka + 1 < nblk && ws.blk_raw[ka + 1] <= pb) ? ka + 1 : ka;
            const bool need_a = (ka % own_step) == own_first && !ws.blk_plain[ka];
            const bool need_b = (kb2 % own_step) == own_first && !ws.blk_plain[kb2];
            if (!need_a && !need_b) continue;
        }
        TileInfo ti;
        tile_analyse(raw, len, tile, ws, s64, s32, ti);
        if (ti.t.nvalid) {
            // block of my first byte: last block with blk_raw <= p0
            uint32_t lo = 0, hi = nblk;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ws.blk_raw[mid] <= ti.t.p0) lo = mid; else hi = mid;
            }
            uint32_t kb = lo;
            uint64_t next_raw = ws.blk_raw[kb + 1];
            uint64_t f0 = ws.blk_f[kb];
            uint8_t *dst = slabs + (size_t)(kb / own_step) * BZX_BLK_STRIDE;      // (slab of an owned block: its local index)
            uint64_t f = ws.tile_off[tile] + ti.f_excl;
            for (int i = 0; i < RL_BYTES; i++) {
                if ((uint32_t)i < ti.t.nvalid) {
                    const uint64_t p = ti.t.p0 + i;
                    if (p >= next_raw) {
                        kb++;
                        next_raw = ws.blk_raw[kb + 1];
                        f0 = ws.blk_f[kb];
                        dst = slabs + (size_t)(kb / own_step) * BZX_BLK_STRIDE;
                    }
                    const uint32_t e = (uint32_t)(ti.e_bits >> (2 * i)) & 3u;
                    if (e && ((kb % own_step) != own_first || ws.blk_plain[kb])) {
                        f += e;
                    } else if (e) {
                        const uint32_t c = tile_byte(ti.t, i);
                        dst[f - f0] = (uint8_t)c;
                        if (e == 2) {
                            // 4th byte of a piece: count the rest of the piece (<= 251 more equal bytes)
                            uint64_t q = p + 1;
                            uint32_t extra = 0;
                            while (q < len && extra < 251 && raw[q] == c) {
                                q++;
                                extra++;
                            }
                            dst[f - f0 + 1] = (uint8_t)extra;
                        }
                        f += e;
                    }
                }
            }
        }
        __syncthreads();
    }
    for (uint32_t b = blockIdx.x * RL_NT + threadIdx.x; b < nblk; b += gridDim.x * RL_NT) {
        blk[b].in_off = (ws.blk_plain[b] || all_plain) ? (BZX_IN_RAW | ws.blk_raw[b]) : (uint64_t)(b / own_step) * BZX_BLK_STRIDE;
        blk[b].n = (uint32_t)(ws.blk_f[b + 1] - ws.blk_f[b]);
        blk[b].status = 0;
    }
}

// ---- E: CRC-32/BZIP2 of every block's raw range
__device__ __forceinline__ uint32_t gf_mulmod(uint32_t a, uint32_t b)
{
    // a(x) * b(x) mod P(x), P = x^32 + 0x04C11DB7, bit 31 = x^31
    uint32_t r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? 0x04C11DB7u : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

// x^(8*nbytes) mod P
__device__ uint32_t gf_xpow8(uint64_t nbytes)
{
    uint32_t result = 1u;            // the polynomial "1"
    uint32_t sq = 0x100u;            // x^8
    while (nbytes) {
        if (nbytes & 1ull) result = gf_mulmod(result, sq);
        sq = gf_mulmod(sq, sq);
        nbytes >>= 1;
    }
    return result;
}

#define CRC_NT 1024
#define CRC_SUB 64                         // bytes per lane per tile
#define CRC_TILE (CRC_NT * CRC_SUB)        // 64 KiB
#define CRC_PITCH 72                       // LDS row pitch of a lane's 64 bytes (8-byte aligned, spreads banks)

// One workgroup per block.  The block's raw bytes are walked in 64 KiB tiles that END at the block end (the
// first tile is padded with virtual leading zero bytes, which do not change a zero-initialised CRC register):
// coalesced 16-byte loads -> LDS -> every lane runs slicing-by-8 over its own 64 contiguous bytes.  A lane keeps
// one running register across tiles (Horner step: R = R * x^(8*65536) + r, the multiplication by table), so the
// GF(2) combination of the 1024 lanes happens once per block: D = sum_t R_t * x^(8*64*(1023-t)), and
// crc = ~(0xffffffff * x^(8*total) + D).
__global__ __launch_bounds__(CRC_NT) void bzx_rl_crc_kernel(const uint8_t *__restrict__ raw, BzxSplitWs ws,
                                                            BzxBlock *__restrict__ blk, uint32_t own_first,
                                                            uint32_t own_step)
{
    __shared__ uint32_t tab[8][256];       // slicing-by-8: register after byte v followed by k zero bytes
    __shared__ uint32_t tabx[4][256];      // (v << 8k) * x^(8*65536) mod P
    __shared__ __attribute__((aligned(16))) uint8_t buf[CRC_NT * CRC_PITCH];
    __shared__ uint32_t red[CRC_NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 256) {
        uint32_t c = tid << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
        tab[0][tid] = c;
    }
    __syncthreads();
    for (int k = 1; k < 8; k++) {
        if (tid < 256) {
            const uint32_t p = tab[k - 1][tid];
            tab[k][tid] = (p << 8) ^ tab[0][p >> 24];
        }
        __syncthreads();
    }
    {
        const uint32_t xt = gf_xpow8(CRC_TILE);
        tabx[tid >> 8][tid & 255u] = gf_mulmod((tid & 255u) << (8 * (tid >> 8)), xt);
    }
    const uint32_t my_weight = gf_xpow8((uint64_t)CRC_SUB * (CRC_NT - 1 - tid));     // x^(8*64*(1023-t))
    __syncthreads();
    const uint32_t nblk = ws.nblk[0];
    for (uint32_t b = own_first + blockIdx.x * own_step; b < nblk; b += gridDim.x * own_step) {
        const uint64_t lo = ws.blk_raw[b], hi = ws.blk_raw[b + 1];
        const uint64_t total = hi - lo;
        const uint64_t ntile = (total + CRC_TILE - 1) / CRC_TILE;
        uint32_t R = 0;
        // piece e of lane t in a tile = bytes [(e*1024 + t)*16, +16) of the tile; tile k starts at hi - (ntile-k)*64K
        uint4 nx[4];
        auto load_piece = [&](int64_t p) -> uint4 {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p >= (int64_t)lo) {
                __builtin_memcpy(&v, raw + p, 16);
            } else if (p + 16 > (int64_t)lo) {
                uint8_t tmp[16];
                for (int j = 0; j < 16; j++) tmp[j] = (p + j >= (int64_t)lo) ? raw[p + j] : (uint8_t)0;
                __builtin_memcpy(&v, tmp, 16);
            }
            return v;
        };
        int64_t t0 = (int64_t)hi - (int64_t)(ntile * CRC_TILE);
#pragma unroll
        for (int e = 0; e < 4; e++) nx[e] = ntile ? load_piece(t0 + (int64_t)(e * CRC_NT + tid) * 16) : make_uint4(0, 0, 0, 0);
        for (uint64_t k = 0; k < ntile; k++) {
            uint4 cur[4];
#pragma unroll
            for (int e = 0; e < 4; e++) cur[e] = nx[e];
            t0 += CRC_TILE;
            if (k + 1 < ntile) {
#pragma unroll
                for (int e = 0; e < 4; e++) nx[e] = load_piece(t0 + (int64_t)(e * CRC_NT + tid) * 16);
            }
            __syncthreads();                       // previous tile fully consumed
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t o = (uint32_t)(e * CRC_NT + tid) * 16;              // byte offset inside the tile
                uint64_t *dst = reinterpret_cast<uint64_t *>(buf + (o / CRC_SUB) * CRC_PITCH + (o % CRC_SUB));
                dst[0] = (uint64_t)cur[e].x | ((uint64_t)cur[e].y << 32);
                dst[1] = (uint64_t)cur[e].z | ((uint64_t)cur[e].w << 32);
            }
            __syncthreads();
            const uint64_t *src = reinterpret_cast<const uint64_t *>(buf + tid * CRC_PITCH);
            uint32_t r = 0;
#pragma unroll
            for (int q = 0; q < CRC_SUB / 8; q++) {
                const uint64_t w = src[q];
                const uint32_t w0 = __builtin_bswap32((uint32_t)w) ^ r, w1 = __builtin_bswap32((uint32_t)(w >> 32));
                r = tab[7][w0 >> 24] ^ tab[6][(w0 >> 16) & 255u] ^ tab[5][(w0 >> 8) & 255u] ^ tab[4][w0 & 255u] ^
                    tab[3][w1 >> 24] ^ tab[2][(w1 >> 16) & 255u] ^ tab[1][(w1 >> 8) & 255u] ^ tab[0][w1 & 255u];
            }
            R = tabx[3][R >> 24] ^ tabx[2][(R >> 16) & 255u] ^ tabx[1][(R >> 8) & 255u] ^ tabx[0][R & 255u] ^ r;
        }
        // D = xor over lanes of R_t * weight_t
        uint32_t d = gf_mulmod(R, my_weight);
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) d ^= __shfl_xor(d, s2);
        __syncthreads();
        if (lane == 0) red[wave] = d;
        __syncthreads();
        if (tid == 0) {
            uint32_t D = 0;
            for (uint32_t i = 0; i < CRC_NT / 64; i++) D ^= red[i];
            blk[b].crc = ~(gf_mulmod(0xffffffffu, gf_xpow8(total)) ^ D);
        }
    }
}

// ---- host orchestration
struct bzx_ctx;
int bzx_ctx_split_scratch(bzx_ctx *ctx, size_t bytes, void **p);   // bzx_api.hip
hipStream_t bzx_ctx_stream(bzx_ctx *ctx);
int bzx_ctx_ncu(bzx_ctx *ctx);

// Scratch of the splitter: the three tile arrays (in the context's scratch, or -- sharded analysis -- in the caller's
// array `tiles` of 3 x tile_stride words, which the ranks all-gather), the block arrays and the scans' segment totals.
static int split_ws(bzx_ctx *ctx, size_t len, uint32_t max_blocks, uint64_t *tiles, size_t tile_stride, BzxSplitWs *ws_out,
                    uint64_t **segtot_out)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
    const uint64_t nsegw = ntiles / SCAN_SEG + 4;                         // scratch of the two-level scans
    const size_t own_tiles = tiles ? 0 : 3 * (ntiles + 2);
    const size_t bytes = (own_tiles + 3 * ((size_t)max_blocks + 2) + nsegw + 8) * sizeof(uint64_t) + 64;
    void *p = nullptr;
    int rc = bzx_ctx_split_scratch(ctx, bytes, &p);
    if (rc) return rc;
    BzxSplitWs ws;
    uint64_t *q = (uint64_t *)p;
    if (tiles) {
        ws.tile_rs = tiles;
        ws.tile_off = tiles + tile_stride;
        ws.tile_np = tiles + 2 * tile_stride;
    } else {
        ws.tile_rs = q;
        ws.tile_off = ws.tile_rs + (ntiles + 2);
        ws.tile_np = ws.tile_off + (ntiles + 2);
        q = ws.tile_np + (ntiles + 2);
    }
    ws.blk_raw = q;
    ws.blk_f = ws.blk_raw + (max_blocks + 2);
    ws.blk_plain = (uint32_t *)(ws.blk_f + (max_blocks + 2));
    ws.nblk = (uint32_t *)((uint64_t *)ws.blk_plain + (max_blocks + 2));
    *segtot_out = (uint64_t *)ws.nblk + 8;
    ws.max_blocks = max_blocks;
    *ws_out = ws;
    return 0;
}

static uint32_t tile_grid(bzx_ctx *ctx, uint64_t ntiles)
{
    const uint64_t g = (uint64_t)bzx_ctx_ncu(ctx) * 8;
    return (uint32_t)(ntiles < g ? (ntiles ? ntiles : 1) : g);
}

// Launches A..C; writes the block count to ws.nblk (device).  max_blocks bounds the descriptor arrays.
int bzx_split_launch_boundaries(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, int level, uint32_t max_blocks,
                                BzxSplitWs *ws_out)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
    BzxSplitWs ws;
    uint64_t *segtot = nullptr;
    int rc = split_ws(ctx, len, max_blocks, nullptr, 0, &ws, &segtot);
    if (rc) return rc;
    hipStream_t st = bzx_ctx_stream(ctx);
    const uint32_t grid = tile_grid(ctx, ntiles);
    const uint32_t nmax = 100000u * (uint32_t)level - 19u;
    hipLaunchKernelGGL(bzx_rl_runstart_kernel, dim3(grid), dim3(RL_NT), 0, st, d_raw, (uint64_t)len, (uint64_t)0, ntiles, ws);
    launch_scan(st, ws.tile_rs, ntiles, 1, segtot);
    hipLaunchKernelGGL(bzx_rl_count_kernel, dim3(grid), dim3(RL_NT), 0, st, d_raw, (uint64_t)len, (uint64_t)0, ntiles, ws);
    launch_scan(st, ws.tile_off, ntiles, 0, segtot);
    launch_scan(st, ws.tile_np, ntiles, 0, segtot);
    hipLaunchKernelGGL(bzx_rl_boundaries_kernel, dim3(1), dim3(RL_NT), 0, st, d_raw, (uint64_t)len, ntiles, nmax, ws);
    *ws_out = ws;
    return 0;
}

// ---- the same analysis with the per-byte scans (kernels A and B) on one rank's share of the tiles.
// Tile arrays: the caller's `tiles`, 3 arrays of `stride` = per_rank * world words; rank r owns the entries
// [r * per_rank, (r + 1) * per_rank) of each and all-gathers them between the steps (the library has no collective).
uint64_t bzx_split_tiles_per_rank(size_t len, uint32_t world)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
    return (ntiles + 2 + world - 1) / world;
}

// step 1: kernel A on this rank's tiles (all three arrays get their provisional entries)
int bzx_split_shard_runs(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, uint32_t rank, uint32_t world, uint64_t *tiles)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE, per = bzx_split_tiles_per_rank(len, world);
    const uint64_t lo = (uint64_t)rank * per < ntiles ? (uint64_t)rank * per : ntiles;
    const uint64_t hi = lo + per < ntiles ? lo + per : ntiles;
    BzxSplitWs ws;
    uint64_t *segtot = nullptr;
    int rc = split_ws(ctx, len, 2, tiles, per * world, &ws, &segtot);
    if (rc) return rc;
    if (hi > lo)
        hipLaunchKernelGGL(bzx_rl_runstart_kernel, dim3(tile_grid(ctx, hi - lo)), dim3(RL_NT), 0, bzx_ctx_stream(ctx), d_raw,
                           (uint64_t)len, lo, hi, ws);
    return 0;
}

// step 2 (array 0 gathered): carry-in scan over ALL tiles (every rank the same, 8 B per 8 KiB of input), kernel B on
// this rank's tiles
int bzx_split_shard_counts(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, uint32_t rank, uint32_t world, uint64_t *tiles)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE, per = bzx_split_tiles_per_rank(len, world);
    const uint64_t lo = (uint64_t)rank * per < ntiles ? (uint64_t)rank * per : ntiles;
    const uint64_t hi = lo + per < ntiles ? lo + per : ntiles;
    BzxSplitWs ws;
    uint64_t *segtot = nullptr;
    int rc = split_ws(ctx, len, 2, tiles, per * world, &ws, &segtot);
    if (rc) return rc;
    hipStream_t st = bzx_ctx_stream(ctx);
    launch_scan(st, ws.tile_rs, ntiles, 1, segtot);
    if (hi > lo)
        hipLaunchKernelGGL(bzx_rl_count_kernel, dim3(tile_grid(ctx, (hi - lo + RL_NT - 1) / RL_NT)), dim3(RL_NT), 0, st, d_raw,
                           (uint64_t)len, lo, hi, ws);
    return 0;
}

// step 3 (arrays 1 and 2 gathered): offsets and the serial chain of block boundaries, on every rank
int bzx_split_shard_boundaries(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, int level, uint32_t max_blocks, uint32_t world,
                               uint64_t *tiles, BzxSplitWs *ws_out)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE, per = bzx_split_tiles_per_rank(len, world);
    BzxSplitWs ws;
    uint64_t *segtot = nullptr;
    int rc = split_ws(ctx, len, max_blocks, tiles, per * world, &ws, &segtot);
    if (rc) return rc;
    hipStream_t st = bzx_ctx_stream(ctx);
    const uint32_t nmax = 100000u * (uint32_t)level - 19u;
    launch_scan(st, ws.tile_off, ntiles, 0, segtot);
    launch_scan(st, ws.tile_np, ntiles, 0, segtot);
    hipLaunchKernelGGL(bzx_rl_boundaries_kernel, dim3(1), dim3(RL_NT), 0, st, d_raw, (uint64_t)len, ntiles, nmax, ws);
    *ws_out = ws;
    return 0;
}

// Launches D and E for nblk blocks (ws.nblk on the device already holds nblk).
void bzx_split_launch_scatter(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, const BzxSplitWs &ws, uint32_t nblk,
                              uint8_t *d_slabs, BzxBlock *d_blk, uint32_t own_first, uint32_t own_step)
{
    const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
    hipStream_t st = bzx_ctx_stream(ctx);
    const uint32_t grid = (uint32_t)(ntiles < (uint64_t)bzx_ctx_ncu(ctx) * 8 ? ntiles : (uint64_t)bzx_ctx_ncu(ctx) * 8);
    hipLaunchKernelGGL(bzx_rl_scatter_kernel, dim3(grid), dim3(RL_NT), 0, st, d_raw, (uint64_t)len, ntiles, ws, d_slabs, d_blk, own_first, own_step);
    const uint32_t mine = nblk > own_first ? (nblk - own_first + own_step - 1) / own_step : 0;
    const uint32_t cgrid = mine < (uint32_t)bzx_ctx_ncu(ctx) ? (mine ? mine : 1) : (uint32_t)bzx_ctx_ncu(ctx);
    hipLaunchKernelGGL(bzx_rl_crc_kernel, dim3(cgrid), dim3(CRC_NT), 0, st, d_raw, ws, d_blk, own_first, own_step);
}

uint32_t *bzx_split_nblk_ptr(const BzxSplitWs &ws) { return ws.nblk; }

// CRC-32/BZIP2 of nblk consecutive byte ranges [bounds[b], bounds[b+1]) of d_raw into blk[b].crc (the decompressor
// checks its output with the compressor's kernel).  d_nblk: device word holding nblk.
void bzx_launch_block_crcs(bzx_ctx *ctx, const uint8_t *d_raw, const uint64_t *d_bounds, uint32_t *d_nblk, BzxBlock *d_blk,
                           uint32_t nblk)
{
    BzxSplitWs ws;
    memset(&ws, 0, sizeof(ws));
    ws.blk_raw = const_cast<uint64_t *>(d_bounds);
    ws.nblk = d_nblk;
    const uint32_t cgrid = nblk < (uint32_t)bzx_ctx_ncu(ctx) ? (nblk ? nblk : 1) : (uint32_t)bzx_ctx_ncu(ctx);
    hipLaunchKernelGGL(bzx_rl_crc_kernel, dim3(cgrid), dim3(CRC_NT), 0, bzx_ctx_stream(ctx), d_raw, ws, d_blk, 0u, 1u);
}
