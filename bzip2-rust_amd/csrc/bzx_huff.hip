// bzx_huff.hip -- multi-table Huffman optimisation of one block's symbol stream on gfx950.
//
// Contract (reference src/huffman_coding/huffman.rs:79-374, huf_encode up to the point where
// bits are written): number of tables from nMTF (huffman.rs:87-93), initial tables from the
// symbol histogram (huffman.rs:472-532 -- libbz2's partition rule), four passes
// of { cost of every 50-symbol group under each table, first-minimum table, per-table symbol
// frequencies, new code lengths } (huffman.rs:114-200), selector MTF (huffman.rs:237-292),
// canonical codes (huffman.rs:361-374).  Code lengths come from a literal restatement of
// libbz2's heap construction (replaces huffman_code_from_weights.rs:17-84),
// run by one lane per table, because its tie-breaking decides output bits.
//
// Parallel shape: one workgroup per block; one lane per 50-symbol group (<= 18002 groups);
// six 10-bit code-length fields packed in two LDS words per symbol so a group cost is 50 x 2
// LDS reads; rfreq in LDS atomics.  Also produces every section size so the emitter and the
// stream layout know all bit offsets.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define HUF_NT 512
#define HUF_NW (HUF_NT / 64)

__shared__ uint32_t h_freq[BZX_MAX_ALPHA + 2];
__shared__ uint8_t h_len[6][BZX_MAX_ALPHA + 2];
// per symbol, ONE 8-byte word: low half len0 | len1<<10 | len2<<20, high half len3 | len4<<10 | len5<<20
__shared__ uint2 h_lenAB[BZX_MAX_ALPHA + 2];
__shared__ uint32_t h_rfreq[6][BZX_MAX_ALPHA + 2];
__shared__ __attribute__((aligned(16))) uint64_t h_heap[6][BZX_MAX_ALPHA + 2];     // [weight:32 | node:32]
__shared__ int32_t h_weight[6][BZX_MAX_ALPHA + 2];                                  // leaves
__shared__ int32_t h_parent[6][BZX_MAX_ALPHA * 2];
__shared__ uint32_t h_code[6][BZX_MAX_ALPHA + 2];
__shared__ uint32_t h_part[6][2];   // initial partition [gs, ge] per table
__shared__ uint32_t h_scratch[2 * HUF_NW];
__shared__ uint32_t h_bcast[4];
__shared__ uint32_t h_acc[4];       // [0] selector bits, [1] table bits, [2] payload bits

// libbz2 hbMakeCodeLengths for table t, by ONE WAVE: the heap (whose order decides ties) is built and
// emptied by lane 0 exactly as libbz2 does it; what has no order -- the initial weights, the depth of every leaf
// (a walk up the parent links), the halving of the weights -- is spread over the 64 lanes.
// The heap walk is one chain of dependent LDS round trips, so it is laid out to need as few as possible: a heap entry
// carries its node's weight ([weight:32 | node:32]; libbz2 looks the weight up through the node), and the two children
// of a node are one aligned 16-byte read.  Weights of inner nodes exist only inside their heap entries.  (At 258
// symbols the walk took 0.8 ms per table and pass with separate heap and weight arrays: three quarters of the stage.)
__device__ void make_code_lengths(int t, int32_t alpha, int32_t max_len)
{
    uint64_t *heap = h_heap[t];
    int32_t *weight = h_weight[t], *parent = h_parent[t];
    const int32_t lane = (int32_t)bzx_lane();
    for (int32_t i = lane; i < alpha; i += 64) {
        const uint32_t f = h_rfreq[t][i];
        weight[i + 1] = (int32_t)((f == 0 ? 1u : f) << 8);
    }
    for (;;) {
        bzx_wave_sync();
        if (lane == 0) {
            int32_t n_nodes = alpha, n_heap = 0;
            heap[0] = 0;                                   // (node 0, weight 0: the sentinel above the root)
            parent[0] = -2;
            auto up = [&](uint64_t e) {                    // libbz2 UPHEAP of the entry placed at n_heap
                int32_t zz = n_heap;
                const uint32_t wt = (uint32_t)(e >> 32);
                for (;;) {
                    const uint64_t pe = heap[zz >> 1];
                    if (!(wt < (uint32_t)(pe >> 32))) break;
                    heap[zz] = pe;
                    zz >>= 1;
                }
                heap[zz] = e;
            };
            for (int32_t i = 1; i <= alpha; i++) {
                parent[i] = -1;
                n_heap++;
                up(((uint64_t)(uint32_t)weight[i] << 32) | (uint32_t)i);
            }
            while (n_heap > 1) {
                uint64_t e12[2];
                for (int rep = 0; rep < 2; rep++) {        // twice: take the root, DOWNHEAP the last entry from there
                    e12[rep] = heap[1];
                    const uint64_t e = heap[n_heap];
                    n_heap--;
                    int32_t zz = 1;
                    const uint32_t wt = (uint32_t)(e >> 32);
                    for (;;) {
                        int32_t yy = zz << 1;
                        if (yy > n_heap) break;
                        const uint4 two = *reinterpret_cast<const uint4 *>(heap + yy);     // children yy, yy + 1 (yy is even)
                        uint32_t cw = two.y, cn = two.x;
                        if (yy < n_heap && two.w < two.y) {
                            yy++;
                            cw = two.w;
                            cn = two.z;
                        }
                        if (wt < cw) break;
                        heap[zz] = ((uint64_t)cw << 32) | cn;
                        zz = yy;
                    }
                    heap[zz] = e;
                }
                n_nodes++;
                parent[(uint32_t)e12[0]] = parent[(uint32_t)e12[1]] = n_nodes;
                const uint32_t w1 = (uint32_t)(e12[0] >> 32), w2 = (uint32_t)(e12[1] >> 32);
                const uint32_t d1 = w1 & 0xffu, d2 = w2 & 0xffu;
                const uint32_t wn = ((w1 & 0xffffff00u) + (w2 & 0xffffff00u)) | (1u + (d1 > d2 ? d1 : d2));
                parent[n_nodes] = -1;
                n_heap++;
                up(((uint64_t)wn << 32) | (uint32_t)n_nodes);
            }
        }
        bzx_wave_sync();
        bool too_long = false;
        for (int32_t i = 1 + lane; i <= alpha; i += 64) {
            int32_t j = 0, k = i;
            while (parent[k] >= 0) {
                k = parent[k];
                j++;
            }
            h_len[t][i - 1] = (uint8_t)j;
            if (j > max_len) too_long = true;
        }
        if (!__ballot(too_long)) break;
        for (int32_t i = 1 + lane; i <= alpha; i += 64) {
            int32_t j = weight[i] >> 8;
            j = 1 + (j / 2);
            weight[i] = j << 8;
        }
    }
}

#define HUF_STAMP(slot)                                                       \
    do {                                                                      \
        if (B.dbg && tid == 0) {                                              \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&B.dbg[slot], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)

#ifndef HUF_WAVES_PER_SIMD
#define HUF_WAVES_PER_SIMD 6            // three workgroups per compute unit (48 KB of LDS each)
#endif
__global__ __launch_bounds__(HUF_NT) __attribute__((amdgpu_waves_per_eu(HUF_WAVES_PER_SIMD, HUF_WAVES_PER_SIMD))) void bzx_huff_kernel(BzxBatch B)
{
    const uint32_t tid = threadIdx.x, lane = bzx_lane(), wave = bzx_wave();
    unsigned long long t_last = 0;

    for (;;) {
        if (tid == 0) h_bcast[0] = atomicAdd(&B.counters[2], 1u);
        __syncthreads();
        const uint32_t j_ = h_bcast[0];
        __syncthreads();
        if (j_ >= B.nblk) break;
        const uint32_t b = B.blk_first + j_ * B.blk_step;

        if (B.dbg && tid == 0) t_last = wall_clock64();
        const uint32_t n_mtf = B.blk[b].n_mtf;
        const uint32_t alpha = B.blk[b].n_in_use + 2;
        const uint16_t *__restrict__ V = B.mtfv + BZX_SLAB(B, b) * BZX_BLK_STRIDE;
        uint8_t *__restrict__ SEL = B.selector + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        uint8_t *__restrict__ SELM = B.selector_mtf + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        uint16_t *__restrict__ GB = B.gbits + BZX_SLAB(B, b) * BZX_SEL_STRIDE;
        const uint32_t n_sel = (n_mtf + BZX_G_SIZE - 1) / BZX_G_SIZE;
        const uint32_t n_groups = n_mtf < 200 ? 2u : n_mtf < 600 ? 3u : n_mtf < 1200 ? 4u : n_mtf < 2400 ? 5u : 6u;

        for (uint32_t i = tid; i < BZX_MAX_ALPHA + 2; i += HUF_NT) h_freq[i] = i < alpha ? B.freq[BZX_SLAB(B, b) * 260 + i] : 0u;
        if (tid < 4) h_acc[tid] = 0;
        __syncthreads();

        // ---- initial tables (libbz2 partition rule)
        if (tid == 0) {
            int32_t n_part = (int32_t)n_groups, rem_f = (int32_t)n_mtf, gs = 0;
            while (n_part > 0) {
                const int32_t t_freq = rem_f / n_part;
                int32_t ge = gs - 1, a_freq = 0;
                while (a_freq < t_freq && ge < (int32_t)alpha - 1) {
                    ge++;
                    a_freq += (int32_t)h_freq[ge];
                }
                if (ge > gs && n_part != (int32_t)n_groups && n_part != 1 && (((int32_t)n_groups - n_part) % 2 == 1)) {
                    a_freq -= (int32_t)h_freq[ge];
                    ge--;
                }
                h_part[n_part - 1][0] = (uint32_t)gs;
                h_part[n_part - 1][1] = (uint32_t)ge;   // may be gs-1 (empty); stored as int bits
                n_part--;
                gs = ge + 1;
                rem_f -= a_freq;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < 6 * (BZX_MAX_ALPHA + 2); i += HUF_NT) {
            const uint32_t t = i / (BZX_MAX_ALPHA + 2), v = i % (BZX_MAX_ALPHA + 2);
            uint8_t l = 15;
            if (t < n_groups && (int32_t)v >= (int32_t)h_part[t][0] && (int32_t)v <= (int32_t)h_part[t][1]) l = 0;
            h_len[t][v] = l;
        }
        __syncthreads();

        HUF_STAMP(40);
        // ---- four refinement passes
        for (int iter = 0; iter < BZX_N_ITERS; iter++) {
            for (uint32_t i = tid; i < 6 * (BZX_MAX_ALPHA + 2); i += HUF_NT) (&h_rfreq[0][0])[i] = 0;
            for (uint32_t v = tid; v < BZX_MAX_ALPHA + 2; v += HUF_NT) {
                h_lenAB[v] = make_uint2((uint32_t)h_len[0][v] | ((uint32_t)h_len[1][v] << 10) | ((uint32_t)h_len[2][v] << 20),
                                        (uint32_t)h_len[3][v] | ((uint32_t)h_len[4][v] << 10) | ((uint32_t)h_len[5][v] << 20));
            }
            __syncthreads();
            for (uint32_t g = tid; g < n_sel; g += HUF_NT) {
                const uint32_t gs = g * BZX_G_SIZE;
                const uint32_t cnt = (n_mtf - gs < BZX_G_SIZE) ? n_mtf - gs : BZX_G_SIZE;
                const uint32_t *__restrict__ vp = reinterpret_cast<const uint32_t *>(V + gs);
                uint32_t sy[BZX_G_SIZE / 2];
#pragma unroll
                for (int k = 0; k < BZX_G_SIZE / 2; k++) sy[k] = vp[k];
                uint32_t accA = 0, accB = 0;
#pragma unroll
                for (int k = 0; k < BZX_G_SIZE; k++) {
                    if ((uint32_t)k < cnt) {
                        const uint32_t s = (sy[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                        const uint2 l2 = h_lenAB[s];
                        accA += l2.x;
                        accB += l2.y;
                    }
                }
                const uint32_t cost[6] = {accA & 1023u, (accA >> 10) & 1023u, (accA >> 20) & 1023u,
                                          accB & 1023u, (accB >> 10) & 1023u, (accB >> 20) & 1023u};
                uint32_t bc = 999999999u, bt = 0;
#pragma unroll
                for (uint32_t t = 0; t < 6; t++)
                    if (t < n_groups && cost[t] < bc) {
                        bc = cost[t];
                        bt = t;
                    }
                SEL[g] = (uint8_t)bt;
#pragma unroll
                for (int k = 0; k < BZX_G_SIZE; k++) {
                    if ((uint32_t)k < cnt) {
                        const uint32_t s = (sy[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                        atomicAdd(&h_rfreq[bt][s], 1u);
                    }
                }
            }
            __syncthreads();
            HUF_STAMP(41);
            if (wave < n_groups) make_code_lengths((int)wave, (int32_t)alpha, 17);      // (a wave per table)
            __syncthreads();
            HUF_STAMP(42);
        }

        // ---- canonical codes (huffman.rs:361-374), one wave per table: for every length in turn, the symbols of that
        // length take consecutive codes in symbol order (ballot + count of the lanes below)
        if (wave < n_groups) {
            const int t = (int)wave;
            uint32_t mx = 0, mn_inv = 0, tb = 0;
            for (uint32_t i = lane; i < alpha; i += 64) {
                const uint32_t l = h_len[t][i], prev = h_len[t][i ? i - 1 : 0];
                mx = l > mx ? l : mx;
                mn_inv = 32u - l > mn_inv ? 32u - l : mn_inv;
                tb += 2u * (l > prev ? l - prev : prev - l) + 1u;      // coding-table section: 5 + sum(2|delta| + 1)
            }
            const uint32_t max_len = (uint32_t)__shfl((int)bzx_wave_incl_max(mx), 63);
            const uint32_t min_len = 32u - (uint32_t)__shfl((int)bzx_wave_incl_max(mn_inv), 63);
            tb = bzx_wave_incl_sum(tb);
            if (lane == 63) atomicAdd(&h_acc[1], 5u + tb);
            uint32_t vec = 0;
            for (uint32_t nn = min_len; nn <= max_len; nn++) {
                for (uint32_t i0 = 0; i0 < alpha; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool hit = i < alpha && h_len[t][i] == nn;
                    const uint64_t m = __ballot(hit);
                    if (hit) h_code[t][i] = vec + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    vec += (uint32_t)__popcll(m);
                }
                vec <<= 1;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < 6 * 260; i += HUF_NT) {
            const uint32_t t = i / 260, v = i % 260;
            const bool live = t < n_groups && v < alpha;
            B.len[BZX_SLAB(B, b) * 6 * 260 + i] = live ? h_len[t][v] : (uint8_t)0;
            B.code[BZX_SLAB(B, b) * 6 * 260 + i] = live ? h_code[t][v] : 0u;
        }
        for (uint32_t v = tid; v < BZX_MAX_ALPHA + 2; v += HUF_NT) {
            h_lenAB[v] = make_uint2((uint32_t)h_len[0][v] | ((uint32_t)h_len[1][v] << 10) | ((uint32_t)h_len[2][v] << 20),
                                    (uint32_t)h_len[3][v] | ((uint32_t)h_len[4][v] << 10) | ((uint32_t)h_len[5][v] << 20));
        }
        __syncthreads();

        HUF_STAMP(43);
        // ---- payload size per group under its final selector; total payload bits
        {
            uint32_t my_bits = 0;
            for (uint32_t g = tid; g < n_sel; g += HUF_NT) {
                const uint32_t gs = g * BZX_G_SIZE;
                const uint32_t cnt = (n_mtf - gs < BZX_G_SIZE) ? n_mtf - gs : BZX_G_SIZE;
                const uint32_t *__restrict__ vp = reinterpret_cast<const uint32_t *>(V + gs);
                const uint32_t bt = SEL[g];
                uint32_t sy[BZX_G_SIZE / 2];
#pragma unroll
                for (int k = 0; k < BZX_G_SIZE / 2; k++) sy[k] = vp[k];
                uint32_t accA = 0, accB = 0;
#pragma unroll
                for (int k = 0; k < BZX_G_SIZE; k++) {
                    if ((uint32_t)k < cnt) {
                        const uint32_t sm = (sy[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                        const uint2 l2 = h_lenAB[sm];
                        accA += l2.x;
                        accB += l2.y;
                    }
                }
                const uint32_t pick = bt < 3 ? accA : accB;
                const uint32_t acc = (pick >> (10 * (bt % 3))) & 1023u;
                GB[g] = (uint16_t)acc;
                my_bits += acc;
            }
            uint32_t tot;
            (void)bzx_block_excl_sum<HUF_NT>(my_bits, h_scratch, tot);
            if (tid == 0) h_acc[2] = tot;
        }

        HUF_STAMP(44);
        // ---- selector MTF (huffman.rs:237-292): chunk per lane, start list rebuilt from earlier chunks
        {
            const uint32_t per = (n_sel + HUF_NT - 1) / HUF_NT;
            const uint32_t lo = tid * per;
            const uint32_t hi = lo + per < n_sel ? lo + per : n_sel;
            // last use (1 + selector index) of every table inside my chunk, then an exclusive max-scan over the
            // lanes: lp[t] = last use of table t before my chunk (0 = never)
            uint32_t lp[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t sg = SEL[i];
#pragma unroll
                for (uint32_t t = 0; t < 6; t++)
                    if (sg == t) lp[t] = i + 1;
            }
#pragma unroll
            for (uint32_t t = 0; t < 6; t++) {
                uint32_t d0, d1, ex, tt;
                bzx_block_scan_sum_max<HUF_NT>(0u, lp[t], h_scratch, d0, d1, ex, tt);
                lp[t] = ex;
            }
            // start list (nibble k = k-th entry): tables by most recent use before lo, then never-used ascending
            uint32_t st = 0, cnt = 0, taken = 0;
            for (uint32_t r = 0; r < n_groups; r++) {
                uint32_t best = 0, bt = 6;
#pragma unroll
                for (uint32_t t = 0; t < 6; t++)
                    if (t < n_groups && !((taken >> t) & 1u) && lp[t] > best) {
                        best = lp[t];
                        bt = t;
                    }
                if (bt == 6) break;
                taken |= 1u << bt;
                st |= bt << (4 * cnt);
                cnt++;
            }
            for (uint32_t sg = 0; sg < n_groups; sg++)
                if (!((taken >> sg) & 1u)) {
                    st |= sg << (4 * cnt);
                    cnt++;
                }
            st |= 0xFFFFFFFFu << (4 * n_groups);      // unused nibbles never match a table number
            uint32_t my_bits = 0;
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t ll = SEL[i];
                const uint32_t x = st ^ (ll * 0x11111111u);
                const uint32_t z = (x - 0x11111111u) & ~x & 0x88888888u;       // lowest flagged nibble = first match
                const uint32_t j = (uint32_t)(__ffs((int)z) - 1) >> 2;
                if (j) {
                    const uint32_t lowmask = (1u << (4 * j)) - 1u;
                    const uint32_t himask = ~((1u << (4 * (j + 1))) - 1u);
                    st = (st & himask) | ((st & lowmask) << 4) | ll;
                }
                SELM[i] = (uint8_t)j;
                my_bits += j + 1;
            }
            uint32_t tot;
            (void)bzx_block_excl_sum<HUF_NT>(my_bits, h_scratch, tot);
            if (tid == 0) h_acc[0] = tot;
        }
        __syncthreads();

        HUF_STAMP(45);
        if (tid == 0) {
            uint32_t map_words = 0;
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t any = 0;
                for (uint32_t j = 0; j < 16; j++) any |= B.in_use[BZX_SLAB(B, b) * 256 + i * 16 + j];
                map_words += any ? 1u : 0u;
            }
            const uint32_t map_bits = 16 + 16 * map_words;
            BzxBlock &d = B.blk[b];
            d.n_groups = n_groups;
            d.n_selectors = n_sel;
            d.sec_bits[0] = h_acc[0];
            d.sec_bits[1] = h_acc[1];
            d.sec_bits[2] = h_acc[2];
            d.sec_bits[3] = map_bits;
            // 48 magic + 32 crc + 1 randomised + 24 origPtr = 105 ; 3 nGroups + 15 nSelectors
            d.bits = 105ull + map_bits + 3 + 15 + h_acc[0] + h_acc[1] + h_acc[2];
        }
        __syncthreads();
    }
}

void bzx_launch_huffman(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_huff_kernel, dim3(grid), dim3(HUF_NT), 0, stream, B);
}
