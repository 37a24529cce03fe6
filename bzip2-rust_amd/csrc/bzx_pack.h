// bzx_pack.h -- the packed block P shared by the two suffix sorters (bzx_bsort.hip, bzx_bwt.hip).
//
// Alphabet packing: the block's bytes are mapped to dense, order-preserving ids of `bits` bits each and written
// once, most significant bit first, as one bit string P (symbol i at bit i*bits), continued cyclically for PK_PAD
// symbols past the end.  Fixed-width ids keep integer order == lexicographic order at ANY bit offset, so every key
// a sorter needs is ONE unaligned 8-byte read of P and a shift (reference contract: bwt_sort.rs:45-57 compares
// rotations byte by byte; comparing their packed bit strings is the same order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PK_PAD 80u            // symbols of cyclic continuation: a 64-bit window at the last symbol stays inside P

// >= 57 valid bits of the packed text starting at bit `b`, most significant first
__device__ __forceinline__ uint64_t pk_window_bit(const uint8_t *__restrict__ P, uint32_t b)
{
    uint64_t w;
    __builtin_memcpy(&w, P + (b >> 3), 8);      // unaligned 8-byte global load
    return __builtin_bswap64(w) << (b & 7u);
}

// >= 57 valid bits of rotation pos, most significant first
__device__ __forceinline__ uint64_t pk_window(const uint8_t *__restrict__ P, uint32_t pos, uint32_t bits)
{
    return pk_window_bit(P, pos * bits);
}

// Builds P from the block: one lane per 64 symbols = `bits` whole 8-byte words (big-endian bit order).
// seq: byte value -> dense symbol id (LDS).  NT = workgroup size.  Ends with a workgroup barrier.
template <int NT>
__device__ __forceinline__ void pk_build_t(const uint8_t *__restrict__ T, uint32_t n, uint32_t bits,
                                           uint8_t *__restrict__ P, const uint8_t *seq)
{
    const uint32_t ngroups = (n + PK_PAD + 63u) / 64u + 1u;       // one group of zeros behind the continuation
    for (uint32_t q = threadIdx.x; q < ngroups; q += NT) {
        const uint32_t pos = 64u * q;
        uint32_t c[16];                                            // 64 block bytes, cyclic
        if (pos + 64u <= n) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint4 v;
                __builtin_memcpy(&v, T + pos + 16 * k, 16);
                c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) c[k] = 0;
            if (pos < n + PK_PAD) {
                uint32_t pm = pos % n;
                for (int j = 0; j < 64; j++) {
                    c[j >> 2] |= (uint32_t)T[pm] << (8 * (j & 3));
                    pm = pm + 1 == n ? 0u : pm + 1;
                }
            }
        }
        uint64_t *out = reinterpret_cast<uint64_t *>(P) + (size_t)q * bits;      // P is 8-byte aligned
        uint64_t acc = 0;
        uint32_t have = 0;                                         // bits in acc
        const bool real = pos < n + PK_PAD;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const uint64_t id = real ? seq[(c[j >> 2] >> (8 * (j & 3))) & 255u] : 0u;
            if (have + bits <= 64u) {
                acc = bits == 64u ? id : (acc << bits) | id;
                have += bits;
            } else {                                               // the symbol straddles a word boundary
                const uint32_t hi = 64u - have, lo = bits - hi;
                *out++ = __builtin_bswap64((acc << hi) | (id >> lo));
                acc = id & ((1ull << lo) - 1ull);
                have = lo;
            }
            if (have == 64u) {
                *out++ = __builtin_bswap64(acc);
                acc = 0;
                have = 0;
            }
        }
    }
    __syncthreads();
}
