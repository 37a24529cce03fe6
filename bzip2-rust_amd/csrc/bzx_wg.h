// bzx_wg.h -- workgroup-level primitives (wave64) shared by the stage kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BZX_WAVE 64

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding
// global loads/stores (vmcnt), so prefetched loads and scatter stores stay in flight across it.
#ifdef BZX_HIP_EMU
#define bzx_lds_barrier() __syncthreads()
#else
#define bzx_lds_barrier() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Wave-level ordering point: on the device the lanes of a wave execute each instruction together and LDS /
// global accesses of one wave are issued in program order, so only the compiler must not reorder; the CPU
// emulator runs lanes one after the other and needs a real rendezvous here.
#ifdef BZX_HIP_EMU
#define bzx_wave_sync() hipemu::wave_sync()
#else
#define bzx_wave_sync()                                      \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)
#endif


// ---- the remaining device / emulator differences (the product sources have no other BZX_HIP_EMU sites) ----------
// lds_order(): ordering point between LDS operations of ONE wave.  The hardware executes a wave's LDS instructions in
// issue order and all its lanes together, so only the compiler has to keep them in program order (no s_waitcnt is
// needed: a fence would drain the LDS queue, and a volatile access through a generic pointer becomes a FLAT load with
// a full wait).  The CPU emulator runs lanes one after the other and needs a real rendezvous.
// bzx_uni(): a value every lane of the workgroup loaded from the same address, kept in a scalar register (values live
// across calls would otherwise be spilled around every call: only 24 of 80 VGPRs are callee-saved).
// bzx_bcast0(): lane 0's value to the whole wave (all lanes active).
// bzx_tid_here(): the lane id, opaque to the optimiser: re-read at the top of a loop body it keeps address arithmetic
// derived from it from being hoisted out of the loop, where dozens of such values would be spilled and reloaded from
// scratch (global memory) inside the hot loops.
// bzx_drain_stores(): all of this lane's global stores have landed (before the wave reads them back).
// bzx_lds_ticket(): one returning LDS add of 1 per valid lane.  On the device the lanes of a wave instruction that name
// the same counter are served in lane order (observed, not promised: the sort kernel checks what it builds on it); the
// emulator runs lanes in whatever order its scheduler resumes them, so it hands out the tickets by lane number itself.
#ifdef BZX_HIP_EMU
__device__ inline uint32_t bzx_lds_ticket(uint32_t *ctr, bool valid)
{
    static uintptr_t who[16][64];                   // (the emulator runs one workgroup at a time)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uintptr_t me = valid ? (uintptr_t)ctr : 0;
    who[wave][lane] = me;
    hipemu::wave_sync();
    uint32_t before = 0, same = 0;
    for (uint32_t l = 0; l < 64; l++) {
        if (valid && who[wave][l] == me) {
            same++;
            if (l < lane) before++;
        }
    }
    const uint32_t base = valid ? *ctr : 0u;
    hipemu::wave_sync();
    if (valid && before == 0) *ctr = base + same;
    hipemu::wave_sync();
    return base + before;
}
#else
__device__ __forceinline__ uint32_t bzx_lds_ticket(uint32_t *ctr, bool valid) { return valid ? atomicAdd(ctr, 1u) : 0u; }
#endif
#ifdef BZX_HIP_EMU
#define lds_order() hipemu::wave_sync()
__device__ __forceinline__ uint32_t bzx_uni(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t bzx_bcast0(uint32_t v) { return __shfl(v, 0); }
__device__ __forceinline__ uint32_t bzx_tid_here() { return threadIdx.x; }
#define bzx_drain_stores() do {} while (0)
#else
#define lds_order() asm volatile("" ::: "memory")
__device__ __forceinline__ uint32_t bzx_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t bzx_bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t bzx_tid_here()
{
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define bzx_drain_stores() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

__device__ __forceinline__ uint32_t bzx_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t bzx_wave() { return threadIdx.x >> 6; }

// Lanes of this wave holding the same nbits-wide digit (invalid lanes match nobody useful).
__device__ __forceinline__ uint64_t bzx_match_any(uint32_t d, int nbits, bool valid)
{
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// Wave inclusive scans and reductions.  On the device they are six DPP steps (row_shr 1,2,4,8 inside the rows of 16
// lanes, then row_bcast 15 / 31 across rows): plain vector instructions, no trip through the LDS crossbar as a
// __shfl_up (ds_bpermute_b32 + wait) per step would be -- these scans sit between two workgroup barriers of every
// radix pass, so their latency is everybody's.  The CPU emulator has no DPP and keeps the shuffle form.
#ifdef BZX_HIP_EMU
#define BZX_DPP_SCAN(v, OP, IDENT)                                           \
    do {                                                                     \
        const uint32_t lane_ = bzx_lane();                                   \
        for (uint32_t d_ = 1; d_ < 64; d_ <<= 1) {                           \
            const uint32_t y_ = __shfl_up(v, d_);                            \
            if (lane_ >= d_) v = OP(v, y_);                                  \
        }                                                                    \
    } while (0)
#else
#define BZX_DPP_STEP(v, OP, IDENT, CTRL, ROWMASK)                                                                 \
    do {                                                                                                         \
        const uint32_t y_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)(v), CTRL, ROWMASK, 0xf, false); \
        v = OP(v, y_);                                                                                           \
    } while (0)
#define BZX_DPP_SCAN(v, OP, IDENT)                                           \
    do {                                                                     \
        BZX_DPP_STEP(v, OP, IDENT, 0x111, 0xf);      /* row_shr:1 */         \
        BZX_DPP_STEP(v, OP, IDENT, 0x112, 0xf);      /* row_shr:2 */         \
        BZX_DPP_STEP(v, OP, IDENT, 0x114, 0xf);      /* row_shr:4 */         \
        BZX_DPP_STEP(v, OP, IDENT, 0x118, 0xf);      /* row_shr:8 */         \
        BZX_DPP_STEP(v, OP, IDENT, 0x142, 0xa);      /* row_bcast:15 */      \
        BZX_DPP_STEP(v, OP, IDENT, 0x143, 0xc);      /* row_bcast:31 */      \
    } while (0)
#endif
#define BZX_OP_ADD(a, b) ((a) + (b))
#define BZX_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define BZX_OP_OR(a, b) ((a) | (b))
#define BZX_OP_AND(a, b) ((a) & (b))

// Wave inclusive sum scan.
__device__ __forceinline__ uint32_t bzx_wave_incl_sum(uint32_t v)
{
    BZX_DPP_SCAN(v, BZX_OP_ADD, 0u);
    return v;
}

// Wave inclusive max scan.
__device__ __forceinline__ uint32_t bzx_wave_incl_max(uint32_t v)
{
    BZX_DPP_SCAN(v, BZX_OP_MAX, 0u);
    return v;
}

// OR / AND over the wave: the result is valid in lane 63 (inclusive scans).
__device__ __forceinline__ uint32_t bzx_wave_incl_or(uint32_t v)
{
    BZX_DPP_SCAN(v, BZX_OP_OR, 0u);
    return v;
}
__device__ __forceinline__ uint32_t bzx_wave_incl_and(uint32_t v)
{
    BZX_DPP_SCAN(v, BZX_OP_AND, 0xFFFFFFFFu);
    return v;
}

// Block exclusive sum scan over NT threads.  scratch: NT/64 words of LDS.  Two barriers.
template <int NT>
__device__ __forceinline__ uint32_t bzx_block_excl_sum(uint32_t v, uint32_t *scratch, uint32_t &total)
{
    const uint32_t incl = bzx_wave_incl_sum(v);
    if (bzx_lane() == 63) scratch[bzx_wave()] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    const uint32_t w = bzx_wave();
#pragma unroll
    for (uint32_t i = 0; i < NT / 64; i++) {
        const uint32_t t = scratch[i];
        if (i < w) pre += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return pre + incl - v;
}

// Block exclusive sum scan and exclusive max scan sharing the same two barriers.
// scratch: 2*NT/64 words.  max_v uses 0 as the identity.
template <int NT>
__device__ __forceinline__ void bzx_block_scan_sum_max(uint32_t sum_v, uint32_t max_v, uint32_t *scratch,
                                                       uint32_t &sum_excl, uint32_t &sum_total,
                                                       uint32_t &max_excl, uint32_t &max_total)
{
    const uint32_t lane = bzx_lane(), w = bzx_wave();
    const uint32_t si = bzx_wave_incl_sum(sum_v);
    const uint32_t mi = bzx_wave_incl_max(max_v);
    uint32_t m_prev = __shfl_up(mi, 1);       // exclusive max within the wave
    if (lane == 0) m_prev = 0;
    if (lane == 63) {
        scratch[w] = si;
        scratch[NT / 64 + w] = mi;
    }
    __syncthreads();
    uint32_t pre = 0, tot = 0, mpre = 0, mtot = 0;
#pragma unroll
    for (uint32_t i = 0; i < NT / 64; i++) {
        const uint32_t t = scratch[i], m = scratch[NT / 64 + i];
        if (i < w) {
            pre += t;
            if (m > mpre) mpre = m;
        }
        tot += t;
        if (m > mtot) mtot = m;
    }
    __syncthreads();
    sum_excl = pre + si - sum_v;
    sum_total = tot;
    max_excl = m_prev > mpre ? m_prev : mpre;
    max_total = mtot;
}

// Same, but the two barriers order LDS only (global loads/stores of the caller stay in flight).
// scratch: 2*NT/64 words.  max_v uses 0 as the identity.
template <int NT>
__device__ __forceinline__ void bzx_block_scan_sum_max_lds(uint32_t sum_v, uint32_t max_v, uint32_t *scratch,
                                                       uint32_t &sum_excl, uint32_t &sum_total,
                                                       uint32_t &max_excl, uint32_t &max_total)
{
    const uint32_t lane = bzx_lane(), w = bzx_wave();
    const uint32_t si = bzx_wave_incl_sum(sum_v);
    const uint32_t mi = bzx_wave_incl_max(max_v);
    uint32_t m_prev = __shfl_up(mi, 1);       // exclusive max within the wave
    if (lane == 0) m_prev = 0;
    if (lane == 63) {
        scratch[w] = si;
        scratch[NT / 64 + w] = mi;
    }
    bzx_lds_barrier();
    uint32_t pre = 0, tot = 0, mpre = 0, mtot = 0;
#pragma unroll
    for (uint32_t i = 0; i < NT / 64; i++) {
        const uint32_t t = scratch[i], m = scratch[NT / 64 + i];
        if (i < w) {
            pre += t;
            if (m > mpre) mpre = m;
        }
        tot += t;
        if (m > mtot) mtot = m;
    }
    bzx_lds_barrier();
    sum_excl = pre + si - sum_v;
    sum_total = tot;
    max_excl = m_prev > mpre ? m_prev : mpre;
    max_total = mtot;
}

// Same with LDS-only barriers.
template <int NT>
__device__ __forceinline__ uint32_t bzx_block_excl_sum_lds(uint32_t v, uint32_t *scratch, uint32_t &total)
{
    const uint32_t incl = bzx_wave_incl_sum(v);
    if (bzx_lane() == 63) scratch[bzx_wave()] = incl;
    bzx_lds_barrier();
    uint32_t pre = 0, tot = 0;
    const uint32_t w = bzx_wave();
#pragma unroll
    for (uint32_t i = 0; i < NT / 64; i++) {
        const uint32_t t = scratch[i];
        if (i < w) pre += t;
        tot += t;
    }
    bzx_lds_barrier();
    total = tot;
    return pre + incl - v;
}

