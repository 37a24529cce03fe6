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

// Wave inclusive sum scan.
__device__ __forceinline__ uint32_t bzx_wave_incl_sum(uint32_t v)
{
    const uint32_t lane = bzx_lane();
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

// Wave inclusive max scan.
__device__ __forceinline__ uint32_t bzx_wave_incl_max(uint32_t v)
{
    const uint32_t lane = bzx_lane();
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d);
        if (lane >= d && y > v) v = y;
    }
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

