/*
 * tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A minimal single-threaded emulation of the HIP programming model (workgroups of fibers,
 * 64-lane wave collectives, block barriers, LDS as static storage) so that the kernels in
 * bzip2-rust_amd/csrc/ can be compiled with g++ and their LOGIC (indexing, scans, loop
 * bounds) debugged in the GPU-less build container:
 *     g++ -I tests/emu -x c++ ... csrc/*.hip
 * It is never linked into libbzx.so; the product has no CPU path.  Performance, memory
 * ordering and scheduling are NOT modelled: only -m gpu tests prove the device code.
 */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <vector>
#include <algorithm>

#define BZX_HIP_EMU 1

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4

struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; };

namespace hipemu {
void wave_sync();
void block_sync();
uint64_t *wave_slots();   /* 64 exchange slots of the calling fiber's wave */
unsigned lane_id();
void launch(void (*trampoline)(void *), void *args, dim3 grid, dim3 block);
}

inline void __syncthreads() { hipemu::block_sync(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

inline unsigned long long __ballot(int pred)
{
    uint64_t *s = hipemu::wave_slots();
    unsigned l = hipemu::lane_id();
    s[l] = pred ? 1 : 0;
    hipemu::wave_sync();
    unsigned long long m = 0;
    unsigned nl = std::min<unsigned>(64u, blockDim.x - (threadIdx.x & ~63u));
    for (unsigned i = 0; i < nl; i++) m |= (unsigned long long)(s[i] & 1) << i;
    hipemu::wave_sync();
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { unsigned nl = std::min<unsigned>(64u, blockDim.x - (threadIdx.x & ~63u));
    unsigned long long full = nl == 64 ? ~0ull : ((1ull << nl) - 1); return __ballot(pred) == full; }

template <typename T> inline T __shfl(T v, int src, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl");
    uint64_t *s = hipemu::wave_slots();
    unsigned l = hipemu::lane_id();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s[l] = raw;
    hipemu::wave_sync();
    unsigned base = l & ~(unsigned)(width - 1);
    uint64_t r = s[base + ((unsigned)src & (unsigned)(width - 1))];
    hipemu::wave_sync();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64)
{
    unsigned l = hipemu::lane_id();
    unsigned base = l & ~(unsigned)(width - 1);
    T r = __shfl(v, (int)((l - base) >= d ? (l - base - d) : (l - base)), width);
    return r;
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64)
{
    unsigned l = hipemu::lane_id();
    unsigned base = l & ~(unsigned)(width - 1);
    unsigned rel = l - base;
    return __shfl(v, (int)(rel + d < (unsigned)width ? rel + d : rel), width);
}
template <typename T> inline T __shfl_xor(T v, int m, int width = 64)
{
    unsigned l = hipemu::lane_id();
    unsigned base = l & ~(unsigned)(width - 1);
    return __shfl(v, (int)(((l - base) ^ (unsigned)m)), width);
}

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) if (x & (1u << i)) r |= 1u << (31 - i); return r; }

template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

/* ---- host runtime subset: device memory is host memory ---- */
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t = 0) { memmove(d, s, n); return 0; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return 0; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = 0; return 0; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *a, int *b) { *a = 0; *b = 0; return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = 0; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 2; strcpy(p->name, "emu"); strcpy(p->gcnArchName, "emu"); return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = 0; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

/* hipLaunchKernelGGL(kernel, grid, block, shmem, stream, args...) */
#include <tuple>
#include <utility>
namespace hipemu {
template <typename F, typename Tup, size_t... I> inline void call_tuple(F f, Tup &t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
template <typename F, typename... A> struct Thunk {
    F f;
    std::tuple<A...> args;
    static void run(void *p) { Thunk *t = (Thunk *)p; call_tuple(t->f, t->args, std::index_sequence_for<A...>{}); }
};
template <typename... P, typename... A>
inline void launch_kernel(void (*k)(P...), dim3 grid, dim3 block, A... a)
{
    Thunk<void (*)(P...), P...> t{k, std::tuple<P...>(static_cast<P>(a)...)};
    launch(&Thunk<void (*)(P...), P...>::run, &t, grid, block);
}
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_kernel(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)
template <typename F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 1; return 0; }
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
inline unsigned long long wall_clock64() { return 0; }
inline int __syncthreads_or(int p)
{
    static int acc;
    __syncthreads();
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    if (p) acc = 1;
    __syncthreads();
    return acc;
}
inline int __syncthreads_and(int p)
{
    static int acc;
    __syncthreads();
    if (threadIdx.x == 0) acc = 1;
    __syncthreads();
    if (!p) acc = 0;
    __syncthreads();
    return acc;
}
