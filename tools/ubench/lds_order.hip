// Probe (not product): are the values returned by ONE wave-wide ds_add_rtn_u32 handed out in lane order among lanes
// that hit the same address?  (Undocumented; the sort kernel only uses it optimistically and verifies the result.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_order(uint32_t seed0, uint32_t iters, uint32_t nctr_mask, unsigned long long *viol, unsigned long long *total)
{
    __shared__ uint32_t ctr[4][2048];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = lane; i < 2048; i += 64) ctr[wave][i] = 0;
    uint32_t s = seed0 ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u) | 1u;
    unsigned long long bad = 0, tot = 0;
    for (uint32_t it = 0; it < iters; it++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        // skew: a quarter of the lanes use one of 4 hot counters
        uint32_t d = ((s >> 8) & 3u) == 0 ? ((s >> 12) & 3u) : ((s >> 12) & nctr_mask);
        const uint32_t before = ctr[wave][d];
        asm volatile("" ::: "memory");
        const uint32_t r = atomicAdd(&ctr[wave][d], 1u);
        asm volatile("" ::: "memory");
        // expected: before + number of lower lanes with the same d
        uint32_t lower = 0;
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t dl = __shfl(d, l);
            lower += (l < lane && dl == d);
        }
        bad += (r != before + lower);
        tot++;
    }
    atomicAdd(viol, bad);
    atomicAdd(total, tot);
}

int main()
{
    unsigned long long *d; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
    for (uint32_t mask : {7u, 63u, 255u, 2047u}) {
        hipLaunchKernelGGL(k_order, dim3(2048), dim3(256), 0, 0, 12345u + mask, 2000u, mask, d, d + 1);
        CK(hipDeviceSynchronize());
        unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("counters %4u: %llu lane-ops checked so far, %llu out of lane order\n", mask + 1, h[1], h[0]);
    }
    return 0;
}
