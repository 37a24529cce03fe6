// Micro-benchmark (probe, not product): rate of scattered 8-byte / 4-byte global stores as the split kernel issues them
// (one workgroup per 900k-record block, ~660 destination cursors per block), against runs of R consecutive records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define NREC 900000u

template <typename T>
__global__ __launch_bounds__(1024) void k_scatter(const uint32_t *__restrict__ perm, T *__restrict__ dst, uint32_t nblk)
{
    for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        T *d = dst + (size_t)b * NREC;
        for (uint32_t i = threadIdx.x; i < NREC; i += 1024 * 4) {
            uint32_t p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = i + u * 1024 < NREC ? perm[i + u * 1024] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++) if (p[u] != 0xFFFFFFFFu) d[p[u]] = (T)(i + u) * (T)0x9E3779B1u;
        }
    }
}

int main()
{
    const uint32_t nblk = 1194;
    uint32_t *d_perm; void *d_dst;
    CK(hipMalloc(&d_perm, NREC * 4)); CK(hipMalloc(&d_dst, (size_t)nblk * NREC * 8));
    std::vector<uint32_t> perm(NREC);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int runs[] = {0, 1, 2, 4, 8, 16, 32};
    for (int ri = 0; ri < 7; ri++) {
        const int R = runs[ri];
        if (R == 0) { for (uint32_t i = 0; i < NREC; i++) perm[i] = i; }
        else {
            // 660 buckets of skewed sizes; the stream visits buckets at random in runs of R records
            const int NB = 660; std::vector<uint32_t> size(NB), base(NB + 1), cur(NB, 0);
            uint64_t s = 88172645463325252ull; double tot = 0; std::vector<double> w(NB);
            for (int k = 0; k < NB; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[k] = 0.2 + ((s >> 11) & 0xFFFF) / 65536.0; tot += w[k]; }
            uint32_t acc = 0; for (int k = 0; k < NB; k++) { size[k] = (uint32_t)(w[k] / tot * NREC); base[k] = acc; acc += size[k]; }
            size[NB - 1] += NREC - acc; base[NB] = NREC;
            uint32_t i = 0;
            while (i < NREC) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                int k = (int)((s >> 20) % NB);
                while (cur[k] >= size[k]) k = (k + 1) % NB;
                for (int r = 0; r < R && i < NREC && cur[k] < size[k]; r++) perm[i++] = base[k] + cur[k]++;
            }
            // a lane handles records i, i+1024, ...: runs must be runs across LANES, which they are (consecutive i)
        }
        CK(hipMemcpy(d_perm, perm.data(), NREC * 4, hipMemcpyHostToDevice));
        for (int w = 0; w < 2; w++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (w == 0) hipLaunchKernelGGL(k_scatter<uint64_t>, dim3(256), dim3(1024), 0, 0, d_perm, (uint64_t *)d_dst, nblk);
                else hipLaunchKernelGGL(k_scatter<uint32_t>, dim3(256), dim3(1024), 0, 0, d_perm, (uint32_t *)d_dst, nblk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("run length %2d (%s) %d-byte records: %7.3f ms for %u blocks  (%6.1f GB/s of records)\n", R, R == 0 ? "sequential" : "scattered",
                   w == 0 ? 8 : 4, ms, nblk, (double)nblk * NREC * (w == 0 ? 8 : 4) / (ms * 1e6));
        }
    }
    return 0;
}
