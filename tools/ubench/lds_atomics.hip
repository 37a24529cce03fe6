// Micro-benchmark (probe, not product): throughput of LDS atomics on gfx950 under the address patterns the split and
// sort kernels would use.  Reports lane-operations per nanosecond per compute unit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int NCTR>
__global__ __launch_bounds__(256) void k_atomics(const uint32_t *__restrict__ idx, uint32_t per_lane, uint32_t iters, uint32_t *out)
{
    __shared__ uint32_t ctr[NCTR];
    for (uint32_t i = threadIdx.x; i < NCTR; i += 256) ctr[i] = 0;
    __syncthreads();
    uint32_t acc = 0;
    // indices for this lane are loaded once into registers (16 per lane), then replayed `iters` times
    uint32_t a[16];
    for (int k = 0; k < 16; k++) a[k] = idx[(blockIdx.x % 64) * 256 * 16 + k * 256 + threadIdx.x] % NCTR;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) atomicAdd(&ctr[a[k]], 1u);                          // no return
            else if (MODE == 1) acc += atomicAdd(&ctr[a[k]], 1u);              // returning
            else if (MODE == 2) acc += ctr[a[k]];                              // plain read
            else if (MODE == 3) ctr[a[k]] = acc + k;                           // plain write
        }
        if (MODE >= 2) __syncthreads();
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NCTR; i += 256) acc += ctr[i];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int NCTR>
static void run(const char *name, const uint32_t *d_idx, uint32_t *d_out, int wgs_per_cu)
{
    const uint32_t iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k_atomics<MODE, NCTR>), dim3(grid), dim3(256), 0, 0, d_idx, 16, 10, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_atomics<MODE, NCTR>), dim3(grid), dim3(256), 0, 0, d_idx, 16, iters, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)grid * 256 * 16 * iters;
    printf("%-44s ctr=%6d wg/cu=%d  %8.3f ms  %7.2f lane-ops/ns/CU  (%5.1f ns per wave-instr per CU)\n", name, NCTR, wgs_per_cu, ms,
           ops / (ms * 1e6) / 256.0, 64.0 / (ops / (ms * 1e6) / 256.0));
}

int main()
{
    const size_t N = 64 * 256 * 16;
    std::vector<uint32_t> h(N);
    uint32_t *d_idx, *d_out;
    CK(hipMalloc(&d_idx, N * 4)); CK(hipMalloc(&d_out, 64));
    for (int pat = 0; pat < 4; pat++) {
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < N; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            uint32_t r = (uint32_t)(s >> 20);
            if (pat == 0) h[i] = r;                                         // uniform
            else if (pat == 1) { double u = (r & 0xFFFFFF) / 16777216.0; h[i] = (uint32_t)(u * u * u * u * 32768.0); }  // skewed
            else if (pat == 2) h[i] = (uint32_t)i;                           // sequential (conflict-free)
            else h[i] = (r & 0xFF) < 64 ? 7u : r;                            // 25 % on one address
        }
        CK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        const char *pn[4] = {"uniform", "skewed u^4", "sequential", "25% one address"};
        printf("---- pattern: %s\n", pn[pat]);
        run<0, 256>("ds_add (no return)", d_idx, d_out, 4);
        run<0, 2048>("ds_add (no return)", d_idx, d_out, 4);
        run<0, 32768>("ds_add (no return)", d_idx, d_out, 1);
        run<1, 256>("ds_add_rtn", d_idx, d_out, 4);
        run<1, 2048>("ds_add_rtn", d_idx, d_out, 4);
        run<1, 32768>("ds_add_rtn", d_idx, d_out, 1);
        run<2, 2048>("ds_read_b32", d_idx, d_out, 4);
        run<3, 2048>("ds_write_b32", d_idx, d_out, 4);
    }
    return 0;
}
