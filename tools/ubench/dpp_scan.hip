// Probe (not product): the DPP wave scans of bzx_wg.h against a serial reference, full and partial EXEC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../bzip2-rust_amd/csrc/bzx_wg.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k(const uint32_t *in, uint32_t *out, uint32_t nact)
{
    const uint32_t t = threadIdx.x;
    out[t] = out[256 + t] = out[512 + t] = out[768 + t] = 0xDEADBEEF;
    if (t < nact) {
        out[t] = bzx_wave_incl_sum(in[t]);
        out[256 + t] = bzx_wave_incl_max(in[t]);
        out[512 + t] = bzx_wave_incl_or(in[t]);
        out[768 + t] = bzx_wave_incl_and(in[t] | 0xFFFF0000u);
    }
}
int main()
{
    uint32_t h[256], o[1024], *di, *dout;
    for (int i = 0; i < 256; i++) h[i] = (i * 2654435761u) >> 20;
    CK(hipMalloc(&di, 1024)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(di, h, 1024, hipMemcpyHostToDevice));
    for (uint32_t nact : {256u, 200u, 64u, 37u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, di, dout, nact);
        CK(hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost));
        int bad[4] = {0, 0, 0, 0};
        for (uint32_t w = 0; w < 4; w++) {
            uint32_t s = 0, m = 0, r = 0, a = ~0u;
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t t = w * 64 + l;
                if (t >= nact) break;
                s += h[t]; m = h[t] > m ? h[t] : m; r |= h[t]; a &= h[t] | 0xFFFF0000u;
                bad[0] += o[t] != s; bad[1] += o[256 + t] != m; bad[2] += o[512 + t] != r; bad[3] += o[768 + t] != a;
            }
        }
        printf("active %3u: mismatches sum %d max %d or %d and %d\n", nact, bad[0], bad[1], bad[2], bad[3]);
    }
    return 0;
}
