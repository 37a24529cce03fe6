// Launchers not written yet (removed as the stages land).
#include <hip/hip_runtime.h>
#include "bzx_device.h"
void bzx_launch_mtf(const BzxBatch &, uint32_t, hipStream_t) {}
void bzx_launch_huffman(const BzxBatch &, uint32_t, hipStream_t) {}
