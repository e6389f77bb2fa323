/*
 * Lab builds only (-DSPGPU_TUNING_VARIANTS): spgpuDebugFillLds(handle, word) fills the LDS of every CU with one 32-bit word (0; or ~0: -1 as an
 * integer, NaN as a float or double) on the handle's current stream.  LDS is not cleared between workgroups: a kernel that reads a word of it before writing it finds what the last
 * workgroup on that CU left there -- in a loop over one matrix the right values of the previous launch, which hides the slip.
 * tools/stress_lds.py runs the ordered SpMV paths with this in front of every call; no product path calls it.
 */
#include <hip/hip_runtime.h>

#include "spgpu/core.h"

#ifdef SPGPU_TUNING_VARIANTS
namespace {

constexpr int kLdsBytes = 80 * 1024; /* two such workgroups hold the CU's 160 KiB */

__global__ __launch_bounds__(256) void fillLdsKernel(unsigned word, unsigned* sink)
{
    extern __shared__ unsigned words[];
    for (int i = threadIdx.x; i < kLdsBytes / 4; i += 256)
        words[i] = word;
    __syncthreads();
    /* stay for a while, so that the second workgroup of the CU gets the other half instead of this one again */
    const unsigned long long until = wall_clock64() + 2000ull; /* 20 us at 100 MHz */
    unsigned seen = 0u;
    while (wall_clock64() < until)
        seen += words[(threadIdx.x * 97u + seen) % (kLdsBytes / 4)];
    if (seen == 0x12345678u && sink)
        *sink = seen;
}

} // namespace

extern "C" int spgpuDebugFillLds(spgpuHandle_t handle, unsigned word)
{
    static bool sized = false;
    if (!sized) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fillLdsKernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
            return -1;
        sized = true;
    }
    hipLaunchKernelGGL(fillLdsKernel, dim3(2 * 256), dim3(256), kLdsBytes, handle->currentStream, word, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif
