/*
 * Device-side COO -> HDIA construction for gfx950 (include/spgpu/convert_device.h).
 * New functionality; the arrays it produces are byte-identical to the host
 * converter's (spgpu_amd/csrc/conv_hdia.cpp; reference hdia.cpp:161-349) for the same COO input given
 * hdiaValues zeroed by the caller: diagonals of a hack in ascending (column - row), duplicates of one
 * (row, column) resolved in favour of the LAST entry in COO order.
 *
 * Plan:  one 64-bit key per entry, (hack << 32) | (column0 - row0 % hackSize + hackSize), sorted and reduced to
 *        its distinct values: entry g of that list IS stored diagonal g (hack-major, ascending offset inside a
 *        hack).  hackOffsets[h] is the position of the first key of hack >= h (binary search).
 *        The sort and the duplicate removal are rocPRIM's (radix_sort_keys, unique); this is format
 *        construction, not the SpMV path.
 * Fill:  an entry finds its diagonal by binary search among its hack's offsets.  Which entry owns a slot is
 *        settled first (atomicMax of the COO position into a per-slot scratch word), then only owners write,
 *        so the result does not depend on scheduling.
 *
 * Plan scratch layout:  misc[16 ints] | distinct keys / sort buffer A [nnz u64] | sort buffer B [nnz u64] | rocPRIM temp
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/convert_device.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace spgpu {

constexpr int kHdThreads = 256;
typedef unsigned long long DiagKey;

struct HdiaPlanWork {
    int* misc;        /* [0] distinct count (written by rocPRIM as size_t in misc[0..1]), [2] out-of-range flag */
    DiagKey* a;
    DiagKey* b;
    void* temp;
    size_t tempBytes;
};

static size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* rocPRIM temporary storage for n keys (needs a device: asks for the target architecture) */
static hipError_t primTempBytes(size_t n, size_t* bytes)
{
    size_t sortBytes = 0, uniqueBytes = 0;
    hipError_t err = rocprim::radix_sort_keys(nullptr, sortBytes, (DiagKey*)nullptr, (DiagKey*)nullptr, n);
    if (err != hipSuccess)
        return err;
    err = rocprim::unique(nullptr, uniqueBytes, (DiagKey*)nullptr, (DiagKey*)nullptr, (size_t*)nullptr, n,
                          rocprim::equal_to<DiagKey>());
    if (err != hipSuccess)
        return err;
    *bytes = alignUp(sortBytes > uniqueBytes ? sortBytes : uniqueBytes, 256);
    return hipSuccess;
}

static HdiaPlanWork carveHdia(void* work, int nnz, size_t tempBytes)
{
    HdiaPlanWork w;
    char* p = static_cast<char*>(work);
    w.misc = reinterpret_cast<int*>(p);
    p += 256;
    w.a = reinterpret_cast<DiagKey*>(p);
    p += alignUp((size_t)nnz * sizeof(DiagKey), 256);
    w.b = reinterpret_cast<DiagKey*>(p);
    p += alignUp((size_t)nnz * sizeof(DiagKey), 256);
    w.temp = p;
    w.tempBytes = tempBytes;
    return w;
}

static unsigned gridOver(long long n)
{
    const long long blocks = (n + kHdThreads - 1) / kHdThreads;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1048576 ? 1048576 : blocks));
}

__device__ inline DiagKey makeKey(int row0, int col0, int hackSize)
{
    const unsigned hack = (unsigned)row0 / (unsigned)hackSize;
    const int inHack = row0 - (int)hack * hackSize;
    return ((DiagKey)hack << 32) | (DiagKey)(unsigned)(col0 - inHack + hackSize); /* > 0: col0 >= 0, inHack < hackSize */
}

__global__ __launch_bounds__(kHdThreads) void diagKeysKernel(DiagKey* keys, int* misc, int rows, int cols, int nnz,
                                                             const int* cooRows, const int* cooCols, int base, int hackSize)
{
    const long long stride = (long long)gridDim.x * kHdThreads;
    for (long long e = (long long)blockIdx.x * kHdThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base, c = cooCols[e] - base;
        if (r < 0 || r >= rows || c < 0 || c >= cols) {
            misc[2] = 1;
            keys[e] = ~(DiagKey)0; /* sorts last; the call reports SPGPU_UNSUPPORTED */
        } else {
            keys[e] = makeKey(r, c, hackSize);
        }
    }
}

/* hackOffsets[h] = number of distinct keys whose hack is < h, h = 0 .. hacks */
__global__ __launch_bounds__(kHdThreads) void hackOffsetsKernel(int* hackOffsets, int hacks, const DiagKey* distinct,
                                                                const int* misc)
{
    const long long count = *reinterpret_cast<const size_t*>(misc);
    const long long stride = (long long)gridDim.x * kHdThreads;
    for (long long h = (long long)blockIdx.x * kHdThreads + threadIdx.x; h <= hacks; h += stride) {
        const DiagKey bound = (DiagKey)h << 32;
        long long lo = 0, hi = count;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (distinct[mid] < bound)
                lo = mid + 1;
            else
                hi = mid;
        }
        hackOffsets[h] = (int)lo;
    }
}

__global__ __launch_bounds__(kHdThreads) void diagOffsetsKernel(int* hdiaOffsets, long long count, const DiagKey* distinct,
                                                                int hackSize)
{
    const long long stride = (long long)gridDim.x * kHdThreads;
    for (long long g = (long long)blockIdx.x * kHdThreads + threadIdx.x; g < count; g += stride) {
        const DiagKey key = distinct[g];
        const int hack = (int)(key >> 32);
        const int shifted = (int)(unsigned)(key & 0xffffffffu); /* column0 - row0 % hackSize + hackSize */
        hdiaOffsets[g] = shifted - hackSize - hack * hackSize;  /* == column - row */
    }
}

/* slot of COO entry e, or -1 */
__device__ inline long long slotOf(long long e, const int* cooRows, const int* cooCols, int base, int hackSize,
                                   const int* hackOffsets, const int* hdiaOffsets)
{
    const int r = cooRows[e] - base, c = cooCols[e] - base;
    const unsigned hack = (unsigned)r / (unsigned)hackSize;
    const int want = c - r;
    int lo = hackOffsets[hack], hi = hackOffsets[hack + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (hdiaOffsets[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (long long)lo * hackSize + (r - (int)hack * hackSize);
}

__global__ __launch_bounds__(kHdThreads) void claimSlotsKernel(int* owner, int nnz, const int* cooRows, const int* cooCols,
                                                               int base, int hackSize, const int* hackOffsets,
                                                               const int* hdiaOffsets)
{
    const long long stride = (long long)gridDim.x * kHdThreads;
    for (long long e = (long long)blockIdx.x * kHdThreads + threadIdx.x; e < nnz; e += stride)
        atomicMax(&owner[slotOf(e, cooRows, cooCols, base, hackSize, hackOffsets, hdiaOffsets)], (int)e + 1);
}

template <typename E>
__global__ __launch_bounds__(kHdThreads) void writeOwnedKernel(E* values, const int* owner, int nnz, const int* cooRows,
                                                               const int* cooCols, const E* cooValues, int base,
                                                               int hackSize, const int* hackOffsets, const int* hdiaOffsets)
{
    const long long stride = (long long)gridDim.x * kHdThreads;
    for (long long e = (long long)blockIdx.x * kHdThreads + threadIdx.x; e < nnz; e += stride) {
        const long long slot = slotOf(e, cooRows, cooCols, base, hackSize, hackOffsets, hdiaOffsets);
        if (owner[slot] == (int)e + 1)
            values[slot] = cooValues[e];
    }
}

} // namespace spgpu

using namespace spgpu;

extern "C" size_t spgpuCooHdiaPlanWorkBytes(int rowsCount, int nonZerosCount)
{
    (void)rowsCount;
    const size_t n = nonZerosCount > 0 ? (size_t)nonZerosCount : 1;
    size_t temp = 0;
    if (primTempBytes(n, &temp) != hipSuccess)
        return 0; /* no device to ask */
    return 256 + 2 * alignUp(n * sizeof(DiagKey), 256) + temp;
}

extern "C" size_t spgpuCooToHdiaScratchBytes(int hackSize, int allocationHeight)
{
    const size_t slots = (size_t)(hackSize > 0 ? hackSize : 0) * (size_t)(allocationHeight > 0 ? allocationHeight : 0);
    return (slots ? slots : 1) * sizeof(int);
}

extern "C" spgpuStatus_t spgpuCooHdiaPlanDevice(spgpuHandle_t handle, int* allocationHeight, int* hackOffsets,
                                                int hackSize, int rowsCount, int columnsCount, int nonZerosCount,
                                                const int* cooRowIndices, const int* cooColsIndices, int cooBaseIndex,
                                                void* work)
{
    if (!handle || !allocationHeight || !hackOffsets || !work || hackSize <= 0 || rowsCount < 0 || nonZerosCount < 0)
        return SPGPU_UNSUPPORTED;
    hipStream_t stream = handle->currentStream;
    const int hacks = (rowsCount + hackSize - 1) / hackSize;
    const size_t n = nonZerosCount > 0 ? (size_t)nonZerosCount : 1;
    size_t tempBytes = 0;
    if (primTempBytes(n, &tempBytes) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    HdiaPlanWork w = carveHdia(work, (int)n, tempBytes);
    if (hipMemsetAsync(w.misc, 0, 256, stream) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    if (nonZerosCount > 0) {
        hipLaunchKernelGGL(diagKeysKernel, dim3(gridOver(nonZerosCount)), dim3(kHdThreads), 0, stream, w.b, w.misc, rowsCount,
                           columnsCount, nonZerosCount, cooRowIndices, cooColsIndices, cooBaseIndex, hackSize);
        size_t bytes = w.tempBytes;
        if (rocprim::radix_sort_keys(w.temp, bytes, w.b, w.a, (size_t)nonZerosCount, 0, 64, stream) != hipSuccess)
            return SPGPU_UNSPECIFIED;
        bytes = w.tempBytes;
        /* distinct keys into B, their number (size_t) into misc[0..1] */
        if (rocprim::unique(w.temp, bytes, w.a, w.b, reinterpret_cast<size_t*>(w.misc), (size_t)nonZerosCount,
                            rocprim::equal_to<DiagKey>(), stream) != hipSuccess)
            return SPGPU_UNSPECIFIED;
        /* keep the list where the fill call expects it: buffer A */
        if (hipMemcpyAsync(w.a, w.b, (size_t)nonZerosCount * sizeof(DiagKey), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return SPGPU_UNSPECIFIED;
    }
    hipLaunchKernelGGL(hackOffsetsKernel, dim3(gridOver((long long)hacks + 1)), dim3(kHdThreads), 0, stream, hackOffsets, hacks,
                       w.a, w.misc);
    int host[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(host, w.misc, sizeof(host), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    if (host[2]) {
        *allocationHeight = 0;
        return SPGPU_UNSUPPORTED; /* an entry outside the matrix */
    }
    *allocationHeight = host[0]; /* low word of the size_t count: fewer than 2^31 diagonals by the format's int offsets */
    spgpuDebugCheck(handle, "cooHdiaPlanDevice");
    return SPGPU_SUCCESS;
}

extern "C" spgpuStatus_t spgpuCooToHdiaDevice(spgpuHandle_t handle, void* hdiaValues, int* hdiaOffsets,
                                              const int* hackOffsets, int hackSize, int rowsCount, int columnsCount,
                                              int nonZerosCount, const int* cooRowIndices, const int* cooColsIndices,
                                              const void* cooValues, int cooBaseIndex, spgpuType_t valuesType,
                                              int allocationHeight, const void* work, void* scratch)
{
    (void)columnsCount;
    if (!handle || !work || !scratch || hackSize <= 0 || rowsCount < 0 || nonZerosCount < 0 || allocationHeight < 0)
        return SPGPU_UNSUPPORTED;
    if (nonZerosCount == 0 || allocationHeight == 0)
        return SPGPU_SUCCESS;
    hipStream_t stream = handle->currentStream;
    const HdiaPlanWork w = carveHdia(const_cast<void*>(work), nonZerosCount, 0);
    int* owner = static_cast<int*>(scratch);
    if (hipMemsetAsync(owner, 0, spgpuCooToHdiaScratchBytes(hackSize, allocationHeight), stream) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    hipLaunchKernelGGL(diagOffsetsKernel, dim3(gridOver(allocationHeight)), dim3(kHdThreads), 0, stream, hdiaOffsets,
                       (long long)allocationHeight, w.a, hackSize);
    const dim3 grid(gridOver(nonZerosCount)), block(kHdThreads);
    hipLaunchKernelGGL(claimSlotsKernel, grid, block, 0, stream, owner, nonZerosCount, cooRowIndices, cooColsIndices,
                       cooBaseIndex, hackSize, hackOffsets, hdiaOffsets);
    switch (spgpuSizeOf(valuesType)) {
    case 4:
        hipLaunchKernelGGL(writeOwnedKernel<uint32_t>, grid, block, 0, stream, static_cast<uint32_t*>(hdiaValues), owner,
                           nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const uint32_t*>(cooValues), cooBaseIndex,
                           hackSize, hackOffsets, hdiaOffsets);
        break;
    case 8:
        hipLaunchKernelGGL(writeOwnedKernel<uint64_t>, grid, block, 0, stream, static_cast<uint64_t*>(hdiaValues), owner,
                           nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const uint64_t*>(cooValues), cooBaseIndex,
                           hackSize, hackOffsets, hdiaOffsets);
        break;
    case 16:
        hipLaunchKernelGGL(writeOwnedKernel<ulonglong2>, grid, block, 0, stream, static_cast<ulonglong2*>(hdiaValues), owner,
                           nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const ulonglong2*>(cooValues), cooBaseIndex,
                           hackSize, hackOffsets, hdiaOffsets);
        break;
    default:
        return SPGPU_UNSUPPORTED;
    }
    spgpuDebugCheck(handle, "cooToHdiaDevice");
    return SPGPU_SUCCESS;
}
