/*
 * Rows ordered by length, in HBM (include/spgpu/oell_device.h).  New functionality next to the reference's host
 * ellToOell (ell.c:85-202); the order is the one the host oellOrder (csrc/conv_ell.c) defines and, for one window
 * and no long-row group, exactly the reference's: (length, row) descending.
 *
 * Pass 1: one 64-bit key per row, (~length << 32) | ~row, sorted ascending = (length, row) descending.
 * Pass 2 (windows or long rows set aside only): a STABLE sort of the pass-1 sequence by group number (the long rows'
 *         windows first, then the others'), which leaves every group in pass-1 order; the groups that ascend are then
 *         read back to front.
 * Both sorts are rocPRIM radix sorts; this is format construction, not the SpMV path.
 *
 * Scratch layout:  A [rows u64] | B [rows u64] | rocPRIM temp.   Pass 2 reuses A as {group in, row in} and B as
 * {group out, row out}, rows u32 each.
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/oell_device.h"
#include "spgpu/ell_conv.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace spgpu {

constexpr int kOeThreads = 256;
typedef unsigned long long LenKey;

static size_t alignUpOe(size_t v, size_t a) { return (v + a - 1) / a * a; }

static unsigned gridOverOe(long long n)
{
    const long long blocks = (n + kOeThreads - 1) / kOeThreads;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1048576 ? 1048576 : blocks));
}

/* 1 for a row that is set aside (longer than the threshold): scanned for the aligned form of the order */
struct LongerThan {
    int threshold;
    __host__ __device__ int operator()(int len) const { return len > threshold ? 1 : 0; }
};

static hipError_t orderTempBytes(size_t n, size_t* bytes)
{
    size_t keysBytes = 0, pairsBytes = 0;
    hipError_t err = rocprim::radix_sort_keys(nullptr, keysBytes, (LenKey*)nullptr, (LenKey*)nullptr, n);
    if (err != hipSuccess)
        return err;
    err = rocprim::radix_sort_pairs(nullptr, pairsBytes, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, n);
    if (err != hipSuccess)
        return err;
    size_t scanBytes = 0;
    err = rocprim::exclusive_scan(nullptr, scanBytes, rocprim::make_transform_iterator((const int*)nullptr, LongerThan{0}), (int*)nullptr, 0, n,
                                  rocprim::plus<int>());
    if (err != hipSuccess)
        return err;
    size_t most = keysBytes > pairsBytes ? keysBytes : pairsBytes;
    most = most > scanBytes ? most : scanBytes;
    *bytes = alignUpOe(most, 256);
    return hipSuccess;
}

/* group number with the direction of the group in bit 0: (group << 1) | ascending */
/* longBefore (aligned form only, else null): rows set aside among the rows before this one; longCount: all of them */
__device__ inline unsigned groupOfRow(int row, int len, int window, int longRows, unsigned longGroups, const int* longBefore,
                                      long long longCount)
{
    unsigned inClass, group;
    if (longRows > 0 && len > longRows) {
        inClass = window > 0 ? (unsigned)((long long)row / ((long long)window * SPGPU_OELL_LONG_WINDOW_FACTOR)) : 0u;
        group = inClass;
    } else {
        if (longBefore) /* runs of `window` of the shorter rows, every run but the first starting on a multiple of `window` */
            inClass = (unsigned)((longCount + (row - longBefore[row])) / window - longCount / window);
        else
            inClass = window > 0 ? (unsigned)row / (unsigned)window : 0u;
        group = longGroups + inClass;
    }
    return (group << 1) | (inClass & 1u);
}

__global__ __launch_bounds__(kOeThreads) void lengthKeysKernel(LenKey* keys, const int* rs, int rows)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long r = (long long)blockIdx.x * kOeThreads + threadIdx.x; r < rows; r += stride)
        keys[r] = ((LenKey)(0xFFFFFFFFu - (unsigned)rs[r]) << 32) | (LenKey)(0xFFFFFFFFu - (unsigned)r);
}

__global__ __launch_bounds__(kOeThreads) void groupKeysKernel(unsigned* groups, unsigned* rowsOut, const LenKey* sorted,
                                                              int rows, int window, int longRows, unsigned longGroups,
                                                              const int* longBefore, const int* srcRs)
{
    const long long longCount = longBefore ? (long long)longBefore[rows - 1] + (srcRs[rows - 1] > longRows ? 1 : 0) : 0;
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long i = (long long)blockIdx.x * kOeThreads + threadIdx.x; i < rows; i += stride) {
        const LenKey key = sorted[i];
        const unsigned row = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
        const unsigned len = 0xFFFFFFFFu - (unsigned)(key >> 32);
        groups[i] = groupOfRow((int)row, (int)len, window, longRows, longGroups, longBefore, longCount);
        rowsOut[i] = row;
    }
}

/* one group, or the two-row case the reference leaves unsorted */
__global__ __launch_bounds__(kOeThreads) void finishWholeKernel(int* rIdx, int* dstRs, const int* srcRs, const LenKey* sorted,
                                                                int rows, int identity)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long i = (long long)blockIdx.x * kOeThreads + threadIdx.x; i < rows; i += stride) {
        const int row = identity ? (int)i : (int)(0xFFFFFFFFu - (unsigned)(sorted[i] & 0xFFFFFFFFu));
        rIdx[i] = row;
        dstRs[i] = srcRs[row];
    }
}

__device__ inline long long firstNotBelow(const unsigned* sortedGroups, long long n, unsigned value)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sortedGroups[mid] < value)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kOeThreads) void finishGroupedKernel(int* rIdx, int* dstRs, const int* srcRs,
                                                                  const unsigned* sortedGroups, const unsigned* sortedRows,
                                                                  int rows)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long i = (long long)blockIdx.x * kOeThreads + threadIdx.x; i < rows; i += stride) {
        const unsigned g = sortedGroups[i]; /* (group << 1) | ascending */
        long long from = i;
        if (g & 1u) { /* an ascending group: its descending run read back to front */
            const long long first = firstNotBelow(sortedGroups, rows, g);
            const long long end = firstNotBelow(sortedGroups, rows, g + 1u);
            from = first + (end - 1 - i);
        }
        const int row = (int)sortedRows[from];
        rIdx[i] = row;
        dstRs[i] = srcRs[row];
    }
}

static spgpuStatus_t orderRows(spgpuHandle_t handle, int* rIdx, int* dstRs, const int* srcRs, int rows, int window,
                               int longRows, void* work, bool aligned = false)
{
    if (rows <= 0)
        return SPGPU_SUCCESS;
    if (!work)
        return SPGPU_UNSPECIFIED;
    hipStream_t s = handle->currentStream;
    size_t tempBytes = 0;
    if (orderTempBytes((size_t)rows, &tempBytes) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    char* p = static_cast<char*>(work);
    LenKey* a = reinterpret_cast<LenKey*>(p);
    p += alignUpOe((size_t)rows * sizeof(LenKey), 256);
    LenKey* b = reinterpret_cast<LenKey*>(p);
    p += alignUpOe((size_t)rows * sizeof(LenKey), 256);
    void* temp = p;
    p += tempBytes;
    int* longBefore = reinterpret_cast<int*>(p); /* aligned form: rows * 4 bytes behind the sort's scratch */
    aligned = aligned && window > 0 && longRows > 0;

    const bool whole = (window <= 0 || window >= rows) && longRows <= 0;
    if (whole && rows == 2) { /* ell.c:131-157 never merges exactly two rows */
        hipLaunchKernelGGL(finishWholeKernel, dim3(1), dim3(kOeThreads), 0, s, rIdx, dstRs, srcRs, (const LenKey*)nullptr, rows, 1);
        return SPGPU_SUCCESS;
    }
    hipLaunchKernelGGL(lengthKeysKernel, dim3(gridOverOe(rows)), dim3(kOeThreads), 0, s, a, srcRs, rows);
    size_t bytes = tempBytes;
    if (rocprim::radix_sort_keys(temp, bytes, a, b, (size_t)rows, 0, 64, s) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    if (whole) {
        hipLaunchKernelGGL(finishWholeKernel, dim3(gridOverOe(rows)), dim3(kOeThreads), 0, s, rIdx, dstRs, srcRs,
                           (const LenKey*)b, rows, 0);
        return SPGPU_SUCCESS;
    }
    unsigned* groupIn = reinterpret_cast<unsigned*>(a);
    unsigned* rowIn = groupIn + rows;
    const long long longWindow = window > 0 ? (long long)window * SPGPU_OELL_LONG_WINDOW_FACTOR : 0;
    const unsigned longGroups = longRows > 0 ? (longWindow > 0 ? (unsigned)((rows - 1) / longWindow) + 1u : 1u) : 0u;
    if (aligned) {
        bytes = tempBytes;
        if (rocprim::exclusive_scan(temp, bytes, rocprim::make_transform_iterator(srcRs, LongerThan{longRows}), longBefore, 0, (size_t)rows,
                                    rocprim::plus<int>(), s) != hipSuccess)
            return SPGPU_UNSPECIFIED;
    }
    hipLaunchKernelGGL(groupKeysKernel, dim3(gridOverOe(rows)), dim3(kOeThreads), 0, s, groupIn, rowIn, (const LenKey*)b, rows,
                       window, longRows, longGroups, aligned ? (const int*)longBefore : (const int*)nullptr, srcRs);
    /* b is free again once groupKeysKernel has read it (same stream) */
    unsigned* groupOut = reinterpret_cast<unsigned*>(b);
    unsigned* rowOut = groupOut + rows;
    const unsigned groups = longGroups + 1u + (window > 0 ? (unsigned)(rows - 1) / (unsigned)window : 0u);
    unsigned bits = 2; /* the key is (group << 1) | direction */
    while (bits < 32 && (1u << bits) < 2u * groups)
        ++bits;
    bytes = tempBytes;
    if (rocprim::radix_sort_pairs(temp, bytes, groupIn, groupOut, rowIn, rowOut, (size_t)rows, 0, bits, s) != hipSuccess)
        return SPGPU_UNSPECIFIED;
    hipLaunchKernelGGL(finishGroupedKernel, dim3(gridOverOe(rows)), dim3(kOeThreads), 0, s, rIdx, dstRs, srcRs,
                       (const unsigned*)groupOut, (const unsigned*)rowOut, rows);
    return SPGPU_SUCCESS;
}

/* real entries of row rIdx[i] -> row i; one lane per destination row, so that the stores of a slab column coalesce */
template <typename E>
__global__ __launch_bounds__(kOeThreads) void copyOrderedRowsKernel(E* dstValues, int* dstIndices, const E* srcValues,
                                                                    const int* srcIndices, const int* rIdx, const int* dstRs,
                                                                    long long valuesPitch, long long indicesPitch, int rows)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long i = (long long)blockIdx.x * kOeThreads + threadIdx.x; i < rows; i += stride) {
        const long long src = rIdx[i];
        const int len = dstRs[i];
        for (int k = 0; k < len; ++k) {
            dstValues[i + k * valuesPitch] = srcValues[src + k * valuesPitch];
            dstIndices[i + k * indicesPitch] = srcIndices[src + k * indicesPitch];
        }
    }
}

__global__ __launch_bounds__(kOeThreads) void invertOrderKernel(int* inverse, const int* rIdx, int rows)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long i = (long long)blockIdx.x * kOeThreads + threadIdx.x; i < rows; i += stride)
        inverse[rIdx[i]] = (int)i;
}

__global__ __launch_bounds__(kOeThreads) void permuteCooRowsKernel(int* dst, const int* src, int nnz, const int* inverse,
                                                                   int rows, int base)
{
    const long long stride = (long long)gridDim.x * kOeThreads;
    for (long long e = (long long)blockIdx.x * kOeThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = src[e] - base;
        dst[e] = (r >= 0 && r < rows) ? inverse[r] + base : src[e]; /* an entry outside the matrix stays outside */
    }
}

struct Bits128 { unsigned long long lo, hi; };

} // namespace spgpu

using namespace spgpu;

extern "C" {

size_t spgpuOellOrderWorkBytes(int rowsCount)
{
    if (rowsCount <= 0)
        return 256;
    size_t temp = 0;
    if (orderTempBytes((size_t)rowsCount, &temp) != hipSuccess)
        return 0;
    return 2 * alignUpOe((size_t)rowsCount * sizeof(LenKey), 256) + temp + alignUpOe((size_t)rowsCount * sizeof(int), 256);
}

spgpuStatus_t spgpuOellOrderDevice(spgpuHandle_t handle, int* rIdx, int* dstRs, const int* srcRs, int rowsCount, int window,
                                   int longRows, void* work)
{
    const spgpuStatus_t status = orderRows(handle, rIdx, dstRs, srcRs, rowsCount, window, longRows, work);
    spgpuDebugCheck(handle, "oellOrder");
    return status;
}

spgpuStatus_t spgpuOellOrderAlignedDevice(spgpuHandle_t handle, int* rIdx, int* dstRs, const int* srcRs, int rowsCount, int window,
                                   int longRows, void* work)
{
    const spgpuStatus_t status = orderRows(handle, rIdx, dstRs, srcRs, rowsCount, window, longRows, work, true);
    spgpuDebugCheck(handle, "oellOrder");
    return status;
}

spgpuStatus_t spgpuEllToOellDevice(spgpuHandle_t handle, int* rIdx, void* dstEllValues, int* dstEllIndices, int* dstRs,
                                   const void* srcEllValues, const int* srcEllIndices, const int* srcRs, int ellValuesPitch,
                                   int ellIndicesPitch, int rowsCount, spgpuType_t valuesType, int window, int longRows,
                                   void* work)
{
    if (rowsCount <= 0)
        return SPGPU_SUCCESS;
    const size_t elem = spgpuSizeOf(valuesType);
    if (elem != 4 && elem != 8 && elem != 16)
        return SPGPU_UNSUPPORTED;
    const spgpuStatus_t status = orderRows(handle, rIdx, dstRs, srcRs, rowsCount, window, longRows, work);
    if (status != SPGPU_SUCCESS)
        return status;
    hipStream_t s = handle->currentStream;
    const dim3 grid(gridOverOe(rowsCount)), block(kOeThreads);
    if (elem == 4)
        hipLaunchKernelGGL((copyOrderedRowsKernel<unsigned>), grid, block, 0, s, (unsigned*)dstEllValues, dstEllIndices,
                           (const unsigned*)srcEllValues, srcEllIndices, (const int*)rIdx, (const int*)dstRs,
                           (long long)ellValuesPitch, (long long)ellIndicesPitch, rowsCount);
    else if (elem == 8)
        hipLaunchKernelGGL((copyOrderedRowsKernel<unsigned long long>), grid, block, 0, s, (unsigned long long*)dstEllValues,
                           dstEllIndices, (const unsigned long long*)srcEllValues, srcEllIndices, (const int*)rIdx,
                           (const int*)dstRs, (long long)ellValuesPitch, (long long)ellIndicesPitch, rowsCount);
    else
        hipLaunchKernelGGL((copyOrderedRowsKernel<Bits128>), grid, block, 0, s, (Bits128*)dstEllValues, dstEllIndices,
                           (const Bits128*)srcEllValues, srcEllIndices, (const int*)rIdx, (const int*)dstRs,
                           (long long)ellValuesPitch, (long long)ellIndicesPitch, rowsCount);
    spgpuDebugCheck(handle, "ellToOell");
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCooPermuteRowsDevice(spgpuHandle_t handle, int* dstCooRowIndices, const int* srcCooRowIndices,
                                        int nonZerosCount, const int* rIdx, int rowsCount, int cooBaseIndex, int* inverse)
{
    if (rowsCount <= 0 || nonZerosCount <= 0)
        return SPGPU_SUCCESS;
    if (!inverse)
        return SPGPU_UNSPECIFIED;
    hipStream_t s = handle->currentStream;
    hipLaunchKernelGGL(invertOrderKernel, dim3(gridOverOe(rowsCount)), dim3(kOeThreads), 0, s, inverse, rIdx, rowsCount);
    hipLaunchKernelGGL(permuteCooRowsKernel, dim3(gridOverOe(nonZerosCount)), dim3(kOeThreads), 0, s, dstCooRowIndices,
                       srcCooRowIndices, nonZerosCount, (const int*)inverse, rowsCount, cooBaseIndex);
    spgpuDebugCheck(handle, "cooPermuteRows");
    return SPGPU_SUCCESS;
}

} // extern "C"
