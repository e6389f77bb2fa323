/*
 * Device-side COO -> ELL / HELL construction for gfx950 (include/spgpu/convert_device.h).
 * New functionality; the arrays it produces are byte-identical to the host
 * converters' (spgpu_amd/csrc/conv_ell.c, conv_hell.c; reference ell.c:5-80, hell.c:4-104).
 *
 * Scratch layout (ints):  misc[16] | rowStart[rows+1] | cursor[rows] | scanTotals[tiles+2] | bucket[nnz]
 *   rowStart  exclusive scan of the row lengths
 *   bucket    COO entry ids grouped by row (order inside a row arbitrary: filled with an atomic cursor)
 *   misc[0]   longest row / total slots, misc[1] out-of-range flag
 * The k-th position of entry e inside its row is #{e' in the row's bucket : e' < e}: exact whatever order
 * the atomics produced, so the result does not depend on scheduling.
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/convert_device.h"

namespace spgpu {

constexpr int kCvThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kCvThreads * kScanPerThread;

struct ConvertWork {
    int* rowStart;
    int* cursor;
    int* bucket;
    int* scanTotals;
    int* misc;
};

static size_t scanBlocks(long long n) { return (size_t)((n + kScanTile - 1) / kScanTile); }

static ConvertWork carve(void* work, int rows, int nnz)
{
    (void)nnz; /* the bucket comes last, so every other area has a position that depends on rows only */
    ConvertWork w;
    int* p = static_cast<int*>(work);
    w.misc = p;
    p += 16;
    w.rowStart = p;
    p += (size_t)rows + 1;
    w.cursor = p;
    p += (size_t)rows;
    w.scanTotals = p;
    p += scanBlocks((long long)rows + 1) + 2;
    w.bucket = p;
    return w;
}

static unsigned gridFor(long long n)
{
    const long long blocks = (n + kCvThreads - 1) / kCvThreads;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1048576 ? 1048576 : blocks));
}

__global__ __launch_bounds__(kCvThreads) void histogramKernel(int* rowLengths, int rows, int nnz, const int* cooRows,
                                                              int base, int* misc)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long e = (long long)blockIdx.x * kCvThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base;
        if (r < 0 || r >= rows)
            misc[1] = 1; /* out-of-range row: reported to the host, entry skipped */
        else
            atomicAdd(&rowLengths[r], 1);
    }
}

__global__ __launch_bounds__(kCvThreads) void maxKernel(const int* values, long long n, int scale, int* misc)
{
    int best = 0;
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long i = (long long)blockIdx.x * kCvThreads + threadIdx.x; i < n; i += stride)
        best = values[i] > best ? values[i] : best;
    best = waveMax(best);
    if ((threadIdx.x & (kWave - 1)) == 0 && best > 0)
        atomicMax(&misc[0], best * scale);
}

/* ---- exclusive scan in three launches: tile scans, scan of the tile totals, add-back ---- */
__device__ inline int blockExclusiveScan(int value, int* ldsWaveTotals, int* blockTotal)
{
    /* inclusive scan inside the wavefront with lane shuffles */
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int incl = value;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int up = __shfl_up(incl, d, kWave);
        if (lane >= d)
            incl += up;
    }
    if (lane == kWave - 1)
        ldsWaveTotals[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kCvThreads / kWave; ++w) {
        if (w < wave)
            before += ldsWaveTotals[w];
        total += ldsWaveTotals[w];
    }
    __syncthreads();
    *blockTotal = total;
    return before + incl - value;
}

/* out[i] = sum_{j<i} in[j]*scale within the tile; tile total to totals[tile].  in == NULL scans totals in place. */
__global__ __launch_bounds__(kCvThreads) void scanTilesKernel(int* out, const int* in, long long n, int scale, int* totals)
{
    __shared__ int waveTotals[kCvThreads / kWave];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanPerThread;
    int v[kScanPerThread], mine = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        v[j] = base + j < n ? in[base + j] * scale : 0;
        mine += v[j];
    }
    int total;
    int run = blockExclusiveScan(mine, waveTotals, &total);
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        if (base + j < n)
            out[base + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 0)
        totals[blockIdx.x] = total;
}

/* single workgroup: exclusive scan of `count` tile totals in place, grand total to totals[count] */
__global__ __launch_bounds__(kCvThreads) void scanTotalsKernel(int* totals, long long count)
{
    __shared__ int waveTotals[kCvThreads / kWave];
    int carry = 0;
    for (long long first = 0; first < count; first += kCvThreads) {
        const long long i = first + threadIdx.x;
        const int v = i < count ? totals[i] : 0;
        int total;
        const int excl = blockExclusiveScan(v, waveTotals, &total);
        if (i < count)
            totals[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0)
        totals[count] = carry;
}

__global__ __launch_bounds__(kCvThreads) void addTotalsKernel(int* out, long long n, const int* totals)
{
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanPerThread;
    const int add = totals[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j)
        if (base + j < n)
            out[base + j] += add;
}

/* out[0..n) = exclusive scan of in[0..n)*scale; the grand total lands in totals[tiles] (device). */
static void exclusiveScan(hipStream_t s, int* out, const int* in, long long n, int scale, int* totals)
{
    const size_t tiles = scanBlocks(n);
    hipLaunchKernelGGL(scanTilesKernel, dim3((unsigned)tiles), dim3(kCvThreads), 0, s, out, in, n, scale, totals);
    hipLaunchKernelGGL(scanTotalsKernel, dim3(1), dim3(kCvThreads), 0, s, totals, (long long)tiles);
    hipLaunchKernelGGL(addTotalsKernel, dim3((unsigned)tiles), dim3(kCvThreads), 0, s, out, n, totals);
}

__global__ __launch_bounds__(kCvThreads) void bucketKernel(int* bucket, int* cursor, const int* rowStart, int rows, int nnz,
                                                           const int* cooRows, int base)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long e = (long long)blockIdx.x * kCvThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base;
        if (r >= 0 && r < rows)
            bucket[rowStart[r] + atomicAdd(&cursor[r], 1)] = (int)e;
    }
}

/* Longest row of every hack (hell.c:71-88). */
__global__ __launch_bounds__(kCvThreads) void hackDepthKernel(int* depth, int hacks, int hackSize, int rows, const int* rowLengths)
{
    const long long h = (long long)blockIdx.x * kCvThreads + threadIdx.x;
    if (h >= hacks)
        return;
    int longest = 0;
    for (int j = 0; j < hackSize; ++j) {
        const long long r = h * hackSize + j;
        if (r >= rows)
            break;
        longest = rowLengths[r] > longest ? rowLengths[r] : longest;
    }
    depth[h] = longest;
}

/* One thread per bucket position: recover k = rank of the entry inside its row, then place it. */
template <typename ELEM, bool TO_HELL>
__global__ __launch_bounds__(kCvThreads) void placeKernel(ELEM* values, int* indices, long long valStride, long long idxStride,
                                                          const int* hackOffsets, int hackSize, int outBase, int rows,
                                                          int nnz, const int* cooRows, const int* cooCols, const ELEM* cooVals,
                                                          int cooBase, const int* rowStart, const int* bucket)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long p = (long long)blockIdx.x * kCvThreads + threadIdx.x; p < nnz; p += stride) {
        if (p >= rowStart[rows])
            continue; /* entries with out-of-range rows were never bucketed */
        const int e = bucket[p];
        const int r = cooRows[e] - cooBase;
        int k = 0;
        for (int q = rowStart[r]; q < rowStart[r + 1]; ++q)
            k += bucket[q] < e;
        long long slot;
        if constexpr (TO_HELL) {
            const int hack = r / hackSize;
            slot = (long long)hackOffsets[hack] + (r - hack * hackSize) + (long long)k * hackSize;
            indices[slot] = cooCols[e] - cooBase + outBase;
            values[slot] = cooVals[e];
        } else {
            indices[r + (long long)k * idxStride] = cooCols[e] - cooBase + outBase;
            values[r + (long long)k * valStride] = cooVals[e];
        }
    }
}

struct Bits128 { unsigned long long lo, hi; };

template <bool TO_HELL>
static spgpuStatus_t place(spgpuHandle_t handle, void* values, int* indices, long long valStride, long long idxStride,
                           const int* hackOffsets, int hackSize, int outBase, int rows, int nnz, const int* cooRows,
                           const int* cooCols, const void* cooVals, int cooBase, spgpuType_t type, const ConvertWork& w)
{
    if (nnz <= 0 || rows <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const dim3 grid(gridFor(nnz)), block(kCvThreads);
    switch (spgpuSizeOf(type)) {
    case 4:
        hipLaunchKernelGGL((placeKernel<unsigned, TO_HELL>), grid, block, 0, s, static_cast<unsigned*>(values), indices, valStride,
                           idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooRows, cooCols,
                           static_cast<const unsigned*>(cooVals), cooBase, w.rowStart, w.bucket);
        break;
    case 8:
        hipLaunchKernelGGL((placeKernel<unsigned long long, TO_HELL>), grid, block, 0, s, static_cast<unsigned long long*>(values),
                           indices, valStride, idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooRows, cooCols,
                           static_cast<const unsigned long long*>(cooVals), cooBase, w.rowStart, w.bucket);
        break;
    case 16:
        hipLaunchKernelGGL((placeKernel<Bits128, TO_HELL>), grid, block, 0, s, static_cast<Bits128*>(values), indices, valStride,
                           idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooRows, cooCols,
                           static_cast<const Bits128*>(cooVals), cooBase, w.rowStart, w.bucket);
        break;
    default:
        return SPGPU_UNSUPPORTED;
    }
    spgpuDebugCheck(handle, "coo conversion");
    return SPGPU_SUCCESS;
}

/* ---- COO -> DIA ----------------------------------------------------------------------------------------------
 * Scratch layout (ints): present[span] | slotOf[span] | scanTotals[tiles+2], span = rows + cols - 1 possible diagonals.
 * present[d] = 1 if some entry has (column - row) = d - (rows - 1); slotOf = its exclusive scan = the position of
 * that diagonal among the stored ones (ascending offset, as dia.c:70-84). */
__global__ __launch_bounds__(kCvThreads) void markDiagonalsKernel(int* present, int* misc, int rows, int cols, int nnz,
                                                                  const int* cooRows, const int* cooCols, int base)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long e = (long long)blockIdx.x * kCvThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base, c = cooCols[e] - base;
        if (r < 0 || r >= rows || c < 0 || c >= cols)
            misc[1] = 1;
        else
            present[(long long)(rows - 1) + c - r] = 1;
    }
}

__global__ __launch_bounds__(kCvThreads) void diaOffsetsKernel(int* offsets, const int* present, const int* slotOf,
                                                               long long span, int rows)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long d = (long long)blockIdx.x * kCvThreads + threadIdx.x; d < span; d += stride)
        if (present[d])
            offsets[slotOf[d]] = (int)(d - (rows - 1));
}

/* owner[slot] = 1 + the largest COO position that maps to the slot (the reference's in-order memcpy: last one wins) */
__global__ __launch_bounds__(kCvThreads) void diaClaimKernel(int* owner, const int* slotOf, int rows, int pitch, int nnz,
                                                             const int* cooRows, const int* cooCols, int base)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long e = (long long)blockIdx.x * kCvThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base, c = cooCols[e] - base;
        atomicMax(&owner[(long long)slotOf[(long long)(rows - 1) + c - r] * pitch + r], (int)e + 1);
    }
}

template <typename E>
__global__ __launch_bounds__(kCvThreads) void diaWriteKernel(E* values, const int* owner, const int* slotOf, int rows, int pitch,
                                                             int nnz, const int* cooRows, const int* cooCols,
                                                             const E* cooValues, int base)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long e = (long long)blockIdx.x * kCvThreads + threadIdx.x; e < nnz; e += stride) {
        const int r = cooRows[e] - base, c = cooCols[e] - base;
        const long long slot = (long long)slotOf[(long long)(rows - 1) + c - r] * pitch + r;
        if (owner[slot] == (int)e + 1)
            values[slot] = cooValues[e];
    }
}

} // namespace spgpu

using namespace spgpu;

extern "C" {

size_t spgpuCooConvertWorkBytes(int rowsCount, int nonZerosCount)
{
    const size_t rows = rowsCount > 0 ? (size_t)rowsCount : 0, nnz = nonZerosCount > 0 ? (size_t)nonZerosCount : 0;
    return (16 + 2 * rows + 1 + scanBlocks((long long)rows + 1) + 2 + nnz) * sizeof(int);
}

spgpuStatus_t spgpuCooRowLengthsDevice(spgpuHandle_t handle, int* rowLengths, int* maxRowSize, int rowsCount,
                                       int nonZerosCount, const int* cooRowIndices, int cooBaseIndex, void* work)
{
    *maxRowSize = 0;
    if (rowsCount <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const ConvertWork w = carve(work, rowsCount, nonZerosCount);
    (void)hipMemsetAsync(rowLengths, 0, (size_t)rowsCount * sizeof(int), s);
    (void)hipMemsetAsync(w.cursor, 0, (size_t)rowsCount * sizeof(int), s);
    (void)hipMemsetAsync(w.misc, 0, 16 * sizeof(int), s);
    if (nonZerosCount > 0)
        hipLaunchKernelGGL(histogramKernel, dim3(gridFor(nonZerosCount)), dim3(kCvThreads), 0, s, rowLengths, rowsCount,
                           nonZerosCount, cooRowIndices, cooBaseIndex, w.misc);
    hipLaunchKernelGGL(maxKernel, dim3(gridFor(rowsCount)), dim3(kCvThreads), 0, s, rowLengths, (long long)rowsCount, 1, w.misc);
    /* rowStart[0..rows] = exclusive scan of the lengths (rowStart[rows] = bucketed entries) */
    (void)hipMemsetAsync(w.rowStart + rowsCount, 0, sizeof(int), s);
    exclusiveScan(s, w.rowStart, rowLengths, (long long)rowsCount, 1, w.scanTotals);
    (void)hipMemcpyAsync(w.rowStart + rowsCount, w.scanTotals + scanBlocks(rowsCount), sizeof(int), hipMemcpyDeviceToDevice, s);
    if (nonZerosCount > 0)
        hipLaunchKernelGGL(bucketKernel, dim3(gridFor(nonZerosCount)), dim3(kCvThreads), 0, s, w.bucket, w.cursor, w.rowStart,
                           rowsCount, nonZerosCount, cooRowIndices, cooBaseIndex);
    int* host = static_cast<int*>(spgpuPrivate(handle)->reduceHost);
    (void)hipMemcpyAsync(host, w.misc, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    *maxRowSize = host[0];
    spgpuDebugCheck(handle, "coo row lengths");
    return host[1] ? SPGPU_UNSUPPORTED : SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCooToEllDevice(spgpuHandle_t handle, void* ellValues, int* ellIndices, int ellValuesPitch,
                                  int ellIndicesPitch, int ellBaseIndex, int rowsCount, int nonZerosCount,
                                  const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
                                  int cooBaseIndex, spgpuType_t valuesType, const int* rowLengths, void* work)
{
    (void)rowLengths;
    const ConvertWork w = carve(work, rowsCount, nonZerosCount);
    return place<false>(handle, ellValues, ellIndices, ellValuesPitch, ellIndicesPitch, nullptr, 0, ellBaseIndex, rowsCount,
                        nonZerosCount, cooRowIndices, cooColsIndices, cooValues, cooBaseIndex, valuesType, w);
}

spgpuStatus_t spgpuHellPlanDevice(spgpuHandle_t handle, int* allocationHeight, int* hackOffsets, int hackSize,
                                  int rowsCount, const int* rowLengths, void* work)
{
    *allocationHeight = 0;
    if (rowsCount <= 0 || hackSize <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const int hacks = (rowsCount + hackSize - 1) / hackSize;
    /* the per-hack depths reuse the cursor area (free once the buckets are filled; hacks <= rows) */
    const ConvertWork w = carve(work, rowsCount, 0);
    int* depth = w.cursor;
    int* totals = w.scanTotals;
    hipLaunchKernelGGL(hackDepthKernel, dim3(gridFor(hacks)), dim3(kCvThreads), 0, s, depth, hacks, hackSize, rowsCount, rowLengths);
    exclusiveScan(s, hackOffsets, depth, (long long)hacks, hackSize, totals);
    int* host = static_cast<int*>(spgpuPrivate(handle)->reduceHost);
    (void)hipMemcpyAsync(host, totals + scanBlocks(hacks), sizeof(int), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    *allocationHeight = host[0] / hackSize;
    spgpuDebugCheck(handle, "hell plan");
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCooToHellDevice(spgpuHandle_t handle, void* hellValues, int* hellIndices, const int* hackOffsets,
                                   int hackSize, int hellBaseIndex, int rowsCount, int nonZerosCount,
                                   const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
                                   int cooBaseIndex, spgpuType_t valuesType, const int* rowLengths, void* work)
{
    (void)rowLengths;
    const ConvertWork w = carve(work, rowsCount, nonZerosCount);
    return place<true>(handle, hellValues, hellIndices, 0, 0, hackOffsets, hackSize, hellBaseIndex, rowsCount, nonZerosCount,
                       cooRowIndices, cooColsIndices, cooValues, cooBaseIndex, valuesType, w);
}

/* ---- COO -> DIA (device counterparts of computeDiaDiagonalsCount and coo2dia, dia.c:11-104) ---- */
static long long diaSpan(int rows, int cols) { return (long long)rows + cols - 1; }

size_t spgpuCooDiaWorkBytes(int rowsCount, int columnsCount)
{
    const long long span = diaSpan(rowsCount, columnsCount) > 0 ? diaSpan(rowsCount, columnsCount) : 1;
    return ((size_t)16 + 2 * (size_t)span + scanBlocks(span) + 2) * sizeof(int);
}

size_t spgpuCooToDiaScratchBytes(int valuesPitch, int diagonals)
{
    const size_t slots = (size_t)(valuesPitch > 0 ? valuesPitch : 0) * (size_t)(diagonals > 0 ? diagonals : 0);
    return (slots ? slots : 1) * sizeof(int);
}

spgpuStatus_t spgpuCooDiaPlanDevice(spgpuHandle_t handle, int* diagonals, int rowsCount, int columnsCount,
                                    int nonZerosCount, const int* cooRowIndices, const int* cooColsIndices,
                                    int cooBaseIndex, void* work)
{
    if (!handle || !diagonals || !work || rowsCount < 0 || columnsCount < 0 || nonZerosCount < 0)
        return SPGPU_UNSUPPORTED;
    *diagonals = 0;
    const long long span = diaSpan(rowsCount, columnsCount);
    if (span <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    int* misc = static_cast<int*>(work);
    int* present = misc + 16;
    int* slotOf = present + span;
    int* totals = slotOf + span;
    (void)hipMemsetAsync(work, 0, (16 + (size_t)span) * sizeof(int), s);
    if (nonZerosCount > 0)
        hipLaunchKernelGGL(markDiagonalsKernel, dim3(gridFor(nonZerosCount)), dim3(kCvThreads), 0, s, present, misc, rowsCount,
                           columnsCount, nonZerosCount, cooRowIndices, cooColsIndices, cooBaseIndex);
    exclusiveScan(s, slotOf, present, span, 1, totals);
    int* host = static_cast<int*>(spgpuPrivate(handle)->reduceHost);
    (void)hipMemcpyAsync(host, totals + scanBlocks(span), sizeof(int), hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(host + 1, misc + 1, sizeof(int), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (host[1])
        return SPGPU_UNSUPPORTED; /* an entry outside the matrix */
    *diagonals = host[0];
    spgpuDebugCheck(handle, "cooDiaPlanDevice");
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCooToDiaDevice(spgpuHandle_t handle, void* values, int* offsets, int valuesPitch, int diagonals,
                                  int rowsCount, int columnsCount, int nonZerosCount, const int* cooRowIndices,
                                  const int* cooColsIndices, const void* cooValues, int cooBaseIndex,
                                  spgpuType_t valuesType, const void* work, void* scratch)
{
    if (!handle || !work || !scratch || valuesPitch < rowsCount || diagonals < 0 || nonZerosCount < 0)
        return SPGPU_UNSUPPORTED;
    const long long span = diaSpan(rowsCount, columnsCount);
    if (span <= 0 || diagonals == 0 || nonZerosCount == 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const int* present = static_cast<const int*>(work) + 16;
    const int* slotOf = present + span;
    int* owner = static_cast<int*>(scratch);
    (void)hipMemsetAsync(owner, 0, spgpuCooToDiaScratchBytes(valuesPitch, diagonals), s);
    hipLaunchKernelGGL(diaOffsetsKernel, dim3(gridFor(span)), dim3(kCvThreads), 0, s, offsets, present, slotOf, span, rowsCount);
    const dim3 grid(gridFor(nonZerosCount)), block(kCvThreads);
    hipLaunchKernelGGL(diaClaimKernel, grid, block, 0, s, owner, slotOf, rowsCount, valuesPitch, nonZerosCount, cooRowIndices,
                       cooColsIndices, cooBaseIndex);
    switch (spgpuSizeOf(valuesType)) {
    case 4:
        hipLaunchKernelGGL(diaWriteKernel<uint32_t>, grid, block, 0, s, static_cast<uint32_t*>(values), owner, slotOf, rowsCount,
                           valuesPitch, nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const uint32_t*>(cooValues),
                           cooBaseIndex);
        break;
    case 8:
        hipLaunchKernelGGL(diaWriteKernel<uint64_t>, grid, block, 0, s, static_cast<uint64_t*>(values), owner, slotOf, rowsCount,
                           valuesPitch, nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const uint64_t*>(cooValues),
                           cooBaseIndex);
        break;
    case 16:
        hipLaunchKernelGGL(diaWriteKernel<ulonglong2>, grid, block, 0, s, static_cast<ulonglong2*>(values), owner, slotOf, rowsCount,
                           valuesPitch, nonZerosCount, cooRowIndices, cooColsIndices, static_cast<const ulonglong2*>(cooValues),
                           cooBaseIndex);
        break;
    default:
        return SPGPU_UNSUPPORTED;
    }
    spgpuDebugCheck(handle, "cooToDiaDevice");
    return SPGPU_SUCCESS;
}

} // extern "C"
