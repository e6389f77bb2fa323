/*
 * Device-side COO -> ELL / HELL construction for gfx950 (include/spgpu/convert_device.h).
 * New functionality; the arrays it produces are byte-identical to the host
 * converters' (spgpu_amd/csrc/conv_ell.c, conv_hell.c; reference ell.c:5-80, hell.c:4-104).
 *
 * Scratch layout (ints):  misc[16] | rowStart[rows+1] | depth[rows] | scanTotals[tiles+2] | rowOf[nnz] | entry[nnz] | rocPRIM temp
 *   entry     COO entry ids sorted by row, ascending inside a row: ONE stable radix sort of (row, entry id) pairs
 *             (rocPRIM; keys through a transform iterator, ids from a counting iterator, so no input copy is made)
 *   rowOf     the sorted row numbers; an entry outside [0, rows) gets the key `rows` and sorts behind all rows
 *   rowStart  rowStart[r] = first position whose row is >= r (binary search in rowOf); rowStart[rows] = valid entries
 *   misc[0]   longest row, misc[1] out-of-range flag
 * The k-th position of the entry at sorted position p is p - rowStart[row]: the reference's encounter order, with no
 * atomics anywhere (round 1 bucketed with an atomic cursor and recovered k by counting smaller ids in the bucket --
 * quadratic in the row length, minutes for a row of 10^5 entries).
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/convert_device.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace spgpu {

constexpr int kCvThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kCvThreads * kScanPerThread;

struct ConvertWork {
    int* rowStart;
    int* cursor; /* per-hack depths of the HELL plan */
    int* scanTotals;
    int* misc;
    unsigned* rowOf;
    int* bucket; /* entry ids sorted by row */
    void* temp;
};

static size_t scanBlocks(long long n) { return (size_t)((n + kScanTile - 1) / kScanTile); }
static size_t alignUpCv(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* zero-based row of a COO entry as the sort key; anything outside the matrix becomes `rows` */
struct RowKeyOf {
    int base, rows;
    __host__ __device__ unsigned operator()(int stored) const
    {
        const unsigned r = (unsigned)(stored - base);
        return r < (unsigned)rows ? r : (unsigned)rows;
    }
};
typedef rocprim::transform_iterator<const int*, RowKeyOf, unsigned> RowKeyIterator;

static hipError_t sortTempBytes(size_t nnz, size_t* bytes)
{
    size_t need = 0;
    const hipError_t err = rocprim::radix_sort_pairs(nullptr, need, RowKeyIterator((const int*)nullptr, RowKeyOf{0, 1}), (unsigned*)nullptr,
                                                     rocprim::counting_iterator<int>(0), (int*)nullptr, nnz);
    *bytes = alignUpCv(need, 256);
    return err;
}

static size_t fixedInts(int rows) { return alignUpCv((size_t)16 + 2 * (size_t)rows + 1 + scanBlocks((long long)rows + 1) + 2, 64); }

static ConvertWork carve(void* work, int rows, int nnz)
{
    ConvertWork w;
    int* p = static_cast<int*>(work);
    w.misc = p;
    p += 16;
    w.rowStart = p;
    p += (size_t)rows + 1;
    w.cursor = p;
    p += (size_t)rows;
    w.scanTotals = p;
    p = static_cast<int*>(work) + fixedInts(rows); /* what follows depends on nnz; the HELL plan only uses what precedes */
    w.rowOf = reinterpret_cast<unsigned*>(p);
    p += alignUpCv((size_t)(nnz > 0 ? nnz : 0), 64);
    w.bucket = p;
    p += alignUpCv((size_t)(nnz > 0 ? nnz : 0), 64);
    w.temp = p;
    return w;
}

static unsigned gridFor(long long n)
{
    const long long blocks = (n + kCvThreads - 1) / kCvThreads;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1048576 ? 1048576 : blocks));
}

/* rowStart[r] = first sorted position whose row is >= r, r = 0 .. rows */
__global__ __launch_bounds__(kCvThreads) void rowStartKernel(int* rowStart, int rows, const unsigned* rowOf, int nnz)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long r = (long long)blockIdx.x * kCvThreads + threadIdx.x; r <= rows; r += stride) {
        int lo = 0, hi = nnz;
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (rowOf[mid] < (unsigned)r)
                lo = mid + 1;
            else
                hi = mid;
        }
        rowStart[r] = lo;
    }
}

__global__ __launch_bounds__(kCvThreads) void rowLengthsKernel(int* rowLengths, const int* rowStart, int rows, int nnz, int* misc)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    for (long long r = (long long)blockIdx.x * kCvThreads + threadIdx.x; r < rows; r += stride)
        rowLengths[r] = rowStart[r + 1] - rowStart[r];
    if (blockIdx.x == 0 && threadIdx.x == 0 && rowStart[rows] < nnz)
        misc[1] = 1; /* entries outside the matrix: reported to the host, the entries skipped */
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

/* One thread per sorted position: the entry, its row, and its place inside the row. */
template <typename ELEM, bool TO_HELL>
__global__ __launch_bounds__(kCvThreads) void placeKernel(ELEM* values, int* indices, long long valStride, long long idxStride,
                                                          const int* hackOffsets, int hackSize, int outBase, int rows,
                                                          int nnz, const int* cooCols, const ELEM* cooVals, int cooBase,
                                                          const int* rowStart, const unsigned* rowOf, const int* bucket)
{
    const long long stride = (long long)gridDim.x * kCvThreads;
    const int valid = rowStart[rows];
    for (long long p = (long long)blockIdx.x * kCvThreads + threadIdx.x; p < valid; p += stride) {
        const int e = bucket[p];
        const int r = (int)rowOf[p];
        const int k = (int)p - rowStart[r];
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
    (void)cooRows; /* the sorted row numbers are in the scratch */
    if (nnz <= 0 || rows <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const dim3 grid(gridFor(nnz)), block(kCvThreads);
    switch (spgpuSizeOf(type)) {
    case 4:
        hipLaunchKernelGGL((placeKernel<unsigned, TO_HELL>), grid, block, 0, s, static_cast<unsigned*>(values), indices, valStride,
                           idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooCols,
                           static_cast<const unsigned*>(cooVals), cooBase, w.rowStart, w.rowOf, w.bucket);
        break;
    case 8:
        hipLaunchKernelGGL((placeKernel<unsigned long long, TO_HELL>), grid, block, 0, s, static_cast<unsigned long long*>(values),
                           indices, valStride, idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooCols,
                           static_cast<const unsigned long long*>(cooVals), cooBase, w.rowStart, w.rowOf, w.bucket);
        break;
    case 16:
        hipLaunchKernelGGL((placeKernel<Bits128, TO_HELL>), grid, block, 0, s, static_cast<Bits128*>(values), indices, valStride,
                           idxStride, hackOffsets, hackSize, outBase, rows, nnz, cooCols,
                           static_cast<const Bits128*>(cooVals), cooBase, w.rowStart, w.rowOf, w.bucket);
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
    const int rows = rowsCount > 0 ? rowsCount : 0;
    const size_t nnz = nonZerosCount > 0 ? (size_t)nonZerosCount : 0;
    size_t temp = 0;
    if (nnz > 0 && sortTempBytes(nnz, &temp) != hipSuccess)
        return 0; /* no GPU to ask for rocPRIM's share */
    return (fixedInts(rows) + 2 * alignUpCv(nnz, 64)) * sizeof(int) + temp + 256;
}

spgpuStatus_t spgpuCooRowLengthsDevice(spgpuHandle_t handle, int* rowLengths, int* maxRowSize, int rowsCount,
                                       int nonZerosCount, const int* cooRowIndices, int cooBaseIndex, void* work)
{
    *maxRowSize = 0;
    if (rowsCount <= 0)
        return SPGPU_SUCCESS;
    hipStream_t s = handle->currentStream;
    const ConvertWork w = carve(work, rowsCount, nonZerosCount);
    (void)hipMemsetAsync(w.misc, 0, 16 * sizeof(int), s);
    if (nonZerosCount > 0) {
        size_t bytes = 0;
        if (sortTempBytes((size_t)nonZerosCount, &bytes) != hipSuccess)
            return SPGPU_UNSPECIFIED;
        unsigned bits = 1; /* keys are 0 .. rows */
        while (bits < 32 && (1u << bits) <= (unsigned)rowsCount)
            ++bits;
        if (rocprim::radix_sort_pairs(w.temp, bytes, RowKeyIterator(cooRowIndices, RowKeyOf{cooBaseIndex, rowsCount}), w.rowOf,
                                      rocprim::counting_iterator<int>(0), w.bucket, (size_t)nonZerosCount, 0, bits, s) != hipSuccess)
            return SPGPU_UNSPECIFIED;
    }
    hipLaunchKernelGGL(rowStartKernel, dim3(gridFor((long long)rowsCount + 1)), dim3(kCvThreads), 0, s, w.rowStart, rowsCount,
                       (const unsigned*)w.rowOf, nonZerosCount > 0 ? nonZerosCount : 0);
    hipLaunchKernelGGL(rowLengthsKernel, dim3(gridFor(rowsCount)), dim3(kCvThreads), 0, s, rowLengths, (const int*)w.rowStart,
                       rowsCount, nonZerosCount > 0 ? nonZerosCount : 0, w.misc);
    hipLaunchKernelGGL(maxKernel, dim3(gridFor(rowsCount)), dim3(kCvThreads), 0, s, rowLengths, (long long)rowsCount, 1, w.misc);
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
