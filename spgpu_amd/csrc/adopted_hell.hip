/*
 * ADOPTED matrices: spgpuHellSpmvAdopt (include/spgpu/tuning.h) -- a HELL matrix whose rows come as they are (no rIdx) and are
 * RAGGED.  The reference's answer to such a matrix is the caller's: order the rows by length (ellToOell, ell.c:85-202) and hand
 * the permutation to spgpu?hellspmv as rIdx, as hellPerf.cpp:324-378 does; its plain kernel takes whatever HELL it is given
 * (hell_spmv_base_template.cuh:112-225) and pays for the padding -- on the north_star target 4.99 slots per nonzero and, here,
 * 2.7 ms against 0.70 for the ordered matrix, a bound of the FORMAT (>= 13 GB fetched), not of a kernel.
 *
 * For a caller who will not order the matrix himself but can promise that he will not touch ANY of its arrays until
 * spgpuSpmvThaw -- coefficients included -- the library does it: Adopt computes the aligned order on the device
 * (spgpuOellOrderAlignedDevice, windows of 2 048 rows, rows longer than 256 set aside: the order bench.py's target uses), lays
 * the matrix out again as HELL in that order in memory of its own (slots per nonzero 4.99 -> 1.08 on the target), freezes
 * that copy (plan + 16-bit column indices, planned_spmv.hip), and from then on a spgpu?hellspmv call on the caller's arrays with
 * rIdx == NULL runs the ordered, planned, packed kernel on the copy and writes z through the copy's rIdx: z in the caller's row
 * order, as ever.  The value of every z[i] is the ordered kernel's sum of the same products (another order of additions
 * than the plain kernel's: equal within rounding, bit-identical to what the caller would get by ordering the matrix himself with
 * the same calls).  Device memory: the ordered matrix (12 bytes per slot in fp64) + 8 bytes per row + the frozen plan.
 *
 * Roofline: the SpMV is raggedSpmvKernel's (HBM-bound); Adopt itself is format construction (one radix sort of the rows, one
 * gather of the entries: 17 ms for the target's 320 M entries), not the hot path.
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/oell_device.h"
#include "spgpu/tuning.h"

#include <string.h>

namespace spgpu {

constexpr int kAdoptThreads = 256;

/* depth of every hack of the ordered matrix: the longest of its rows */
__global__ __launch_bounds__(kAdoptThreads) void adoptHackDepthsKernel(int* __restrict__ depths, const int* __restrict__ lengths, int rows, int hackSize)
{
    const long long hack = (long long)blockIdx.x * (kAdoptThreads / kWave) + (threadIdx.x >> 6);
    const long long hacks = ((long long)rows + hackSize - 1) / hackSize;
    if (hack >= hacks)
        return;
    const int lane = threadIdx.x & (kWave - 1);
    int longest = 0;
    for (long long r = hack * hackSize + lane; r < (hack + 1) * hackSize && r < rows; r += kWave)
        longest = lengths[r] > longest ? lengths[r] : longest;
    longest = waveMax(longest);
    if (lane == 0)
        depths[hack] = longest;
}

/* hackOffsets[h] = hackSize * (depths[0] + ... + depths[h - 1]) (hell.c:64-75,100: no trailing total); the total slot count to
 * total[0].  One workgroup (312 500 hacks for 10 M rows: 306 rounds of 1 024). */
__global__ __launch_bounds__(1024) void adoptHackOffsetsKernel(int* __restrict__ hackOffsets, const int* __restrict__ depths, long long hacks, int hackSize,
                                                             unsigned long long* total)
{
    constexpr int BLOCK = 1024, WAVES = BLOCK / kWave;
    __shared__ unsigned long long waveTotals[WAVES];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry = 0ull;
    __syncthreads();
    for (long long base = 0; base < hacks; base += BLOCK) {
        const long long h = base + threadIdx.x;
        const unsigned long long mine = h < hacks ? (unsigned long long)depths[h] * (unsigned long long)hackSize : 0ull;
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned long long below = __shfl_up(incl, d, kWave);
            incl += lane >= d ? below : 0ull;
        }
        if (lane == kWave - 1)
            waveTotals[wave] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w)
            before += waveTotals[w];
        if (h < hacks)
            hackOffsets[h] = (int)(unsigned)(before + incl - mine);
        __syncthreads();
        if (threadIdx.x == BLOCK - 1)
            carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        total[0] = carry;
}

/* the entries of row order[i] of the caller's matrix become row i of the copy, slab column for slab column */
/* (source HELL: hackOffsets != NULL, slot of (row r, k) = hackOffsets[r / hs] + r % hs + k * hs; source ELL: r + k * pitch, with one pitch
 * for the coefficients and one for the indices (ell.h:46-61)) */
template <typename Raw>
__global__ __launch_bounds__(kAdoptThreads) void adoptCopyKernel(Raw* __restrict__ values, int* __restrict__ indices, const int* __restrict__ hackOffsetsOrdered,
                                                               int copyHackSize, const int* __restrict__ order, const int* __restrict__ lengths,
                                                               const Raw* __restrict__ cM, const int* __restrict__ rP, const int* __restrict__ hackOffsets,
                                                               int hackSize, long long valPitch, long long idxPitch, int rows)
{
    const long long i = (long long)blockIdx.x * kAdoptThreads + threadIdx.x;
    if (i >= rows)
        return;
    const unsigned from = (unsigned)order[i], to = (unsigned)i, chs = (unsigned)copyHackSize;
    long long srcV, srcI, stepV, stepI;
    if (hackOffsets) {
        const unsigned hs = (unsigned)hackSize;
        srcV = srcI = (long long)((unsigned)hackOffsets[from / hs] + from % hs);
        stepV = stepI = hs;
    } else {
        srcV = srcI = from;
        stepV = valPitch;
        stepI = idxPitch;
    }
    long long dst = (long long)((unsigned)hackOffsetsOrdered[to / chs] + to % chs);
    const int len = lengths[i];
    for (int k = 0; k < len; ++k, srcV += stepV, srcI += stepI, dst += chs) {
        values[dst] = cM[srcV];
        indices[dst] = rP[srcI];
    }
}

template <typename Raw>
static void launchCopy(hipStream_t s, void* values, int* indices, const int* hackOffsetsOrdered, int copyHackSize, const int* order, const int* lengths,
                       const void* cM, const int* rP, const int* hackOffsets, int hackSize, long long valPitch, long long idxPitch, int rows)
{
    hipLaunchKernelGGL((adoptCopyKernel<Raw>), dim3((unsigned)(((long long)rows + kAdoptThreads - 1) / kAdoptThreads)), dim3(kAdoptThreads), 0, s,
                       static_cast<Raw*>(values), indices, hackOffsetsOrdered, copyHackSize, order, lengths, static_cast<const Raw*>(cM), rP, hackOffsets,
                       hackSize, valPitch, idxPitch, rows);
}

} // namespace spgpu

/* The caller's matrix: HELL (hackOffsets != NULL; pitches 0) or ELL (hackOffsets NULL, callerHackSize 0, the two pitches and maxNnz).
 * The copy is HELL with copyHackSize rows per hack (the caller's hack size; 32 for an ELL source). */
static int adoptMatrix(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int callerHackSize, const int* hackOffsets, const int* rS,
                       int rows, int baseIndex, long long valPitch, long long idxPitch, int maxNnz)
{
    using namespace spgpu;
    const size_t elem = spgpuSizeOf(type);
    if (elem != 4 && elem != 8 && elem != 16)
        return SPGPU_UNSPECIFIED;
    const int hackSize = hackOffsets ? callerHackSize : 32;
    if (hackSize % 32 != 0)
        return SPGPU_UNSUPPORTED; /* (the ordered kernels' 32-row sub-groups are whole parts of a hack) */
    hipStream_t stream = handle->currentStream;
    if (spgpuAdoptedFind(handle, stream, cM, rP, rS, hackOffsets, rows, callerHackSize, baseIndex, valPitch, idxPitch))
        return SPGPU_SUCCESS; /* already adopted */
    const long long hacks = ((long long)rows + hackSize - 1) / hackSize;
    int previous = 0;
    (void)hipGetDevice(&previous);
    (void)hipSetDevice(handle->device);
    SpgpuAdopted e{};
    void *work = nullptr, *depths = nullptr;
    unsigned long long* total = nullptr;
    const size_t workBytes = spgpuOellOrderWorkBytes(rows);
    bool ok = workBytes > 0 && hipMalloc((void**)&e.order, (size_t)rows * sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&e.lengths, (size_t)rows * sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&e.hackOffsetsOrdered, (size_t)hacks * sizeof(int)) == hipSuccess && hipMalloc(&work, workBytes) == hipSuccess &&
              hipMalloc(&depths, (size_t)hacks * sizeof(int) + 16) == hipSuccess;
    unsigned long long slots = 0;
    if (ok) {
        total = reinterpret_cast<unsigned long long*>(static_cast<char*>(depths) + ((size_t)hacks * sizeof(int) + 7) / 8 * 8);
        ok = spgpuOellOrderAlignedDevice(handle, e.order, e.lengths, rS, rows, 2048, 256, work) == SPGPU_SUCCESS;
    }
    if (ok) {
        hipLaunchKernelGGL(adoptHackDepthsKernel, dim3((unsigned)((hacks + kAdoptThreads / kWave - 1) / (kAdoptThreads / kWave))), dim3(kAdoptThreads), 0, stream,
                           static_cast<int*>(depths), e.lengths, rows, hackSize);
        hipLaunchKernelGGL(adoptHackOffsetsKernel, dim3(1), dim3(1024), 0, stream, e.hackOffsetsOrdered, static_cast<const int*>(depths), hacks, hackSize, total);
        ok = hipMemcpyAsync(&slots, total, sizeof(slots), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    }
    ok = ok && slots > 0 && slots < 0x7FFFFFFFull; /* hackOffsets is an int array */
    if (ok) {
        /* what the caller's layout stores: its last hack's offset + hackSize x that hack's depth.  A matrix whose rows are about equally
         * long gains nothing from another order (and would pay the ordered kernel's row indirection): not adopted -- Freeze is the
         * call for it. */
        unsigned long long callerSlots = 0;
        if (hackOffsets) {
            int lastOffset = 0, lastDepth = 0;
            hipLaunchKernelGGL(adoptHackDepthsKernel, dim3((unsigned)((hacks + kAdoptThreads / kWave - 1) / (kAdoptThreads / kWave))), dim3(kAdoptThreads), 0,
                               stream, static_cast<int*>(depths), rS, rows, hackSize);
            ok = hipMemcpyAsync(&lastDepth, static_cast<int*>(depths) + (hacks - 1), sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipMemcpyAsync(&lastOffset, hackOffsets + (hacks - 1), sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
            callerSlots = (unsigned long long)(unsigned)lastOffset + (unsigned long long)hackSize * (unsigned)lastDepth;
        } else {
            callerSlots = (unsigned long long)rows * (unsigned long long)(maxNnz > 0 ? maxNnz : 0); /* ELL: every row as long as the longest */
        }
        ok = ok && callerSlots * 4ull >= slots * 5ull; /* at least 1.25 x the ordered matrix' slots */
    }
    if (ok)
        ok = hipMalloc(&e.values, (size_t)slots * elem) == hipSuccess && hipMalloc((void**)&e.indices, (size_t)slots * sizeof(int)) == hipSuccess;
    if (ok) {
        if (spgpuTuning()->poisonScratch) { /* testing: the padding slots of the copy are never used */
            (void)hipMemsetAsync(e.values, 0xFF, (size_t)slots * elem, stream);
            (void)hipMemsetAsync(e.indices, 0xFF, (size_t)slots * sizeof(int), stream);
        }
        if (elem == 4)
            launchCopy<uint32_t>(stream, e.values, e.indices, e.hackOffsetsOrdered, hackSize, e.order, e.lengths, cM, rP, hackOffsets, callerHackSize, valPitch, idxPitch, rows);
        else if (elem == 8)
            launchCopy<unsigned long long>(stream, e.values, e.indices, e.hackOffsetsOrdered, hackSize, e.order, e.lengths, cM, rP, hackOffsets, callerHackSize, valPitch, idxPitch, rows);
        else
            launchCopy<RawBits<16>::type>(stream, e.values, e.indices, e.hackOffsetsOrdered, hackSize, e.order, e.lengths, cM, rP, hackOffsets, callerHackSize, valPitch, idxPitch, rows);
        ok = hipStreamSynchronize(stream) == hipSuccess;
    }
    if (work)
        (void)hipFree(work);
    if (depths)
        (void)hipFree(depths);
    (void)hipSetDevice(previous);
    if (ok) {
        e.cM = cM;
        e.rP = rP;
        e.rS = rS;
        e.hackOffsets = hackOffsets;
        e.rows = rows;
        e.hackSize = callerHackSize;
        e.valPitch = valPitch;
        e.idxPitch = idxPitch;
        e.baseIndex = baseIndex;
        e.type = (int)type;
        e.bytes = (long long)((size_t)slots * (elem + sizeof(int)) + (size_t)rows * 2 * sizeof(int) + (size_t)hacks * sizeof(int));
        /* the copy frozen: plan + 16-bit indices (complex fp64: the plan alone -- SPGPU_UNSUPPORTED from Freeze is no failure here) */
        if (spgpuHellSpmvFreeze(handle, type, e.values, e.indices, hackSize, e.hackOffsetsOrdered, e.lengths, e.order, rows, baseIndex) != SPGPU_SUCCESS)
            (void)spgpuHellSpmvPrepare(handle, type, e.values, e.indices, hackSize, e.hackOffsetsOrdered, e.lengths, e.order, rows, baseIndex);
        ok = spgpuAdoptedAdd(handle, &e) == SPGPU_SUCCESS;
        if (!ok)
            (void)spgpuSpmvThaw(handle, e.indices);
    }
    if (!ok) {
        (void)hipGetLastError();
        (void)hipFree(e.values);
        (void)hipFree(e.indices);
        (void)hipFree(e.hackOffsetsOrdered);
        (void)hipFree(e.lengths);
        (void)hipFree(e.order);
        return SPGPU_UNSUPPORTED;
    }
    return SPGPU_SUCCESS;
}

extern "C" int spgpuHellSpmvAdopt(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets,
                                  const int* rS, int rows, int baseIndex)
{
    if (!handle || !cM || !rP || !hackOffsets || !rS || rows <= 0 || hackSize <= 0)
        return SPGPU_UNSPECIFIED;
    return adoptMatrix(handle, type, cM, rP, hackSize, hackOffsets, rS, rows, baseIndex, 0, 0, 0);
}

/* The ELL flavour: the ordered copy is HELL (hack size 32) -- the memory argument of BASELINE configs[2] inside the library: a ragged ELL
 * matrix stores rows x maxNnzPerRow slots, its ordered HELL copy ~1.1 per nonzero.  rS is needed (without row lengths ELL has no
 * ragged rows to order: SPGPU_UNSUPPORTED). */
extern "C" int spgpuEllSpmvAdopt(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS,
                                 int maxNnzPerRow, int rows, int baseIndex)
{
    if (!handle || !cM || !rP || rows <= 0 || cMPitch < rows || rPPitch < rows || maxNnzPerRow < 0)
        return SPGPU_UNSPECIFIED;
    if (!rS)
        return SPGPU_UNSUPPORTED;
    return adoptMatrix(handle, type, cM, rP, 0, nullptr, rS, rows, baseIndex, cMPitch, rPPitch, maxNnzPerRow);
}

/* One call for a solver that will multiply by this matrix many times and touch none of its arrays in between (Adopt's promise, which
 * includes Freeze's): the ordered copy if the rows are ragged and come without an order, else the 16-bit index copy if the columns
 * allow one, else nothing -- says which. */
extern "C" int spgpuHellSpmvOptimize(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets,
                                     const int* rS, const int* rIdx, int rows, int baseIndex)
{
    if (!rIdx && spgpuHellSpmvAdopt(handle, type, cM, rP, hackSize, hackOffsets, rS, rows, baseIndex) == SPGPU_SUCCESS)
        return SPGPU_SPMV_ADOPTED;
    if (spgpuHellSpmvFreeze(handle, type, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, baseIndex) == SPGPU_SUCCESS)
        return SPGPU_SPMV_FROZEN;
    return SPGPU_SPMV_AS_IS;
}
