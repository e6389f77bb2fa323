/*
 * ELL / HELL SpMV with a row order (rIdx) and a per-matrix PLAN -- the north_star target: HELL fp64 on a 10 M-row power-law
 * matrix whose rows were ordered by length (ellToOell, reference ell.c:85-202; hellPerf.cpp:333-378 runs exactly this:
 * order once, then spgpu?hellspmv with rIdx thousands of times).  C ABI unchanged: spgpu{S,D,C,Z}{hell,ell}spmv
 * (include/spgpu/hell.h, ell.h); the plan is found by the arrays' addresses, nobody passes it.
 *
 * What the plan is and what it may and may not decide: spgpu_internal.h (SpgpuSpmvPlan).  This file holds
 *   - the analysis: planBlocksKernel (a workgroup per block of SUBS 32-row sub-groups: which of them are deeper than
 *     deepCap, and the window of columns the others reach) and planListKernel (one workgroup: the ascending list of deep
 *     sub-groups; their number goes to a pinned word).  Started on the SpMV's stream by the first call that sees the matrix,
 *     never waited for: the call itself, and every later one until the analysis' event has completed, runs the path
 *     without a plan (registration in the stream's deep list, two launches behind the main kernel: ellpack_spmv.hip).
 *   - the launch with a plan: raggedSpmvKernel<..., PLAN> (ragged_spmv.hip.h) over planMainBlocks + ceil(deep / G)
 *     workgroups -- ONE launch, no list, nothing behind it.
 * Both paths add every row's products in the same order (deep_rows.hip.h: chunksOf), so which of them a call takes -- first
 * call, settled, stale plan, captured graph (never planned: a graph outlives a plan) -- does not change a bit of z.
 *
 * Roofline: HBM bandwidth.  Algorithmic bytes as for every ELL/HELL SpMV (ellpack_spmv.hip) + 4 per row for rIdx; the plan
 * itself is 32 bytes per 1 024 or 2 048 rows.
 */
#include "numeric.hip.h"
#include "slab_args.hip.h"

#include <stdio.h>

namespace spgpu {

#ifdef SPGPU_TRACE_BLOCKS
static __device__ unsigned long long* spgpuTraceBuffer; /* this translation unit's copy (spgpuDebugSetTrace sets both) */
#endif
#include "ragged_spmv.hip.h"

/* ---- analysis ------------------------------------------------------------------------------------------------------ */

template <bool IS_HELL, int SUBS>
__global__ __launch_bounds__(256) void planBlocksKernel(const int* __restrict__ rP, const int* __restrict__ rS,
                                                       const int* __restrict__ hackOffsets, int hackSize, long long idxStride,
                                                       int maxNnz, int rows, int baseIndex, int deepCap, SpgpuPlanBlock* blocks,
                                                       int* counts)
{
    constexpr int BLOCK = 256, WAVES = BLOCK / kWave, ROWS = SUBS * 32, RPT = ROWS / BLOCK;
    static_assert(ROWS % BLOCK == 0 && SUBS <= 64, "whole rounds of 256 rows; a 64-bit mask");
    __shared__ unsigned maskWords[2];
    __shared__ ColumnProbe seen[WAVES];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long blockRow0 = (long long)blockIdx.x * ROWS;
    if (threadIdx.x < 2)
        maskWords[threadIdx.x] = 0u;
    int myLen[RPT];
    unsigned myBase[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const long long r = blockRow0 + threadIdx.x + j * BLOCK;
        myLen[j] = r < rows ? (rS ? rS[r] : maxNnz) : 0;
        myBase[j] = 0;
        if (r < rows) {
            if constexpr (IS_HELL) {
                const unsigned u = (unsigned)r, hs = (unsigned)hackSize;
                myBase[j] = (unsigned)hackOffsets[u / hs] + u % hs;
            } else {
                myBase[j] = (unsigned)r;
            }
        }
    }
    __syncthreads();
    /* the window of columns, by the rule of raggedSpmvKernel's own probes: first and last entry of every row, the mean of their
     * middles -- over the rows the block will walk itself, i.e. not those of its deep sub-groups */
    ColumnProbe mine{0x7fffffff, -0x7fffffff - 1, 0, 0};
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int i = threadIdx.x + j * BLOCK;
        const int depth = halfReduce(myLen[j], MaxOf{});
        const bool deep = depth > deepCap;
        if (deep && (lane & 31) == 0)
            atomicOr(&maskWords[(i >> 5) >> 5], 1u << ((i >> 5) & 31));
        if (!deep && myLen[j] > 0) {
            const int f = rP[(long long)myBase[j]] - baseIndex;
            const int l = rP[(long long)myBase[j] + (long long)(myLen[j] - 1) * idxStride] - baseIndex;
            const int low = f < l ? f : l, high = f < l ? l : f;
            mine.lowest = low < mine.lowest ? low : mine.lowest;
            mine.highest = high > mine.highest ? high : mine.highest;
            mine.middles += ((long long)f + l) >> 1;
            mine.rows += 1;
        }
    }
    mine.lowest = waveReduce(mine.lowest, MinOf{});
    mine.highest = waveReduce(mine.highest, MaxOf{});
    mine.rows = waveReduce(mine.rows, SumOf{});
    mine.middles = (long long)waveSumExact((double)mine.middles);
    if (lane == 0)
        seen[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        ColumnProbe all = seen[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            all.lowest = seen[w].lowest < all.lowest ? seen[w].lowest : all.lowest;
            all.highest = seen[w].highest > all.highest ? seen[w].highest : all.highest;
            all.rows += seen[w].rows;
            all.middles += seen[w].middles;
        }
        SpgpuPlanBlock out;
        out.lowest = all.lowest;
        out.highest = all.highest;
        out.middle = all.rows > 0 ? (int)(all.middles / all.rows) : 0;
        out.probed = all.rows;
        out.deepMask = (unsigned long long)maskWords[0] | ((unsigned long long)maskWords[1] << 32);
        out.firstDeep = 0;
        out.packBase = 0;
        blocks[blockIdx.x] = out;
        counts[blockIdx.x] = __popcll(out.deepMask);
    }
}

/* One workgroup: where each block's deep sub-groups start in the list, the list itself (ascending), and their number where the
 * host can read it once the stream's event says the analysis has finished. */
__global__ __launch_bounds__(1024) void planListKernel(SpgpuPlanBlock* blocks, const int* __restrict__ counts, int nBlocks, int subsPerBlock,
                                                      int* deepSubs, int* deepCountHost)
{
    constexpr int BLOCK = 1024, WAVES = BLOCK / kWave;
    __shared__ int waveTotals[WAVES];
    __shared__ int carry;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (int base = 0; base < nBlocks; base += BLOCK) {
        const int b = base + (int)threadIdx.x;
        const int c = b < nBlocks ? counts[b] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int below = __shfl_up(incl, d, kWave);
            incl += lane >= d ? below : 0;
        }
        if (lane == kWave - 1)
            waveTotals[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w)
            before += waveTotals[w];
        if (b < nBlocks) {
            int at = before + incl - c;
            blocks[b].firstDeep = at;
            unsigned long long mask = blocks[b].deepMask;
            while (mask != 0ull) {
                deepSubs[at++] = b * subsPerBlock + (__ffsll((long long)mask) - 1);
                mask &= mask - 1;
            }
        }
        __syncthreads();
        if (threadIdx.x == BLOCK - 1)
            carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        deepCountHost[0] = carry;
}

/* ---- freeze: the 16-bit copy of the column indices (spgpu?SpmvFreeze, include/spgpu/tuning.h) -------------------------------- */

/* HELL does not say how many slots its arrays have (hackOffsets has no trailing total, hell.c:64,75): the last hack's offset plus
 * hackSize times its longest row.  One wavefront; the answer (< 2^32: slot numbers are what hackOffsets, an int array, can name)
 * goes to a pinned word. */
__global__ __launch_bounds__(kWave) void planSlotsKernel(const int* __restrict__ rS, const int* __restrict__ hackOffsets, int hackSize, int rows,
                                                        int* slotsHost)
{
    const int lastHack = (rows - 1) / hackSize;
    int longest = 0;
    for (long long r = (long long)lastHack * hackSize + threadIdx.x; r < rows; r += kWave)
        longest = rS[r] > longest ? rS[r] : longest;
    longest = waveReduce(longest, MaxOf{});
    if (threadIdx.x == 0)
        slotsHost[0] = (int)((unsigned)hackOffsets[lastHack] + (unsigned)hackSize * (unsigned)longest);
}

/* A workgroup per block of the plan: where the block's 16-bit words count from (packBase: the block's lowest column when all
 * of them are within 16 bits of it -- the x tile then starts there too -- else 32 767 below the middle), and the words of every
 * row the block walks itself.  0xFFFF = "ask rP": a column below the base, 65 535 or more above it, or negative (a slot the
 * kernels skip).  Rows of deep sub-groups are left out: their workgroups read rP. */
template <bool IS_HELL, int SUBS>
__global__ __launch_bounds__(256) void planPackKernel(const int* __restrict__ rP, const int* __restrict__ rS, const int* __restrict__ hackOffsets,
                                                     int hackSize, long long idxStride, int maxNnz, int rows, int baseIndex,
                                                     SpgpuPlanBlock* blocks, unsigned short* __restrict__ packed, unsigned long long* counts)
{
    constexpr int BLOCK = 256, ROWS = SUBS * 32, RPT = ROWS / BLOCK;
    const SpgpuPlanBlock record = blocks[blockIdx.x]; /* (workgroup-uniform) */
    int packBase = 0;
    if (record.probed > 0) {
        const long long span = (long long)record.highest - record.lowest;
        if (span <= 0xFFFE) {
            packBase = record.lowest;
        } else {
            long long start = (long long)record.middle - 0x7FFF;
            start = start < record.lowest ? record.lowest : start;
            start = start > (long long)record.highest - 0xFFFE ? (long long)record.highest - 0xFFFE : start;
            packBase = (int)start;
        }
    }
    if (threadIdx.x == 0)
        blocks[blockIdx.x].packBase = packBase;
    const long long blockRow0 = (long long)blockIdx.x * ROWS;
    unsigned entries = 0, escapes = 0; /* counts[0], counts[1]: the host keeps the copy only if few entries are escapes */
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int i = threadIdx.x + j * BLOCK;
        const long long r = blockRow0 + i;
        if (r >= rows || ((record.deepMask >> (i >> 5)) & 1ull) != 0ull)
            continue;
        const int len = rS ? rS[r] : maxNnz;
        long long at;
        if constexpr (IS_HELL) {
            const unsigned u = (unsigned)r, hs = (unsigned)hackSize;
            at = (long long)((unsigned)hackOffsets[u / hs] + u % hs);
        } else {
            at = r;
        }
        for (int k = 0; k < len; ++k, at += idxStride) {
            const int col = rP[at] - baseIndex;
            const long long off = (long long)col - packBase;
            const bool fits = col >= 0 && off >= 0 && off < 0xFFFF;
            packed[at] = fits ? (unsigned short)off : (unsigned short)0xFFFF;
            entries += 1;
            escapes += fits ? 0 : 1;
        }
    }
    entries = (unsigned)waveReduce((int)entries, SumOf{});
    escapes = (unsigned)waveReduce((int)escapes, SumOf{});
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(&counts[0], (unsigned long long)entries);
        atomicAdd(&counts[1], (unsigned long long)escapes);
    }
}

/* Lock held, plan READY.  Builds the plan's 16-bit index copy on `stream` and waits for it.  False: no memory, or a launch failed
 * (the plan stays as it is, unfrozen). */
template <bool IS_HELL>
static bool packPlan(SpgpuSpmvPlan* plan, spgpuHandle_t handle, hipStream_t stream)
{
    const int* rP = static_cast<const int*>(plan->rP);
    const int* rS = static_cast<const int*>(plan->rS);
    const int* hackOffsets = static_cast<const int*>(plan->hackOffsets);
    long long slots = 0;
    if constexpr (IS_HELL) {
        plan->pinned[2] = 0;
        hipLaunchKernelGGL(planSlotsKernel, dim3(1), dim3(kWave), 0, stream, rS, hackOffsets, plan->hackSize, plan->rows, plan->pinned + 2);
        if (hipStreamSynchronize(stream) != hipSuccess)
            return false;
        slots = (long long)(unsigned)((volatile int*)plan->pinned)[2];
    } else {
        slots = plan->idxStride * (long long)plan->maxNnz;
    }
    if (slots <= 0)
        return false;
    const size_t bytes = ((size_t)slots * sizeof(unsigned short) + 255) / 256 * 256 + 256; /* (a lane's last pack may reach past the last real slot's word; + the two counters) */
    void* packed = nullptr;
    int previous = 0;
    (void)hipGetDevice(&previous);
    (void)hipSetDevice(handle->device);
    const hipError_t allocated = hipMalloc(&packed, bytes);
    (void)hipSetDevice(previous);
    if (allocated != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (spgpuTuning()->poisonScratch) /* testing: no word is read that the pack kernel has not written */
        (void)hipMemsetAsync(packed, 0xA5, bytes, stream);
    SpgpuPlanBlock* records = static_cast<SpgpuPlanBlock*>(plan->device);
    unsigned short* words = static_cast<unsigned short*>(packed);
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(static_cast<char*>(packed) + bytes - 256);
    (void)hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), stream);
    if (plan->subs == 64)
        hipLaunchKernelGGL((planPackKernel<IS_HELL, 64>), dim3((unsigned)plan->blocks), dim3(256), 0, stream, rP, rS, hackOffsets, plan->hackSize,
                           plan->idxStride, plan->maxNnz, plan->rows, plan->baseIndex, records, words, counts);
    else
        hipLaunchKernelGGL((planPackKernel<IS_HELL, 32>), dim3((unsigned)plan->blocks), dim3(256), 0, stream, rP, rS, hackOffsets, plan->hackSize,
                           plan->idxStride, plan->maxNnz, plan->rows, plan->baseIndex, records, words, counts);
    unsigned long long said[2] = {0, 0};
    if (hipMemcpyAsync(said, counts, sizeof(said), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(packed);
        return false;
    }
    /* scattered columns: the copy would save nothing (an escape costs its rP word on top of the 16-bit one) -- the matrix keeps its
     * plan, unfrozen */
    if (said[1] * 100ull > said[0] * (unsigned long long)(spgpuTuning()->freezeEscapesPct < 0 ? 0 : spgpuTuning()->freezeEscapesPct)) {
        (void)hipFree(packed);
        return false;
    }
    plan->packed = packed;
    plan->packedBytes = (long long)bytes;
    return true;
}

/* ---- host ---------------------------------------------------------------------------------------------------------- */

/* Wave-wide loads per stage of the PACKED kernels beyond the unpacked kernels' (SPGPU_RAGGED_UNROLL).  A stage of the queue kernel
 * is a memory round trip and a stage with 16-bit indices is a sixth smaller, so the same bytes in flight could be a third more
 * columns per round trip; the order of additions would not depend on it (a phase adds its columns in ascending order whatever
 * the stage size; the chunks of a split sub-group are sized by the unpacked kernels' stage, raggedSplit).  MEASURED with 1
 * (fp64: 4 loads per stage): 36-44 bytes of scratch per lane, and the target goes from 0.68-0.72 to 0.80-0.83 ms (band) -- a
 * spill is re-read behind a vmcnt(0) wait, which drains the prefetch at every item.  Consuming the stage in two halves (8 x values
 * alive instead of 16) leaves the scratch where it is: the ring of three stages is what does not fit.  0 it stays.
 * Workgroups of 6 wavefronts at 3 per SIMD (168 VGPRs: 4 or 5 loads per stage without scratch) instead: 0.66-0.70 -> 1.01-1.05 ms
 * on the frozen target -- fewer, longer-lived wavefronts lose more than the deeper stage wins.  Not kept either. */
#ifndef SPGPU_PACKED_MORE_UNROLL
#define SPGPU_PACKED_MORE_UNROLL 0
#endif

static size_t roundUp16(size_t v)
{
    return (v + 15) / 16 * 16;
}

/* Lock held.  Starts the analysis of the record's matrix on `stream`; the record is BUILDING afterwards (EMPTY if the
 * allocation failed: the next call tries again). */
template <bool IS_HELL>
static void startPlan(spgpuHandle_t handle, SpgpuSpmvPlan* plan, hipStream_t stream)
{
    SpgpuPrivateHandle* h = spgpuPrivate(handle);
    const long long subs = ((long long)plan->rows + 31) / 32;
    const int blocks = (int)((subs + plan->subs - 1) / plan->subs);
    const size_t blockBytes = roundUp16((size_t)blocks * sizeof(SpgpuPlanBlock)), countBytes = roundUp16((size_t)blocks * sizeof(int));
    void* device = nullptr;
    int previous = 0;
    (void)hipGetDevice(&previous); /* the plan lives where the handle's streams run, whatever device the caller has current */
    (void)hipSetDevice(handle->device);
    const hipError_t allocated = hipMalloc(&device, blockBytes + countBytes + (size_t)subs * sizeof(int));
    (void)hipSetDevice(previous);
    if (allocated != hipSuccess) {
        (void)hipGetLastError();
        plan->state = SPGPU_PLAN_EMPTY;
        return;
    }
    if (spgpuTuning()->poisonScratch) /* testing: the analysis must write every word a launch reads */
        (void)hipMemsetAsync(device, 0xFF, blockBytes + countBytes + (size_t)subs * sizeof(int), stream);
    plan->device = device;
    plan->blocks = blocks;
    plan->deep = 0;
    plan->uses = 0;
    plan->pinned[0] = 0;
    plan->pinned[1] = 0;
    SpgpuPlanBlock* blockRecords = static_cast<SpgpuPlanBlock*>(device);
    int* counts = reinterpret_cast<int*>(static_cast<char*>(device) + blockBytes);
    int* deepSubs = reinterpret_cast<int*>(static_cast<char*>(device) + blockBytes + countBytes);
    const int* rP = static_cast<const int*>(plan->rP);
    const int* rS = static_cast<const int*>(plan->rS);
    const int* hackOffsets = static_cast<const int*>(plan->hackOffsets);
    if (plan->subs == 64)
        hipLaunchKernelGGL((planBlocksKernel<IS_HELL, 64>), dim3((unsigned)blocks), dim3(256), 0, stream, rP, rS, hackOffsets, plan->hackSize,
                           plan->idxStride, plan->maxNnz, plan->rows, plan->baseIndex, plan->deepCap, blockRecords, counts);
    else if (plan->subs == 16)
        hipLaunchKernelGGL((planBlocksKernel<IS_HELL, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, rP, rS, hackOffsets, plan->hackSize,
                           plan->idxStride, plan->maxNnz, plan->rows, plan->baseIndex, plan->deepCap, blockRecords, counts);
    else
        hipLaunchKernelGGL((planBlocksKernel<IS_HELL, 32>), dim3((unsigned)blocks), dim3(256), 0, stream, rP, rS, hackOffsets, plan->hackSize,
                           plan->idxStride, plan->maxNnz, plan->rows, plan->baseIndex, plan->deepCap, blockRecords, counts);
    hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, stream, blockRecords, counts, blocks, plan->subs, deepSubs, plan->pinned);
    if (hipEventRecord(plan->built, stream) != hipSuccess) {
        (void)hipGetLastError();
        spgpuPlanRetire(handle, plan); /* (the kernels may run: the buffer waits in the graveyard) */
        plan->state = SPGPU_PLAN_GIVEN_UP;
        return;
    }
    plan->state = SPGPU_PLAN_BUILDING;
    h->planBuilds += 1;
}

/*
 * The ordered SpMV of launchSlabFamily (ellpack_spmv.hip) with the matrix's plan, if it has one that is ready: true = launched,
 * nothing is to follow; false = the caller runs the path with the deep list (and, where that is possible, the analysis has been
 * started behind the scenes).  `shape`: launchRagged's (4: 2 048 rows per workgroup with staged results; 5: 1 024 rows,
 * staged; otherwise 1 024 rows and a 64 KiB tile); tiled = false: the gather form (512 rows, no x tile).
 * mustLaunch: the caller has no deep list for this stream -- without a ready plan the same kernel runs with NO plan: nothing
 * is listed, every sub-group deeper than the cap is worked off by its own block behind its stream, no x tile.  Stateless,
 * slower, the same bits.
 * prepareMode 1 (spgpu?SpmvPrepare): nothing is launched but the analysis, and that is waited for: true = the plan is ready.
 * prepareMode 2 (spgpu?SpmvFreeze): the same, and the plan gets its 16-bit copy of the column indices (packPlan): true = the
 * matrix is frozen -- later launches read 2 bytes of index per stored entry instead of 4 (raggedSpmvKernel<..., PACKED>).
 */
template <typename T, bool IS_HELL>
bool launchPlanned(spgpuHandle_t handle, hipStream_t stream, const SlabArgs<T>& in, int shape, bool tiled, bool mustLaunch, int prepareMode)
{
    const bool prepareOnly = prepareMode != 0;
    constexpr int RPL = 16 / (int)sizeof(T);
    constexpr int UNROLL = SPGPU_RAGGED_UNROLL(RPL);
    const SpgpuTuning* tune = spgpuTuning();
    const bool staged = tiled && sizeof(T) <= 8 && (shape == 4 || shape == 5);
    const int subs = !tiled ? 16 : (staged && shape == 4 ? 64 : 32);
    SlabArgs<T> a = in;
    a.split = raggedSplit<T>(a.deepCap, (kWave / (32 / RPL)) * UNROLL, tune->raggedSplit);
    a.deepHeader = nullptr;
    a.planBlocks = nullptr;
    a.planDeepSubs = nullptr;
    a.planDeep = 0;
    a.planFlags = nullptr;
    a.planPacked = nullptr;
    int perBlock = tune->planDeepPerBlock;
    perBlock = perBlock < 1 ? 1 : (perBlock > kPlanDeepMost ? kPlanDeepMost : perBlock);
    a.planDeepPerBlock = perBlock;
    a.planDeepRuns = tune->planDeepRuns;
    a.planDeepStride = 0;
    const long long subGroups = ((long long)a.rows + 31) / 32;
    a.planMainBlocks = (int)((subGroups + subs - 1) / subs);
    auto launch = [&]() {
        const unsigned deepBlocks = (unsigned)((a.planDeep + perBlock - 1) / perBlock);
        const unsigned grid = (unsigned)a.planMainBlocks + deepBlocks;
        /* the workgroups of deep sub-groups over the first planDeepSpread per cent of the grid (0: all in front; < 0: all behind) */
        if (tune->planDeepSpread < 0 || deepBlocks == 0u) {
            a.planDeepStride = 0;
        } else {
            const unsigned long long reach = (unsigned long long)grid * (unsigned)(tune->planDeepSpread > 100 ? 100 : tune->planDeepSpread) / 100ull;
            const unsigned stride = (unsigned)(reach / deepBlocks);
            /* odd: the hardware deals workgroup ids round-robin over the 8 XCDs -- with an even stride the long-lived workgroups
             * would pile up on one or two of them (measured: stride 8, 0.73 -> 0.94 ms, one XCD still busy 250 us after the others) */
            a.planDeepStride = (int)(stride < 1u ? 1u : (stride | 1u));
        }
#define SPGPU_PLANNED(WAVES, TILE, SUBS, ZB)                                                                          \
    hipLaunchKernelGGL((raggedSpmvKernel<T, RPL, IS_HELL, UNROLL, WAVES, TILE, SUBS, true, ZB, true>), dim3(grid), dim3((WAVES) * kWave), 0, stream, a)
        if (!tiled) {
            SPGPU_PLANNED(4, 0, 16, 0);
        } else if constexpr (sizeof(T) <= 8) {
#define SPGPU_PLANNED_PACKED(WAVES, TILE, SUBS, ZB)                                                                   \
    hipLaunchKernelGGL((raggedSpmvKernel<T, RPL, IS_HELL, UNROLL + SPGPU_PACKED_MORE_UNROLL, WAVES, TILE, SUBS, true, ZB, true, true>), dim3(grid), dim3((WAVES) * kWave), 0, stream, a)
            if (a.planPacked) { /* a frozen matrix: 16-bit indices */
                if (staged && shape == 4)
                    SPGPU_PLANNED_PACKED(8, 49152, 64, 17408);
                else if (staged)
                    SPGPU_PLANNED_PACKED(8, 49152, 32, 17408);
                else
                    SPGPU_PLANNED_PACKED(8, 65536, 32, 0);
            } else if (staged && shape == 4)
                SPGPU_PLANNED(8, 49152, 64, 17408);
            else if (staged)
                SPGPU_PLANNED(8, 49152, 32, 17408);
            else
                SPGPU_PLANNED(8, 65536, 32, 0);
        } else {
            SPGPU_PLANNED(8, 65536, 32, 0);
        }
#undef SPGPU_PLANNED
#undef SPGPU_PLANNED_PACKED
    };

    /* a captured launch would carry the plan's addresses for as long as the graph lives; plans are retired: no plan there */
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
        (void)hipGetLastError();
        capturing = hipStreamCaptureStatusActive;
    }
    bool launched = false;
    if (tune->plan && in.rIdx && capturing == hipStreamCaptureStatusNone) {
        SpgpuSpmvPlan key{};
        key.rP = in.rP;
        key.rS = in.rS;
        key.rIdx = in.rIdx;
        key.hackOffsets = in.hackOffsets;
        key.idxStride = in.idxStride;
        key.rows = in.rows;
        key.hackSize = in.hackSize;
        key.baseIndex = in.baseIndex;
        key.maxNnz = in.maxNnz;
        key.deepCap = in.deepCap;
        key.subs = subs;
        SpgpuPrivateHandle* h = spgpuPrivate(handle);
        spgpuPlanLock(handle);
        SpgpuSpmvPlan* plan = spgpuPlanRecord(handle, &key);
        if (plan) {
            if (plan->state == SPGPU_PLAN_BUILDING && spgpuEventDone(plan->built)) {
                plan->deep = ((volatile int*)plan->pinned)[0];
                plan->state = SPGPU_PLAN_READY;
            }
            if (plan->state == SPGPU_PLAN_READY && ((volatile int*)plan->pinned)[1] != 0) {
                /* a kernel met a sub-group whose depth contradicts the plan: another matrix lives at these addresses now.  (Its
                 * results were right all the same.)  A matrix that keeps changing under a young plan is left alone after the third time. */
                h->planStales += 1;
                plan->stales += 1;
                const bool giveUp = plan->uses < 16 && plan->stales >= 3;
                spgpuPlanRetire(handle, plan);
                if (giveUp)
                    plan->state = SPGPU_PLAN_GIVEN_UP;
            }
            if (plan->state == SPGPU_PLAN_EMPTY)
                startPlan<IS_HELL>(handle, plan, stream);
            if (prepareOnly) {
                if (plan->state == SPGPU_PLAN_BUILDING && hipEventSynchronize(plan->built) == hipSuccess) {
                    plan->deep = ((volatile int*)plan->pinned)[0];
                    plan->state = SPGPU_PLAN_READY;
                }
                launched = plan->state == SPGPU_PLAN_READY;
                if (prepareMode == 2) { /* freeze: true = the plan has its 16-bit indices */
                    if (launched && !plan->packed && tiled && sizeof(T) <= 8 && (subs == 64 || subs == 32))
                        (void)packPlan<IS_HELL>(plan, handle, stream);
                    launched = launched && plan->packed != nullptr;
                    if (launched)
                        h->planFreezes += 1;
                }
            } else if (plan->state == SPGPU_PLAN_READY) {
                const size_t blockBytes = roundUp16((size_t)plan->blocks * sizeof(SpgpuPlanBlock)), countBytes = roundUp16((size_t)plan->blocks * sizeof(int));
                a.planBlocks = static_cast<const SpgpuPlanBlock*>(plan->device);
                a.planDeepSubs = reinterpret_cast<const int*>(static_cast<const char*>(plan->device) + blockBytes + countBytes);
                a.planDeep = plan->deep;
                a.planMainBlocks = plan->blocks;
                a.planFlags = plan->pinned;
                if (tiled && sizeof(T) <= 8)
                    a.planPacked = static_cast<const unsigned short*>(plan->packed);
                launch(); /* (under the lock: a retirement on another host thread waits for this launch to be queued) */
                plan->uses += 1;
                h->planUses += 1;
                launched = true;
            }
        }
        spgpuPlanUnlock(handle);
    }
    if (!launched && mustLaunch && !prepareOnly) {
        launch();
        launched = true;
    }
    return launched;
}

template bool launchPlanned<float, true>(spgpuHandle_t, hipStream_t, const SlabArgs<float>&, int, bool, bool, int);
template bool launchPlanned<float, false>(spgpuHandle_t, hipStream_t, const SlabArgs<float>&, int, bool, bool, int);
template bool launchPlanned<double, true>(spgpuHandle_t, hipStream_t, const SlabArgs<double>&, int, bool, bool, int);
template bool launchPlanned<double, false>(spgpuHandle_t, hipStream_t, const SlabArgs<double>&, int, bool, bool, int);
template bool launchPlanned<cfloat, true>(spgpuHandle_t, hipStream_t, const SlabArgs<cfloat>&, int, bool, bool, int);
template bool launchPlanned<cfloat, false>(spgpuHandle_t, hipStream_t, const SlabArgs<cfloat>&, int, bool, bool, int);
template bool launchPlanned<cdouble, true>(spgpuHandle_t, hipStream_t, const SlabArgs<cdouble>&, int, bool, bool, int);
template bool launchPlanned<cdouble, false>(spgpuHandle_t, hipStream_t, const SlabArgs<cdouble>&, int, bool, bool, int);

} // namespace spgpu

extern "C" {
#ifdef SPGPU_TRACE_BLOCKS
void spgpuPlannedSetTrace(unsigned long long* buffer)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(spgpu::spgpuTraceBuffer), &buffer, sizeof(buffer));
}
#endif
}
