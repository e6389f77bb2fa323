#pragma once
/*
 * What the ELL / HELL SpMV kernels are handed (one struct for every kernel of the family), and the pieces more than one
 * translation unit of the family needs: ellpack_spmv.hip (the kernels for rows as they come, the queue kernel for ordered
 * rows with its deep list) and planned_spmv.hip (the queue kernel driven by a per-matrix plan).
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

namespace spgpu {

template <typename T> struct SlabArgs {
    T* z;
    const T* y;
    const T* x;
    const T* cM;
    const int* rP;
    const int* rS;          /* NULL: every row has maxNnz slots (ELL only) */
    const int* rIdx;        /* NULL: identity */
    const int* hackOffsets; /* HELL only */
    T alpha, beta;
    int rows;
    int baseIndex;
    int hackSize; /* HELL only */
    int maxNnz;   /* ELL without rS */
    long long valStride, idxStride; /* elements between two slab columns */
    int wideIO;   /* y and z are aligned for RPL-wide access */
    int tailLanes; /* TAIL kernels: switch to whole-wave rows when <= this many lanes are busy */
    int* feedback; /* pinned host ints the sample wavefronts report the form they saw to, or NULL */
    int feedbackTag; /* or-ed into every report: the generation of the table entry the report is for (spgpuFormFeedback) */
    long long tileSpanLimit; /* a sample group whose columns span at most this many counts as "local" (x-tile form) */
    /* DEEP kernels: 32-row sub-groups deeper than deepCap hand their columns >= deepCap to the deep kernels */
    int deepCap;
    int deepKeep;                /* of a sub-group deeper than deepCap the main kernel walks the first deepKeep columns (<= deepCap); the rest are items */
    int deepChunk;               /* columns per item */
    int* deepHeader;             /* entries registered, items handed out (may exceed the capacities), finish ticket */
    SpgpuDeepEntry* deepEntries; /* [SPGPU_DEEP_ENTRIES] */
    SpgpuDeepItem* deepItems;    /* [SPGPU_DEEP_ITEMS] */
    T* deepPartials;             /* [SPGPU_DEEP_ENTRIES][32] row sums over the columns < deepCap */
    T* deepItemSums;             /* [SPGPU_DEEP_ITEMS][32] */
    int* deepOverflow;           /* pinned: calls that overflowed the list, and what the last of them asked for */
    int xcdRun;                  /* raggedSpmvKernel: row blocks per XCD run (0: hardware order) */
    int pipeRanges;              /* pipeSpmvKernel: ranges per workgroup */
    int avgNnzPerRow;            /* the caller's hint (0: none) */
    int split;                   /* raggedSpmvKernel: columns per chunk of a split sub-group (0: none) */
    int stageLate;               /* raggedSpmvKernel: the destinations are staged under the tile's round trip (SPGPU_STAGE_LATE=0: before the first requests) */
    /* raggedSpmvKernel<..., PLAN> (planned_spmv.hip): the matrix's plan (spgpu_internal.h) */
    const SpgpuPlanBlock* planBlocks; /* [planMainBlocks] */
    const int* planDeepSubs;          /* [planDeep] sub-groups that get workgroups of their own, ascending */
    int planDeep;
    int planMainBlocks;               /* workgroups that own blocks of rows; the rest of the grid owns deep sub-groups */
    int planDeepPerBlock;             /* deep sub-groups per such workgroup (<= kPlanDeepMost) */
    int planDeepRuns;                 /* 0: such a workgroup takes sub-groups deepId, deepId + deepBlocks, ... of the list; 1: a run of consecutive ones */
    int planDeepStride;               /* such a workgroup at every planDeepStride-th place of the grid, from the front (0: all of them at the end) */
    int* planFlags;                   /* pinned; [1] = 1: a kernel found the plan contradicting the matrix */
    const int* packBases;             /* slabSpmvKernel<..., PACKED> (a frozen matrix without a row order): the column the 16-bit words of
                                       * every group of rows (one wavefront's) count from */
    const unsigned short* planPacked; /* raggedSpmvKernel<..., PACKED>: a frozen matrix' column indices as 16-bit offsets from the block's
                                       * packBase, slot for slot as in rP (0xFFFF: ask rP); NULL: the matrix is not frozen */
};

constexpr int kBlockThreads = 256;
constexpr int kTailLanes = 16; /* switch to whole-wave row processing when <= this many lanes are busy
                                  (measured flat between 4 and 16 for the 1-phase kernel, worse above) */
constexpr int kTailUnroll = 4; /* entries per lane in flight in tail mode */

/* Cache policy of the x gathers (experiments, -DSPGPU_TUNING_VARIANTS): 0 default, 1 non-temporal,
 * 2 agent-scope (sc1: bypasses the per-CU L1). */
template <int POLICY, typename T> __device__ inline T loadX(const T* p)
{
    if constexpr (POLICY == 1) {
        using Raw = typename RawBits<sizeof(T)>::type;
        Raw raw = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(p));
        T out;
        __builtin_memcpy(&out, &raw, sizeof(T));
        return out;
    } else if constexpr (POLICY == 2 && sizeof(T) == 8) {
        unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
        T out;
        __builtin_memcpy(&out, &raw, sizeof(T));
        return out;
    } else {
        return *p;
    }
}

/* One lane registers a 32-row sub-group deeper than deepCap in the handle's deep list: an entry, and one item per
 * deepChunk columns beyond the cap.  Returns the entry, or -1 when the list is full -- the sub-group then stays with
 * the main kernel.  (The list is global: which entry a sub-group gets depends on scheduling, its sum does not.) */
template <typename T> __device__ inline int deepRegister(const SlabArgs<T>& a, int row0, int depth, unsigned base)
{
    const int items = (depth - a.deepKeep + a.deepChunk - 1) / a.deepChunk;
    const int entry = atomicAdd(&a.deepHeader[SPGPU_DEEP_HEAD_ENTRIES], 1);
    if (entry >= SPGPU_DEEP_ENTRIES)
        return -1;
    const int first = atomicAdd(&a.deepHeader[SPGPU_DEEP_HEAD_ITEMS], items);
    const bool fits = first + items <= SPGPU_DEEP_ITEMS;
    a.deepEntries[entry] = SpgpuDeepEntry{row0, depth, first, fits ? items : 0};
    if (!fits) {
        if (first < SPGPU_DEEP_ITEMS) /* the slots from `first` on keep what an earlier call left: nobody may read them */
            atomicMax(&a.deepHeader[SPGPU_DEEP_HEAD_CUT], SPGPU_DEEP_ITEMS - first);
        return -1;
    }
    for (int c = 0; c < items; ++c)
        a.deepItems[first + c] = SpgpuDeepItem{row0, base, depth, c};
    return entry;
}

/* What one wavefront saw of its rows' columns (XTILE probe). */
struct ColumnProbe {
    int lowest, highest, rows;
    long long middles; /* sum over sampled rows of (first + last column) / 2 */
};


/* planned_spmv.hip: the ordered SpMV with the matrix's plan, if it has one that is ready (true: launched, nothing follows) */
template <typename T, bool IS_HELL> bool launchPlanned(spgpuHandle_t handle, hipStream_t stream, const SlabArgs<T>& in, int shape, bool tiled, bool mustLaunch, int prepareMode);

constexpr int kDeepChunk = 64; /* columns per deep item; measured: items of 32 / 64 / 128 columns and stages of 16 / 32 within 8 % -- the kernel is bound by the lines its gathers pull */

} // namespace spgpu
