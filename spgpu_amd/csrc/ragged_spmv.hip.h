/*
 * ELL / HELL SpMV for matrices whose rows were ORDERED BY LENGTH (rIdx given: spgpuOellOrderDevice, ellToOell) --
 * the north_star target, power-law row lengths.  Included by ellpack_spmv.hip (namespace spgpu, after SlabArgs).
 *
 * What is different from slabSpmvKernel: there a wavefront owns a fixed block of rows, which is right when rows are
 * about equally long.  After an ordering by length the depth changes along the rows -- steeply at the head of every
 * window -- and a workgroup whose wavefronts own fixed rows waits for its deepest ones with the LDS tile and the
 * wavefront slots of the others idle.  Here a workgroup owns SUBS sub-groups of 32 rows and its wavefronts take them
 * from a queue, deepest first (that is the order the rows are in), one wavefront per sub-group:
 *
 *   - "one wavefront per hack": 32/RPL lanes with RPL rows each cover a slab column, PH = 64 / (32/RPL) columns per
 *     load instruction (1 KiB contiguous for hackSize 32), UNROLL of them per stage; the PH phase sums of a row are
 *     combined with lane-xor shuffles.  Summation order = the reference's multi-thread-per-row order generalised to
 *     PH phases (hell_spmv_base_template.cuh:59-101): phase p adds the entries k = p (mod PH) in ascending k, the
 *     phase sums are combined pairwise -- orc_?hellspmv / orc_?ellspmv with phases = PH.
 *   - row lengths and slab bases of all the workgroup's rows sit in LDS, so a wavefront that finishes a sub-group
 *     knows the addresses of the next one at once: the first stage of the next sub-group is requested during the last
 *     stage of the current one, and the stream never stops for a dependent look-up.
 *   - the slice of x the workgroup's rows touch is staged in LDS (as in the x-tile form of slabSpmvKernel); entries
 *     outside it are gathered from global memory.
 *   - sub-groups deeper than deepCap keep their first deepCap columns here and hand the rest to the deep kernels
 *     through the handle's deep list (see slabSpmvKernel, DEEP).
 *
 * Algorithmic bytes as for slabSpmvKernel, plus 4 per row for rIdx.
 */

/* The value lane (l ^ M) holds, delivered to the lanes whose bit M is clear (what the phase sums' fold needs: only phase 0 uses the
 * result), without the LDS queue: M = 8 a rotation inside the DPP row, M = 16 / 32 gfx950's row and half swaps
 * (v_permlane16_swap / v_permlane32_swap).  In the other lanes the result is unspecified. */
template <int M> __device__ inline unsigned partnerWord(unsigned w)
{
    static_assert(M == 8 || M == 16 || M == 32, "the folds of 8-, 4-, 2- and 1-row strips");
    if constexpr (M == 8) {
        return (unsigned)dppMove<0x128, 0xF>((int)w, (int)w); /* row_ror:8 */
    } else if constexpr (M == 16) {
        return __builtin_amdgcn_permlane16_swap(w, w, false, false)[1]; /* rows 0 and 2 receive rows 1 and 3 */
    } else {
        return __builtin_amdgcn_permlane32_swap(w, w, false, false)[1]; /* the lower half receives the upper half */
    }
}
template <int M, typename T> __device__ inline T partnerOf(T v)
{
    constexpr int WORDS = (int)sizeof(T) / 4;
    unsigned w[WORDS];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < WORDS; ++i)
        w[i] = partnerWord<M>(w[i]);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}

/* Most chunks a split sub-group may have (see SPLIT below): what the tightest tiled shape -- 48 KiB of LDS behind 2 048 rows --
 * can park when every one of its sub-groups is split.  fp64 / complex fp32: 3, fp32: 6, complex fp64: 1 (never split). */
template <typename T> constexpr int kRaggedMostChunks = 49152 / (int)sizeof(T) / 2048;

#include "deep_rows.hip.h"

/* PLAN (planned_spmv.hip): the matrix has a plan (spgpu_internal.h).  Two of the prologue's three dependent round trips
 * are then one -- the block's record says where its slice of x lies, so nobody asks every row for its first and last column
 * -- and nothing registers anywhere: the sub-groups the plan lists as deep are skipped here and worked off by workgroups of
 * their own at the end (or the start) of the same grid (deep_rows.hip.h).  What the plan says is never trusted for WHAT is
 * computed: lengths, bases, destinations and the depth that decides a sub-group's chunks are read from the matrix; a
 * sub-group deeper than deepCap that the plan does not list (a stale plan) is worked off behind the block's own stream, by
 * the same routine, in the same order of additions. */
/* PACKED (a FROZEN matrix: spgpu?SpmvFreeze, include/spgpu/tuning.h -- the caller has promised that its index arrays stay as they
 * are): the blocks of rows read their column indices from the plan's 16-bit copy (a.planPacked: offsets from the block's packBase,
 * slot for slot as in rP) instead of from rP -- 2 bytes per stored entry instead of 4, a sixth of an fp64 matrix' stream.  An offset
 * is a position in the LDS tile once the tile's base is subtracted; 0xFFFF says "ask rP" (a column out of the 16-bit reach of the
 * base, or negative).  The same columns, the same x, the same order of additions: the same bits.  The workgroups of deep
 * sub-groups read rP as ever. */
template <typename T, int RPL, bool IS_HELL, int UNROLL, int WAVES, int TILE_BYTES, int SUBS, bool DEEP, int ZBYTES = 0, bool PLAN = false, bool PACKED = false>
__global__ __launch_bounds__(WAVES * kWave) __attribute__((amdgpu_waves_per_eu(4)))
void raggedSpmvKernel(const SlabArgs<T> a) /* 4 wavefronts per SIMD: two 8-wavefront workgroups per CU (what LDS admits) */
{
    static_assert(!PACKED || (PLAN && TILE_BYTES > 0 && RPL >= 2), "packed indices come with a plan and a tile; 4- and 8-byte elements");
    using ColumnWord = typename std::conditional<PACKED, unsigned short, int>::type;
    constexpr int LPC = 32 / RPL;   /* lanes per slab column of a sub-group */
    constexpr int PH = kWave / LPC; /* slab columns per wave-wide load */
    constexpr int STEP = PH * UNROLL;
    constexpr int BLOCK = WAVES * kWave;
    constexpr int ROWS = SUBS * 32;
    constexpr bool XTILE = TILE_BYTES > 0;
    /* (the gather form has no tile, only room for the chunk sums of its split sub-groups: the sums must not depend on the form) */
    /* BEHIND: deep sub-groups nobody else takes are worked off by this workgroup behind its stream; that routine borrows the tile */
    constexpr bool BEHIND = DEEP;
    constexpr int CHUNK_ROOM = kRaggedMostChunks<T> >= 2 ? SUBS * 32 * kRaggedMostChunks<T> : 1;
    constexpr int BEHIND_ROOM = BEHIND ? 16384 / (int)sizeof(T) : 1;
    constexpr int TILE_ELEMS = XTILE ? TILE_BYTES / (int)sizeof(T) : (CHUNK_ROOM > BEHIND_ROOM ? CHUNK_ROOM : BEHIND_ROOM);
    static_assert(TILE_ELEMS >= SUBS * 32 * kRaggedMostChunks<T> || kRaggedMostChunks<T> < 2, "room for every chunk sum");
    /* ZSTAGE: the workgroup's results wait in LDS, placed by destination, and leave in whole lines when its queue is empty.
     * The L2 does not merge stores over time (every store's bytes leave it at once): written from the lanes that hold the
     * sums, z[rIdx[r]] is one 8-byte fabric write per row -- 3 % of the kernel's time with the rows ordered in windows of
     * 1 024, 6 % at 2 048, 12 % at 4 096 (profiles/r03_exp_z_scatter.txt).  ZW destinations from the workgroup's lowest. */
    constexpr bool ZSTAGE = ZBYTES > 0;
    constexpr int ZW = ZSTAGE ? ZBYTES / (int)sizeof(T) : 32;
    static_assert(ZW < 0xFFFF && ZW % 32 == 0, "a staged destination is a 16-bit offset");
    static_assert(SUBS >= WAVES, "every wavefront starts with a sub-group of its own");
    static_assert(!PLAN || (DEEP && SUBS <= 64), "a plan record holds a 64-bit mask");
    static_assert(!BEHIND || TILE_ELEMS * (int)sizeof(T) >= 16384, "the deep routine borrows the tile");

    __shared__ __attribute__((aligned(16))) T tile[TILE_ELEMS];
    /* row lengths as walked here (deep sub-groups: cut at deepCap); 16 bits each, 0xFFFF = "65 535 or more: ask rS" (a
     * row that long is walked whole only when the deep list was full) */
    __shared__ __attribute__((aligned(8))) unsigned short lens[ROWS];
    /* rIdx of the workgroup's rows (fetched from global memory at the end of a sub-group it would be waited for with vmcnt(0)
     * -- counters retire in order -- and drain the wavefront's prefetch); ZSTAGE: as offsets from the workgroup's lowest
     * destination instead, 0xFFFF = beyond the staging buffer */
    __shared__ int dests[ZSTAGE ? 1 : ROWS];
    __shared__ __attribute__((aligned(8))) unsigned short destOffset[ZSTAGE ? ROWS : 4];
    __shared__ __attribute__((aligned(16))) T staged[ZW];
    __shared__ unsigned stagedMask[ZW / 32];
    __shared__ int waveLowestDest[WAVES];
    __shared__ int2 subFacts[SUBS]; /* first item of the sub-group; first parked chunk sum, or -1 (one chunk) */
    __shared__ unsigned bases[ROWS / RPL]; /* first slot of every RPL-row strip, in elements (hackOffsets is an int array: a slot
                                              number plus the offset inside the hack fits 32 unsigned bits; 64-bit from here on) */
    __shared__ int depths[SUBS];      /* longest walked row of every sub-group */
    __shared__ int deepSlots[SUBS];   /* its entry in the deep list, or -1 */
    __shared__ int nextItem;
    __shared__ ColumnProbe seen[WAVES];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane % LPC, phase = lane / LPC;
#ifdef SPGPU_TRACE_BLOCKS
    /* experiment builds only: start / end time of every workgroup (100 MHz wall clock) into a caller-provided buffer */
    if (spgpuTraceBuffer && threadIdx.x == 0)
        spgpuTraceBuffer[8 * (size_t)blockIdx.x] = wall_clock64();
#endif
    /* Consecutive workgroups (in row order) read overlapping slices of x.  The hardware deals workgroup ids round-robin
     * over the 8 XCDs, each with an L2 of its own: left alone, every slice is fetched from memory by all eight.  With
     * a.xcdRun > 0 the id is permuted so that runs of xcdRun consecutive row blocks share an XCD (speed only). */
    unsigned mainId = blockIdx.x, mainBlocks = gridDim.x;
    if constexpr (PLAN) {
        /* the grid: planMainBlocks workgroups that own blocks of rows, and one for every planDeepPerBlock deep sub-groups.  The
         * latter live long on little bandwidth (their x comes from global memory, a round trip per stage): they are spread
         * over the front part of the grid, one at every planDeepStride-th place, so that they run beside many blocks of rows and
         * are done long before the launch ends (planDeepStride 0: all of them behind the blocks of rows).  Each takes a RUN of
         * consecutive sub-groups of the list (planDeepRuns; 0: every deepBlocks-th -- "a bit of everything", the first form).  After
         * an ordering by length the list's neighbours are one window of set-aside long rows: its sub-groups' gathers fall into one
         * stretch of x and meet in the workgroup's L2, and every window holds about the same work.  Measured on the target, same
         * process and allocations, four rounds each: band 0.7534 -> 0.7465 ms, +-2 048 0.8345 -> 0.8205. */
        mainBlocks = (unsigned)a.planMainBlocks;
        const unsigned deepBlocks = gridDim.x - mainBlocks, stride = (unsigned)a.planDeepStride;
        bool deepBlock;
        unsigned deepId;
        if (stride == 0u) {
            deepBlock = blockIdx.x >= mainBlocks;
            deepId = blockIdx.x - mainBlocks;
        } else {
            deepId = blockIdx.x / stride;
            deepBlock = blockIdx.x % stride == 0u && deepId < deepBlocks;
            const unsigned before = (blockIdx.x + stride - 1u) / stride;
            mainId = blockIdx.x - (before < deepBlocks ? before : deepBlocks);
        }
        if (deepBlock) {
            int count = 0;
            const bool runs = a.planDeepRuns != 0;
            for (int j = 0; j < a.planDeepPerBlock; ++j)
                count += (runs ? (long long)deepId * a.planDeepPerBlock + j : (long long)deepId + (long long)j * deepBlocks) < a.planDeep ? 1 : 0;
            wholeSubgroups<T, RPL, IS_HELL, WAVES, TILE_ELEMS * (int)sizeof(T)>(a, tile, count, [&](int j) {
                return a.planDeepSubs[runs ? deepId * (unsigned)a.planDeepPerBlock + (unsigned)j : deepId + (unsigned)j * deepBlocks];
            }, true);
#ifdef SPGPU_TRACE_BLOCKS
            if (spgpuTraceBuffer && lane == 0)
                atomicMax(&spgpuTraceBuffer[8 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
            return;
        }
    }
    const unsigned logicalBlock = a.xcdRun > 0 ? xcdRuns(mainId, mainBlocks, (unsigned)a.xcdRun) : mainId;
    const long long blockRow0 = (long long)logicalBlock * ROWS;
    const T* __restrict__ x = a.x;

    /* ---- prologue.  A workgroup lives ~35 us and streams nothing while it finds out where its rows are, so the
     * dependent memory round trips here are counted: (1) row lengths and hack offsets, (2) the first and last column
     * of every row (where is the slice of x?) and the deep registrations, (3) the slice of x itself, in ONE round of
     * loads, together with the first stages of the stream.  Measured with 4 round trips and 4 barriers: 12.7 us, a
     * third of a workgroup's life (profiles/r02b_ragged_workgroup_trace.txt). ------------------------------------- */
    constexpr int RPT = (ROWS + BLOCK - 1) / BLOCK; /* rows a thread looks at; 32 consecutive rows = 32 consecutive lanes */
    int myLen[RPT], myDest[RPT];
    unsigned myBase[RPT];
    bool stagedHere[RPT]; /* the row's result is this stream's to finish */
    constexpr int kHere = 0, kElsewhere = 1, kBehind = 2;
    SpgpuPlanBlock planned{};
    if constexpr (PLAN) {
        if (a.planBlocks) /* (workgroup-uniform: scalar loads, no exchange; no plan: nothing listed, no tile -- the stateless fallback) */
            planned = a.planBlocks[logicalBlock];
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) { /* round trip 1 */
        const int i = threadIdx.x + j * BLOCK;
        const long long r = blockRow0 + i;
        const bool live = i < ROWS && r < a.rows;
        myLen[j] = live ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
        myDest[j] = live && a.rIdx ? a.rIdx[r] : (int)r;
        myBase[j] = 0;
        if (live) {
            if constexpr (IS_HELL) {
                const unsigned u0 = (unsigned)r, hs = (unsigned)a.hackSize;
                myBase[j] = (unsigned)a.hackOffsets[u0 / hs] + u0 % hs;
            } else {
                myBase[j] = (unsigned)r;
            }
        }
    }
    if constexpr (ZSTAGE) {
        if (threadIdx.x < ZW / 32)
            stagedMask[threadIdx.x] = 0u; /* (long before the barrier below) */
    }
#ifdef SPGPU_TRACE_BLOCKS
    /* (trace builds: 8 words per workgroup -- 0 start, 1 end, 2 tile in place, 3 lengths here, 4 probes here and tables written,
     * 5 destinations staged, 6 first stages requested) */
#define SPGPU_STAMP(word)                                                                                             \
    do {                                                                                                              \
        if (spgpuTraceBuffer && threadIdx.x == 0)                                                                     \
            spgpuTraceBuffer[8 * (size_t)blockIdx.x + (word)] = wall_clock64();                                       \
    } while (0)
#else
#define SPGPU_STAMP(word) do { } while (0)
#endif
    if (myLen[0] < 0)
        return; /* (never: makes the stamp below wait for round trip 1) */
    SPGPU_STAMP(3);
    ColumnProbe mine{0x7fffffff, -0x7fffffff - 1, 0, 0};
#pragma unroll
    for (int j = 0; j < RPT; ++j) { /* round trip 2 */
        const int i = threadIdx.x + j * BLOCK;
        int first = 0, last = 0;
        if (XTILE && !PLAN && myLen[j] > 0) {
            first = a.rP[(long long)myBase[j]];
            last = a.rP[(long long)myBase[j] + (long long)(myLen[j] - 1) * a.idxStride];
        }
        /* depth of the 32-row sub-group these 32 lanes hold; the deep ones register and are cut at deepCap */
        const int depth = halfReduce(myLen[j], MaxOf{});
        /* Who works the sub-group off.  kHere: this workgroup's stream (a sub-group registered in the deep list: its first
         * deepKeep columns).  kElsewhere: a workgroup of deep sub-groups, as the plan says.  kBehind: this workgroup, behind its
         * stream, whole (deep_rows.hip.h) -- deeper than the cap with nobody else to take it: a stale or absent plan, a
         * full list.  The chunks and their order are the same in all three. */
        int slot = -1, who = kHere;
        const bool deep = DEEP && depth > a.deepCap;
        if constexpr (PLAN) {
            const bool listed = ((planned.deepMask >> ((i >> 5) & 63)) & 1ull) != 0ull;
            who = listed ? kElsewhere : (deep ? kBehind : kHere);
            if (listed != deep && (lane & 31) == 0 && i < ROWS && blockRow0 + i < a.rows && a.planFlags)
                a.planFlags[1] = 1; /* the plan is of another matrix: the host retires it before its next call */
        } else if constexpr (DEEP) {
            if ((lane & 31) == 0 && deep && i < ROWS)
                slot = deepRegister(a, (int)(blockRow0 + i), depth, myBase[j]);
            const int slotLow = __builtin_amdgcn_readlane(slot, 0), slotHigh = __builtin_amdgcn_readlane(slot, 32);
            slot = (lane & 32) ? slotHigh : slotLow;
            who = (BEHIND && deep && slot < 0) ? kBehind : kHere;
        }
        stagedHere[j] = who == kHere && slot < 0;
        if (i < ROWS) {
            const int walked = who != kHere ? 0 : ((slot >= 0 && myLen[j] > a.deepKeep) ? a.deepKeep : myLen[j]);
            lens[i] = (unsigned short)(walked < 0xFFFF ? walked : 0xFFFF);
            if constexpr (!ZSTAGE)
                dests[i] = myDest[j];
            if (i % RPL == 0)
                bases[i / RPL] = myBase[j];
            if ((lane & 31) == 0) {
                depths[i >> 5] = who != kHere ? -who : (slot >= 0 ? a.deepKeep : depth); /* < 0: not walked by this stream */
                deepSlots[i >> 5] = who != kHere ? -2 : slot; /* -1: finished here (staged or stored); >= 0: the deep kernels finish it */
            }
        }
        if (XTILE && !PLAN && myLen[j] > 0) {
            const int f = first - a.baseIndex, l = last - a.baseIndex;
            const int low = f < l ? f : l, high = f < l ? l : f;
            mine.lowest = low < mine.lowest ? low : mine.lowest;
            mine.highest = high > mine.highest ? high : mine.highest;
            mine.middles += ((long long)f + l) >> 1;
            mine.rows += 1;
        }
    }
    if constexpr (XTILE && !PLAN) {
        mine.lowest = waveReduce(mine.lowest, MinOf{});
        mine.highest = waveReduce(mine.highest, MaxOf{});
        mine.rows = waveReduce(mine.rows, SumOf{});
        mine.middles = (long long)waveSumExact((double)mine.middles); /* (a lane's sum of up to 4 column numbers: exact as a double) */
        if (lane == 0)
            seen[wave] = mine;
    }
    SPGPU_STAMP(4);
    if (threadIdx.x == 0)
        nextItem = WAVES; /* the first WAVES sub-groups are dealt out statically */
    if constexpr (ZSTAGE) { /* every wavefront's lowest destination: a word each, met behind the prologue's one barrier */
        int low = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int i = threadIdx.x + j * BLOCK;
            if (i < ROWS && blockRow0 + i < a.rows && stagedHere[j])
                low = myDest[j] < low ? myDest[j] : low;
        }
        low = waveReduce(low, MinOf{});
        if (lane == 0)
            waveLowestDest[wave] = low;
    }
    __syncthreads();
    int zBase = 0;
    /* Where each row's result will wait (ZSTAGE).  Nothing needs this before the end of a wavefront's first item, and the
     * tile's barrier lies between: it is done while the tile and the first stages are on their way (round 4: it used to stand
     * here, 2.4 us of LDS traffic between the lengths' arrival and the first request of the stream). */
    auto stageDestinations = [&]() {
        if constexpr (ZSTAGE) {
            static_assert(WAVES <= 16, "one DPP row holds the wavefronts' words");
            zBase = waveReduce(waveLowestDest[lane < WAVES ? lane : 0], MinOf{}); /* one LDS read per lane, the minimum by DPP */
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = threadIdx.x + j * BLOCK;
                if (i < ROWS) {
                    const long long off = (long long)myDest[j] - zBase;
                    /* (rows of a deep sub-group are finished by the deep kernels: nothing of theirs is staged) */
                    const bool in = blockRow0 + i < a.rows && off >= 0 && off < ZW && stagedHere[j];
                    destOffset[i] = in ? (unsigned short)off : (unsigned short)0xFFFF;
                    if (in)
                        atomicOr(&stagedMask[off >> 5], 1u << (off & 31));
                }
            }
        }
    };
    if (!XTILE || !a.stageLate)
        stageDestinations();

    SPGPU_STAMP(5);
    /* ---- the per-sub-group state of a lane, and the stage loads --------------------------------------------------- */
    /* SPLIT (a.split > 0 columns): after the ordering every window has one sub-group some 250 columns deep at its head; walked
     * by one wavefront that is ~21 stages, 90 us, while the other seven finish the window's remaining 63 sub-groups in ~50.  A
     * sub-group deeper than a.split (and not deeper than deepCap: at most a few chunks) is therefore cut into chunks of a.split
     * columns, each an item of its own; the chunk sums wait behind the x tile and are added in chunk order when the queue is
     * empty (orc_?spmv_deep, mainChunk).  Every wavefront works the table out for itself, a lane per sub-group. */
    const int split = a.split;
    const int myDepthHere = lane < SUBS ? depths[lane] : 0;
    const bool mySplit = split > 0 && myDepthHere > split && myDepthHere <= a.deepCap;
    const int myChunks = lane < SUBS ? (mySplit ? (myDepthHere + split - 1) / split : ((BEHIND && myDepthHere < 0) ? 0 : 1)) : 0;
    /* inclusive prefix sums over the lanes by counting bits: a sub-group has at most kRaggedMostChunks chunks, "lane l has at least
     * K chunks" is one ballot per K, and the bits of it below a lane are one mbcnt (no cross-lane data movement at all) */
    int itemIncl = myChunks, parkIncl = mySplit ? myChunks : 0, totalItems = 0, totalParks = 0;
    {
        constexpr int MOST = kRaggedMostChunks<T> > 1 ? kRaggedMostChunks<T> : 1;
#pragma unroll
        for (int K = 1; K <= MOST; ++K) {
            const unsigned long long has = __ballot(myChunks >= K), parks = __ballot(mySplit && myChunks >= K);
            itemIncl += bitsBelowLane(has);
            parkIncl += bitsBelowLane(parks);
            totalItems += __popcll(has);
            totalParks += __popcll(parks);
        }
    }
    const int tileRoom = TILE_ELEMS - totalParks * 32; /* >= 0: the host sized a.split for it (launchRagged) */
    T* const parked = tile + (tileRoom > 0 ? tileRoom : 0);
    if (lane < SUBS && wave == 0) /* (for the combine behind the stream, past a barrier) */
        subFacts[lane] = int2{itemIncl - myChunks, mySplit ? parkIncl - myChunks : -1};
    /* first parked chunk sum + 1 (0: one chunk) in the top 10 bits, the walked depth below: what loadItem asks lane s for */
    static_assert(SUBS * kRaggedMostChunks<T> < 1023, "a park number fits 10 bits");
    const int parkAndDepth = (int)(((unsigned)(mySplit ? parkIncl - myChunks + 1 : 0) << 22) |
                                   (unsigned)(myDepthHere < 0x3FFFFF ? ((BEHIND && myDepthHere < 0) ? 0 : myDepthHere) : 0x3FFFFF));
    const unsigned long long splitOnes = __ballot(mySplit);

    struct Item {
        long long slab; /* first slot of this lane's strip */
        int len[RPL];   /* cut at the end of the chunk */
        int longest;    /* of this lane's rows */
        int kEnd;       /* end of the chunk (wave-uniform) */
        int s;          /* sub-group */
        int park;       /* where the chunk sum waits, or -1: the sub-group is one item and finished on the spot */
    };
    struct Stage {
        Pack<T, RPL> v[UNROLL];
        Pack<ColumnWord, RPL> c[UNROLL];
    };
    /* item -> (sub-group, chunk); returns the chunk's first column */
    auto loadItem = [&](int item, Item& it) -> int {
        /* the lane's strip number, opaque here: the LDS addresses and the row pointer built from it below are loop-invariant
         * per lane, and hoisted out of the stream loop they were kept in registers the loop does not have -- spilled, and
         * re-read from scratch with a vmcnt(0) wait (which drains the prefetch) at every item */
        int strip = sub;
        asm volatile("" : "+v"(strip));
        const int s = __popcll(__ballot(lane < SUBS && itemIncl <= item));
        /* lane s holds the sub-group's facts: read from its registers (v_readlane), not from LDS */
        const int itemFirst = s > 0 ? __builtin_amdgcn_readlane(itemIncl, s - 1) : 0;
        const int packedFacts = __builtin_amdgcn_readlane(parkAndDepth, s);
        const int chunk = item - itemFirst;
        const int parkFirst = (int)((unsigned)packedFacts >> 22) - 1;
        const int depth = packedFacts & 0x3FFFFF;
        it.s = s;
        it.park = parkFirst >= 0 ? parkFirst + chunk : -1;
        it.kEnd = parkFirst >= 0 && (chunk + 1) * split < depth ? (chunk + 1) * split : depth;
        it.slab = (long long)bases[s * LPC + strip];
        it.longest = 0;
        unsigned short stripLens[RPL]; /* the strip's RPL lengths in ONE LDS read */
        __builtin_memcpy(stripLens, &lens[s * 32 + strip * RPL], sizeof(stripLens));
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            it.len[t] = stripLens[t];
            if (it.len[t] == 0xFFFF) { /* see lens: not cut (the sub-group has no deep slot), so the row's own length */
                const long long r = blockRow0 + s * 32 + strip * RPL + t;
                it.len[t] = a.rS ? a.rS[r] : a.maxNnz;
            }
            it.len[t] = it.len[t] < it.kEnd ? it.len[t] : it.kEnd;
            it.longest = it.len[t] > it.longest ? it.len[t] : it.longest;
        }
        return parkFirst >= 0 ? chunk * split : 0;
    };
    auto fetch = [&](const Item& it, int kBase, Stage& st) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u * PH + phase;
            if (k < it.longest) {
                st.v[u] = loadPack<true, T, RPL>(a.cM + it.slab + (long long)k * a.valStride);
                if constexpr (PACKED)
                    st.c[u] = loadPack<true, unsigned short, RPL>(a.planPacked + it.slab + (long long)k * a.idxStride);
                else
                    st.c[u] = loadPack<true, int, RPL>(a.rP + it.slab + (long long)k * a.idxStride);
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    st.v[u].v[t] = zeroOf<T>();
                    st.c[u].v[t] = PACKED ? (ColumnWord)0 : (ColumnWord)a.baseIndex;
                }
            }
        }
    };

    /* The stream of stages a wavefront walks: the stages of its first sub-group, then of the sub-groups it takes from the
     * queue.  A fetch cursor runs AHEAD stages in front of the stage being consumed, so that AHEAD stage loads (3 KiB
     * each for 8-byte elements at UNROLL 2 ... 6 KiB at 4) are in flight per wavefront: with one stage ahead a
     * wavefront waits a full memory round trip per stage (measured: 1.25 GB/s per wavefront, 10 GB/s per workgroup).
     * A ring slot holds a stage and what consuming it needs to know. */
    constexpr int AHEAD = 2;
    struct Slot {
        Stage st;
        int len[RPL];
        int s;      /* sub-group (< 0: nothing, the stream has ended) */
        int park;
        int kBase;
        bool last;  /* last stage of its item */
    };
    auto grab = [&]() -> int {
        int got = 0;
        if (lane == 0)
            got = atomicAdd(&nextItem, 1);
        return __builtin_amdgcn_readfirstlane(got);
    };
    /* fetch cursor */
    int fItem = wave, fk = 0, fThen = totalItems; /* fThen: taken one item ahead (after the prologue's barrier) */
    Item fit;
    if (fItem < totalItems)
        fk = loadItem(fItem, fit);
    auto fetchNext = [&](Slot& slot) {
        slot.s = -1;
        slot.last = false;
        if (fItem < totalItems) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                slot.len[t] = fit.len[t];
            slot.s = fit.s;
            slot.park = fit.park;
            slot.kBase = fk;
            fetch(fit, fk, slot.st);
            fk += STEP;
            slot.last = fk >= fit.kEnd;
            if (slot.last) { /* wavefront-uniform: on to the next item */
                fItem = fThen;
                if (fItem < totalItems)
                    fk = loadItem(fItem, fit);
                fThen = fItem < totalItems ? grab() : totalItems;
            }
        }
    };
    Slot ring[AHEAD + 1];

    fThen = grab();
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        fetchNext(ring[i]); /* on their way while the tile is being placed and filled */

    SPGPU_STAMP(6);
    /* ---- the slice of x (round trip 3; the stages requested just above travel with it) ---------------------------- */
    int tileBase = 0;
    unsigned tileCount = 0;
    int packBase = 0;       /* PACKED: the column the block's 16-bit words count from, */
    unsigned packDelta = 0; /* and the same minus the tile's base: word + packDelta = position in the tile */
    if constexpr (XTILE) {
        /* the wavefronts' probes meet: a lane reads ONE of them, the rest is DPP (every lane reading all WAVES of them was
         * 24 LDS reads per lane behind the neighbour's traffic) */
        ColumnProbe all;
        if constexpr (PLAN) {
            all = ColumnProbe{planned.lowest, planned.highest, planned.probed, (long long)planned.middle * planned.probed};
        } else {
            const ColumnProbe one = lane < WAVES ? seen[lane] : ColumnProbe{0x7fffffff, -0x7fffffff - 1, 0, 0};
            all.lowest = waveReduce(one.lowest, MinOf{});
            all.highest = waveReduce(one.highest, MaxOf{});
            all.rows = waveReduce(one.rows, SumOf{});
            all.middles = (long long)waveSumExact((double)one.middles);
        }
        if (all.rows > 0 && all.lowest >= 0) {
            const long long span = (long long)all.highest - all.lowest + 1;
            if (span <= tileRoom) {
                tileBase = all.lowest;
                tileCount = (unsigned)span;
            } else if (tileRoom > 0) {
                long long start = all.middles / all.rows - tileRoom / 2;
                start = start < all.lowest ? all.lowest : start;
                start = start + tileRoom > (long long)all.highest + 1 ? (long long)all.highest + 1 - tileRoom : start;
                tileBase = (int)start;
                tileCount = (unsigned)tileRoom;
            }
        }
        if constexpr (PACKED) {
            packBase = planned.packBase;
            packDelta = (unsigned)packBase - (unsigned)tileBase;
        }
        constexpr int PIECE = 16 / (int)sizeof(T);
        constexpr int ROUND = (TILE_ELEMS / PIECE + BLOCK - 1) / BLOCK; /* 16-byte pieces per lane: all in flight at once */
        const T* __restrict__ from = x + tileBase;
        const unsigned pieces = tileCount / PIECE;
        Pack<T, PIECE> w[ROUND];
#pragma unroll
        for (int q = 0; q < ROUND; ++q)
            if (threadIdx.x + q * BLOCK < pieces)
                w[q] = loadPackElementAligned<T, PIECE>(from + (size_t)(threadIdx.x + q * BLOCK) * PIECE);
        if (a.stageLate)
            stageDestinations(); /* under the tile's and the first stages' round trip */
#pragma unroll
        for (int q = 0; q < ROUND; ++q)
            if (threadIdx.x + q * BLOCK < pieces)
                storePack<T, PIECE>(tile + (size_t)(threadIdx.x + q * BLOCK) * PIECE, w[q]);
        if (pieces * PIECE + threadIdx.x < tileCount)
            tile[pieces * PIECE + threadIdx.x] = from[pieces * PIECE + threadIdx.x];
        __syncthreads();
    }

#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && threadIdx.x == 0)
        spgpuTraceBuffer[8 * (size_t)blockIdx.x + 2] = wall_clock64(); /* tile in place */
#endif
    /* ---- 4: the stage stream -------------------------------------------------------------------------------------- */
    const bool hasBeta = isNotZero(a.beta);
    T sum[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t)
        sum[t] = zeroOf<T>();

    /* the sum of row `rowInSub` of sub-group s over the columns walked here is complete */
    auto finishRow = [&](int s, int rowInSub, T rowSum, int knownOffset = -1) { /* knownOffset: the row's destOffset if the caller has read it */
        const long long r = blockRow0 + s * 32 + rowInSub;
        if (r >= a.rows)
            return;
        const int deepSlot = (DEEP && !PLAN) ? deepSlots[s] : -1;
        if (deepSlot >= 0) {
            a.deepPartials[(size_t)deepSlot * 32 + (size_t)rowInSub] = rowSum; /* the deep kernels finish the row */
        } else if constexpr (ZSTAGE) {
            const unsigned off = knownOffset >= 0 ? (unsigned)knownOffset : destOffset[s * 32 + rowInSub];
            if (off != 0xFFFFu) {
                staged[off] = mul(a.alpha, rowSum); /* beta * y joins when the line is written */
            } else { /* beyond the staging buffer: the scattered store, its destination from global memory */
                const int outRow = a.rIdx ? a.rIdx[r] : (int)r;
                a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, rowSum, a.beta, a.y[outRow]) : epilogue<false>(a.alpha, rowSum, a.beta, zeroOf<T>());
            }
        } else {
            const int outRow = dests[s * 32 + rowInSub];
            a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, rowSum, a.beta, a.y[outRow]) : epilogue<false>(a.alpha, rowSum, a.beta, zeroOf<T>());
        }
    };
    /* consume the stage in `cur`; request the stage AHEAD further on into `refill` (the slot consumed last) */
    auto step = [&](Slot& cur, Slot& refill) {
        T xv[UNROLL][RPL];
        bool use[UNROLL][RPL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = cur.kBase + u * PH + phase;
            if constexpr (PACKED) {
                bool outside = false;
                unsigned at[RPL];
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const unsigned word = cur.st.c[u].v[t];
                    use[u][t] = k < cur.len[t];
                    at[t] = word + packDelta; /* packBase + word - tileBase */
                    const bool inside = at[t] < tileCount && word != 0xFFFFu;
                    outside |= use[u][t] && !inside;
                    xv[u][t] = tile[inside ? at[t] : 0u];
                }
                if (__ballot(outside) != 0ull) { /* beside the tile, or beyond the 16 bits: rare, and the one place that asks rP */
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        const unsigned word = cur.st.c[u].v[t];
                        if (use[u][t] && !(at[t] < tileCount && word != 0xFFFFu)) {
                            int col = packBase + (int)word;
                            if (word == 0xFFFFu)
                                col = a.rP[(long long)bases[cur.s * LPC + sub] + (long long)k * a.idxStride + t] - a.baseIndex;
                            use[u][t] = col >= 0;
                            xv[u][t] = x[col >= 0 ? col : 0];
                        }
                    }
                }
            } else if constexpr (XTILE) {
                bool outside = false;
                unsigned at[RPL];
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = cur.st.c[u].v[t] - a.baseIndex;
                    use[u][t] = k < cur.len[t] && col >= 0;
                    at[t] = (unsigned)(col - tileBase);
                    const bool inside = at[t] < tileCount;
                    outside |= use[u][t] && !inside;
                    xv[u][t] = tile[inside ? at[t] : 0u];
                }
                if (__ballot(outside) != 0ull) {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        if (use[u][t] && at[t] >= tileCount)
                            xv[u][t] = x[cur.st.c[u].v[t] - a.baseIndex];
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = cur.st.c[u].v[t] - a.baseIndex;
                    use[u][t] = k < cur.len[t] && col >= 0;
                    xv[u][t] = x[use[u][t] ? col : 0];
                }
            }
        }
        const int s = cur.s, park = cur.park;
        const bool last = cur.last;
        fetchNext(refill); /* behind the x reads in issue order */
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = pick(use[u][t], mulAdd(cur.st.v[u].v[t], xv[u][t], sum[t]), sum[t]);
        }
        if (last) { /* wavefront-uniform: the item is complete */
            /* the PH phase sums of a row meet in phase 0: pairwise, own + partner, the order of the lane-xor tree */
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                if constexpr (LPC <= 8)
                    sum[t] = add(sum[t], partnerOf<8>(sum[t]));
                if constexpr (LPC <= 16)
                    sum[t] = add(sum[t], partnerOf<16>(sum[t]));
                sum[t] = add(sum[t], partnerOf<32>(sum[t]));
            }
            if (phase == 0) {
                /* the strip's staged offsets in ONE LDS read (16 bits each, RPL of them side by side) */
                unsigned short offs[RPL];
                if constexpr (ZSTAGE)
                    __builtin_memcpy(offs, &destOffset[s * 32 + sub * RPL], sizeof(offs));
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    if (park >= 0)
                        parked[park * 32 + sub * RPL + t] = sum[t]; /* a chunk of a split sub-group: combined below */
                    else
                        finishRow(s, sub * RPL + t, sum[t], ZSTAGE ? (int)offs[t] : -1);
                }
            }
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = zeroOf<T>();
        }
    };
    static_assert(AHEAD == 2, "the rotation below is written out for a ring of three");
    for (;;) { /* the ring rotates by name, not by copying registers */
        if (ring[0].s < 0) break;
        step(ring[0], ring[2]);
        if (ring[1].s < 0) break;
        step(ring[1], ring[0]);
        if (ring[2].s < 0) break;
        step(ring[2], ring[1]);
    }
    if (splitOnes != 0ull) { /* workgroup-uniform (every wavefront computed the same table) */
        __syncthreads();     /* every chunk sum is in LDS */
        const int splitCount = __popcll(splitOnes);
        for (int q0 = wave * 2; q0 < splitCount; q0 += WAVES * 2) { /* a half-wave per split sub-group */
            const int q = q0 + (lane >> 5);
            unsigned long long rest = splitOnes;
            for (int skip = 0; skip < (q < splitCount ? q : 0); ++skip)
                rest &= rest - 1;
            const int s = __ffsll((long long)rest) - 1;
            if (q < splitCount) {
                const int first = subFacts[s].y, count = (depths[s] + split - 1) / split;
                T total = parked[first * 32 + (lane & 31)];
                for (int c = 1; c < count; ++c)
                    total = add(total, parked[(first + c) * 32 + (lane & 31)]);
                finishRow(s, lane & 31, total);
            }
        }
    }
    if constexpr (ZSTAGE) {
        __syncthreads(); /* every staged sum is in LDS */
        for (int e = threadIdx.x; e < ZW; e += BLOCK) {
            if ((stagedMask[e >> 5] >> (e & 31)) & 1u) { /* consecutive lanes, consecutive destinations: whole lines where the workgroup holds them */
                const long long outRow = (long long)zBase + e;
                const T v = staged[e];
                a.z[outRow] = hasBeta ? mulAdd(a.beta, a.y[outRow], v) : v;
            }
        }
    }
    if constexpr (BEHIND) {
        /* sub-groups deeper than the cap that nobody else takes (stale or absent plan, full list): whole, by this workgroup, in batches */
        unsigned long long orphans = __ballot(lane < SUBS && depths[lane < SUBS ? lane : 0] == -kBehind);
        while (orphans != 0ull) { /* workgroup-uniform: every wavefront read the same table */
            const unsigned long long batch = orphans;
            int count = __popcll(batch);
            count = count < kPlanDeepMost ? count : kPlanDeepMost;
            const int firstSub = (int)logicalBlock * SUBS;
            wholeSubgroups<T, RPL, IS_HELL, WAVES, TILE_ELEMS * (int)sizeof(T)>(a, tile, count, [&](int j) {
                unsigned long long rest = batch;
                for (int skip = 0; skip < j; ++skip)
                    rest &= rest - 1;
                return firstSub + __ffsll((long long)rest) - 1;
            }, false);
            for (int done = 0; done < count; ++done)
                orphans &= orphans - 1;
        }
    }
#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && lane == 0)
        atomicMax(&spgpuTraceBuffer[8 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
}

#ifndef SPGPU_RAGGED_UNROLL
#define SPGPU_RAGGED_UNROLL(RPL) ((RPL) >= 4 ? 2 : 3) /* wave-wide loads per stage; 3 keeps the 8-byte kernels at 112 VGPRs (4 wavefronts per SIMD: two 8-wavefront workgroups per CU) with two stages in flight */
#endif
/* Shapes (SPGPU_RAGGED_SHAPE; 0 is the default): workgroup lanes / tile / sub-groups per workgroup. */
/* Returns true if the deep kernels have to follow (false: the launch runs the deep list's items itself). */
/* Columns per chunk of a split sub-group (raggedSpmvKernel, SPLIT): about 96 (8 stages of the 8-byte kernels), and large
 * enough that the chunk sums of a workgroup whose sub-groups are ALL deepCap deep -- the hacks of set-aside long rows -- fit
 * behind the x tile, which is of no use to such a workgroup anyway.  The SAME value for every shape and form (the sums must not
 * depend on which of them AUTO takes): tests/oracle_api.py ragged_split restates this. */
template <typename T> static int raggedSplit(int deepCap, int step, int asked)
{
    constexpr int most = kRaggedMostChunks<T>;
    if (most < 2 || deepCap <= 0 || asked == 0)
        return 0;
    const int want = ((asked > 0 ? asked : 96) + step - 1) / step * step;
    const int need = ((deepCap + most - 1) / most + step - 1) / step * step;
    return want > need ? want : need;
}

template <typename T, int RPL, bool IS_HELL, bool DEEP>
static bool launchRagged(hipStream_t stream, const SlabArgs<T>& in, int shape, bool tiled)
{
    constexpr int UNROLL = SPGPU_RAGGED_UNROLL(RPL);
    SlabArgs<T> a = in;
    a.split = raggedSplit<T>(a.deepCap, (kWave / (32 / RPL)) * UNROLL, spgpuTuning()->raggedSplit);
    const long long subs = ((long long)a.rows + 31) / 32;
#define SPGPU_RAGGED(WAVES, TILE, SUBS)                                                                               \
    hipLaunchKernelGGL((raggedSpmvKernel<T, RPL, IS_HELL, UNROLL, WAVES, TILE, SUBS, DEEP>),                          \
                       dim3((unsigned)((subs + (SUBS) - 1) / (SUBS))), dim3((WAVES) * kWave), 0, stream, a)
#define SPGPU_RAGGED_Z(WAVES, TILE, SUBS, ZB)                                                                         \
    hipLaunchKernelGGL((raggedSpmvKernel<T, RPL, IS_HELL, UNROLL, WAVES, TILE, SUBS, DEEP, ZB>),                      \
                       dim3((unsigned)((subs + (SUBS) - 1) / (SUBS))), dim3((WAVES) * kWave), 0, stream, a)
    if (!tiled) {
        SPGPU_RAGGED(4, 0, 16);
        return true;
    }
    switch (shape) {
#ifdef SPGPU_TUNING_VARIANTS
    case 1: SPGPU_RAGGED(8, 65536, 64); break;
    case 2: SPGPU_RAGGED(4, 49152, 32); break;
    case 3: SPGPU_RAGGED(4, 32768, 16); break;
#endif
    case 4: /* 2 048 rows per workgroup, results staged by destination */
    case 5: /* 1 024 rows, staged */
        if constexpr (sizeof(T) <= 8) { /* (16-byte elements: tile + staging leave room for one workgroup per CU -- the default shape) */
            if (shape == 4)
                SPGPU_RAGGED_Z(8, 49152, 64, 17408);
            else
                SPGPU_RAGGED_Z(8, 49152, 32, 17408);
            break;
        }
        [[fallthrough]];
    default: SPGPU_RAGGED(8, 65536, 32); break;
    }
    return true;
#undef SPGPU_RAGGED
#undef SPGPU_RAGGED_Z
}
