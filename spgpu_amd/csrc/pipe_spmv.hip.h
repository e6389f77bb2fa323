/*
 * ELL / HELL SpMV for matrices whose rows were ORDERED BY LENGTH (rIdx given: spgpuOellOrderDevice, ellToOell) --
 * the north_star target, power-law row lengths.  Included by ellpack_spmv.hip (namespace spgpu, after SlabArgs).
 * ONE launch, no state outside the kernel's own LDS: nothing is shared between two calls in flight.
 *
 * What bounds this path is not the stream rate of a wavefront but the time a workgroup spends NOT streaming: finding its
 * rows, their lengths, the slice of x they touch, filling the LDS tile -- three to five dependent memory round trips,
 * 10-18 us of a 35-45 us workgroup life (profiles/r02b_ragged_workgroup_trace.txt, r03_share_trace.txt), during which half
 * of a CU's wavefront slots hold nothing in flight.  Here that work runs BESIDE the stream:
 *
 *   - one workgroup of WAVES wavefronts per CU, resident for the whole launch.  The matrix is cut into G*J ranges of equal
 *     work -- range r holds the hacks h with r*Q <= M(h) < (r+1)*Q,  M(h) = hackOffsets[h] + kRowCost*hackSize*h -- and
 *     workgroup w owns the ranges w, w+G, w+2G, ...: a static assignment (nothing is shared between workgroups), but
 *     interleaved, so that a stretch of the matrix that streams slower than its slots suggest (the hacks of set-aside
 *     long rows at the head of the order, whose x values come from all over the matrix) is spread over all workgroups
 *     instead of making a few of them the stragglers the launch waits for.  All 2J boundaries of a workgroup are found at
 *     its start by ONE two-level search over hackOffsets (every lane probes one of BLOCK evenly spaced hacks; then the
 *     brackets that contain the boundaries are read): two dependent round trips.  Layouts without hackOffsets take
 *     ranges of equal row counts.
 *   - the range is walked in BLOCKS of up to 64 sub-groups (32 rows each).  The last wavefront is the SCOUT: it prepares
 *     block b+1 in the second set of LDS buffers -- row lengths, slab bases, the (sub-group, chunk) item table, the column
 *     window, the x tile (global_load_lds: no registers) -- while the other wavefronts stream block b.  Blocks are handed
 *     over through LDS words (ready / left counters); there is no workgroup barrier after the search.
 *   - the STREAMERS take (sub-group, chunk) items from the block's queue exactly as in shareSpmvKernel; a wavefront's fetch
 *     cursor runs two stages ahead and crosses into the next block when that is ready, so the stream does not drain at a
 *     block boundary.  The wavefront that completes the last item of a block adds the chunk sums of its deep sub-groups
 *     in chunk order (orc_?spmv_deep with deepCap = deepChunk = CHUNK: the same bits as shareSpmvKernel).
 *   - consecutive blocks of a workgroup read overlapping slices of x through ONE L2, and write z lines of one window.
 *
 * Every wait spins on an LDS word written by a wavefront of the same (resident) workgroup that is never itself waiting
 * for the spinner: the scout only waits for streamers to leave a block they can finish, streamers only wait for the scout.
 */

constexpr int kPipeRanges = 32; /* ranges a workgroup can own */
constexpr int kSpinLimit = 1 << 23; /* polls of an LDS word (~0.1 us each) before a wavefront gives up: a second */

template <typename T, int RPL, bool IS_HELL, bool BY_WORK, int UNROLL, int WAVES, int TILE_BYTES, bool XTILE, int CHUNK_STAGES, int AHEAD>
__global__ __launch_bounds__(WAVES * kWave) void pipeSpmvKernel(const SlabArgs<T> a)
{
    constexpr int LPC = 32 / RPL;   /* lanes per slab column of a sub-group */
    constexpr int PH = kWave / LPC; /* slab columns per wave-wide load */
    constexpr int STEP = PH * UNROLL;
    constexpr int CHUNK = STEP * CHUNK_STAGES;
    constexpr int BLOCK = WAVES * kWave;
    constexpr int STREAMERS = WAVES - 1;
    constexpr int MAXSUBS = 64; /* sub-groups of one block: a lane each in the scout's scans */
    constexpr int MAXROWS = MAXSUBS * 32;
    constexpr int TILE_ELEMS = TILE_BYTES / (int)sizeof(T);
    constexpr int PMAX = TILE_ELEMS / 32 / 2; /* chunk sums (32 values each) a block may park in its tile buffer */
    static_assert(PMAX >= 2, "room for the chunk sums of a two-chunk sub-group");
    static_assert(!BY_WORK || IS_HELL, "ranges of equal work are cut along hackOffsets");

    struct BlockInfo {
        long long row0;    /* first row */
        int items;         /* (sub-group, chunk) items; < 0: no more blocks */
        int used;          /* sub-groups */
        int tileRoom;      /* elements of the buffer the tile may use; the chunk sums sit behind */
        int tileBase;
        unsigned tileCount;
        int chunksDone;    /* of the first sub-group, in earlier blocks */
        int cutShort;      /* the first sub-group does not finish in this block */
        unsigned long long parking; /* sub-groups whose chunk sums go through LDS */
    };
    __shared__ __attribute__((aligned(16))) T buffers[2][TILE_ELEMS]; /* x tile from the front, chunk sums from the back */
    __shared__ T carry[32];                                          /* a sub-group cut by a block boundary */
    __shared__ int lens[2][MAXROWS];
    __shared__ int dests[2][MAXROWS]; /* rIdx of the block's rows: a streamer that fetched it from global memory at the end of an
                                         item would wait for it with vmcnt(0) -- counters retire in order -- and drain its own prefetch */
    __shared__ unsigned bases[2][BY_WORK ? MAXSUBS : MAXROWS / RPL]; /* first slot of a sub-group / of an RPL-row strip */
    __shared__ int4 subFacts[2][MAXSUBS]; /* first item, first chunk sum or -1, depth, chunks of this block */
    __shared__ BlockInfo infos[2];
    __shared__ int control[16]; /* 4,5 item queues; 6,7 items completed; 8,9 ready; 10,11 left; 12 folded */
    __shared__ int searchCount[4 * kPipeRanges]; /* probes below every boundary; hacks below it inside its bracket */
    __shared__ int rangeCut[2 * kPipeRanges];    /* first and end sub-group of this workgroup's ranges */

    const T* __restrict__ x = a.x;
    const long long totalSubs = ((long long)a.rows + 31) / 32;
#ifdef SPGPU_TRACE_BLOCKS
    /* experiment builds only: start / end of every workgroup (100 MHz wall clock) and the polls its streamers spent
     * waiting for the scout, into a caller-provided buffer */
    if (spgpuTraceBuffer && threadIdx.x == 0)
        spgpuTraceBuffer[3 * (size_t)blockIdx.x] = wall_clock64();
#endif

    /* ---- which sub-groups does this workgroup own? (all wavefronts, once) ------------------------------------------- */
    /* ranges blockIdx.x + j * gridDim.x, j < perGroup; rangeCut[2j], rangeCut[2j+1] = first and end sub-group of range j */
    const int perGroup = a.pipeRanges < 1 ? 1 : (a.pipeRanges > kPipeRanges ? kPipeRanges : a.pipeRanges);
    if (threadIdx.x < 16)
        control[threadIdx.x] = 0;
    if (threadIdx.x < 4 * kPipeRanges)
        searchCount[threadIdx.x] = 0;
    if constexpr (BY_WORK) {
        const int lane = threadIdx.x & (kWave - 1);
        const long long hs = a.hackSize, hacks = ((long long)a.rows + hs - 1) / hs, perHack = hs / 32;
        const long long rowCost = (long long)kRowCost * hs;
        /* the last hack's own slots are left out of the total (hackOffsets has no closing entry): every M(h) is below it */
        const long long total = (long long)a.hackOffsets[hacks - 1] + rowCost * hacks;
        const long long ranges = (long long)gridDim.x * perGroup;
        const long long quota = (total + ranges - 1) / ranges;
        const long long stride = (hacks + BLOCK - 1) / BLOCK;
        const long long probe = (long long)threadIdx.x * stride;
        const long long mine = probe < hacks ? (long long)a.hackOffsets[probe] + rowCost * probe : 0x7fffffffffffffffll;
        auto target = [&](int t) -> long long { /* boundary t: start (even t) or end (odd t) of the workgroup's range t / 2 */
            return ((long long)blockIdx.x + (long long)(t >> 1) * gridDim.x + (t & 1)) * quota;
        };
        __syncthreads();
        for (int t = 0; t < 2 * perGroup; ++t) { /* round trip 1 is in `mine`: how many probes lie below each boundary */
            const int below = __popcll(__ballot(mine < target(t)));
            if (lane == 0 && below > 0)
                atomicAdd(&searchCount[t], below);
        }
        __syncthreads();
        /* round trip 2: the hacks between the last probe below a boundary and the next probe, for all boundaries at once */
        const long long inner = stride - 1; /* hacks strictly between two probes */
        for (long long p = threadIdx.x; p < 2ll * perGroup * inner; p += BLOCK) {
            const int t = (int)(p / inner);
            const int probes = searchCount[t];
            const long long h = (long long)(probes - 1) * stride + 1 + p % inner;
            if (probes > 0 && h < hacks && (long long)a.hackOffsets[h] + rowCost * h < target(t))
                atomicAdd(&searchCount[2 * kPipeRanges + t], 1);
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * perGroup) {
            const int t = threadIdx.x, probes = searchCount[t];
            const long long hack = probes > 0 ? (long long)(probes - 1) * stride + 1 + searchCount[2 * kPipeRanges + t] : 0;
            rangeCut[t] = (int)(hack * perHack < totalSubs ? hack * perHack : totalSubs);
        }
    } else {
        if ((int)threadIdx.x < 2 * perGroup) {
            const long long ranges = (long long)gridDim.x * perGroup;
            const long long per = (totalSubs + ranges - 1) / ranges;
            const long long r = (long long)blockIdx.x + (long long)(threadIdx.x >> 1) * gridDim.x + (threadIdx.x & 1);
            rangeCut[threadIdx.x] = (int)(r * per < totalSubs ? r * per : totalSubs);
        }
    }
    __syncthreads();

    volatile int* const flags = control;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    if (wave == STREAMERS) {
        /* ================================ the scout ==================================================================== */
        int range = 0;                                  /* of this workgroup */
        long long blockFirst = rangeCut[0], rangeEnd = rangeCut[1]; /* first sub-group of the block being prepared */
        int chunksDone = 0;                             /* of that sub-group, in earlier blocks */
        for (int blk = 0;; ++blk) {
            unsigned tid = threadIdx.x; /* opaque per block: keeps lane-derived values out of registers held across the loop */
            asm volatile("" : "+v"(tid));
            const int lane = tid & (kWave - 1);
            const int buf = blk & 1;
            /* the buffers of block blk-2 are free once every streamer has left it */
            for (int spins = 0; __builtin_amdgcn_readfirstlane(flags[10 + buf]) < STREAMERS * (blk / 2); ++spins) {
                if (spins > kSpinLimit) {
#ifdef SPGPU_TRACE_BLOCKS
                    if (spgpuTraceBuffer && lane == 0)
                        spgpuTraceBuffer[4096 + 32 * (size_t)blockIdx.x + 16] = (1ull << 40) | ((unsigned long long)flags[10 + buf] << 20) | (unsigned)blk;
#endif
                    return; /* cannot happen while the streamers make progress; never hang the device */
                }
#ifdef SPGPU_TRACE_BLOCKS
                if (spgpuTraceBuffer && lane == 0)
                    atomicAdd(&spgpuTraceBuffer[3 * (size_t)blockIdx.x + 2], 1ull << 32); /* the scout's polls: upper half */
#endif
                __builtin_amdgcn_s_sleep(4);
            }
            while (blockFirst >= rangeEnd && range + 1 < perGroup) { /* on to this workgroup's next range */
                ++range;
                blockFirst = __builtin_amdgcn_readfirstlane(rangeCut[2 * range]);
                rangeEnd = __builtin_amdgcn_readfirstlane(rangeCut[2 * range + 1]);
            }
            BlockInfo info;
            if (blockFirst >= rangeEnd) { /* no more blocks */
                if (lane == 0) {
                    infos[buf].items = -1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    flags[8 + buf] = blk + 1;
                }
                break;
            }
            const int candidates = rangeEnd - blockFirst < MAXSUBS ? (int)(rangeEnd - blockFirst) : MAXSUBS;
            const long long row0 = blockFirst * 32;

            /* Round trip 1 -- every dependent round trip of the scout costs ~5 us under the stream's load, and a block
             * streams in ~20, so each step has ALL its loads in flight at once: the row lengths (32 consecutive rows = 32
             * consecutive lanes; a lane holds up to 32 of them), slab bases. */
            constexpr int PER_LANE = MAXROWS / kWave; /* rows a lane looks at */
            int len[PER_LANE], dest[PER_LANE];
            unsigned stripBase[BY_WORK ? 1 : PER_LANE]; /* dead once written to LDS */
            unsigned myBase = 0; /* BY_WORK: lane s holds the first slot of sub-group s */
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                len[j] = dest[j] = 0;
                if (j * kWave < candidates * 32) { /* wavefront-uniform */
                    const int i = lane + j * kWave;
                    const long long r = row0 + i;
                    const bool live = i < candidates * 32 && r < a.rows;
                    len[j] = live ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
                    dest[j] = live && a.rIdx ? a.rIdx[r] : (int)r;
                    if constexpr (!BY_WORK) {
                        stripBase[j] = 0;
                        if (live) {
                            if constexpr (IS_HELL) {
                                const unsigned u0 = (unsigned)r, hs = (unsigned)a.hackSize;
                                stripBase[j] = (unsigned)a.hackOffsets[u0 / hs] + u0 % hs;
                            } else {
                                stripBase[j] = (unsigned)r;
                            }
                        }
                    }
                }
            }
            if constexpr (BY_WORK) { /* behind the loads above: its wait is theirs */
                if (lane < candidates) {
                    const long long s = blockFirst + lane, perHack = a.hackSize / 32;
                    myBase = (unsigned)a.hackOffsets[s / perHack] + (unsigned)(s % perHack) * 32u;
                }
            }
            int myDepth = 0; /* lane s: depth of sub-group s */
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                if (j * kWave < candidates * 32) {
                    const int i = lane + j * kWave;
                    lens[buf][i] = len[j];
                    dests[buf][i] = dest[j];
                    if constexpr (!BY_WORK) {
                        if (i % RPL == 0)
                            bases[buf][i / RPL] = stripBase[j];
                    }
                    int depth = len[j];
#pragma unroll
                    for (int m = 1; m < 32; m <<= 1) {
                        const int other = laneXor(depth, m);
                        depth = other > depth ? other : depth;
                    }
                    /* sub-groups 2j and 2j+1: hand their depths to the lanes of those numbers */
                    const int lowHalf = __shfl(depth, 0, kWave), highHalf = __shfl(depth, 32, kWave);
                    if (lane == 2 * j)
                        myDepth = lowHalf;
                    if (lane == 2 * j + 1)
                        myDepth = highHalf;
                }
            }
            if constexpr (BY_WORK) {
                if (lane < candidates)
                    bases[buf][lane] = myBase;
            }
            if (lane >= candidates)
                myDepth = 0;

            /* the block: chunks per sub-group, which sub-groups fit (their chunk sums have to find room behind the tile),
             * where each one's items and chunk sums start */
            const int myChunksAll = lane < candidates ? (myDepth + CHUNK - 1) / CHUNK + (myDepth == 0 ? 1 : 0) : 0;
            int myChunks = lane == 0 ? myChunksAll - chunksDone : myChunksAll;
            const bool myParks = myChunksAll > 1;
            bool cutShort = false;
            if (__shfl(myParks && myChunks > PMAX, 0, kWave)) {
                cutShort = true;
                myChunks = lane == 0 ? PMAX : 0;
            }
            int parkIncl = myParks ? myChunks : 0;
#pragma unroll
            for (int m = 1; m < kWave; m <<= 1) {
                const int other = __shfl_up(parkIncl, m, kWave);
                parkIncl += lane >= m ? other : 0;
            }
            const unsigned long long fits = __ballot(lane < candidates && parkIncl <= PMAX);
            const int used = cutShort ? 1 : (~fits == 0ull ? kWave : __ffsll((long long)~fits) - 1);
            if (lane >= used)
                myChunks = 0;
            int itemIncl = myChunks;
#pragma unroll
            for (int m = 1; m < kWave; m <<= 1) {
                const int other = __shfl_up(itemIncl, m, kWave);
                itemIncl += lane >= m ? other : 0;
            }
            info.row0 = row0;
            info.items = __builtin_amdgcn_readfirstlane(__shfl(itemIncl, kWave - 1, kWave));
            info.used = used;
            const int parked = __builtin_amdgcn_readfirstlane(__shfl(lane < used ? parkIncl : 0, used - 1, kWave));
            info.tileRoom = TILE_ELEMS - parked * 32;
            info.chunksDone = chunksDone;
            info.cutShort = cutShort ? 1 : 0;
            info.parking = __ballot(lane < used && myParks);
            subFacts[buf][lane] = int4{itemIncl - myChunks, myParks ? parkIncl - myChunks : -1, myDepth, myChunks};

            /* where is the slice of x?  first and last column of every row (the extremes of a row whose columns ascend; any
             * order is still correct: entries outside the tile are gathered from global memory) */
            info.tileBase = 0;
            info.tileCount = 0;
            if constexpr (XTILE) {
                /* Round trip 2: first and last column of every fourth row (the rows of a sub-group are sorted by length and
                 * come from one window of the order; lanes take turns so that every sub-group is sampled).  A row whose
                 * columns reach beyond the sampled ones' gathers those from global memory. */
                int lowest = 0x7fffffff, highest = -0x7fffffff - 1, counted = 0;
                long long middles = 0;
                constexpr int SAMPLES = PER_LANE / 4;
                int first[SAMPLES], last[SAMPLES];
                bool sampled[SAMPLES];
#pragma unroll
                for (int q = 0; q < SAMPLES; ++q) {
                    first[q] = last[q] = 0;
                    sampled[q] = false;
                    if (4 * q * kWave < used * 32) { /* wavefront-uniform */
                        const int i = lane + (4 * q + (lane & 3)) * kWave; /* of the lane's rows 4q .. 4q+3 the one numbered by the lane */
                        long long at;
                        if constexpr (BY_WORK)
                            at = (long long)__shfl(myBase, (i >> 5) & (kWave - 1), kWave) + (i & 31); /* every lane takes part in the shuffle */
                        else
                            at = (long long)bases[buf][(i & (MAXROWS - 1)) / RPL] + i % RPL;
                        const int lenOf = i < used * 32 ? lens[buf][i & (MAXROWS - 1)] : 0;
                        sampled[q] = lenOf > 0;
                        if (sampled[q]) {
                            first[q] = a.rP[at];
                            last[q] = a.rP[at + (long long)(lenOf - 1) * a.idxStride];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < SAMPLES; ++q) {
                    if (sampled[q]) {
                        const int f = first[q] - a.baseIndex, l = last[q] - a.baseIndex;
                        const int low = f < l ? f : l, high = f < l ? l : f;
                        lowest = low < lowest ? low : lowest;
                        highest = high > highest ? high : highest;
                        middles += ((long long)f + l) >> 1;
                        counted += 1;
                    }
                }
                lowest = waveMin(lowest);
                highest = waveMax(highest);
#pragma unroll
                for (int m = 1; m < kWave; m <<= 1) {
                    counted += laneXor(counted, m);
                    const int lowHalf = laneXor((int)(unsigned)(middles & 0xffffffffll), m);
                    const int highHalf = laneXor((int)(middles >> 32), m);
                    middles += ((long long)highHalf << 32) | (unsigned)lowHalf;
                }
                const int room = info.tileRoom;
                if (counted > 0 && lowest >= 0) {
                    const long long span = (long long)highest - lowest + 1;
                    if (span <= room) {
                        info.tileBase = lowest;
                        info.tileCount = (unsigned)span;
                    } else if (span <= 4ll * room) { /* rows from all over the matrix (a hack of set-aside long rows): no tile */
                        long long start = middles / counted - room / 2;
                        start = start < lowest ? lowest : start;
                        start = start + room > (long long)highest + 1 ? (long long)highest + 1 - room : start;
                        info.tileBase = (int)start;
                        info.tileCount = (unsigned)room;
                    }
                }
                /* straight from global memory into LDS (global_load_lds_dwordx4: one wave-wide instruction writes 1 KiB of
                 * LDS in lane order from 64 per-lane addresses), from a 16-byte aligned source: the tile starts a few
                 * elements early if it has to.  The ragged end goes through registers. */
                constexpr int PIECE = 16 / (int)sizeof(T);
                if constexpr (PIECE > 1) {
                    const int early = (int)(((uintptr_t)(x + info.tileBase) % 16) / sizeof(T));
                    if (early <= info.tileBase && info.tileCount > 0) {
                        info.tileBase -= early;
                        info.tileCount = info.tileCount + early <= (unsigned)room ? info.tileCount + early : (unsigned)room;
                    }
                }
                const T* __restrict__ from = x + info.tileBase;
                T* const tile = buffers[buf];
                const unsigned pieces = info.tileCount / PIECE;
                const bool direct = ((uintptr_t)from % 16) == 0;
                unsigned piece0 = 0;
                if (direct) {
                    for (; piece0 + kWave <= pieces; piece0 += kWave) {
#if defined(__HIP_DEVICE_COMPILE__) /* the host pass of hipcc parses the kernel body too and has no such builtin */
                        __builtin_amdgcn_global_load_lds(from + (size_t)(piece0 + lane) * PIECE, tile + (size_t)(piece0 + lane) * PIECE, 16, 0, 0);
#endif
                    }
                }
                for (unsigned piece = piece0 + lane; piece < pieces; piece += kWave) {
                    const Pack<T, PIECE> w = loadPackElementAligned<T, PIECE>(from + (size_t)piece * PIECE);
                    storePack<T, PIECE>(tile + (size_t)piece * PIECE, w);
                }
                for (unsigned e = pieces * PIECE + lane; e < info.tileCount; e += kWave)
                    tile[e] = from[e];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the tile has landed */
            }
            if (lane == 0) {
                infos[buf] = info;
                flags[4 + buf] = 0; /* the item queue */
                flags[6 + buf] = 0; /* items completed */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                flags[8 + buf] = blk + 1;
            }
            if (cutShort) {
                chunksDone += PMAX;
            } else {
                blockFirst += used;
                chunksDone = 0;
            }
        }
        return;
    }

    /* ==================================== the streamers ================================================================ */
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % LPC, phase = lane / LPC;
    const bool hasBeta = isNotZero(a.beta);
    auto finishRow = [&](int buf, int inBlock, T sum) { /* row inBlock of the block in buffer set buf */
        const int outRow = dests[buf][inBlock];
        a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, sum, a.beta, a.y[outRow]) : epilogue<false>(a.alpha, sum, a.beta, zeroOf<T>());
    };
    struct Item {
        long long slab; /* first slot of this lane's strip */
        int len[RPL];   /* cut at the end of the chunk */
        int longest;    /* of this lane's rows */
        int kEnd;       /* end of the chunk (wave-uniform) */
        int row;        /* first row of the sub-group, relative to the block */
        int park;       /* where the chunk sum goes, or -1: the sub-group has one chunk and is finished on the spot */
    };
    struct Stage {
        Pack<T, RPL> v[UNROLL];
        Pack<int, RPL> c[UNROLL];
    };
    /* A ring slot: a stage and what consuming it needs to know. */
    struct Slot {
        Stage st;
        int len[RPL];
        int kBase;
        int row;   /* -1: nothing (the cursor waits for the scout); -2: nothing, there are no more blocks */
        int park;
        int blk;   /* the block the stage belongs to */
        bool last; /* last stage of its item */
    };
    /* AHEAD stages are in flight per wavefront while one is consumed (template parameter: the fewer wavefronts a workgroup
     * has, the more registers each may spend on stages in flight) */

    /* ---- fetch cursor --------------------------------------------------------------------------------------------- */
    int cBlk = -1, cItems = 0, cUsed = 0, cChunksDone = 0; /* the block the cursor is in */
    bool cOpen = false;    /* its queue may still hold items (once it is empty this wavefront never touches it again: the
                              scout may be reusing the buffer for the block after next) */
    bool cEnded = false;   /* there are no more blocks */
    bool cStalled = false; /* the next block was not ready: nothing more is fetched until the ring has drained, so that
                              no stage is requested behind an empty slot */
    Item fit;
    int fk = 0;
    bool fHave = false; /* fit / fk describe an item with stages left */
    auto loadItem = [&](int buf, int item) -> int {
        /* item -> (sub-group, chunk): the sub-group is the number of sub-groups whose items end at or before it */
        const int4 mine = subFacts[buf][lane];
        const int s = __popcll(__ballot(lane < cUsed && mine.x + mine.w <= item));
        const int4 facts = subFacts[buf][s];
        const int inSub = item - __builtin_amdgcn_readfirstlane(facts.x);
        const int parkFirst = __builtin_amdgcn_readfirstlane(facts.y);
        const int depth = __builtin_amdgcn_readfirstlane(facts.z);
        const int chunk = inSub + (s == 0 ? cChunksDone : 0);
        if constexpr (BY_WORK)
            fit.slab = (long long)bases[buf][s] + sub * RPL;
        else
            fit.slab = bases[buf][s * LPC + sub];
        fit.kEnd = (chunk + 1) * CHUNK < depth ? (chunk + 1) * CHUNK : depth;
        fit.longest = 0;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            const int len = lens[buf][s * 32 + sub * RPL + t];
            fit.len[t] = len < fit.kEnd ? len : fit.kEnd;
            fit.longest = fit.len[t] > fit.longest ? fit.len[t] : fit.longest;
        }
        fit.row = s * 32;
        fit.park = parkFirst >= 0 ? parkFirst + inSub : -1;
        return chunk * CHUNK;
    };
    /* the next item of the cursor's block, or of the next block if that is ready; false: nothing for now (or ever) */
    auto nextItem = [&]() -> bool {
        for (;;) {
            if (cOpen) {
                int got = 0;
                if (lane == 0)
                    got = atomicAdd(&control[4 + (cBlk & 1)], 1);
                got = __builtin_amdgcn_readfirstlane(got);
                if (got < cItems) {
                    fk = loadItem(cBlk & 1, got);
                    return true;
                }
                cOpen = false;
            }
            const int nb = cBlk + 1;
            if (__builtin_amdgcn_readfirstlane(flags[8 + (nb & 1)]) != nb + 1)
                return false; /* not prepared yet */
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int items = __builtin_amdgcn_readfirstlane(infos[nb & 1].items);
            if (items < 0) {
                cEnded = true;
                return false;
            }
            cBlk = nb;
            cOpen = true;
            cItems = items;
            cUsed = __builtin_amdgcn_readfirstlane(infos[nb & 1].used);
            cChunksDone = __builtin_amdgcn_readfirstlane(infos[nb & 1].chunksDone);
        }
    };
    auto fetch = [&](int kBase, Stage& st) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u * PH + phase;
            if (k < fit.longest) {
                st.v[u] = loadPack<true, T, RPL>(a.cM + fit.slab + (long long)k * a.valStride);
                st.c[u] = loadPack<true, int, RPL>(a.rP + fit.slab + (long long)k * a.idxStride);
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    st.v[u].v[t] = zeroOf<T>();
                    st.c[u].v[t] = a.baseIndex;
                }
            }
        }
    };
    auto fetchNext = [&](Slot& slot) {
        slot.last = false;
        if (!fHave && !cEnded && !cStalled) {
            fHave = nextItem();
            cStalled = !fHave && !cEnded;
        }
        if (!fHave) {
            slot.row = cEnded ? -2 : -1;
            return;
        }
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            slot.len[t] = fit.len[t];
        slot.kBase = fk;
        slot.row = fit.row;
        slot.park = fit.park;
        slot.blk = cBlk;
        fetch(fk, slot.st);
        fk += STEP;
        slot.last = fk >= fit.kEnd;
        if (slot.last)
            fHave = false; /* the next call takes the next item */
    };

    /* ---- consumer state: the block whose stages are being consumed ---------------------------------------------------- */
    int uBlk = 0; /* blocks below this one have been left by this wavefront */
    long long uRow0 = 0;
    int uItems = 0, uTileBase = 0, uTileRoom = 0, uKnown = -1;
    unsigned uTileCount = 0;
    auto leaveUpTo = [&](int b) { /* this wavefront will not touch the buffers of blocks <= b again */
        for (; uBlk <= b; ++uBlk)
            if (lane == 0)
                atomicAdd(&control[10 + (uBlk & 1)], 1);
    };
    /* the chunk sums of a block's deep sub-groups, in chunk order: by the wavefront that completed the block's last item */
    auto foldBlock = [&](int blk) {
        const int buf = blk & 1;
        const BlockInfo info = infos[buf];
        const T* const sums = buffers[buf] + info.tileRoom;
        const int deepOnes = __popcll(info.parking);
        if (info.chunksDone > 0) /* a sub-group resumed from the block before: its carry has to be there */
            for (int spins = 0; __builtin_amdgcn_readfirstlane(flags[12]) < blk && spins < kSpinLimit; ++spins)
                __builtin_amdgcn_s_sleep(2);
        for (int q0 = 0; q0 < deepOnes; q0 += 2) { /* a half-wave per sub-group */
            const int q = q0 + (lane >> 5);
            const bool live = q < deepOnes;
            unsigned long long rest = info.parking;
            for (int skip = 0; skip < (live ? q : 0); ++skip)
                rest &= rest - 1;
            const int s = __ffsll((long long)rest) - 1;
            const int4 facts = subFacts[buf][s];
            const int first = facts.y, count = facts.w;
            const int rowInSub = lane & 31;
            const bool resumed = s == 0 && info.chunksDone > 0;
            if (live) {
                T total = resumed ? carry[rowInSub] : sums[first * 32 + rowInSub];
                for (int c = resumed ? 0 : 1; c < count; ++c)
                    total = add(total, sums[(first + c) * 32 + rowInSub]);
                if (s == 0 && info.cutShort) {
                    carry[rowInSub] = total;
                } else {
                    if (info.row0 + s * 32 + rowInSub < a.rows)
                        finishRow(buf, s * 32 + rowInSub, total);
                }
            }
        }
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            atomicMax(&control[12], blk + 1); /* at most two blocks are in progress, and a resumed block waits for the one before */
        }
    };

    T sum[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t)
        sum[t] = zeroOf<T>();
    /* consume the stage in `cur`; request the stage AHEAD further on into `refill` (the slot consumed last) */
    auto step = [&](Slot& cur, Slot& refill) {
        const int buf = cur.blk & 1;
        if (cur.blk != uKnown) { /* wavefront-uniform: first stage of a block */
            leaveUpTo(cur.blk - 1);
            uKnown = cur.blk;
            uRow0 = infos[buf].row0;
            uItems = __builtin_amdgcn_readfirstlane(infos[buf].items);
            uTileBase = __builtin_amdgcn_readfirstlane(infos[buf].tileBase);
            uTileCount = (unsigned)__builtin_amdgcn_readfirstlane((int)infos[buf].tileCount);
            uTileRoom = __builtin_amdgcn_readfirstlane(infos[buf].tileRoom);
        }
        const T* const tile = buffers[buf];
        T xv[UNROLL][RPL];
        bool use[UNROLL][RPL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = cur.kBase + u * PH + phase;
            if constexpr (XTILE) {
                bool outside = false;
                unsigned at[RPL];
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = cur.st.c[u].v[t] - a.baseIndex;
                    use[u][t] = k < cur.len[t] && col >= 0;
                    at[t] = (unsigned)(col - uTileBase);
                    const bool inside = at[t] < uTileCount;
                    outside |= use[u][t] && !inside;
                    xv[u][t] = tile[inside ? at[t] : 0u];
                }
                if (__ballot(outside) != 0ull) {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        if (use[u][t] && at[t] >= uTileCount)
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
        const int row = cur.row, park = cur.park, blk = cur.blk;
        const bool last = cur.last;
        fetchNext(refill); /* behind the x reads in issue order */
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = pick(use[u][t], mulAdd(cur.st.v[u].v[t], xv[u][t], sum[t]), sum[t]);
        }
        if (last) { /* wavefront-uniform: the chunk is complete */
#pragma unroll
            for (int m = LPC; m < kWave; m <<= 1) {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = add(sum[t], laneXor(sum[t], m));
            }
            if (phase == 0) {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    if (park >= 0) {
                        buffers[buf][uTileRoom + park * 32 + sub * RPL + t] = sum[t];
                    } else {
                        if (uRow0 + row + sub * RPL + t < a.rows)
                            finishRow(buf, row + sub * RPL + t, sum[t]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = zeroOf<T>();
            int before = 0;
            if (lane == 0)
                before = atomicAdd(&control[6 + buf], 1); /* behind this wavefront's chunk sum in LDS order */
            if (__builtin_amdgcn_readfirstlane(before) == uItems - 1)
                foldBlock(blk);
        }
    };

    Slot ring[AHEAD + 1];
    for (;;) {
        /* (re)start the stream: wait for the scout if the cursor has nothing */
        cStalled = false;
        for (int spins = 0; !fHave && !cEnded; ++spins) {
            fHave = nextItem();
            if (!fHave) /* the cursor may have walked through blocks whose queues were already empty: with the ring empty
                           this wavefront is done with every block up to the cursor's, and the scout waits to hear it */
                leaveUpTo(cBlk);
            if (!fHave && !cEnded) {
                if (spins > kSpinLimit) {
                    cEnded = true; /* cannot happen while the scout makes progress; never hang the device */
#ifdef SPGPU_TRACE_BLOCKS
                    if (spgpuTraceBuffer && lane == 0)
                        spgpuTraceBuffer[4096 + 32 * (size_t)blockIdx.x + wave] = (3ull << 40) | ((unsigned long long)uBlk << 20) | (unsigned)cBlk;
#endif
                }
#ifdef SPGPU_TRACE_BLOCKS
                if (spgpuTraceBuffer && lane == 0)
                    atomicAdd(&spgpuTraceBuffer[3 * (size_t)blockIdx.x + 2], 1ull);
#endif
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (!fHave)
            break;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            fetchNext(ring[i]);
        for (bool more = true; more;) { /* the ring rotates by name (the loop below is unrolled), not by copying registers */
#pragma unroll
            for (int i = 0; i <= AHEAD; ++i) {
                if (more) {
                    if (ring[i].row < 0)
                        more = false;
                    else
                        step(ring[i], ring[(i + AHEAD) % (AHEAD + 1)]);
                }
            }
        }
        /* the ring is empty: every block up to the cursor's has been consumed as far as this wavefront is concerned */
        leaveUpTo(cBlk);
        if (cEnded)
            break;
    }
    leaveUpTo(cBlk);
#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && lane == 0)
        atomicMax(&spgpuTraceBuffer[3 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
}

#ifndef SPGPU_PIPE_UNROLL
#define SPGPU_PIPE_UNROLL(RPL) ((RPL) >= 4 ? 2 : 3)
#endif
/* One workgroup per CU: the grid is the number of CUs (fewer for small matrices: a workgroup should own some blocks).
 * Shapes (SPGPU_RAGGED_SHAPE; 0 is the default): wavefronts per workgroup / stages in flight per wavefront. */
template <typename T, int RPL, bool IS_HELL, int WAVES, int AHEAD>
static void launchPipeShape(hipStream_t stream, const SlabArgs<T>& a, int computeUnits, bool tiled)
{
    constexpr int UNROLL = SPGPU_PIPE_UNROLL(RPL);
    constexpr int STAGES = 48 / ((kWave / (32 / RPL)) * UNROLL) > 0 ? 48 / ((kWave / (32 / RPL)) * UNROLL) : 1; /* chunks of 48 columns */
    const long long subs = ((long long)a.rows + 31) / 32;
    const bool byWork = IS_HELL && a.hackSize % 32 == 0;
    long long groups = (subs + 31) / 32; /* at least 1 024 rows each */
    groups = groups < 1 ? 1 : (groups > computeUnits ? computeUnits : groups);
    groups = groups > 4096 ? 4096 : groups;
    /* ranges of about 48 sub-groups: mostly one block each (a block holds up to 64) */
    long long perGroup = (subs + groups * 48 - 1) / (groups * 48);
    perGroup = perGroup < 1 ? 1 : (perGroup > kPipeRanges ? kPipeRanges : perGroup);
    SlabArgs<T> b = a;
    b.pipeRanges = (int)perGroup;
    const dim3 grid((unsigned)groups), block(WAVES * kWave);
#define SPGPU_PIPE(BYTES, XT)                                                                                         \
    do {                                                                                                              \
        if constexpr (IS_HELL) {                                                                                      \
            if (byWork) {                                                                                             \
                hipLaunchKernelGGL((pipeSpmvKernel<T, RPL, true, true, UNROLL, WAVES, BYTES, XT, STAGES, AHEAD>), grid, block, 0, stream, b); \
                break;                                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((pipeSpmvKernel<T, RPL, IS_HELL, false, UNROLL, WAVES, BYTES, XT, STAGES, AHEAD>), grid, block, 0, stream, b); \
    } while (0)
    if (tiled)
        SPGPU_PIPE(49152, true);
    else
        SPGPU_PIPE(16384, false);
#undef SPGPU_PIPE
}

template <typename T, int RPL, bool IS_HELL>
static void launchPipe(hipStream_t stream, const SlabArgs<T>& a, int computeUnits, bool tiled, int shape)
{
    switch (shape) {
#ifdef SPGPU_TUNING_VARIANTS
    case 1: launchPipeShape<T, RPL, IS_HELL, 12, 3>(stream, a, computeUnits, tiled); break;
    case 2: launchPipeShape<T, RPL, IS_HELL, 12, 4>(stream, a, computeUnits, tiled); break;
    case 3: launchPipeShape<T, RPL, IS_HELL, 8, 6>(stream, a, computeUnits, tiled); break;
    case 4: launchPipeShape<T, RPL, IS_HELL, 8, 8>(stream, a, computeUnits, tiled); break;
#endif
    default: launchPipeShape<T, RPL, IS_HELL, 16, 2>(stream, a, computeUnits, tiled); break;
    }
}
