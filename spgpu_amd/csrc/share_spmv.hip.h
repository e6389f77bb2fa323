/*
 * ELL / HELL SpMV for matrices whose rows were ORDERED BY LENGTH (rIdx given: spgpuOellOrderDevice, ellToOell) --
 * the north_star target, power-law row lengths.  Included by ellpack_spmv.hip (namespace spgpu, after SlabArgs).
 * ONE launch, no state outside the kernel's own LDS: nothing is shared between two calls in flight.
 *
 * Why the slab kernel is wrong here: after an ordering by length the depth changes along the rows -- steeply at the head
 * of every window, and the rows set aside as "long" form hacks thousands of columns deep.  A wavefront that owns fixed rows
 * streams a few GB/s, so one deep hack must not be one wavefront's job, and a workgroup that owns fixed rows must not be
 * ten times heavier than its neighbour.
 *
 *   - SHARES OF EQUAL WORK (HELL, hackSize a multiple of 32).  The grid has ceil(rows / SHARE_ROWS) workgroups; workgroup t
 *     owns the hacks h with  t*Q <= M(h) < (t+1)*Q,  M(h) = hackOffsets[h] + kRowCost*hackSize*h  (slots in front of the
 *     hack plus a fixed cost per row) and Q = ceil(M(hacks) / workgroups).  hackOffsets ascends, so both ends are found by
 *     a two-level search: every lane probes one of BLOCK evenly spaced hacks, then the lanes read the bracket that contains
 *     the boundary -- two dependent round trips, no scratch memory, and both neighbours compute a common boundary with the
 *     same arithmetic.  Other layouts (ELL, other hack sizes) take SHARE_ROWS consecutive rows per workgroup.
 *   - ITEMS.  A share's 32-row sub-groups are cut into chunks of CHUNK slab columns; the wavefronts take (sub-group, chunk)
 *     items from an LDS queue, so the longest job of a wavefront is CHUNK columns however deep the hack.  "One wavefront
 *     per hack" while it works on it: 32/RPL lanes with RPL rows each cover a slab column, PH = 64 / (32/RPL) columns per
 *     load instruction (1 KiB contiguous for hackSize 32), UNROLL of them per stage, two stages in flight per wavefront,
 *     the first stage of the next item requested during the last stages of the current one (row lengths and slab bases of
 *     the share sit in LDS).
 *   - SUMMATION ORDER.  A chunk sum = its PH phase sums (phase p adds the entries k = p mod PH in ascending k, the
 *     reference's multi-thread-per-row order, hell_spmv_base_template.cuh:59-101) combined pairwise with lane-xor
 *     shuffles.  A sub-group of one chunk is finished by its wavefront; deeper ones leave their chunk sums in LDS and the
 *     workgroup adds them in chunk order once the queue is empty: orc_?spmv_deep with deepCap = deepChunk = CHUNK.
 *     Which wavefront computes a chunk depends on timing, the sums do not.
 *   - x: the slice the share's rows touch is staged in LDS (as in the x-tile form of slabSpmvKernel); entries outside it
 *     are gathered from global memory.  The chunk sums of deep sub-groups borrow the end of the same buffer.
 *   - A share that does not fit (more than MAXSUBS sub-groups, more chunk sums than half the buffer, one sub-group deeper
 *     than that) runs in several passes; the carry of a sub-group cut by a pass boundary waits in LDS.  Any matrix is
 *     computed correctly; the shapes are sized for the ordered layouts.
 *
 * Algorithmic bytes as for slabSpmvKernel, plus 4 per row for rIdx.
 */

constexpr int kRowCost = 4; /* what a row costs besides its slots (rS, rIdx, z, probes), in slots of sizeof(T) + 4 bytes */

template <typename T, int RPL, bool IS_HELL, bool BY_WORK, int UNROLL, int WAVES, int BUFFER_BYTES, bool XTILE, int SHARE_SUBS,
          int CHUNK_STAGES>
__global__ __launch_bounds__(WAVES * kWave) __attribute__((amdgpu_waves_per_eu(4, 4))) void shareSpmvKernel(const SlabArgs<T> a)
{
    constexpr int LPC = 32 / RPL;   /* lanes per slab column of a sub-group */
    constexpr int PH = kWave / LPC; /* slab columns per wave-wide load */
    constexpr int STEP = PH * UNROLL;
    constexpr int CHUNK = STEP * CHUNK_STAGES;
    constexpr int BLOCK = WAVES * kWave;
    constexpr int MAXSUBS = BY_WORK ? 64 : SHARE_SUBS; /* sub-groups of one pass: a lane each in the scans below */
    constexpr int MAXROWS = MAXSUBS * 32;
    constexpr int BUFFER_ELEMS = BUFFER_BYTES / (int)sizeof(T);
    constexpr int PMAX = BUFFER_ELEMS / 32 / 2; /* chunk sums (32 values each) a pass may park in the buffer */
    static_assert(MAXSUBS <= kWave && SHARE_SUBS <= MAXSUBS, "one lane per sub-group");
    static_assert(PMAX >= 2, "room for the chunk sums of a two-chunk sub-group");
    static_assert(!BY_WORK || IS_HELL, "shares of equal work are cut along hackOffsets");

    __shared__ __attribute__((aligned(16))) T buffer[BUFFER_ELEMS]; /* x tile from the front, chunk sums from the back */
    __shared__ T carry[32];                                        /* a sub-group cut by a pass boundary */
    __shared__ int lens[MAXROWS];
    __shared__ int dests[MAXROWS]; /* rIdx of the pass's rows: fetched from global memory at the end of an item it would be waited for
                                      with vmcnt(0) -- counters retire in order -- and drain the wavefront's prefetch */
    __shared__ unsigned bases[BY_WORK ? MAXSUBS : MAXROWS / RPL]; /* first slot of a sub-group / of an RPL-row strip */
    __shared__ int depths[MAXSUBS];
    __shared__ int4 subFacts[MAXSUBS]; /* first item, first chunk sum or -1, depth, chunks done earlier */
    __shared__ int control[8]; /* 0..3 the search's counters, 4 the item queue */
    __shared__ ColumnProbe seen[WAVES];

    const T* __restrict__ x = a.x;
    const long long totalSubs = ((long long)a.rows + 31) / 32;
#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && threadIdx.x == 0)
        spgpuTraceBuffer[3 * (size_t)blockIdx.x] = wall_clock64();
#endif

    /* ---- which sub-groups? ------------------------------------------------------------------------------------------ */
    long long shareFirst, shareEnd; /* sub-groups [shareFirst, shareEnd) */
    if constexpr (BY_WORK) {
        const int lane = threadIdx.x & (kWave - 1);
        const long long hs = a.hackSize, hacks = ((long long)a.rows + hs - 1) / hs, perHack = hs / 32;
        const long long rowCost = (long long)kRowCost * hs;
        if (threadIdx.x < 4)
            control[threadIdx.x] = 0;
        /* round trip 1: BLOCK evenly spaced hacks.  The last hack's own slots are left out of the total (hackOffsets has
         * no closing entry): every M(h) is still below it. */
        const long long total = (long long)a.hackOffsets[hacks - 1] + rowCost * hacks;
        const long long quota = (total + gridDim.x - 1) / gridDim.x;
        const long long lo = (long long)blockIdx.x * quota, hi = lo + quota;
        const long long stride = (hacks + BLOCK - 1) / BLOCK;
        const long long probe = (long long)threadIdx.x * stride;
        const long long mine = probe < hacks ? (long long)a.hackOffsets[probe] + rowCost * probe : 0x7fffffffffffffffll;
        __syncthreads();
        const int belowLo = __popcll(__ballot(mine < lo)), belowHi = __popcll(__ballot(mine < hi));
        if (lane == 0) {
            atomicAdd(&control[0], belowLo);
            atomicAdd(&control[1], belowHi);
        }
        __syncthreads();
        /* round trip 2: the hacks between the last probe below a boundary and the next probe */
        const int probesLo = control[0], probesHi = control[1];
        int fineLo = 0, fineHi = 0;
        if (probesLo > 0) {
            const long long from = (long long)(probesLo - 1) * stride + 1;
            const long long to = from + stride - 1 < hacks ? from + stride - 1 : hacks;
            for (long long h = from + threadIdx.x; h < to; h += BLOCK)
                fineLo += ((long long)a.hackOffsets[h] + rowCost * h < lo) ? 1 : 0;
        }
        if (probesHi > 0) {
            const long long from = (long long)(probesHi - 1) * stride + 1;
            const long long to = from + stride - 1 < hacks ? from + stride - 1 : hacks;
            for (long long h = from + threadIdx.x; h < to; h += BLOCK)
                fineHi += ((long long)a.hackOffsets[h] + rowCost * h < hi) ? 1 : 0;
        }
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) {
            fineLo += laneXor(fineLo, m);
            fineHi += laneXor(fineHi, m);
        }
        if (lane == 0) {
            atomicAdd(&control[2], fineLo);
            atomicAdd(&control[3], fineHi);
        }
        __syncthreads();
        const long long hackFirst = probesLo > 0 ? (long long)(probesLo - 1) * stride + 1 + control[2] : 0;
        const long long hackEnd = probesHi > 0 ? (long long)(probesHi - 1) * stride + 1 + control[3] : 0;
        shareFirst = hackFirst * perHack;
        shareEnd = hackEnd * perHack < totalSubs ? hackEnd * perHack : totalSubs;
    } else {
        shareFirst = (long long)blockIdx.x * SHARE_SUBS;
        shareEnd = shareFirst + SHARE_SUBS < totalSubs ? shareFirst + SHARE_SUBS : totalSubs;
    }

    const bool hasBeta = isNotZero(a.beta);
    auto finishRow = [&](int inPass, T sum) { /* row inPass of the pass */
        const int outRow = dests[inPass];
        a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, sum, a.beta, a.y[outRow]) : epilogue<false>(a.alpha, sum, a.beta, zeroOf<T>());
    };

    /* ---- passes (one, unless the share does not fit) ------------------------------------------------------------------ */
    long long passFirst = shareFirst; /* first sub-group of the pass */
    int chunksDone = 0;               /* of that sub-group, in earlier passes */
    while (passFirst < shareEnd) {    /* workgroup-uniform */
        /* The lane's id, made opaque per pass: otherwise everything derived from it that does not change between passes
         * (row numbers, strip and phase offsets, comparisons) is hoisted out of this loop and kept in registers across
         * the whole body -- 380 bytes of scratch in the stream loop for a loop that runs once. */
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & (kWave - 1);
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int sub = lane % LPC, phase = lane / LPC;
        const int candidates = shareEnd - passFirst < MAXSUBS ? (int)(shareEnd - passFirst) : MAXSUBS;
        const long long passRow0 = passFirst * 32;

        /* round trip 3: row lengths, slab bases */
        constexpr int RPT = (MAXROWS + BLOCK - 1) / BLOCK; /* rows a lane looks at; 32 consecutive rows = 32 consecutive lanes */
        int myLen[RPT], myDest[RPT];
        unsigned myBase[RPT];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int i = tid + j * BLOCK;
            const long long r = passRow0 + i;
            const bool live = i < candidates * 32 && r < a.rows;
            myLen[j] = live ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
            myDest[j] = live && a.rIdx ? a.rIdx[r] : (int)r;
            myBase[j] = 0;
            if constexpr (!BY_WORK) {
                if (live) {
                    if constexpr (IS_HELL) {
                        const unsigned u0 = (unsigned)r, hs = (unsigned)a.hackSize;
                        myBase[j] = (unsigned)a.hackOffsets[u0 / hs] + u0 % hs;
                    } else {
                        myBase[j] = (unsigned)r;
                    }
                }
            }
        }
        if constexpr (BY_WORK) {
            if ((int)tid < candidates) {
                const long long s = passFirst + tid, perHack = a.hackSize / 32;
                bases[tid] = (unsigned)a.hackOffsets[s / perHack] + (unsigned)(s % perHack) * 32u;
            }
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int i = tid + j * BLOCK;
            int depth = myLen[j];
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                const int other = laneXor(depth, m);
                depth = other > depth ? other : depth;
            }
            if (i < MAXROWS) {
                lens[i] = myLen[j];
                dests[i] = myDest[j];
                if constexpr (!BY_WORK) {
                    if (i % RPL == 0)
                        bases[i / RPL] = myBase[j];
                }
                if ((lane & 31) == 0)
                    depths[i >> 5] = depth;
            }
        }
        if (tid == 0)
            control[4] = WAVES; /* the first WAVES items are dealt out statically */
        __syncthreads();

        /* Every wavefront works out the pass for itself, a lane per sub-group: chunks, which sub-groups fit (their chunk
         * sums have to find room in the buffer), where each one's items and chunk sums start. */
        const int myDepth = lane < candidates ? depths[lane] : 0;
        const int myChunksAll = lane < candidates ? (myDepth + CHUNK - 1) / CHUNK + (myDepth == 0 ? 1 : 0) : 0; /* an empty sub-group is one item */
        int myChunks = lane == 0 ? myChunksAll - chunksDone : myChunksAll; /* still to do */
        const bool myParks = myChunksAll > 1;                              /* chunk sums go through LDS */
        bool cutShort = false;                                             /* the first sub-group does not finish in this pass */
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
        const int used = cutShort ? 1 : (~fits == 0ull ? kWave : __ffsll((long long)~fits) - 1); /* leading sub-groups that fit */
        if (lane >= used)
            myChunks = 0;
        int itemIncl = myChunks;
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) {
            const int other = __shfl_up(itemIncl, m, kWave);
            itemIncl += lane >= m ? other : 0;
        }
        const int passItems = __builtin_amdgcn_readfirstlane(__shfl(itemIncl, kWave - 1, kWave));
        const int parked = __builtin_amdgcn_readfirstlane(__shfl(lane < used ? parkIncl : 0, used - 1, kWave)); /* chunk sums of this pass */
        const int tileRoom = BUFFER_ELEMS - parked * 32;
        T* const sums = buffer + tileRoom;
        /* what loadItem needs to know about a sub-group, where a wave-uniform read finds it (kept out of the registers of
         * the stream loop).  Every wavefront writes the same values and reads them behind its own writes (LDS is in order
         * per wavefront), so no barrier is needed before the first use. */
        if (lane < MAXSUBS)
            subFacts[lane] = int4{itemIncl - myChunks, myParks ? parkIncl - myChunks : -1, myDepth, lane == 0 ? chunksDone : 0};
        const unsigned long long parking = __ballot(lane < used && myParks);
        const int chunksOfMine = myChunks; /* the fold below wants it per lane */

        /* round trip 4: where is the slice of x?  first and last column of every row (the extremes of a row whose columns
         * ascend; any order is still correct) */
        ColumnProbe mine{0x7fffffff, -0x7fffffff - 1, 0, 0};
        int first[RPT], last[RPT];
        if constexpr (XTILE) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = tid + j * BLOCK;
                first[j] = last[j] = 0;
                if (i < used * 32 && myLen[j] > 0) {
                    long long at;
                    if constexpr (BY_WORK)
                        at = (long long)bases[i >> 5] + (i & 31);
                    else
                        at = myBase[j];
                    first[j] = a.rP[at];
                    last[j] = a.rP[at + (long long)(myLen[j] - 1) * a.idxStride];
                }
            }
        }

        /* ---- the per-item state of a lane, and the stage loads ----------------------------------------------------- */
        struct Item {
            long long slab; /* first slot of this lane's strip */
            int len[RPL];   /* cut at the end of the chunk */
            int longest;    /* of this lane's rows */
            int kEnd;       /* end of the chunk (wave-uniform) */
            int row;        /* first row of the sub-group, relative to the pass */
            int park;       /* where the chunk sum goes, or -1: the sub-group has one chunk and is finished on the spot */
        };
        struct Stage {
            Pack<T, RPL> v[UNROLL];
            Pack<int, RPL> c[UNROLL];
        };
        /* item -> (sub-group, chunk): the sub-group is the number of lanes whose items end at or before it */
        auto loadItem = [&](int item, Item& it) -> int {
            const int s = __popcll(__ballot(lane < used && itemIncl <= item));
            const int4 facts = subFacts[s];
            const int inSub = item - __builtin_amdgcn_readfirstlane(facts.x);
            const int parkFirst = __builtin_amdgcn_readfirstlane(facts.y);
            const int depth = __builtin_amdgcn_readfirstlane(facts.z);
            const int chunk = inSub + __builtin_amdgcn_readfirstlane(facts.w);
            if constexpr (BY_WORK)
                it.slab = (long long)bases[s] + sub * RPL;
            else
                it.slab = bases[s * LPC + sub];
            it.kEnd = (chunk + 1) * CHUNK < depth ? (chunk + 1) * CHUNK : depth;
            it.longest = 0;
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const int len = lens[s * 32 + sub * RPL + t];
                it.len[t] = len < it.kEnd ? len : it.kEnd;
                it.longest = it.len[t] > it.longest ? it.len[t] : it.longest;
            }
            it.row = s * 32;
            it.park = parkFirst >= 0 ? parkFirst + inSub : -1;
            return chunk * CHUNK; /* first column */
        };
        auto fetch = [&](const Item& it, int kBase, Stage& st) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = kBase + u * PH + phase;
                if (k < it.longest) {
                    st.v[u] = loadPack<true, T, RPL>(a.cM + it.slab + (long long)k * a.valStride);
                    st.c[u] = loadPack<true, int, RPL>(a.rP + it.slab + (long long)k * a.idxStride);
                } else {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        st.v[u].v[t] = zeroOf<T>();
                        st.c[u].v[t] = a.baseIndex;
                    }
                }
            }
        };

        /* The stream of stages a wavefront walks: the stages of its first item, then of the items it takes from the queue.
         * A fetch cursor runs AHEAD stages in front of the stage being consumed (with one stage ahead a wavefront waits a
         * full memory round trip per stage: measured 1.25 GB/s per wavefront).  A ring slot holds a stage and what
         * consuming it needs to know. */
        constexpr int AHEAD = 2;
        struct Slot {
            Stage st;
            int len[RPL];
            int kBase;
            int row;    /* < 0: nothing, the stream has ended */
            int park;
            bool last;  /* last stage of its item */
        };
        auto grab = [&]() -> int {
            int got = 0;
            if (lane == 0)
                got = atomicAdd(&control[4], 1);
            return __builtin_amdgcn_readfirstlane(got);
        };
        int fItem = wave, fThen = passItems, fk = 0; /* fThen: taken one item ahead */
        Item fit;
        if (fItem < passItems)
            fk = loadItem(fItem, fit);
        auto fetchNext = [&](Slot& slot) {
            slot.row = -1;
            slot.last = false;
            if (fItem < passItems) {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    slot.len[t] = fit.len[t];
                slot.kBase = fk;
                slot.row = fit.row;
                slot.park = fit.park;
                fetch(fit, fk, slot.st);
                fk += STEP;
                slot.last = fk >= fit.kEnd;
                if (slot.last) { /* wavefront-uniform: on to the next item */
                    fItem = fThen;
                    if (fItem < passItems)
                        fk = loadItem(fItem, fit);
                    fThen = fItem < passItems ? grab() : passItems;
                }
            }
        };
        Slot ring[AHEAD + 1];
        fThen = grab();
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            fetchNext(ring[i]); /* on their way while the tile is being placed and filled */
        /* the probes' answers (the stage loads above are younger: waiting for the probes leaves them in flight) */
        if constexpr (XTILE) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = tid + j * BLOCK;
                if (i < used * 32 && myLen[j] > 0) {
                    const int f = first[j] - a.baseIndex, l = last[j] - a.baseIndex;
                    const int low = f < l ? f : l, high = f < l ? l : f;
                    mine.lowest = low < mine.lowest ? low : mine.lowest;
                    mine.highest = high > mine.highest ? high : mine.highest;
                    mine.middles += ((long long)f + l) >> 1;
                    mine.rows += 1;
                }
            }
            mine.lowest = waveMin(mine.lowest);
            mine.highest = waveMax(mine.highest);
#pragma unroll
            for (int m = 1; m < kWave; m <<= 1) {
                mine.rows += laneXor(mine.rows, m);
                const int lowHalf = laneXor((int)(unsigned)(mine.middles & 0xffffffffll), m);
                const int highHalf = laneXor((int)(mine.middles >> 32), m);
                mine.middles += ((long long)highHalf << 32) | (unsigned)lowHalf;
            }
            if (lane == 0)
                seen[wave] = mine;
        }

        __syncthreads();

        /* ---- the slice of x (round trip 5; the stages requested just above travel with it) -------------------------- */
        int tileBase = 0;
        unsigned tileCount = 0;
        if constexpr (XTILE) {
            ColumnProbe all{0x7fffffff, -0x7fffffff - 1, 0, 0};
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const ColumnProbe other = seen[w];
                all.lowest = other.lowest < all.lowest ? other.lowest : all.lowest;
                all.highest = other.highest > all.highest ? other.highest : all.highest;
                all.rows += other.rows;
                all.middles += other.middles;
            }
            if (all.rows > 0 && all.lowest >= 0) {
                const long long span = (long long)all.highest - all.lowest + 1;
                if (span <= tileRoom) {
                    tileBase = all.lowest;
                    tileCount = (unsigned)span;
                } else {
                    long long start = all.middles / all.rows - tileRoom / 2;
                    start = start < all.lowest ? all.lowest : start;
                    start = start + tileRoom > (long long)all.highest + 1 ? (long long)all.highest + 1 - tileRoom : start;
                    tileBase = (int)start;
                    tileCount = (unsigned)tileRoom;
                }
            }
            /* The copy goes straight from global memory into LDS (global_load_lds_dwordx4: no registers, no ds_write; with
             * 8 x 16 bytes per lane staged in registers beside the two stages already in flight the kernel spilled, and
             * every spilled piece was a round trip of its own).  One wave-wide instruction writes 1 KiB of LDS in lane
             * order from 64 per-lane addresses, so only whole wavefronts take it; the source is 16-byte aligned (the tile
             * starts a few elements early if it has to).  The ragged end is copied through registers. */
            constexpr int PIECE = 16 / (int)sizeof(T);
            if constexpr (PIECE > 1) {
                const int early = (int)(((uintptr_t)(x + tileBase) % 16) / sizeof(T));
                if (early <= tileBase && tileCount > 0) {
                    tileBase -= early;
                    tileCount = tileCount + early <= (unsigned)tileRoom ? tileCount + early : (unsigned)tileRoom;
                }
            }
            constexpr int ROUND = (BUFFER_ELEMS / PIECE + BLOCK - 1) / BLOCK;
            const T* __restrict__ from = x + tileBase;
            const unsigned pieces = tileCount / PIECE;
            const bool direct = ((uintptr_t)from % 16) == 0;
#pragma unroll
            for (int q = 0; q < ROUND; ++q) {
                const unsigned piece = tid + q * BLOCK;
                const unsigned waveFirst = (unsigned)(wave * kWave + q * BLOCK);
                if (direct && waveFirst + kWave <= pieces) { /* wavefront-uniform */
#if defined(__HIP_DEVICE_COMPILE__) /* the host pass of hipcc parses the kernel body too and has no such builtin */
                    __builtin_amdgcn_global_load_lds(from + (size_t)piece * PIECE, buffer + (size_t)piece * PIECE, 16, 0, 0);
#endif
                } else if (piece < pieces) {
                    const Pack<T, PIECE> w = loadPackElementAligned<T, PIECE>(from + (size_t)piece * PIECE);
                    storePack<T, PIECE>(buffer + (size_t)piece * PIECE, w);
                }
            }
            if (pieces * PIECE + tid < tileCount)
                buffer[pieces * PIECE + tid] = from[pieces * PIECE + tid];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* global_load_lds retires on vmcnt; the barrier hands the tile over */
            __syncthreads();
        }
#ifdef SPGPU_TRACE_BLOCKS
        if (spgpuTraceBuffer && tid == 0 && passFirst == shareFirst)
            spgpuTraceBuffer[3 * (size_t)blockIdx.x + 2] = wall_clock64(); /* tile in place */
#endif

        /* ---- the stage stream ---------------------------------------------------------------------------------------- */
        T sum[RPL];
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            sum[t] = zeroOf<T>();
        /* consume the stage in `cur`; request the stage AHEAD further on into `refill` (the slot consumed last) */
        auto step = [&](Slot& cur, Slot& refill) {
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
                        at[t] = (unsigned)(col - tileBase);
                        const bool inside = at[t] < tileCount;
                        outside |= use[u][t] && !inside;
                        xv[u][t] = buffer[inside ? at[t] : 0u];
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
            const int row = cur.row, park = cur.park;
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
                            sums[park * 32 + sub * RPL + t] = sum[t];
                        } else {
                            if (passRow0 + row + sub * RPL + t < a.rows)
                                finishRow(row + sub * RPL + t, sum[t]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = zeroOf<T>();
            }
        };
        static_assert(AHEAD == 2, "the rotation below is written out for a ring of three");
        for (;;) { /* the ring rotates by name, not by copying registers */
            if (ring[0].row < 0) break;
            step(ring[0], ring[2]);
            if (ring[1].row < 0) break;
            step(ring[1], ring[0]);
            if (ring[2].row < 0) break;
            step(ring[2], ring[1]);
        }
        __syncthreads(); /* every chunk sum of the pass is in LDS */

        /* ---- deep sub-groups: chunk sums in chunk order (32 lanes per sub-group) ------------------------------------- */
        {
            const int deepOnes = __popcll(parking);
            for (int q0 = wave * 2; q0 < deepOnes; q0 += WAVES * 2) { /* wavefront-uniform: a half-wave per sub-group */
                const int q = q0 + (lane >> 5);
                const bool live = q < deepOnes;
                unsigned long long rest = parking;
                for (int skip = 0; skip < (live ? q : 0); ++skip)
                    rest &= rest - 1;
                const int s = __ffsll((long long)rest) - 1;
                const int first = subFacts[s].y, count = __shfl(chunksOfMine, s, kWave);
                const int rowInSub = lane & 31;
                const bool resumed = s == 0 && chunksDone > 0;
                if (live) {
                    T total = resumed ? carry[rowInSub] : sums[first * 32 + rowInSub];
                    for (int c = resumed ? 0 : 1; c < count; ++c)
                        total = add(total, sums[(first + c) * 32 + rowInSub]);
                    if (s == 0 && cutShort) {
                        carry[rowInSub] = total;
                    } else {
                        if (passRow0 + s * 32 + rowInSub < a.rows)
                            finishRow(s * 32 + rowInSub, total);
                    }
                }
            }
        }
        if (cutShort) {
            chunksDone += PMAX;
        } else {
            passFirst += used;
            chunksDone = 0;
        }
        if (passFirst < shareEnd)
            __syncthreads(); /* LDS is about to be rewritten */
    }
#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && (threadIdx.x & 63) == 0)
        atomicMax(&spgpuTraceBuffer[3 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
}

#ifndef SPGPU_SHARE_UNROLL
#define SPGPU_SHARE_UNROLL(RPL) ((RPL) >= 4 ? 2 : 3) /* wave-wide loads per stage; 3 keeps the 8-byte kernels at two 8-wavefront workgroups per CU with two stages in flight */
#endif
/* Shapes (SPGPU_RAGGED_SHAPE; 0 is the default): workgroup lanes / buffer / rows per share. */
template <typename T, int RPL, bool IS_HELL>
static void launchShare(hipStream_t stream, const SlabArgs<T>& a, int shape, bool tiled)
{
    constexpr int UNROLL = SPGPU_SHARE_UNROLL(RPL);
    constexpr int STAGES = 48 / ((kWave / (32 / RPL)) * UNROLL) > 0 ? 48 / ((kWave / (32 / RPL)) * UNROLL) : 1; /* chunks of 48 columns */
    const long long subs = ((long long)a.rows + 31) / 32;
    const bool byWork = IS_HELL && a.hackSize % 32 == 0;
#define SPGPU_SHARE(WAVES, BYTES, XT, SUBS)                                                                           \
    do {                                                                                                              \
        const dim3 grid((unsigned)((subs + (SUBS) - 1) / (SUBS))), block((WAVES) * kWave);                            \
        if constexpr (IS_HELL) {                                                                                      \
            if (byWork) {                                                                                             \
                hipLaunchKernelGGL((shareSpmvKernel<T, RPL, true, true, UNROLL, WAVES, BYTES, XT, SUBS, STAGES>), grid, block, 0, stream, a); \
                break;                                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((shareSpmvKernel<T, RPL, IS_HELL, false, UNROLL, WAVES, BYTES, XT, SUBS, STAGES>), grid, block, 0, stream, a); \
    } while (0)
    if (!tiled) {
        SPGPU_SHARE(4, 16384, false, 16);
        return;
    }
    switch (shape) {
#ifdef SPGPU_TUNING_VARIANTS
    case 1: SPGPU_SHARE(8, 57344, true, 64); break;
    case 2: SPGPU_SHARE(8, 49152, true, 32); break;
#endif
    default: SPGPU_SHARE(8, 57344, true, 32); break;
    }
#undef SPGPU_SHARE
}
