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

template <typename T, int RPL, bool IS_HELL, int UNROLL, int WAVES, int TILE_BYTES, int SUBS, bool DEEP, int ZBYTES = 0>
__global__ __launch_bounds__(WAVES * kWave) __attribute__((amdgpu_waves_per_eu(4)))
void raggedSpmvKernel(const SlabArgs<T> a) /* 4 wavefronts per SIMD: two 8-wavefront workgroups per CU (what LDS admits) */
{
    constexpr int LPC = 32 / RPL;   /* lanes per slab column of a sub-group */
    constexpr int PH = kWave / LPC; /* slab columns per wave-wide load */
    constexpr int STEP = PH * UNROLL;
    constexpr int BLOCK = WAVES * kWave;
    constexpr int ROWS = SUBS * 32;
    constexpr bool XTILE = TILE_BYTES > 0;
    constexpr int TILE_ELEMS = XTILE ? TILE_BYTES / (int)sizeof(T) : 1;
    /* ZSTAGE: the workgroup's results wait in LDS, placed by destination, and leave in whole lines when its queue is empty.
     * The L2 does not merge stores over time (every store's bytes leave it at once): written from the lanes that hold the
     * sums, z[rIdx[r]] is one 8-byte fabric write per row -- 3 % of the kernel's time with the rows ordered in windows of
     * 1 024, 6 % at 2 048, 12 % at 4 096 (profiles/r03_exp_z_scatter.txt).  ZW destinations from the workgroup's lowest. */
    constexpr bool ZSTAGE = ZBYTES > 0;
    constexpr int ZW = ZSTAGE ? ZBYTES / (int)sizeof(T) : 32;
    static_assert(ZW < 0xFFFF && ZW % 32 == 0, "a staged destination is a 16-bit offset");
    static_assert(SUBS >= WAVES, "every wavefront starts with a sub-group of its own");

    __shared__ __attribute__((aligned(16))) T tile[TILE_ELEMS];
    /* row lengths as walked here (deep sub-groups: cut at deepCap); 16 bits each, 0xFFFF = "65 535 or more: ask rS" (a
     * row that long is walked whole only when the deep list was full) */
    __shared__ unsigned short lens[ROWS];
    /* rIdx of the workgroup's rows (fetched from global memory at the end of a sub-group it would be waited for with vmcnt(0)
     * -- counters retire in order -- and drain the wavefront's prefetch); ZSTAGE: as offsets from the workgroup's lowest
     * destination instead, 0xFFFF = beyond the staging buffer */
    __shared__ int dests[ZSTAGE ? 1 : ROWS];
    __shared__ unsigned short destOffset[ZSTAGE ? ROWS : 1];
    __shared__ __attribute__((aligned(16))) T staged[ZW];
    __shared__ unsigned stagedMask[ZW / 32];
    __shared__ int lowestDest;
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
        spgpuTraceBuffer[3 * (size_t)blockIdx.x] = wall_clock64();
#endif
    /* Consecutive workgroups (in row order) read overlapping slices of x.  The hardware deals workgroup ids round-robin
     * over the 8 XCDs, each with an L2 of its own: left alone, every slice is fetched from memory by all eight.  With
     * a.xcdRun > 0 the id is permuted so that runs of xcdRun consecutive row blocks share an XCD (speed only). */
    const unsigned logicalBlock = a.xcdRun > 0 ? xcdRuns(blockIdx.x, gridDim.x, (unsigned)a.xcdRun) : blockIdx.x;
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
            stagedMask[threadIdx.x] = 0u;
        if (threadIdx.x == 0)
            lowestDest = 0x7fffffff;
    }
    ColumnProbe mine{0x7fffffff, -0x7fffffff - 1, 0, 0};
#pragma unroll
    for (int j = 0; j < RPT; ++j) { /* round trip 2 */
        const int i = threadIdx.x + j * BLOCK;
        int first = 0, last = 0;
        if (XTILE && myLen[j] > 0) {
            first = a.rP[(long long)myBase[j]];
            last = a.rP[(long long)myBase[j] + (long long)(myLen[j] - 1) * a.idxStride];
        }
        /* depth of the 32-row sub-group these 32 lanes hold; the deep ones register and are cut at deepCap */
        int depth = myLen[j];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const int other = laneXor(depth, m);
            depth = other > depth ? other : depth;
        }
        int slot = -1;
        if constexpr (DEEP) {
            if ((lane & 31) == 0 && depth > a.deepCap && i < ROWS)
                slot = deepRegister(a, (int)(blockRow0 + i), depth);
            slot = __shfl(slot, lane & 32, kWave);
        }
        if (i < ROWS) {
            const int walked = (slot >= 0 && myLen[j] > a.deepCap) ? a.deepCap : myLen[j];
            lens[i] = (unsigned short)(walked < 0xFFFF ? walked : 0xFFFF);
            if constexpr (!ZSTAGE)
                dests[i] = myDest[j];
            if (i % RPL == 0)
                bases[i / RPL] = myBase[j];
            if ((lane & 31) == 0) {
                depths[i >> 5] = (slot >= 0 && depth > a.deepCap) ? a.deepCap : depth;
                deepSlots[i >> 5] = slot;
            }
        }
        if (XTILE && myLen[j] > 0) {
            const int f = first - a.baseIndex, l = last - a.baseIndex;
            const int low = f < l ? f : l, high = f < l ? l : f;
            mine.lowest = low < mine.lowest ? low : mine.lowest;
            mine.highest = high > mine.highest ? high : mine.highest;
            mine.middles += ((long long)f + l) >> 1;
            mine.rows += 1;
        }
    }
    if constexpr (XTILE) {
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
    if (threadIdx.x == 0)
        nextItem = WAVES; /* the first WAVES sub-groups are dealt out statically */
    if constexpr (ZSTAGE) {
        __syncthreads(); /* lowestDest and stagedMask initialised */
        int low = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int i = threadIdx.x + j * BLOCK;
            if (i < ROWS && blockRow0 + i < a.rows)
                low = myDest[j] < low ? myDest[j] : low;
        }
        low = waveMin(low);
        if (lane == 0)
            atomicMin(&lowestDest, low);
    }
    __syncthreads();
    int zBase = 0;
    if constexpr (ZSTAGE) {
        zBase = lowestDest;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int i = threadIdx.x + j * BLOCK;
            if (i < ROWS) {
                const long long off = (long long)myDest[j] - zBase;
                /* (rows of a deep sub-group are finished by the deep kernels: nothing of theirs is staged) */
                const bool in = blockRow0 + i < a.rows && off >= 0 && off < ZW && !(DEEP && deepSlots[i >> 5] >= 0);
                destOffset[i] = in ? (unsigned short)off : (unsigned short)0xFFFF;
                if (in)
                    atomicOr(&stagedMask[off >> 5], 1u << (off & 31));
            }
        }
        /* (read by the wavefronts at the end of their first sub-group at the earliest: the tile's barrier lies between) */
    }

    /* ---- the per-sub-group state of a lane, and the stage loads --------------------------------------------------- */
    struct Item {
        long long slab; /* first slot of this lane's strip */
        int len[RPL];
        int longest;    /* of this lane's rows */
        int depth;      /* of the sub-group (wave-uniform) */
    };
    struct Stage {
        Pack<T, RPL> v[UNROLL];
        Pack<int, RPL> c[UNROLL];
    };
    auto loadItem = [&](int s, Item& it) {
        it.slab = (long long)bases[s * LPC + sub];
        it.longest = 0;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            it.len[t] = lens[s * 32 + sub * RPL + t];
            if (it.len[t] == 0xFFFF) { /* see lens: not cut (the sub-group has no deep slot), so the row's own length */
                const long long r = blockRow0 + s * 32 + sub * RPL + t;
                it.len[t] = a.rS ? a.rS[r] : a.maxNnz;
            }
            it.longest = it.len[t] > it.longest ? it.len[t] : it.longest;
        }
        it.depth = depths[s];
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

    /* The stream of stages a wavefront walks: the stages of its first sub-group, then of the sub-groups it takes from the
     * queue.  A fetch cursor runs AHEAD stages in front of the stage being consumed, so that AHEAD stage loads (3 KiB
     * each for 8-byte elements at UNROLL 2 ... 6 KiB at 4) are in flight per wavefront: with one stage ahead a
     * wavefront waits a full memory round trip per stage (measured: 1.25 GB/s per wavefront, 10 GB/s per workgroup).
     * A ring slot holds a stage and what consuming it needs to know. */
    constexpr int AHEAD = 2;
    struct Slot {
        Stage st;
        int len[RPL];
        int s;      /* sub-group (>= SUBS: nothing, the stream has ended) */
        int kBase;
        bool last;  /* last stage of its sub-group */
    };
    auto grab = [&]() -> int {
        int got = 0;
        if (lane == 0)
            got = atomicAdd(&nextItem, 1);
        return __builtin_amdgcn_readfirstlane(got);
    };
    /* fetch cursor */
    int fs = wave, fk = 0, fsThen = SUBS; /* fsThen: grabbed one sub-group ahead (after the prologue's barrier) */
    Item fit;
    loadItem(fs, fit);
    auto fetchNext = [&](Slot& slot) {
        slot.s = fs;
        slot.kBase = fk;
        if (fs < SUBS) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                slot.len[t] = fit.len[t];
            fetch(fit, fk, slot.st);
            fk += STEP;
            slot.last = fk >= fit.depth;
            if (slot.last) { /* wavefront-uniform: on to the next sub-group */
                fs = fsThen;
                fk = 0;
                if (fs < SUBS)
                    loadItem(fs, fit);
                fsThen = fs < SUBS ? grab() : SUBS;
            }
        } else {
            slot.last = false;
        }
    };
    Slot ring[AHEAD + 1];

    fsThen = grab();
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        fetchNext(ring[i]); /* on their way while the tile is being placed and filled */

    /* ---- the slice of x (round trip 3; the stages requested just above travel with it) ---------------------------- */
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
            if (span <= TILE_ELEMS) {
                tileBase = all.lowest;
                tileCount = (unsigned)span;
            } else {
                long long start = all.middles / all.rows - TILE_ELEMS / 2;
                start = start < all.lowest ? all.lowest : start;
                start = start + TILE_ELEMS > (long long)all.highest + 1 ? (long long)all.highest + 1 - TILE_ELEMS : start;
                tileBase = (int)start;
                tileCount = TILE_ELEMS;
            }
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
        spgpuTraceBuffer[3 * (size_t)blockIdx.x + 2] = wall_clock64(); /* tile in place */
#endif
    /* ---- 4: the stage stream -------------------------------------------------------------------------------------- */
    const bool hasBeta = isNotZero(a.beta);
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
        const int s = cur.s;
        const bool last = cur.last;
        fetchNext(refill); /* behind the x reads in issue order */
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = pick(use[u][t], mulAdd(cur.st.v[u].v[t], xv[u][t], sum[t]), sum[t]);
        }
        if (last) { /* wavefront-uniform: the sub-group is complete */
#pragma unroll
            for (int m = LPC; m < kWave; m <<= 1) {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = add(sum[t], laneXor(sum[t], m));
            }
            if (phase == 0) {
                const int deepSlot = deepSlots[s];
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const long long r = blockRow0 + s * 32 + sub * RPL + t;
                    if (r < a.rows) {
                        if (DEEP && deepSlot >= 0) {
                            a.deepPartials[(size_t)deepSlot * 32 + (size_t)(sub * RPL + t)] = sum[t];
                        } else if constexpr (ZSTAGE) {
                            const unsigned off = destOffset[s * 32 + sub * RPL + t];
                            if (off != 0xFFFFu) {
                                staged[off] = mul(a.alpha, sum[t]); /* beta * y joins when the line is written */
                            } else { /* beyond the staging buffer: the scattered store, its destination from global memory */
                                const int outRow = a.rIdx ? a.rIdx[r] : (int)r;
                                a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, sum[t], a.beta, a.y[outRow])
                                                      : epilogue<false>(a.alpha, sum[t], a.beta, zeroOf<T>());
                            }
                        } else {
                            const int outRow = dests[s * 32 + sub * RPL + t];
                            a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, sum[t], a.beta, a.y[outRow])
                                                  : epilogue<false>(a.alpha, sum[t], a.beta, zeroOf<T>());
                        }
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
        if (ring[0].s >= SUBS) break;
        step(ring[0], ring[2]);
        if (ring[1].s >= SUBS) break;
        step(ring[1], ring[0]);
        if (ring[2].s >= SUBS) break;
        step(ring[2], ring[1]);
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
#ifdef SPGPU_TRACE_BLOCKS
    if (spgpuTraceBuffer && lane == 0)
        atomicMax(&spgpuTraceBuffer[3 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
#endif
}

#ifndef SPGPU_RAGGED_UNROLL
#define SPGPU_RAGGED_UNROLL(RPL) ((RPL) >= 4 ? 2 : 3) /* wave-wide loads per stage; 3 keeps the 8-byte kernels at 112 VGPRs (4 wavefronts per SIMD: two 8-wavefront workgroups per CU) with two stages in flight */
#endif
/* Shapes (SPGPU_RAGGED_SHAPE; 0 is the default): workgroup lanes / tile / sub-groups per workgroup. */
/* Returns true if the deep kernels have to follow (false: the launch runs the deep list's items itself). */
template <typename T, int RPL, bool IS_HELL, bool DEEP>
static bool launchRagged(hipStream_t stream, const SlabArgs<T>& a, int shape, bool tiled)
{
    constexpr int UNROLL = SPGPU_RAGGED_UNROLL(RPL);
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
    case 4: SPGPU_RAGGED_Z(8, 49152, 64, 17408); break; /* 2 048 rows per workgroup, results staged by destination */
    case 5: SPGPU_RAGGED_Z(8, 49152, 32, 17408); break;
    default: SPGPU_RAGGED(8, 65536, 32); break;
    }
    return true;
#undef SPGPU_RAGGED
#undef SPGPU_RAGGED_Z
}
