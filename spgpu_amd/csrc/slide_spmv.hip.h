#pragma once
/*
 * EXPERIMENT (lab builds only, SPGPU_SLIDE=1; measured in round 4, not adopted: profiles/r04_exp_slide_tile.txt).
 *
 * A moving x tile for rows as they come whose columns ascend inside a row and lie inside a window around the row that is
 * several LDS tiles wide -- too wide for the x-tile form, whose tile stands still, and narrow enough that the gathers of
 * neighbouring rows meet in L2, where every one of them still costs a 128-byte line on the L2 -> L1 path (the 65 536-column
 * window of BASELINE configs[1]: 41 GB of lines for 2.6 GB of x values, 1.41 ms).
 *
 * All rows of a workgroup walk their slab columns in step.  At slab column k the k-th entries of 2 048 neighbouring rows
 * lie around one place of x (ascending columns: the k-th of L entries sits about (k + 1/2)/L of the way from the row's
 * first to its last column), so the workgroup keeps a CIRCULAR tile of x in LDS (128 KiB: one workgroup per CU) and moves
 * it along with k: after the gathers of a slab column every lane appends one 16-byte piece of x behind the tile's end, over
 * the piece that fell out at its start.  An entry whose column lies outside the tile at that moment is gathered from global
 * memory -- so the result never depends on where the tile stands; the schedule of the tile (a straight line from the
 * workgroup's lowest first column to its highest last column) only decides how many gathers stay in LDS (86 % on the
 * 65 536-column window).
 *
 * A lane owns RPL neighbouring rows (16-byte coefficient loads), a wavefront 64 * RPL consecutive rows, as in
 * slabSpmvKernel with one phase; a row's products are added in ascending k, no whole-wave tail rows: the order of the
 * SWEEP form (orc_?hellspmv with phases = 1; reference: one thread per row, hell_spmv_base_template.cuh:104-215).
 *
 * Per stage of UNROLL slab columns a lane has in flight: the coefficients and indices of the next stage, the pieces of x
 * that extend the tile during the next stage, and the global gathers of this stage's outside entries; the barriers between
 * slab columns wait for LDS only.
 *
 * Result: 1.41 -> 1.30 ms on the window pattern, bit for bit the 1-phase order; 0.80 ms on the band (0.64 in its default
 * form).  With the LDS taken by the tile a CU holds ONE workgroup, and nothing overlaps its prologue (three dependent round
 * trips and the first tile), its barriers (two per slab column) or the outside gathers (loads retire in issue order: a gather
 * requested behind a prefetch waits for it).  Deeper pipelines (requests two stages ahead, 4 stages of coefficients in
 * registers; 512 lanes with two strips each) did not fit the register file: 184 VGPRs wanted where 16 wavefronts leave 128,
 * and a spilled value's reload waits for every load in flight.  Left as this experiment.
 */
/* (included inside namespace spgpu by ellpack_spmv.hip) */

/* LDS writes and reads of this wavefront done, then the workgroup's barrier.  (__syncthreads() also waits for every global
 * load in flight -- the prefetches this kernel lives on.) */
__device__ inline void ldsBarrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T, int RPL, bool IS_HELL, int BLOCK, int TILE_ELEMS>
__global__ __launch_bounds__(BLOCK) void slideSpmvKernel(const SlabArgs<T> a)
{
    static_assert((TILE_ELEMS & (TILE_ELEMS - 1)) == 0, "the circular tile is indexed by column & (TILE_ELEMS - 1)");
    constexpr int WAVES = BLOCK / kWave;
    constexpr int GROUP_ROWS = kWave * RPL;
    constexpr int UNROLL = 4;
    constexpr int PIECE = 16 / (int)sizeof(T);
    constexpr int EXT = BLOCK * PIECE; /* the most the tile moves per slab column: one piece per lane */
    constexpr unsigned MASK = TILE_ELEMS - 1;

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const T* __restrict__ x = a.x;
    T* tile = ldsArray<T, TILE_ELEMS>();
    struct Seen {
        int lowest, highest, longest, rows;
    };
    Seen* seen = ldsArray<Seen, WAVES>();

    /* Workgroup ids go round the 8 XCDs; neighbouring row blocks read neighbouring pieces of x, so a run of 32 blocks goes to
     * one XCD (its 32 CUs hold one workgroup each: what the neighbour appends to its tile is then in this XCD's L2). */
    unsigned block = blockIdx.x;
    {
        const unsigned whole = gridDim.x / 256u * 256u;
        if (block < whole) {
            const unsigned within = block % 256u;
            block = block - within + (within % 8u) * 32u + within / 8u;
        }
    }
    const long long row0 = ((long long)block * WAVES + wave) * GROUP_ROWS + (long long)lane * RPL;
    const bool live = row0 < a.rows;
    long long slab = 0;
    if (live) {
        if constexpr (IS_HELL) {
            const unsigned r0 = (unsigned)row0, hs = (unsigned)a.hackSize;
            const unsigned hack = r0 / hs;
            slab = (long long)a.hackOffsets[hack] + (r0 - hack * hs);
        } else {
            slab = row0;
        }
    }
    int len[RPL];
    int laneLongest = 0;
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const long long r = row0 + t;
        len[t] = r < a.rows ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
        laneLongest = len[t] > laneLongest ? len[t] : laneLongest;
    }
    const T* __restrict__ vals = a.cM + slab;
    const int* __restrict__ idxs = a.rP + slab;

    /* Where do the workgroup's rows begin and end?  (first and last entry of every row: the extremes of a row whose
     * columns ascend; any other row is still added up correctly, through the global gathers) */
    Seen mine{0x7fffffff, -0x7fffffff - 1, laneLongest, 0};
    {
        int first[RPL], last[RPL];
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            first[t] = len[t] > 0 ? idxs[t] : 0;
            last[t] = len[t] > 0 ? idxs[t + (long long)(len[t] - 1) * a.idxStride] : 0;
        }
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            if (len[t] > 0) {
                const int f = first[t] - a.baseIndex, l = last[t] - a.baseIndex;
                const int low = f < l ? f : l, high = f < l ? l : f;
                mine.lowest = low < mine.lowest ? low : mine.lowest;
                mine.highest = high > mine.highest ? high : mine.highest;
                mine.rows += 1;
            }
        }
    }
    mine.lowest = waveMin(mine.lowest);
    mine.highest = waveMax(mine.highest);
    mine.longest = waveMax(mine.longest);
    mine.rows = waveMax(mine.rows);
    if (lane == 0)
        seen[wave] = mine;
    __syncthreads();
    int lo = 0x7fffffff, hi = -0x7fffffff - 1, longest = 0, any = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const Seen other = seen[w];
        lo = other.lowest < lo ? other.lowest : lo;
        hi = other.highest > hi ? other.highest : hi;
        longest = other.longest > longest ? other.longest : longest;
        any |= other.rows;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    longest = __builtin_amdgcn_readfirstlane(longest);
    const bool tiled = any != 0 && lo >= 0; /* an index below the base: no tile, every entry through the global gathers */

    /* The tile's schedule.  tileAt(k): first column of the tile while slab column k is consumed. */
    const long long span = tiled ? (long long)hi - lo + 1 : 0;
    const long long stepQ = (tiled && span > TILE_ELEMS && longest > 0) ? (span << 16) / longest : 0; /* columns per slab column, 16 fractional bits */
    auto wanted = [&](int k) -> int {
        if (stepQ == 0)
            return lo;
        long long start = (long long)lo + ((stepQ * (2 * (long long)k + 1)) >> 17) - TILE_ELEMS / 2;
        start = start < lo ? lo : start;
        start = start + TILE_ELEMS > (long long)hi + 1 ? (long long)hi + 1 - TILE_ELEMS : start;
        return (int)start;
    };
    auto next = [&](int base, int k) -> int { /* the tile never moves back, and at most EXT columns at a time */
        const int want = wanted(k);
        return want <= base ? base : (want - base > EXT ? base + EXT : want);
    };
    /* columns of x that exist for sure: lo .. hi */
    auto loadPiece = [&](int col) -> Pack<T, PIECE> {
        Pack<T, PIECE> w;
        if (col + PIECE - 1 <= hi) {
            w = loadPackElementAligned<T, PIECE>(x + col);
        } else {
#pragma unroll
            for (int q = 0; q < PIECE; ++q)
                w.v[q] = col + q <= hi ? x[col + q] : zeroOf<T>();
        }
        return w;
    };
    auto storePiece = [&](int col, const Pack<T, PIECE>& w, int count) { /* the first `count` elements of the piece */
#pragma unroll
        for (int q = 0; q < PIECE; ++q)
            if (q < count)
                tile[(unsigned)(col + q) & MASK] = w.v[q];
    };

    int base = tiled ? wanted(0) : 0;
    if (tiled) {
        const long long have = (long long)hi + 1 - base;
        const int count = have < TILE_ELEMS ? (int)have : TILE_ELEMS;
        for (int p0 = threadIdx.x * PIECE; p0 < count; p0 += 4 * EXT) {
            Pack<T, PIECE> w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p0 + q * EXT < count)
                    w[q] = loadPiece(base + p0 + q * EXT);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p0 + q * EXT < count)
                    storePiece(base + p0 + q * EXT, w[q], count - (p0 + q * EXT));
        }
    }

    struct Stage {
        Pack<T, RPL> v[UNROLL];
        Pack<int, RPL> c[UNROLL];
    };
    auto fetch = [&](int kBase, Stage& s) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u;
            if (k < laneLongest) {
                s.v[u] = loadPack<true, T, RPL>(vals + (long long)k * a.valStride);
                s.c[u] = loadPack<true, int, RPL>(idxs + (long long)k * a.idxStride);
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    s.v[u].v[t] = zeroOf<T>();
                    s.c[u].v[t] = a.baseIndex;
                }
            }
        }
    };
    /* the pieces that extend the tile behind the slab columns kBase .. kBase + UNROLL - 1; bases[u] = the tile's start at
     * kBase + u (bases[UNROLL]: at the next stage's first column) */
    struct Extension {
        Pack<T, PIECE> w[UNROLL];
    };
    auto schedule = [&](int kBase, int from, int (&bases)[UNROLL + 1]) {
        bases[0] = from;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            bases[u + 1] = tiled ? next(bases[u], kBase + u + 1) : from;
    };
    auto request = [&](const int (&bases)[UNROLL + 1], Extension& e) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int moved = bases[u + 1] - bases[u];
            const int mineAt = (int)threadIdx.x * PIECE;
            if (mineAt < moved)
                e.w[u] = loadPiece(bases[u] + TILE_ELEMS + mineAt);
        }
    };

    T sum[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t)
        sum[t] = zeroOf<T>();

    Stage cur, nxt;
    Extension extNow, extNext;
    int basesNow[UNROLL + 1], basesNext[UNROLL + 1];
    fetch(0, cur);
    schedule(0, base, basesNow);
    request(basesNow, extNow);
    __syncthreads(); /* the first tile is in place (this one waits for the global loads too: they are the fill) */

    for (int kBase = 0; kBase < longest; kBase += UNROLL) { /* workgroup-uniform */
        /* this stage's entries outside the tile: their gathers first (the oldest requests of the stage: what follows stays in
         * flight while they are consumed) */
        T xv[UNROLL][RPL];
        bool use[UNROLL][RPL], inside[UNROLL][RPL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long have = (long long)hi + 1 - basesNow[u];
            const unsigned count = !tiled ? 0u : (have < TILE_ELEMS ? (unsigned)have : (unsigned)TILE_ELEMS);
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const int col = cur.c[u].v[t] - a.baseIndex;
                use[u][t] = kBase + u < len[t] && col >= 0;
                inside[u][t] = (unsigned)(col - basesNow[u]) < count;
                if (use[u][t] && !inside[u][t])
                    xv[u][t] = x[col];
            }
        }
        schedule(kBase + UNROLL, basesNow[UNROLL], basesNext);
        request(basesNext, extNext);
        fetch(kBase + UNROLL, nxt);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const int col = cur.c[u].v[t] - a.baseIndex;
                const T fromTile = tile[(unsigned)col & MASK];
                const T xvNow = inside[u][t] ? fromTile : xv[u][t];
                sum[t] = pick(use[u][t], mulAdd(cur.v[u].v[t], xvNow, sum[t]), sum[t]);
            }
            const int moved = basesNow[u + 1] - basesNow[u];
            if (moved > 0) { /* workgroup-uniform */
                ldsBarrier(); /* every wavefront has read what it wanted of the piece that falls out */
                const int mineAt = (int)threadIdx.x * PIECE;
                if (mineAt < moved) {
                    const int col = basesNow[u] + TILE_ELEMS + mineAt;
                    const int left = moved - mineAt;
                    const long long exist = (long long)hi + 1 - col;
                    storePiece(col, extNow.w[u], left < exist ? left : (int)exist);
                }
                ldsBarrier(); /* the tile of the next slab column is complete */
            }
        }
        cur = nxt;
        extNow = extNext;
#pragma unroll
        for (int u = 0; u <= UNROLL; ++u)
            basesNow[u] = basesNext[u];
    }

    if (!live)
        return;
    const bool hasBeta = isNotZero(a.beta);
    if (!a.rIdx && a.wideIO && row0 + RPL <= a.rows) {
        Pack<T, RPL> out;
        if (hasBeta) {
            const Pack<T, RPL> yv = loadPack<false, T, RPL>(a.y + row0);
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                out.v[t] = epilogue<true>(a.alpha, sum[t], a.beta, yv.v[t]);
        } else {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                out.v[t] = epilogue<false>(a.alpha, sum[t], a.beta, zeroOf<T>());
        }
        storePack<T, RPL>(a.z + row0, out);
    } else {
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            const long long r = row0 + t;
            if (r < a.rows) {
                const int outRow = a.rIdx ? a.rIdx[r] : (int)r;
                a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, sum[t], a.beta, a.y[outRow])
                                      : epilogue<false>(a.alpha, sum[t], a.beta, zeroOf<T>());
            }
        }
    }
}

template <typename T, int RPL, bool IS_HELL> static void launchSlide(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int BLOCK = 1024;
    constexpr int TILE_ELEMS = 131072 / (int)sizeof(T);
    const long long rowsPerBlock = (long long)BLOCK * RPL;
    const unsigned blocks = (unsigned)(((long long)a.rows + rowsPerBlock - 1) / rowsPerBlock);
    hipLaunchKernelGGL((slideSpmvKernel<T, RPL, IS_HELL, BLOCK, TILE_ELEMS>), dim3(blocks), dim3(BLOCK), 0, stream, a);
}

