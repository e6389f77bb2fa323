/*
 * ELL and HELL SpMV for gfx950 (MI355X):  z = alpha*A*x + beta*y.
 *
 * C ABI: spgpu{S,D,C,Z}hellspmv (include/spgpu/hell.h, reference hell.h:45-169)
 *        spgpu{S,D,C,Z}ellspmv  (include/spgpu/ell.h,  reference ell.h:46-173)
 * Behaviour follows the reference dispatchers/kernels
 * (kernels/hell_spmv_base.cuh:103-157, hell_spmv_base_template.cuh:19-357,
 *  kernels/ell_spmv_base.cuh:99-146, ell_spmv_base_template.cuh:102-425,
 *  ell_spmv_base_nors.cuh:17-340); the kernel design below is new.
 *
 * ---- Wavefront design ("slab" kernel) ------------------------------------
 * Both formats store a block of 32 consecutive rows as a column-major slab:
 * element (row r, k-th entry) sits at  slabBase + r%32 + k*stride  with
 * stride = hackSize (HELL) or the pitch (ELL).  One 64-lane wavefront owns
 * one such 32-row group (for hackSize == 32: exactly one hack).
 *
 *   RPL   = rows per lane = 16 B / sizeof(T)  (S:4  D:2  C:2  Z:1)
 *   LPC   = 32 / RPL lanes cover one slab column with one 16-B load each
 *   PH    = 64 / LPC = 2*RPL "phases": lane group p handles entries k = p, p+PH, ...
 *
 * A wave-wide load therefore moves PH slab columns at once: 1 KiB of
 * coefficients (global_load_dwordx4 per lane) plus the matching indices,
 * and for hackSize == 32 those PH columns are contiguous in memory, so the
 * wave streams the hack front to back in 1-KiB pieces.  Each lane gathers
 * x for its RPL rows, keeps RPL running sums, and the PH partial sums of a
 * row are combined with log2(PH) lane-xor shuffles (DPP / ds_bpermute; no LDS,
 * no barrier).  Lanes of phase 0 apply alpha/beta and write RPL consecutive z
 * values with one wide store.
 *
 * Summation order of one row: entries k = p (mod PH) are accumulated in
 * ascending k per phase p, then phases are added pairwise (xor tree).  For
 * PH == 2 (double complex) this is exactly the reference's two-threads-per-row
 * order (hell_spmv_base_template.cuh:59-101).
 *
 * The same kernel template with RPL == 1 (element loads) and/or PH == 1 (a
 * lane walks whole rows) takes the cases the wide form cannot: streams that
 * are not 16-byte aligned, odd pitches, hackSize not a multiple of RPL
 * (every lane derives its hack from its own first row, so any hackSize works).
 *
 * Roofline: HBM bandwidth.  Algorithmic bytes per nonzero: sizeof(T) + 4;
 * per row: 4 (rS) + sizeof(T) (z) [+ sizeof(T) for y when beta != 0]
 * [+ 4 for rIdx]; per column: sizeof(T) (x once); per hack: 4.
 */
#include "numeric.hip.h"
#include <type_traits>
#include "spgpu_internal.h"
#include "slab_args.hip.h"

#include "spgpu/ell.h"
#include "spgpu/hell.h"

#include <stdio.h>
#include <stdlib.h>

namespace spgpu {

/* Function-scope LDS: only kernels that call this allocate it (the forms without a tile keep 0 bytes of LDS). */
template <typename E, int N> __device__ inline E* ldsArray()
{
    __shared__ __attribute__((aligned(16))) E buffer[N];
    return buffer;
}

/* The wavefronts that report the form they ran in: about the quarter points of the matrix, nudged off them -- grid
 * problems put their boundary rows (the ones that never qualify) exactly on power-of-two row numbers. */
__device__ inline long long sampleGroup(long long groups, int q)
{
    const long long at = groups * q / 4 + 2 * q + 1;
    return at < groups ? at : groups - 1;
}

/*
 * RPL    rows per lane (1, or 16/sizeof(T) with 16-byte loads)
 * PH     phases: lane groups that split the entries of a row by k mod PH
 *        (PH == 1: a lane walks all entries of its rows, no cross-lane sum)
 * UNROLL slab-column loads issued back to back before the first gather
 * One wavefront owns 64/PH strips = (64/PH)*RPL consecutive rows.
 * STRIPS compiles the strip-load form in (see consume below); the form without it exists as well because the mere
 *        presence of the second loop costs the gather loop ~8 % on scattered matrices (measured; same instruction
 *        counts, so a placement / allocation effect), and the host picks per matrix (launchSlabFamily).
 * PACKED a FROZEN matrix without a row order (spgpu?SpmvFreeze, include/spgpu/tuning.h; frozen_slab below): the stage loads read
 *        the column indices from the library's 16-bit copy (a.planPacked: offsets from the group's a.packBases[group], slot for
 *        slot as in rP; 0xFFFF = "ask rP") -- 2 bytes per stored entry instead of 4.  Same columns, same order: same bits.  The
 *        rare paths (whole-wave tail rows, the sample wavefronts' span) read rP itself, which the caller's promise keeps valid.
 */
template <typename T, int RPL, int PH, bool IS_HELL, bool NT, int UNROLL, int PIPE, bool TAIL, int XPOLICY = 0, bool STRIPS = false,
          int BLOCK = kBlockThreads, int TILE_BYTES = 0, bool DEEP = false, int GPW = 1, int TAIL_EVERY = 0, bool PACKED = false>
__global__ __launch_bounds__(BLOCK) void slabSpmvKernel(const SlabArgs<T> a)
{
    /* (PACKED, measured: the fp64 kernel needs 140 VGPRs -- 3 wavefronts per SIMD, as the unpacked kernel's 146.  Capped at 128 for a
     * fourth wavefront -- amdgpu_waves_per_eu(4, 4) -- it spills 52-64 bytes per lane into its stage loop: 0.575 -> 0.896 ms; with
     * the stage consumed in two halves (16 instead of 32 registers of x alive) 56-152 bytes still.) */
    static_assert(!PACKED || (TILE_BYTES == 0 && !DEEP && RPL >= 2 && XPOLICY == 0), "packed indices: the gather and strip forms of 4- and 8-byte elements");
    using ColumnWord = typename std::conditional<PACKED, unsigned short, int>::type;
    constexpr int LPC = kWave / PH;         /* lanes that cover one slab column */
    constexpr int GROUP_ROWS = LPC * RPL;   /* rows owned by the wavefront */
    constexpr int WAVES = BLOCK / kWave;
    constexpr bool XTILE = TILE_BYTES > 0;  /* the workgroup stages the slice of x its rows touch in LDS */
    constexpr int TILE_ELEMS = TILE_BYTES / (int)sizeof(T);

    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % LPC;   /* which RPL-row strip of the group */
    const int phase = lane / LPC; /* which residue class of k */
    const T* __restrict__ x = a.x;

    /* GPW groups per wavefront (x-tile forms: 2).  A workgroup owns GPW * WAVES consecutive groups and wavefront w takes
     * the groups w and 2 * WAVES - 1 - w: when the rows were ordered by length the depths along a workgroup's groups
     * rise or fall monotonically, and pairing the two ends gives every wavefront about the same work -- the tile stays
     * allocated until the slowest wavefront is done.  It also halves the tile fills per row. */
    static_assert(GPW == 1 || GPW == 2, "one group per wavefront, or the two ends of the workgroup's range");
    auto groupOfTurn = [&](int turn) -> long long {
        const int wave = threadIdx.x >> 6;
        return (long long)blockIdx.x * (WAVES * GPW) + (turn == 0 ? wave : 2 * WAVES - 1 - wave);
    };

    /* XTILE: x[tileBase .. tileBase + tileCount) lives in `tile` once the prologue below has run */
    T* tile = nullptr;
    int tileBase = 0;
    unsigned tileCount = 0;
    if constexpr (XTILE) {
        tile = ldsArray<T, TILE_ELEMS>();
        /* Which slice of x?  Every row of the workgroup is sampled at its first and its last entry (the extremes of a
         * row whose columns ascend; any row order is still correct, entries outside the tile are gathered from global
         * memory).  If the span of the workgroup's rows fits the tile it starts at the lowest column, otherwise it is
         * centred on the mean of the rows' middles (a few far-away rows then do not drag it off). */
        ColumnProbe mine{0x7fffffff, -0x7fffffff - 1, 0, 0};
        if (phase == 0) {
            int first[GPW][RPL], last[GPW][RPL], lenAt[GPW][RPL];
#pragma unroll
            for (int turn = 0; turn < GPW; ++turn) {
                const long long r0 = groupOfTurn(turn) * GROUP_ROWS + (long long)sub * RPL;
                long long at = 0;
                if (r0 < a.rows) {
                    if constexpr (IS_HELL) {
                        const unsigned u0 = (unsigned)r0, hs = (unsigned)a.hackSize;
                        at = (long long)a.hackOffsets[u0 / hs] + (u0 % hs);
                    } else {
                        at = r0;
                    }
                }
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const long long r = r0 + t;
                    lenAt[turn][t] = r < a.rows ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
                    first[turn][t] = lenAt[turn][t] > 0 ? a.rP[at + t] : 0;
                    last[turn][t] = lenAt[turn][t] > 0 ? a.rP[at + t + (long long)(lenAt[turn][t] - 1) * a.idxStride] : 0;
                }
            }
#pragma unroll
            for (int turn = 0; turn < GPW; ++turn) {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    if (lenAt[turn][t] > 0) {
                        const int f = first[turn][t] - a.baseIndex, l = last[turn][t] - a.baseIndex;
                        const int low = f < l ? f : l, high = f < l ? l : f;
                        mine.lowest = low < mine.lowest ? low : mine.lowest;
                        mine.highest = high > mine.highest ? high : mine.highest;
                        mine.middles += ((long long)f + l) >> 1;
                        mine.rows += 1;
                    }
                }
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
        ColumnProbe* seen = ldsArray<ColumnProbe, WAVES>();
        if (lane == 0)
            seen[threadIdx.x >> 6] = mine;
        __syncthreads();
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
        /* coalesced copy: 16-byte pieces (global memory takes them at any element address), 4 per lane in flight */
        constexpr int PIECE = 16 / (int)sizeof(T);
        const T* __restrict__ from = x + tileBase;
        const unsigned pieces = tileCount / PIECE;
        for (unsigned p0 = threadIdx.x; p0 < pieces; p0 += 4u * BLOCK) {
            Pack<T, PIECE> w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p0 + q * BLOCK < pieces)
                    w[q] = loadPackElementAligned<T, PIECE>(from + (size_t)(p0 + q * BLOCK) * PIECE);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p0 + q * BLOCK < pieces)
                    storePack<T, PIECE>(tile + (size_t)(p0 + q * BLOCK) * PIECE, w[q]);
        }
        if (pieces * PIECE + threadIdx.x < tileCount)
            tile[pieces * PIECE + threadIdx.x] = from[pieces * PIECE + threadIdx.x];
        __syncthreads();
    }

    auto processGroup = [&](const long long group) {
    const long long groupRow0 = group * GROUP_ROWS;
    if (groupRow0 >= a.rows)
        return; /* whole wavefront leaves together (the workgroup's barriers are behind it) */
    const long long row0 = groupRow0 + (long long)sub * RPL;
    const bool stripLive = row0 < a.rows;

    /* First slot of this lane's strip, in elements. */
    long long slab = 0;
    if (stripLive) {
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
    /* DEEP: one very long row (or a hack of them, after the rows were ordered by length) would keep this wavefront
     * streaming long after the rest of the grid has drained -- a single wavefront moves a few GB/s.  A 32-row
     * sub-group deeper than deepCap therefore keeps only its first deepCap columns here; it registers itself in the
     * deep list and the deep kernels, launched right behind this kernel, spread the remaining columns -- in chunks,
     * a wavefront each -- add the sums below and write z.  A full list: the sub-group stays here. */
    int deepSlot = -1;
    if constexpr (DEEP) {
        static_assert(PH == 1, "the deep split is built for the shapes in which a lane walks whole rows");
        constexpr int SUB = 32 / RPL; /* lanes that hold one 32-row sub-group */
        int subDepth = laneLongest;
#pragma unroll
        for (int m = 1; m < SUB; m <<= 1) {
            const int other = laneXor(subDepth, m);
            subDepth = other > subDepth ? other : subDepth;
        }
        int slot = -1;
        if (lane % SUB == 0 && subDepth > a.deepCap)
            slot = deepRegister(a, (int)row0, subDepth, (unsigned)slab);
        deepSlot = __shfl(slot, lane & ~(SUB - 1), kWave);
        if (deepSlot >= 0) {
            laneLongest = 0;
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                len[t] = len[t] < a.deepKeep ? len[t] : a.deepKeep;
                laneLongest = len[t] > laneLongest ? len[t] : laneLongest;
            }
        }
    }
    const int groupLongest = waveMax(laneLongest); /* wave-uniform trip count */

    T sum[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t)
        sum[t] = zeroOf<T>();

    const T* __restrict__ vals = a.cM + slab;
    const int* __restrict__ idxs = a.rP + slab;
    /* PACKED: the group's 16-bit words count from here (wave-uniform: one scalar load) */
    int packBase = 0;
    if constexpr (PACKED)
        packBase = a.packBases[group];

    /* One stage = UNROLL slab columns per phase: the coefficient/index loads of a stage are
     * issued back to back (fetch), its x gathers and multiply-adds follow (consume).  With
     * PIPE the next stage is fetched BEFORE the current one is consumed, so the stream loads
     * of stage s+1 are in flight while the gathers of stage s wait for x. */
    struct Stage {
        Pack<T, RPL> v[UNROLL];
        Pack<ColumnWord, RPL> c[UNROLL];
    };
    auto fetch = [&](int kBase, Stage& s) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u * PH + phase;
            if (k < laneLongest) {
                s.v[u] = loadPack<NT, T, RPL>(vals + (long long)k * a.valStride);
                if constexpr (PACKED)
                    s.c[u] = loadPack<NT, unsigned short, RPL>(a.planPacked + slab + (long long)k * a.idxStride);
                else
                    s.c[u] = loadPack<NT, int, RPL>(idxs + (long long)k * a.idxStride);
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    s.v[u].v[t] = zeroOf<T>();
                    s.c[u].v[t] = PACKED ? (ColumnWord)0xFFFF : (ColumnWord)a.baseIndex;
                }
            }
        }
    };
    /* the 0-based column of a stage's word (PACKED: base + offset; an escape asks rP) */
    auto columnOf = [&](const Stage& s, int u, int t, int k) -> int {
        if constexpr (PACKED) {
            const unsigned word = s.c[u].v[t];
            if (word == 0xFFFFu)
                return k < len[t] ? idxs[t + (long long)k * a.idxStride] - a.baseIndex : 0;
            return packBase + (int)word;
        } else {
            return s.c[u].v[t] - a.baseIndex;
        }
    };
    /* consume(form, kBase, stage, between): the x values of the stage, then `between()`, then the multiply-adds.
     * vmcnt retires in issue order: loads issued BEFORE the x loads are waited for together with them,
     * loads issued AFTER them (in `between`) stay in flight while the x values are consumed.
     *
     * Strip form: in a stencil or band matrix in natural order neighbouring rows name neighbouring columns, so the RPL
     * x values of a strip are consecutive and come with ONE element-aligned 16-byte load instead of RPL gathers.
     * Whether a stage qualifies is a wavefront-uniform test (stageIsStrips; a per-lane choice is folded back into
     * element loads by the compiler), and a wavefront that meets scattered columns once stops testing.  The two
     * forms are separate loops on purpose: joined in one loop body their wait counts have to cover both load
     * patterns and the gathers end up waited for together with the prefetch (windowed pattern 1.38 -> 1.64 ms). */
    auto consume = [&](auto stripsTag, int kBase, const Stage& s, auto&& between) {
        constexpr bool AS_STRIPS = decltype(stripsTag)::value;
        T xv[UNROLL][RPL];
        bool use[UNROLL][RPL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u * PH + phase;
            if constexpr (AS_STRIPS) {
                /* stageIsStrips: in this slab column the rows of the strip are all present (consecutive columns) or
                 * all past their end */
                const bool present = k < len[0];
                /* an absent strip still issues its load (no divergence in the stage): from the coefficient array, which
                 * holds at least one whole strip whenever a stage runs -- x itself may be shorter than RPL elements */
                const Pack<T, RPL> w = loadPackElementAligned<T, RPL>(present ? x + (PACKED ? packBase + (int)s.c[u].v[0] : (int)s.c[u].v[0] - a.baseIndex) : a.cM);
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    use[u][t] = present;
                    xv[u][t] = w.v[t];
                }
            } else if constexpr (XTILE) {
                /* from the tile where the column lies inside it (LDS reads retire on lgkmcnt: the stream prefetch,
                 * on vmcnt, stays in flight); the branch over the global gathers is wavefront-uniform per slab
                 * column and not taken when the tile covers the workgroup's columns */
                bool outside = false;
                unsigned at[RPL];
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = s.c[u].v[t] - a.baseIndex;
                    use[u][t] = k < len[t] && col >= 0;
                    at[t] = (unsigned)(col - tileBase);
                    const bool inside = at[t] < tileCount;
                    outside |= use[u][t] && !inside;
                    xv[u][t] = tile[inside ? at[t] : 0u];
                }
                if (__ballot(outside) != 0ull) {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        if (use[u][t] && at[t] >= tileCount)
                            xv[u][t] = x[s.c[u].v[t] - a.baseIndex];
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = columnOf(s, u, t, k);
                    use[u][t] = k < len[t] && col >= 0;
                    xv[u][t] = loadX<XPOLICY>(x + (use[u][t] ? col : 0));
                }
            }
        }
        between();
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                sum[t] = pick(use[u][t], mulAdd(s.v[u].v[t], xv[u][t], sum[t]), sum[t]);
            }
        }
    };
    auto stageIsStrips = [&](int kBase, const Stage& s) -> bool { /* wavefront-uniform */
        bool scattered = false;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u * PH + phase;
            const bool present = k < len[0];
#pragma unroll
            for (int t = 0; t < RPL; ++t) { /* all rows of the strip present with consecutive columns, or all absent */
                if constexpr (PACKED) /* (an escape -- 0xFFFF: the column is in rP -- is never part of a strip; a word's column is >= 0) */
                    scattered |= (k < len[t]) != present ||
                                 (present && (s.c[u].v[t] == 0xFFFFu || (unsigned)s.c[u].v[t] != (unsigned)s.c[u].v[0] + (unsigned)t));
                else
                    scattered |= (k < len[t]) != present ||
                                 (present && (s.c[u].v[0] - a.baseIndex < 0 || s.c[u].v[t] != s.c[u].v[0] + t));
            }
        }
        return __ballot(scattered) == 0ull;
    };

    /* the sample wavefronts: first to last column over the group's rows (their first and last entries: the extremes of
     * rows whose columns ascend), or "unbounded" if an index lies below the base */
    auto columnSpan = [&]() -> long long { /* call with the whole wavefront */
        int lowest = 0x7fffffff, highest = -1;
        bool below = false;
        if (phase == 0) {
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                if (len[t] > 0) {
                    const int f = idxs[t] - a.baseIndex, l = idxs[t + (long long)(len[t] - 1) * a.idxStride] - a.baseIndex;
                    below |= f < 0 || l < 0;
                    lowest = f < lowest ? f : lowest;
                    lowest = l < lowest ? l : lowest;
                    highest = f > highest ? f : highest;
                    highest = l > highest ? l : highest;
                }
            }
        }
        lowest = waveMin(lowest);
        highest = waveMax(highest);
        if (__ballot(below) != 0ull)
            return 1ll << 40;
        return highest < lowest ? 0ll : (long long)highest - lowest + 1;
    };

    constexpr int STEP = PH * UNROLL;
    /* TAIL: when at most kTailLanes lanes of the wavefront still have entries left, the
     * slab loop would run on with >= 7/8 of its lanes idle (ragged matrices: one long row keeps a whole
     * group looping).  The loop stops there and the few remaining rows are finished one at a time by the
     * WHOLE wavefront: lane l takes entries tailFrom + l, + 64, ...; the 64 partial sums are combined
     * with lane-xor shuffles and added to the owner lane's running sum. */
    int tailFrom = groupLongest;
    /* TAIL_EVERY: the switch is only considered at multiples of that many columns -- a kernel with shorter stages then
     * adds every row in exactly the order of the kernel whose stage is TAIL_EVERY columns (the x-tile form of the fp64
     * kernels has 4-column stages and must give the bits of the 8-column gather / strip kernels it alternates with) */
    constexpr int TAIL_STRIDE = TAIL_EVERY > 0 ? TAIL_EVERY : PH * UNROLL;
    auto switchToTail = [&](int kBase) -> bool {
        if constexpr (TAIL) {
            if (kBase % TAIL_STRIDE == 0 && __popcll(__ballot(kBase < laneLongest)) <= a.tailLanes) {
                tailFrom = kBase;
                return true;
            }
        }
        return false;
    };
    constexpr bool STRIPS_POSSIBLE = STRIPS && RPL > 1 && XPOLICY == 0;
    int kBase = 0;
    bool done = false; /* tail taken */
    if constexpr (PIPE) {
        Stage cur, nxt;
        fetch(0, cur);
        /* one stage: `form` says how its x values are fetched */
        auto stage = [&](auto form) {
            if constexpr (PIPE == 2) {
                /* prefetch issued after the current x loads: younger in vmcnt order, stays in flight */
                consume(form, kBase, cur, [&] { fetch(kBase + STEP, nxt); });
            } else {
                fetch(kBase + STEP, nxt); /* lanes past their rows' end fetch nothing */
                consume(form, kBase, cur, [] {});
            }
            cur = nxt;
        };
        if constexpr (STRIPS_POSSIBLE) {
            for (; kBase < groupLongest; kBase += STEP) {
                if (switchToTail(kBase)) {
                    done = true;
                    break;
                }
                if (!stageIsStrips(kBase, cur))
                    break; /* scattered columns: the gather loop takes over from this stage */
                stage(std::true_type{});
            }
            /* three sample wavefronts tell the host which form this matrix runs in (launchSlabFamily): 2 strips,
             * 3 columns inside a window an LDS tile holds, 1 scattered */
            if (a.feedback) {
                const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
                if (group == sampleGroup(groups, 1) || group == sampleGroup(groups, 2) || group == sampleGroup(groups, 3)) {
                    /* rows that fit one stage: placing and filling an LDS tile costs two round trips more than the row's
                     * one stage of gathers (1 M-row 5-point Laplacian: 25.9 us through the tile, 19.4 as gathers) */
                    const int other = groupLongest > STEP && columnSpan() <= a.tileSpanLimit ? 3 : 1;
                    for (int q = 1; q <= 3; ++q)
                        if (group == sampleGroup(groups, q) && lane == 0)
                            /* at least half of it as strips -- and more than one stage of it: the test costs about a
                             * third of a stage, which a single stage of strips does not earn back (5-point Laplacian,
                             * 16.7 M rows: 258 us with it, 251 us as gathers) */
                            a.feedback[q - 1] = a.feedbackTag | (2 * kBase >= groupLongest && groupLongest > STEP ? 2 : other);
                }
            }
        }
        /* (the gather-only and x-tile forms do not report: a walk over the sample wavefronts' indices compiled into this
         * kernel cost its hot loop 7-8 % on scattered columns although three wavefronts ran it -- 1.40 -> 1.51 ms on the
         * 65 536-wide window pattern, profiles/r03_ab_gather_feedback.txt; formProbeKernel below looks instead) */
        if (!done) {
            /* kBase is wavefront-uniform; saying so keeps the loop counter (and every k derived from it) scalar */
            for (kBase = __builtin_amdgcn_readfirstlane(kBase); kBase < groupLongest; kBase += STEP) {
                if (switchToTail(kBase))
                    break;
                stage(std::false_type{});
            }
        }
    } else {
        for (; kBase < groupLongest; kBase += STEP) {
            if (switchToTail(kBase))
                break;
            Stage cur;
            fetch(kBase, cur);
            consume(std::false_type{}, kBase, cur, [] {});
        }
    }

    if constexpr (TAIL) {
        /* all PH lanes of a strip share laneLongest, so they enter and leave `pending` together */
        unsigned long long pending = __ballot(tailFrom < laneLongest);
        while (pending) { /* wave-uniform */
            const int owner = (__ffsll((long long)pending) - 1) % LPC; /* the strip's phase-0 lane */
            pending &= ~__ballot(sub == owner);
            const long long ownerSlab = __shfl(slab, owner, kWave);
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const int rowLen = __shfl(len[t], owner, kWave);
                if (rowLen <= tailFrom)
                    continue;
                const T* __restrict__ rowVals = a.cM + ownerSlab + t;
                const int* __restrict__ rowIdxs = a.rP + ownerSlab + t;
                T part = zeroOf<T>();
                for (int k0 = tailFrom + lane; k0 < rowLen + (kTailUnroll - 1) * kWave; k0 += kTailUnroll * kWave) {
                    T tv[kTailUnroll];
                    int tc[kTailUnroll];
#pragma unroll
                    for (int u = 0; u < kTailUnroll; ++u) {
                        const int k = k0 + u * kWave;
                        const bool in = k < rowLen;
                        tv[u] = in ? rowVals[(long long)k * a.valStride] : zeroOf<T>();
                        tc[u] = in ? rowIdxs[(long long)k * a.idxStride] - a.baseIndex : -1;
                    }
                    T tx[kTailUnroll];
#pragma unroll
                    for (int u = 0; u < kTailUnroll; ++u)
                        tx[u] = x[tc[u] >= 0 ? tc[u] : 0];
#pragma unroll
                    for (int u = 0; u < kTailUnroll; ++u)
                        part = pick(tc[u] >= 0, mulAdd(tv[u], tx[u], part), part);
                }
#pragma unroll
                for (int m = 1; m < kWave; m <<= 1)
                    part = add(part, laneXor(part, m));
                if (lane == owner)
                    sum[t] = add(sum[t], part);
            }
        }
    }

    /* Combine the PH phase partials of every row. */
#pragma unroll
    for (int m = LPC; m < kWave; m <<= 1) {
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            sum[t] = add(sum[t], laneXor(sum[t], m));
    }

    if (phase != 0 || !stripLive)
        return;

    if constexpr (DEEP) {
        if (deepSlot >= 0) { /* raw sums: deepFinishKernel finishes these rows */
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                a.deepPartials[(size_t)deepSlot * 32 + (size_t)((row0 + t) & 31)] = sum[t];
            return;
        }
    }

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
    }; /* processGroup */

#pragma unroll 1
    for (int turn = 0; turn < GPW; ++turn)
        processGroup(groupOfTurn(turn));
}

/*
 * The columns >= deepCap of the sub-groups (32 rows) a DEEP main kernel registered, in two launches right behind it.
 *
 * deepItemsKernel: a wavefront per item, an item = CHUNK columns of one sub-group.  The wavefront reads the chunk the
 * way the format stores it -- 32/RPL lanes with RPL rows each cover a slab column, PH = 64 / (32/RPL) columns per load
 * instruction -- UNROLL load instructions per stage, the next stage requested behind the current stage's gathers.  The
 * items of one very deep sub-group and of many shallow ones alike spread over the whole chip (measured before, with a
 * workgroup per hashed queue of sub-groups: 87-139 us for 108 MB, the grid waiting for its fullest queue).
 * A chunk sum = its PH phase sums (each over ascending k) combined pairwise; it goes to deepItemSums.
 * x comes from global memory: these are the few long rows, their own columns give them their locality.
 *
 * deepFinishKernel: 32 lanes per entry.  Sum of one row = what the main kernel left in deepPartials, plus the chunk
 * sums in chunk order (orc_?spmv_deep restates exactly this); then the SpMV epilogue and the store through rIdx.  The
 * workgroup that finishes last zeroes the header: every workgroup has read it by then, and the next call finds an
 * empty list (a captured graph can be replayed).
 */
template <typename T, int RPL, bool IS_HELL, int UNROLL, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void deepItemsKernel(const SlabArgs<T> a)
{
    constexpr int LPC = 32 / RPL;    /* lanes per slab column of 32 rows */
    constexpr int PH = kWave / LPC;  /* columns per wave-wide load */
    constexpr int STEP = PH * UNROLL;
    static_assert(CHUNK % STEP == 0, "a chunk is a whole number of stages");
    constexpr int WAVES = kBlockThreads / kWave;

    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % LPC, phase = lane / LPC;
    /* Round trip 1: the header and the item's record together (the grid has a wavefront for every item the list can hold, and
     * the record lies inside the array whatever the header says).  Round trip 2: the row lengths and ALL of the item's slab
     * columns -- the addresses come from the record, and a column below the sub-group's depth exists in the arrays whether a
     * given row reaches it or not (what lies there is never used: the test is k < len).  Round trips 3 and 4: the gathers of
     * the two halves.  (Before: header, entry number, entry, lengths and hack offset, first half, gathers, second half, gathers.) */
    const int item = (int)blockIdx.x * WAVES + (int)(threadIdx.x >> 6);
    const int handedOut = a.deepHeader[SPGPU_DEEP_HEAD_ITEMS];
    const int cut = a.deepHeader[SPGPU_DEEP_HEAD_CUT];
    const SpgpuDeepItem mine = item < SPGPU_DEEP_ITEMS ? a.deepItems[item] : SpgpuDeepItem{0, 0u, 0, 0};
    const int fresh = SPGPU_DEEP_ITEMS - cut; /* items below this were written by this call */
    const int items = handedOut < fresh ? handedOut : fresh;
    if (item >= items)
        return;
    {
        const int kFirst = a.deepKeep + mine.chunk * CHUNK;
        const int kEnd = kFirst + CHUNK < mine.depth ? kFirst + CHUNK : mine.depth;
        const long long row0 = (long long)mine.row0 + (long long)sub * RPL;
        long long slab = (long long)mine.base + (long long)sub * RPL;
        if constexpr (IS_HELL) {
            if ((a.hackSize & 31) != 0 && row0 < a.rows) { /* the sub-group may straddle hacks: the lane's own hack (wavefront-uniform test) */
                const unsigned r0 = (unsigned)row0, hs = (unsigned)a.hackSize;
                const unsigned hack = r0 / hs;
                slab = (long long)a.hackOffsets[hack] + (r0 - hack * hs);
            }
        }
        int len[RPL];
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            const long long r = row0 + t;
            len[t] = r < a.rows ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
        }
        const bool rowsExist = row0 < a.rows; /* a strip beyond the last row: nothing of it is loaded */
        /* how far this lane may load: the item's end -- except where the sub-group straddles hacks (hackSize not a multiple of
         * 32): the lane's own hack may be shallower than the sub-group, so there its own rows' lengths bound the loads (and are
         * waited for first) */
        int loadEnd = kEnd;
        if constexpr (IS_HELL) {
            if ((a.hackSize & 31) != 0) {
                int own = 0;
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    own = len[t] > own ? len[t] : own;
                loadEnd = own < kEnd ? own : kEnd;
            }
        }
        const T* __restrict__ vals = a.cM + slab;
        const int* __restrict__ idxs = a.rP + slab;
        constexpr int STAGES = CHUNK / STEP;
        Pack<T, RPL> v[STAGES][UNROLL];
        Pack<int, RPL> c[STAGES][UNROLL];
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = kFirst + s * STEP + u * PH + phase;
                if (k < loadEnd && rowsExist) {
                    v[s][u] = loadPack<true, T, RPL>(vals + (long long)k * a.valStride);
                    c[s][u] = loadPack<true, int, RPL>(idxs + (long long)k * a.idxStride);
                } else {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        v[s][u].v[t] = zeroOf<T>();
                        c[s][u].v[t] = a.baseIndex;
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            len[t] = len[t] < kEnd ? len[t] : kEnd;
        T sum[RPL];
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            sum[t] = zeroOf<T>();
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            T xv[UNROLL][RPL];
            bool use[UNROLL][RPL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = kFirst + s * STEP + u * PH + phase;
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = c[s][u].v[t] - a.baseIndex;
                    use[u][t] = k < len[t] && col >= 0;
                    xv[u][t] = a.x[use[u][t] ? col : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = pick(use[u][t], mulAdd(v[s][u].v[t], xv[u][t], sum[t]), sum[t]);
            }
            __builtin_amdgcn_sched_barrier(0); /* one half's gathers at a time */
        }
#pragma unroll
        for (int m = LPC; m < kWave; m <<= 1) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = add(sum[t], laneXor(sum[t], m));
        }
        if (phase == 0) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                a.deepItemSums[(size_t)item * 32 + (size_t)(sub * RPL + t)] = sum[t];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void deepFinishKernel(const SlabArgs<T> a)
{
    const int registered = a.deepHeader[SPGPU_DEEP_HEAD_ENTRIES];
    const int entries = registered < SPGPU_DEEP_ENTRIES ? registered : SPGPU_DEEP_ENTRIES;
    const bool hasBeta = isNotZero(a.beta);
    const int rowInGroup = threadIdx.x & 31;
    constexpr int PER_BLOCK = kBlockThreads / 32;
    for (int e = (int)blockIdx.x * PER_BLOCK + (int)(threadIdx.x >> 5); e < entries; e += (int)gridDim.x * PER_BLOCK) {
        const SpgpuDeepEntry entry = a.deepEntries[e];
        const long long r = (long long)entry.row0 + rowInGroup;
        if (entry.items <= 0 || r >= a.rows)
            continue;
        T total = a.deepPartials[(size_t)e * 32 + rowInGroup];
        constexpr int BATCH = 8; /* item sums requested together; added in item order */
        for (int c0 = 0; c0 < entry.items; c0 += BATCH) {
            T part[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (c0 + u < entry.items)
                    part[u] = a.deepItemSums[(size_t)(entry.firstItem + c0 + u) * 32 + rowInGroup];
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (c0 + u < entry.items)
                    total = add(total, part[u]);
        }
        const int outRow = a.rIdx ? a.rIdx[r] : (int)r;
        a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, total, a.beta, a.y[outRow])
                              : epilogue<false>(a.alpha, total, a.beta, zeroOf<T>());
    }
    __syncthreads(); /* every wavefront of this workgroup has used the header */
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&a.deepHeader[SPGPU_DEEP_HEAD_TICKET], 1);
        if (ticket == (int)gridDim.x - 1) {
            /* a list that overflowed: say so where the host can see it (spgpuDeepListOverflows, include/spgpu/tuning.h) */
            const int handedOut = a.deepHeader[SPGPU_DEEP_HEAD_ITEMS];
            if (a.deepOverflow && (registered > SPGPU_DEEP_ENTRIES || handedOut > SPGPU_DEEP_ITEMS)) {
                a.deepOverflow[1] = registered;
                a.deepOverflow[2] = handedOut;
                atomicAdd_system(&a.deepOverflow[0], 1); /* the streams of a handle share the word: two of them may overflow at once */
            }
            a.deepHeader[SPGPU_DEEP_HEAD_ENTRIES] = 0;
            a.deepHeader[SPGPU_DEEP_HEAD_ITEMS] = 0;
            a.deepHeader[SPGPU_DEEP_HEAD_TICKET] = 0;
            a.deepHeader[SPGPU_DEEP_HEAD_CUT] = 0;
        }
    }
}

/*
 * What do the columns of this matrix look like?  Three wavefronts (the sample groups of slabSpmvKernel) walk their rows'
 * indices and report what the strip-capable kernel's samples would: 2 = neighbouring rows name consecutive columns (strip x
 * loads), 3 = the columns of a group lie inside a window an LDS tile holds, 1 = scattered.  Launched by AUTO with every
 * fourth call of the forms that do not report themselves, and by spgpu?SpmvForm (include/spgpu/tuning.h) for a caller who
 * wants to hold the answer.  STEP = the columns per stage of the strip-capable kernel of the type (its strip test is per stage).
 */
template <typename T, int RPL, int PH, bool IS_HELL, int STEP>
__global__ __launch_bounds__(kWave) void formProbeKernel(const SlabArgs<T> a)
{
    constexpr int LPC = kWave / PH, GROUP_ROWS = LPC * RPL;
    const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const long long group = sampleGroup(groups, (int)blockIdx.x + 1);
    const int lane = threadIdx.x;
    const long long row0 = group * GROUP_ROWS + (long long)lane * RPL;
    int len[RPL], longest = 0;
    long long slab = 0;
    const bool live = lane < LPC && row0 < a.rows;
    if (live) {
        if constexpr (IS_HELL) {
            const unsigned r0 = (unsigned)row0, hs = (unsigned)a.hackSize;
            slab = (long long)a.hackOffsets[r0 / hs] + (r0 % hs);
        } else {
            slab = row0;
        }
    }
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const long long r = row0 + t;
        len[t] = live && r < a.rows ? (a.rS ? a.rS[r] : a.maxNnz) : 0;
        longest = len[t] > longest ? len[t] : longest;
    }
    const int groupLongest = waveMax(longest);
    /* first column at which this lane's strip is neither "all rows present with consecutive columns" nor "all past their end" */
    int firstBad = 0x7fffffff, lowest = 0x7fffffff, highest = -1;
    bool below = false;
    for (int k = 0; k < longest; ++k) {
        const bool present = k < len[0];
        const int c0 = present ? a.rP[slab + (long long)k * a.idxStride] : 0;
        bool bad = present && c0 - a.baseIndex < 0;
#pragma unroll
        for (int t = 1; t < RPL; ++t) {
            const bool here = k < len[t];
            bad |= here != present || (present && a.rP[slab + t + (long long)k * a.idxStride] != c0 + t);
        }
        if (bad) {
            firstBad = k;
            break;
        }
    }
#pragma unroll
    for (int t = 0; t < RPL; ++t) { /* the span of the group's columns: first and last entry of every row */
        if (len[t] > 0) {
            const int f = a.rP[slab + t] - a.baseIndex, l = a.rP[slab + t + (long long)(len[t] - 1) * a.idxStride] - a.baseIndex;
            below |= f < 0 || l < 0;
            lowest = f < lowest ? f : lowest;
            lowest = l < lowest ? l : lowest;
            highest = f > highest ? f : highest;
            highest = l > highest ? l : highest;
        }
    }
    firstBad = waveMin(firstBad);
    lowest = waveMin(lowest);
    highest = waveMax(highest);
    const long long span = __ballot(below) != 0ull ? (1ll << 40) : (highest < lowest ? 0ll : (long long)highest - lowest + 1);
    const int asStrips = firstBad == 0x7fffffff ? groupLongest : firstBad / STEP * STEP; /* whole stages of strips in front */
    /* 4 = a matrix for the SWEEP form: the columns of the group reach over half of x and more (the matrix is taken to be about
     * square: the API does not say how long x is), ascend inside every sampled row (its first 64 entries), and the rows are about
     * equally long (rows walked in step wait for the longest) */
    bool sweepable = false;
    if constexpr (PH == 1 && sizeof(T) == 8) {
        bool ascends = true;
        int total = 0;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            total += len[t];
            const int look = len[t] < 64 ? len[t] : 64;
            int before = -0x7fffffff - 1;
            for (int k = 0; k < look; ++k) {
                const int c = a.rP[slab + t + (long long)k * a.idxStride];
                ascends &= c >= before;
                before = c;
            }
        }
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1)
            total += laneXor(total, m);
        const long long slots = (long long)groupLongest * GROUP_ROWS;
        sweepable = __ballot(!ascends) == 0ull && span < (1ll << 40) && 2 * span >= (long long)a.rows && groupLongest >= 2 * STEP &&
                    2 * slots <= 3 * (long long)total;
    }
    if (lane == 0 && a.feedback)
        a.feedback[blockIdx.x] = a.feedbackTag | ((RPL > 1 && 2 * asStrips >= groupLongest && groupLongest > STEP) ? 2
                                                  : (groupLongest > STEP && span <= a.tileSpanLimit ? 3 : (sweepable ? 4 : 1))); /* one stage of rows: no tile */
}

/*
 * Rows with a row order (rIdx): how far from the diagonal -- in the ORIGINAL numbering, rIdx[row] -- do their columns lie?
 * The queue kernel for ordered rows has two product shapes (ragged_spmv.hip.h, launchRagged): 2 048 rows per workgroup with
 * the results staged in LDS by destination (whole-line stores of z; 48 KiB left for the x tile) wins when the columns of a
 * window of rows fit that tile, 1 024 rows with a 64 KiB tile when they spread further (columns +-2 048 of the row: 2 048
 * rows would need 64 KiB and more).  192 sampled rows answer: 4 = three quarters of them keep within 1 024 of the
 * diagonal, 5 = they do not; 6 = whatever the columns do, the kernel's 2 048-row blocks are the windows of the order.
 * Launched by AUTO when it has no answer for the matrix, and again every 64th call.
 */
template <bool IS_HELL>
__global__ __launch_bounds__(kWave) void orderedProbeKernel(const int* rP, const int* rS, const int* hackOffsets, const int* rIdx, int hackSize,
                                                           long long idxStride, int maxNnz, int rows, int baseIndex, int* answer, int tag)
{
    const int lane = threadIdx.x;
    int near = 0, seen = 0;
    for (int q = 1; q <= 3; ++q) {
        const long long r = (long long)rows * q / 4 + 2 * q + lane;
        if (r >= rows)
            continue;
        const int len = rS ? rS[r] : maxNnz;
        if (len <= 0)
            continue;
        long long slot;
        if constexpr (IS_HELL)
            slot = (long long)hackOffsets[(unsigned)r / (unsigned)hackSize] + (unsigned)r % (unsigned)hackSize;
        else
            slot = r;
        const long long dest = rIdx[r];
        const long long first = (long long)rP[slot] - baseIndex - dest, last = (long long)rP[slot + (long long)(len - 1) * idxStride] - baseIndex - dest;
        const long long reach = (first < 0 ? -first : first) > (last < 0 ? -last : last) ? (first < 0 ? -first : first) : (last < 0 ? -last : last);
        seen += 1;
        near += reach <= 1024 ? 1 : 0;
    }
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        near += laneXor(near, m);
        seen += laneXor(seen, m);
    }
    /* Are the kernel's 2 048-row blocks the windows of the order (spgpuOellOrderAlignedDevice)?  64 rows spread over each of
     * three blocks: the rows of ONE window come from a stretch of the original numbering little longer than the window, the
     * rows of a block that straddles two windows from twice that.  Then the 2 048-row shape serves wide columns too: its tile
     * holds the one window +- 2 048 such a block touches, and the block's results are whole lines of z. */
    int blocksAreWindows = 0, blocksSeen = 0;
    for (int q = 1; q <= 3; ++q) {
        const long long block0 = ((long long)rows * q / 4) / 2048 * 2048;
        if (block0 + 2048 > rows)
            continue;
        const int dest = rIdx[block0 + lane * 32 + (lane & 31)];
        const int low = waveMin(dest), high = waveMax(dest);
        blocksSeen += 1;
        blocksAreWindows += high - low < 2048 + 512 ? 1 : 0;
    }
    if (lane == 0)
        *answer = tag | ((blocksSeen > 0 && blocksAreWindows == blocksSeen) ? 6 : (seen > 0 && 4 * near >= 3 * seen) ? 4 : 5);
}

/*
 * SWEEP form (include/spgpu/tuning.h; the caller's hint, and AUTO's choice for 8-byte elements when the probe finds such a
 * matrix): for matrices whose columns are scattered over all of x but ascend inside a row.  A lane owns PACKS packs of VEC
 * neighbouring rows (32 rows for 4- and 8-byte elements) and carries all of them through the slab columns in step; the grid
 * is small enough to be resident at once and walks the rows with a tile stride.  At any moment the rows in flight are at
 * about the same k, i.e. they gather from about the same quantile of x, and meet in L2: 10 M x 32 scattered, fp64: L2 hits
 * 22 M -> 54 M of 320 M gathers, 5.85 -> 4.5 ms.  No LDS; coefficient and index streams non-temporal.
 *
 * Order of additions.  TAIL = false: a row's products in ascending k (orc_?hellspmv / orc_?ellspmv with one phase), the
 * reference's one-thread-per-row order (hell_spmv_base_template.cuh:104-215).  TAIL = true (the types whose default kernel
 * walks whole rows: 8-byte elements): exactly that kernel's order -- pack u of a wavefront is the 64 * VEC consecutive rows
 * one of its wavefronts owns, the group hands its last rows to the whole wavefront at the slab column at which that
 * kernel would (first multiple of 8 with at most tailLanes lanes still busy; slabSpmvKernel, TAIL), and they are finished
 * the same way: so AUTO may pick this form without changing a bit of z.
 */
template <typename T, int VEC, int PACKS, bool IS_HELL, bool HAS_BETA, bool TAIL>
__global__ __launch_bounds__(kBlockThreads) void sweepSpmvKernel(const SlabArgs<T> a)
{
    const long long packs = ((long long)a.rows + VEC - 1) / VEC;
    constexpr long long TILE = (long long)kBlockThreads * PACKS;
    constexpr int TAIL_STRIDE = 8; /* the stage of the default kernel of the 8-byte types (launchSlabFamily: 1 phase x 8 columns) */
    const int lane = threadIdx.x & (kWave - 1);
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        T sums[PACKS][VEC];
        int len[PACKS][VEC];
        long long slot[PACKS];
        int tailFrom[PACKS]; /* TAIL: pack u walks the slab columns below tailFrom[u] here (wavefront-uniform) */
        int longest = 0;
        unsigned tails = 0u; /* TAIL: packs whose wavefront has tail rows (wavefront-uniform) */
#pragma unroll
        for (int u = 0; u < PACKS; ++u) {
            const long long row = (base + u * kBlockThreads + threadIdx.x) * VEC;
            slot[u] = 0;
            if (row < a.rows) {
                if constexpr (IS_HELL) {
                    const unsigned r0 = (unsigned)row, hs = (unsigned)a.hackSize;
                    const unsigned hack = r0 / hs;
                    slot[u] = (long long)a.hackOffsets[hack] + (r0 - hack * hs);
                } else {
                    slot[u] = row;
                }
            }
            int packLongest = 0;
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                sums[u][t] = zeroOf<T>();
                len[u][t] = row + t < a.rows ? (a.rS ? a.rS[row + t] : a.maxNnz) : 0;
                packLongest = len[u][t] > packLongest ? len[u][t] : packLongest;
            }
            tailFrom[u] = 0x7fffffff;
            if constexpr (TAIL) {
                const int groupLongest = waveMax(packLongest);
                for (int kBase = 0; kBase < groupLongest; kBase += TAIL_STRIDE) {
                    if (__popcll(__ballot(kBase < packLongest)) <= a.tailLanes) {
                        tailFrom[u] = kBase;
                        tails |= 1u << u;
                        break;
                    }
                }
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    len[u][t] = len[u][t] < tailFrom[u] ? len[u][t] : tailFrom[u];
                packLongest = packLongest < tailFrom[u] ? packLongest : tailFrom[u];
            }
            longest = packLongest > longest ? packLongest : longest;
        }
        for (int k = 0; k < longest; ++k) {
            Pack<T, VEC> v[PACKS];
            Pack<int, VEC> c[PACKS];
#pragma unroll
            for (int u = 0; u < PACKS; ++u) {
                bool any = false;
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    any |= k < len[u][t];
                if (any) {
                    v[u] = loadPack<true, T, VEC>(a.cM + slot[u] + (long long)k * a.valStride);
                    c[u] = loadPack<true, int, VEC>(a.rP + slot[u] + (long long)k * a.idxStride);
                } else {
#pragma unroll
                    for (int t = 0; t < VEC; ++t)
                        c[u].v[t] = a.baseIndex;
                }
            }
#pragma unroll
            for (int u = 0; u < PACKS; ++u) {
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    const int col = c[u].v[t] - a.baseIndex;
                    const bool use = k < len[u][t] && col >= 0;
                    const T xv = a.x[use ? col : 0];
                    if (use)
                        sums[u][t] = mulAdd(v[u].v[t], xv, sums[u][t]);
                }
            }
        }
        if constexpr (TAIL) {
            /* the rows a group handed over: one at a time by the WHOLE wavefront, as slabSpmvKernel's tail does -- lane l takes
             * the entries tailFrom + l, + 64, ..., the 64 partial sums are combined with lane-xor shuffles and added to the
             * owner's running sum */
            if (tails != 0u) { /* wavefront-uniform */
#pragma unroll
                for (int u = 0; u < PACKS; ++u) {
                    if (!(tails & (1u << u)))
                        continue;
                    const long long row = (base + u * kBlockThreads + threadIdx.x) * VEC;
                    int full[VEC], fullLongest = 0; /* the lengths again: len[] was cut at tailFrom */
#pragma unroll
                    for (int t = 0; t < VEC; ++t) {
                        full[t] = row + t < a.rows ? (a.rS ? a.rS[row + t] : a.maxNnz) : 0;
                        fullLongest = full[t] > fullLongest ? full[t] : fullLongest;
                    }
                    const int from = tailFrom[u];
                    unsigned long long pending = __ballot(from < fullLongest);
                    while (pending) { /* wavefront-uniform */
                        const int owner = __ffsll((long long)pending) - 1;
                        pending &= pending - 1;
                        const long long ownerSlot = __shfl(slot[u], owner, kWave);
#pragma unroll
                        for (int t = 0; t < VEC; ++t) {
                            const int rowLen = __shfl(full[t], owner, kWave);
                            if (rowLen <= from)
                                continue;
                            const T* __restrict__ rowVals = a.cM + ownerSlot + t;
                            const int* __restrict__ rowIdxs = a.rP + ownerSlot + t;
                            T part = zeroOf<T>();
                            for (int k0 = from + lane; k0 < rowLen + (kTailUnroll - 1) * kWave; k0 += kTailUnroll * kWave) {
                                T tv[kTailUnroll];
                                int tc[kTailUnroll];
#pragma unroll
                                for (int q = 0; q < kTailUnroll; ++q) {
                                    const int k = k0 + q * kWave;
                                    const bool in = k < rowLen;
                                    tv[q] = in ? rowVals[(long long)k * a.valStride] : zeroOf<T>();
                                    tc[q] = in ? rowIdxs[(long long)k * a.idxStride] - a.baseIndex : -1;
                                }
                                T tx[kTailUnroll];
#pragma unroll
                                for (int q = 0; q < kTailUnroll; ++q)
                                    tx[q] = a.x[tc[q] >= 0 ? tc[q] : 0];
#pragma unroll
                                for (int q = 0; q < kTailUnroll; ++q)
                                    part = pick(tc[q] >= 0, mulAdd(tv[q], tx[q], part), part);
                            }
#pragma unroll
                            for (int m = 1; m < kWave; m <<= 1)
                                part = add(part, laneXor(part, m));
                            if (lane == owner)
                                sums[u][t] = add(sums[u][t], part);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PACKS; ++u) {
            const long long row = (base + u * kBlockThreads + threadIdx.x) * VEC;
            if (a.wideIO && row + VEC <= a.rows) {
                Pack<T, VEC> out, yv;
                if constexpr (HAS_BETA)
                    yv = loadPack<false, T, VEC>(a.y + row);
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    out.v[t] = epilogue<HAS_BETA>(a.alpha, sums[u][t], a.beta, HAS_BETA ? yv.v[t] : zeroOf<T>());
                storePack<T, VEC>(a.z + row, out);
            } else {
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    if (row + t < a.rows)
                        a.z[row + t] = epilogue<HAS_BETA>(a.alpha, sums[u][t], a.beta, HAS_BETA ? a.y[row + t] : zeroOf<T>());
            }
        }
    }
}

#ifdef SPGPU_TRACE_BLOCKS
__device__ unsigned long long* spgpuTraceBuffer;
#endif
#include "ragged_spmv.hip.h"
#ifdef SPGPU_TUNING_VARIANTS
#include "slide_spmv.hip.h"
#include "share_spmv.hip.h"
#include "pipe_spmv.hip.h"
#endif

/* ---- host side ----------------------------------------------------------- */

static bool alignedTo(const void* p, size_t bytes)
{
    return ((uintptr_t)p % bytes) == 0;
}

template <typename T, int RPL, int PH, bool IS_HELL, int UNROLL, int PIPE = 0, bool TAIL = false, int XPOLICY = 0, bool STRIPS = false,
          int BLOCK = kBlockThreads, int TILE_BYTES = 0>
static void launchSlab(hipStream_t stream, const SlabArgs<T>& a, bool nt)
{
    constexpr int GROUP_ROWS = (kWave / PH) * RPL;
    constexpr int WAVES = BLOCK / kWave;
    const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const unsigned blocks = (unsigned)((groups + WAVES - 1) / WAVES);
    if (nt)
        hipLaunchKernelGGL((slabSpmvKernel<T, RPL, PH, IS_HELL, true, UNROLL, PIPE, TAIL, XPOLICY, STRIPS, BLOCK, TILE_BYTES>),
                           dim3(blocks), dim3(BLOCK), 0, stream, a);
    else
        hipLaunchKernelGGL((slabSpmvKernel<T, RPL, PH, IS_HELL, false, UNROLL, PIPE, TAIL, XPOLICY, STRIPS, BLOCK, TILE_BYTES>),
                           dim3(blocks), dim3(BLOCK), 0, stream, a);
}

/* The x-tile forms.  Workgroup size and tile size go together: the tile has to hold the columns of the workgroup's
 * rows, and LDS (160 KiB per CU) divided by the tile is the number of workgroups a CU overlaps.  A lane walks whole
 * rows (PH 1) with 4 slab columns per stage -- half the stage of the gather kernel: LDS gathers are short, and at 8 the
 * kernel needs 148 VGPRs, which leaves room for one 512-lane workgroup per CU only.  Shape 0 is the default; the
 * others exist for A/B runs (SPGPU_X_TILE_SHAPE):
 *   1  one wavefront per 32-row group (PH = 2 * RPL, 2 columns per stage), 512 lanes, 64 KiB  (no deep split)
 *   2  512 lanes, 64 KiB      3  256 lanes, 48 KiB
 * The coefficient/index streams always carry the non-temporal hint here. */
template <typename T, int RPL, int PH, bool IS_HELL, int UNROLL, bool TAIL, int BLOCK, int TILE_BYTES, bool DEEP, int GPW = 1,
          int TAIL_EVERY = 0>
static void launchShape(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int GROUP_ROWS = (kWave / PH) * RPL;
    constexpr int WAVES = BLOCK / kWave;
    const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const unsigned blocks = (unsigned)((groups + WAVES * GPW - 1) / (WAVES * GPW));
    hipLaunchKernelGGL((slabSpmvKernel<T, RPL, PH, IS_HELL, true, UNROLL, 2, TAIL, 0, false, BLOCK, TILE_BYTES, DEEP, GPW, TAIL_EVERY>),
                       dim3(blocks), dim3(BLOCK), 0, stream, a);
}

template <typename T, int RPL, bool IS_HELL, bool DEEP>
static void launchTiled(hipStream_t stream, const SlabArgs<T>& a, int shape)
{
    constexpr int PH1 = (sizeof(T) == 16 && !DEEP) ? 2 : 1; /* 16-byte elements keep the 2-phase shape of their default kernel */
    constexpr bool TAIL = PH1 == 1;
    switch (shape) {
#ifdef SPGPU_TUNING_VARIANTS
    case 1:
        if constexpr (!DEEP) {
            launchShape<T, RPL, (RPL > 1 ? 2 * RPL : 2), IS_HELL, (RPL > 1 ? 2 : 4), (RPL > 1), 512, 65536, false>(stream, a);
            break;
        }
        [[fallthrough]];
    case 2: launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 512, 65536, DEEP>(stream, a); break;
    case 3: launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 256, 49152, DEEP, 2>(stream, a); break;
    case 4: launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 256, 65536, DEEP, 2>(stream, a); break;
    case 5: launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 512, 65536, DEEP, 2>(stream, a); break;
#endif
    default:
        /* the default: same summation order as the type's gather / strip kernel (launchSlabFamily), so that the form
         * AUTO settles on never changes a bit of the result: 8-byte elements walk whole rows and consider the tail every
         * 8 columns; fp32 keeps its 8 phases x 2 columns; complex fp64 its 2 phases */
        if constexpr (DEEP)
            launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 256, 32768, true>(stream, a);
        else if constexpr (sizeof(T) == 4 && RPL == 4)
            launchShape<T, RPL, 2 * RPL, IS_HELL, 2, true, 512, 32768, false>(stream, a);
        else if constexpr (sizeof(T) == 8 && RPL == 2)
            launchShape<T, RPL, 1, IS_HELL, 4, true, 256, 32768, false, 1, 8>(stream, a);
        else
            launchShape<T, RPL, PH1, IS_HELL, 4, TAIL, 256, 32768, false>(stream, a);
        break;
    }
}

constexpr int kAutoSweepRows = 2 * 1024 * 1024; /* AUTO: the SWEEP form wants a grid that fills the chip (8 192 rows per workgroup); measured, scattered
                                                  * fp64, 16 and 32 per row: 1 Mi rows 1.5 x SLOWER than the gathers (x fits the L2s), 2 Mi ... 16 Mi rows 0.61 ... 0.89 x their
                                                  * time (profiles/r04_exp_sweep_rows.txt) */

/* SWEEP: 32 rows per lane (16 for 16-byte elements), at most 2 048 workgroups.  8-byte elements add in the order of their
 * default kernel (whole-wave tail rows), the others in one phase. */
template <typename T, int VEC, bool IS_HELL>
static void launchSweep(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int PACKS = sizeof(T) == 16 ? 16 : 32 / VEC;
    constexpr bool TAIL = sizeof(T) == 8;
    const long long packs = ((long long)a.rows + VEC - 1) / VEC;
    long long blocks = (packs + (long long)kBlockThreads * PACKS - 1) / ((long long)kBlockThreads * PACKS);
    blocks = blocks > 2048 ? 2048 : blocks;
    if (isNotZero(a.beta))
        hipLaunchKernelGGL((sweepSpmvKernel<T, VEC, PACKS, IS_HELL, true, TAIL>), dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((sweepSpmvKernel<T, VEC, PACKS, IS_HELL, false, TAIL>), dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, a);
}

/* Right behind a DEEP kernel.  Fixed grids (the number of items is known on the device only): with nothing
 * registered both kernels read the header and leave. */
template <typename T, int RPL, bool IS_HELL>
static void launchDeep(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int PH = kWave / (32 / RPL);
    constexpr int UNROLL = 32 / PH; /* 32 columns per stage, two stages per item */
    hipLaunchKernelGGL((deepItemsKernel<T, RPL, IS_HELL, UNROLL, kDeepChunk>), dim3(SPGPU_DEEP_ITEMS / (kBlockThreads / kWave)), dim3(kBlockThreads), 0, stream, a);
    hipLaunchKernelGGL((deepFinishKernel<T>), dim3(256), dim3(kBlockThreads), 0, stream, a);
}

/* Short rows (see launchSlabFamily): a lane walks whole rows, 4 columns per stage, no prefetch; 8-byte element types. */
template <typename T, int RPL, bool IS_HELL>
static void launchLean(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int GROUP_ROWS = kWave * RPL;
    constexpr int WAVES = kBlockThreads / kWave;
    const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const unsigned blocks = (unsigned)((groups + WAVES - 1) / WAVES);
    hipLaunchKernelGGL((slabSpmvKernel<T, RPL, 1, IS_HELL, true, 4, 0, true, 0, false, kBlockThreads, 0, false, 1, 8>), dim3(blocks),
                       dim3(kBlockThreads), 0, stream, a);
}

/* The probe of the type's default kernel shape (launchSlabFamily): D/C walk whole rows, 8 columns per stage; S 8 phases x 2. */
template <typename T, bool IS_HELL>
static void launchFormProbe(hipStream_t stream, const SlabArgs<T>& a, bool wideOk)
{
    constexpr int WIDE = 16 / (int)sizeof(T);
    if constexpr (WIDE > 1) {
        if (wideOk) {
            if constexpr (sizeof(T) == 4)
                hipLaunchKernelGGL((formProbeKernel<T, WIDE, 2 * WIDE, IS_HELL, 2 * WIDE * 2>), dim3(3), dim3(kWave), 0, stream, a);
            else
                hipLaunchKernelGGL((formProbeKernel<T, WIDE, 1, IS_HELL, 8>), dim3(3), dim3(kWave), 0, stream, a);
            return;
        }
    }
    hipLaunchKernelGGL((formProbeKernel<T, 1, 2, IS_HELL, 8>), dim3(3), dim3(kWave), 0, stream, a);
}

/* ---- frozen matrices WITHOUT a row order (spgpu?SpmvFreeze with rIdx == NULL, include/spgpu/tuning.h) ------------------------
 * The ordered matrices' frozen form lives with their plan (planned_spmv.hip).  A matrix that runs in the default kernels --
 * BASELINE configs[1], the headline -- gets the same: a 16-bit copy of its column indices, counted per GROUP of rows (the
 * rows one wavefront of slabSpmvKernel owns: 128 for the 8-byte types, 32 for fp32) from the group's lowest column, 0xFFFF where
 * a column lies 65 535 or more above it (or is negative).  The record sits in the handle's plan table under the arrays'
 * addresses with subs = -groupRows (no analysis, no blocks: `device` holds the groups' bases).  A matrix with more than one
 * escape in a hundred entries is not frozen: its columns are scattered, the gathers bound its SpMV, and every escape costs the
 * rP word the copy was to save. */
template <bool IS_HELL>
__global__ __launch_bounds__(256) void slabPackKernel(const int* __restrict__ rP, const int* __restrict__ rS, const int* __restrict__ hackOffsets,
                                                     int hackSize, long long idxStride, int maxNnz, int rows, int baseIndex, int groupRows,
                                                     int* __restrict__ packBases, unsigned short* __restrict__ packed, unsigned long long* counts)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long group = (long long)blockIdx.x * (256 / kWave) + (threadIdx.x >> 6); /* a wavefront per group */
    const long long groupRow0 = group * groupRows;
    if (groupRow0 >= rows)
        return;
    constexpr int MOST = 2; /* rows per lane: groups of up to 128 rows */
    long long at[MOST];
    int len[MOST];
    int lowest = 0x7fffffff;
    for (int j = 0; j < MOST; ++j) {
        const long long r = groupRow0 + lane + j * kWave;
        len[j] = (lane + j * kWave < groupRows && r < rows) ? (rS ? rS[r] : maxNnz) : 0;
        at[j] = 0;
        if (len[j] > 0) {
            if constexpr (IS_HELL) {
                const unsigned u = (unsigned)r, hs = (unsigned)hackSize;
                at[j] = (long long)((unsigned)hackOffsets[u / hs] + u % hs);
            } else {
                at[j] = r;
            }
        }
        for (int k = 0; k < len[j]; ++k) {
            const int col = rP[at[j] + (long long)k * idxStride] - baseIndex;
            lowest = (col >= 0 && col < lowest) ? col : lowest;
        }
    }
    lowest = waveMin(lowest);
    const int base = lowest == 0x7fffffff ? 0 : lowest;
    if (lane == 0)
        packBases[group] = base;
    unsigned entries = 0, escapes = 0;
    for (int j = 0; j < MOST; ++j) {
        for (int k = 0; k < len[j]; ++k) {
            const long long slot = at[j] + (long long)k * idxStride;
            const int col = rP[slot] - baseIndex;
            const long long off = (long long)col - base;
            const bool fits = col >= 0 && off < 0xFFFF;
            packed[slot] = fits ? (unsigned short)off : (unsigned short)0xFFFF;
            entries += 1;
            escapes += fits ? 0 : 1;
        }
    }
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        entries += (unsigned)laneXor((int)entries, m);
        escapes += (unsigned)laneXor((int)escapes, m);
    }
    if (lane == 0) {
        atomicAdd(&counts[0], (unsigned long long)entries);
        atomicAdd(&counts[1], (unsigned long long)escapes);
    }
}

__global__ __launch_bounds__(kWave) void slabSlotsKernel(const int* __restrict__ rS, const int* __restrict__ hackOffsets, int hackSize, int rows, unsigned long long* out)
{
    /* HELL does not state its slot count (hell.c:64,75: no trailing total): last hack's offset + hackSize x its longest row */
    const int lastHack = (rows - 1) / hackSize;
    int longest = 0;
    for (long long r = (long long)lastHack * hackSize + threadIdx.x; r < rows; r += kWave)
        longest = rS[r] > longest ? rS[r] : longest;
    longest = waveMax(longest);
    if (threadIdx.x == 0)
        out[0] = (unsigned long long)(unsigned)hackOffsets[lastHack] + (unsigned long long)hackSize * (unsigned)longest;
}

template <typename T> static SpgpuSpmvPlan slabPlanKey(const SlabArgs<T>& a, int groupRows)
{
    SpgpuSpmvPlan key{};
    key.rP = a.rP;
    key.rS = a.rS;
    key.rIdx = nullptr;
    key.hackOffsets = a.hackOffsets;
    key.idxStride = a.idxStride;
    key.rows = a.rows;
    key.hackSize = a.hackSize;
    key.baseIndex = a.baseIndex;
    key.maxNnz = a.maxNnz;
    key.deepCap = 0;
    key.subs = -groupRows;
    return key;
}

/* spgpu?SpmvFreeze of a matrix without a row order: synchronous; true = frozen (or was already). */
template <typename T, bool IS_HELL>
static bool freezeSlab(spgpuHandle_t handle, hipStream_t stream, const SlabArgs<T>& a, int groupRows)
{
    SpgpuPrivateHandle* h = spgpuPrivate(handle);
    const SpgpuSpmvPlan key = slabPlanKey(a, groupRows);
    spgpuPlanLock(handle);
    SpgpuSpmvPlan* plan = spgpuPlanRecord(handle, &key);
    bool frozen = plan && plan->packed && plan->state == SPGPU_PLAN_READY;
    if (plan && !frozen && plan->state != SPGPU_PLAN_GIVEN_UP) {
        const long long groups = ((long long)a.rows + groupRows - 1) / groupRows;
        const size_t baseBytes = ((size_t)groups * sizeof(int) + 255) / 256 * 256;
        void *device = nullptr, *packed = nullptr;
        int previous = 0;
        (void)hipGetDevice(&previous);
        (void)hipSetDevice(handle->device);
        bool ok = hipMalloc(&device, baseBytes + 256) == hipSuccess;
        unsigned long long* counts = ok ? reinterpret_cast<unsigned long long*>(static_cast<char*>(device) + baseBytes) : nullptr;
        unsigned long long said[2] = {0, 0};
        long long slots = IS_HELL ? 0 : a.idxStride * (long long)a.maxNnz;
        if (ok && IS_HELL) {
            hipLaunchKernelGGL(slabSlotsKernel, dim3(1), dim3(kWave), 0, stream, a.rS, a.hackOffsets, a.hackSize, a.rows, counts);
            ok = hipMemcpyAsync(said, counts, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
            slots = (long long)said[0];
        }
        ok = ok && slots > 0;
        const size_t packedBytes = ok ? ((size_t)slots * sizeof(unsigned short) + 255) / 256 * 256 : 0;
        ok = ok && hipMalloc(&packed, packedBytes) == hipSuccess;
        (void)hipSetDevice(previous);
        if (ok) {
            if (spgpuTuning()->poisonScratch)
                (void)hipMemsetAsync(packed, 0xA5, packedBytes, stream);
            (void)hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), stream);
            hipLaunchKernelGGL((slabPackKernel<IS_HELL>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, stream, a.rP, a.rS, a.hackOffsets, a.hackSize,
                               a.idxStride, a.maxNnz, a.rows, a.baseIndex, groupRows, static_cast<int*>(device), static_cast<unsigned short*>(packed), counts);
            ok = hipMemcpyAsync(said, counts, sizeof(said), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        }
        const int mostPct = spgpuTuning()->freezeEscapesPct < 0 ? 0 : spgpuTuning()->freezeEscapesPct;
        if (ok && said[1] * 100ull <= said[0] * (unsigned long long)mostPct) { /* at most one escape in a hundred entries (SPGPU_FREEZE_MAX_ESCAPES_PCT) */
            plan->device = device;
            plan->packed = packed;
            plan->packedBytes = (long long)packedBytes;
            plan->blocks = 0;
            plan->deep = 0;
            plan->uses = 0;
            plan->state = SPGPU_PLAN_READY;
            h->planFreezes += 1;
            frozen = true;
        } else {
            (void)hipGetLastError();
            if (device)
                (void)hipFree(device);
            if (packed)
                (void)hipFree(packed);
        }
    }
    h->planFrozenSlabs = 0;
    if (h->plans)
        for (int i = 0; i < SPGPU_PLANS; ++i)
            h->planFrozenSlabs += (h->plans[i].rows > 0 && h->plans[i].subs < 0 && h->plans[i].packed) ? 1 : 0;
    spgpuPlanUnlock(handle);
    return frozen;
}

/* The SpMV side: a.planPacked / a.packBases of the matrix' frozen record, if it has one (else they stay NULL).  Not inside a
 * stream capture: a graph would carry the copy's address beyond a Thaw. */
template <typename T>
static void findFrozenSlab(spgpuHandle_t handle, hipStream_t stream, SlabArgs<T>& a, int groupRows)
{
    SpgpuPrivateHandle* h = spgpuPrivate(handle);
    a.planPacked = nullptr;
    a.packBases = nullptr;
    if (__atomic_load_n(&h->planFrozenSlabs, __ATOMIC_RELAXED) <= 0)
        return;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return;
    }
    const SpgpuSpmvPlan key = slabPlanKey(a, groupRows);
    spgpuPlanLock(handle);
    SpgpuSpmvPlan* plan = spgpuPlanFind(handle, &key);
    if (plan && plan->packed && plan->state == SPGPU_PLAN_READY) {
        a.planPacked = static_cast<const unsigned short*>(plan->packed);
        a.packBases = static_cast<const int*>(plan->device);
        plan->uses += 1;
        h->planUses += 1;
    }
    spgpuPlanUnlock(handle);
}

template <typename T, int RPL, int PH, bool IS_HELL, int UNROLL, bool STRIPS>
static void launchSlabPacked(hipStream_t stream, const SlabArgs<T>& a)
{
    constexpr int GROUP_ROWS = (kWave / PH) * RPL;
    constexpr int WAVES = kBlockThreads / kWave;
    const long long groups = ((long long)a.rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const unsigned blocks = (unsigned)((groups + WAVES - 1) / WAVES);
    hipLaunchKernelGGL((slabSpmvKernel<T, RPL, PH, IS_HELL, true, UNROLL, 2, true, 0, STRIPS, kBlockThreads, 0, false, 1, 0, true>),
                       dim3(blocks), dim3(kBlockThreads), 0, stream, a);
}

template <typename T, bool IS_HELL>
static void launchSlabFamily(spgpuHandle_t handle, const SlabArgs<T>& in, int* prepared = nullptr)
{
    /* prepared != NULL (spgpu?SpmvPrepare, include/spgpu/tuning.h): nothing is multiplied -- the choices a first SpMV would leave
     * to later calls are made now and waited for: the ordered matrix' workgroup shape (the probe) and its plan.
     * *prepared: 1 = the next SpMV on these arrays runs from a plan; 0 = this kind of call has none. */
    const bool freeze = prepared && *prepared == 2; /* spgpu?SpmvFreeze: the plan also gets its 16-bit copy of the indices */
    if (prepared)
        *prepared = 0;
    if (in.rows <= 0)
        return;
    SlabArgs<T> a = in;
    hipStream_t stream = handle->currentStream;

    constexpr int WIDE = 16 / (int)sizeof(T);
    /* A lane reads WIDE consecutive rows of a slab column with one 16-byte
     * load: the strip must not straddle a hack (HELL) or run past the pitch
     * (ELL), and the streams must be 16-byte aligned. */
    const long long stripRows = ((long long)a.rows + WIDE - 1) / WIDE * WIDE;
    const bool layoutOk = IS_HELL ? (a.hackSize > 0 && a.hackSize % WIDE == 0)
                                  : (a.valStride >= stripRows && a.idxStride >= stripRows);
    const bool wideOk = layoutOk && alignedTo(a.cM, 16) && alignedTo(a.rP, 4 * WIDE) &&
                        a.valStride % WIDE == 0 && a.idxStride % WIDE == 0;

    /* Kernel shape.  Measured on MI355X, 10 M rows x 32 nnz (tools/sweep_hell.py, profiles/): D/C stream
     * fastest with a lane walking whole rows, 8 slab columns per stage, the next stage prefetched AFTER
     * the current gathers are issued, and whole-wave tail rows (banded 5.9 TB/s, windowed columns +13 %
     * over prefetch-before); S with 8 phases x 2 columns, same prefetch and tail (5.4-6.0 TB/s); 16-byte
     * elements (Z) and unaligned streams take RPL = 1 with 2 phases x 4 columns (5.9 TB/s).
     * SPGPU_SPMV_VARIANT (experiments; 0 = this table):
     *   1 wide PHx2 | 2 wide 1x4 | 3 narrow 2x4 | 4 narrow 1x4 | 6 wide PHx2 pipe | 12 wide 1x8 pipe |
     *   13 narrow 2x4 pipe | 17 wide 1x8 pipe + whole-wave tail rows | 18 wide PHx2 pipe + tail rows |
     *   21 = 17 and 22 = 18 with the prefetch issued after the gathers (defaults for D/C and S)
     *   (5,7..11,14..16 exist only in -DSPGPU_TUNING_VARIANTS builds)
     * SPGPU_NT_LOADS 0/1: non-temporal hint on the coefficient/index streams (default 1). */
    const SpgpuTuning* tune = spgpuTuning();
    a.tailLanes = tune->tailLanes >= 0 ? tune->tailLanes : kTailLanes;
    int variant = tune->spmvVariant;
    const bool nt = tune->ntLoads != 0;
#ifndef SPGPU_TUNING_VARIANTS
    variant = 0; /* the product build carries the default shapes only (13, 21, 22); the others: -DSPGPU_TUNING_VARIANTS */
#endif
    if (variant < 1 || variant > 24)
        variant = !wideOk ? 13 : (sizeof(T) == 4 ? 22 : 21);
    const bool narrowVariant = variant == 3 || variant == 4 || (variant >= 13 && variant <= 16);
    if (!wideOk && !narrowVariant)
        variant = 13;

    /* Strip x loads (consume<STRIPS>): which form a matrix runs in is learnt from the kernel itself.  The
     * strip-capable kernel's sample wavefronts write "ran as strips / as gathers" into pinned host memory; a
     * later call on the same matrix (same rP, same rows) reads that -- no synchronisation, whatever is there --
     * and takes the gather-only kernel when at least two of the three samples said gathers.  Both kernels are
     * correct for every matrix; a stale or missing answer only costs speed.  SPGPU_X_STRIPS = 0 / 1 fixes the form. */
    /* How x is fetched (include/spgpu/tuning.h): the handle's hint, overridden by the environment knobs. */
    int form = spgpuGetSpmvForm(handle);
    if (tune->xStrips >= 0)
        form = tune->xStrips ? SPGPU_SPMV_FORM_STRIPS : SPGPU_SPMV_FORM_GATHER;
    if (tune->xTile >= 0)
        form = tune->xTile ? SPGPU_SPMV_FORM_XTILE : (form == SPGPU_SPMV_FORM_XTILE ? SPGPU_SPMV_FORM_AUTO : form);
    if (form == SPGPU_SPMV_FORM_SWEEP) {
        /* the caller's choice for scattered columns that ascend inside a row; needs 16-byte slab accesses and no row order */
        if (wideOk && !a.rIdx && tune->spmvVariant < 1) {
            if (prepared)
                return;
            a.wideIO = alignedTo(a.z, 16) && alignedTo(a.y, 16);
            a.feedback = nullptr;
            spgpuNoteSpmvForm(handle, SPGPU_SPMV_FORM_SWEEP);
            launchSweep<T, WIDE, IS_HELL>(stream, a);
            return;
        }
        form = SPGPU_SPMV_FORM_AUTO;
    }
#ifdef SPGPU_TUNING_VARIANTS
    if (tune->slide && form != SPGPU_SPMV_FORM_SWEEP) {
        /* experiment (SPGPU_SLIDE=1, lab build only; slide_spmv.hip.h, profiles/r04_exp_slide_tile.txt): the x tile moves along with
         * the slab columns; 8-byte elements, 16-byte slab accesses, no row order.  One phase, no whole-wave tail rows: the order
         * of additions of the SWEEP form. */
        if constexpr (sizeof(T) == 8) {
            if (wideOk && !a.rIdx && tune->spmvVariant < 1 && !prepared) {
                a.wideIO = alignedTo(a.z, 16) && alignedTo(a.y, 16);
                a.feedback = nullptr;
                spgpuNoteSpmvForm(handle, SPGPU_SPMV_FORM_XTILE);
                launchSlide<T, WIDE, IS_HELL>(stream, a);
                return;
            }
        }
    }
#endif
    const bool tiled = form == SPGPU_SPMV_FORM_XTILE && (variant == 13 || variant == 21 || variant == 22);
    /* Deep split (see slabSpmvKernel, DEEP): on when the caller passes a row order -- rows ordered by length are what
     * one does to a ragged matrix, and then whole hacks are deep -- or when SPGPU_DEEP_SPLIT says so. */
    bool deepSplit = (tune->deepSplit >= 0 ? tune->deepSplit != 0 : a.rIdx != nullptr) && wideOk &&
                     (variant == 21 || variant == 22);
    a.deepCap = tune->deepCap > 0 ? tune->deepCap : 256;
    a.deepKeep = tune->deepKeep >= 0 && tune->deepKeep < a.deepCap ? tune->deepKeep : a.deepCap;
    a.xcdRun = tune->xcdOrder;
    a.stageLate = tune->stageLate != 0;
    a.deepChunk = kDeepChunk;
    a.deepHeader = nullptr;
    a.deepEntries = nullptr;
    a.deepItems = nullptr;
    a.deepPartials = nullptr;
    a.deepItemSums = nullptr;
    a.deepOverflow = spgpuDeepOverflowWords(handle);
    bool noDeepList = false;
    SpgpuDeepList list;
    list.idle = nullptr;
    if (deepSplit) {
        if (spgpuDeepScratch(handle, &list) == SPGPU_SUCCESS) {
            a.deepHeader = list.header;
            a.deepEntries = list.entries;
            a.deepItems = list.items;
            a.deepPartials = static_cast<T*>(list.partials);
            a.deepItemSums = static_cast<T*>(list.itemSums);
        } else {
            deepSplit = false;
            noDeepList = true;
        }
    }
#ifdef SPGPU_TUNING_VARIANTS /* two stateless one-launch kernels of round 3, kept for A/B runs (another order of additions: chunks of 48 columns) */
    if (prepared && (tune->ragged == 2 || tune->ragged == 3))
        return;
    if (a.rIdx != nullptr && wideOk && (variant == 21 || variant == 22) && tune->ragged == 3) {
        /* rows ordered by length: one resident workgroup per CU, the next block prepared beside the stream (pipe_spmv.hip.h) */
        a.wideIO = 0;
        a.feedback = nullptr;
        spgpuNoteSpmvForm(handle, form != SPGPU_SPMV_FORM_GATHER ? SPGPU_SPMV_FORM_XTILE : SPGPU_SPMV_FORM_GATHER);
        launchPipe<T, WIDE, IS_HELL>(stream, a, tune->pipeGroups > 0 ? tune->pipeGroups : handle->multiProcessorCount, form != SPGPU_SPMV_FORM_GATHER, tune->raggedShape);
        return;
    }
    if (a.rIdx != nullptr && wideOk && (variant == 21 || variant == 22) && tune->ragged == 2) {
        /* rows ordered by length: shares of equal work, one launch, no state (share_spmv.hip.h) */
        a.wideIO = 0;
        a.feedback = nullptr;
        spgpuNoteSpmvForm(handle, form != SPGPU_SPMV_FORM_GATHER ? SPGPU_SPMV_FORM_XTILE : SPGPU_SPMV_FORM_GATHER);
        launchShare<T, WIDE, IS_HELL>(stream, a, tune->raggedShape, form != SPGPU_SPMV_FORM_GATHER);
        return;
    }
#endif
    /* ELL says how long its longest row is: when none can exceed the cap nothing registers and the two launches behind
     * the main kernel (~5 us each when empty) are left out; HELL does not say */
    const bool deepPossible = IS_HELL || a.maxNnz > a.deepCap;
#ifndef SPGPU_TUNING_VARIANTS
    constexpr bool queueKernelOnly = true; /* the deep split with fixed rows per wavefront (SPGPU_RAGGED=0) is a lab shape */
#else
    constexpr bool queueKernelOnly = false;
#endif
    if ((deepSplit || noDeepList) && (tune->ragged != 0 || queueKernelOnly)) {
        /* the queue-driven kernel for rows ordered by length (ragged_spmv.hip.h); x through an LDS tile unless the
         * caller asked for plain gathers */
        a.wideIO = 0;
        a.feedback = nullptr;
        spgpuNoteSpmvForm(handle, form != SPGPU_SPMV_FORM_GATHER ? SPGPU_SPMV_FORM_XTILE : SPGPU_SPMV_FORM_GATHER);
        int shape = tune->raggedShape;
        if (shape == 0 && form != SPGPU_SPMV_FORM_GATHER && a.rIdx != nullptr && sizeof(T) <= 8) {
            /* which of the two product shapes?  (orderedProbeKernel; the answer is kept with AUTO's per-matrix words and
             * read without synchronisation: a first call runs the 1 024-row shape, which is never far off) */
            int calls = 0, tag = 0;
            int* seen = spgpuFormFeedback(handle, a.rP, a.rows, &calls, &tag);
            int said = spgpuFeedbackSaid(((volatile int*)seen)[3], tag);
            if (said == 0 || calls % 64 == 0)
                hipLaunchKernelGGL((orderedProbeKernel<IS_HELL>), dim3(1), dim3(kWave), 0, stream, a.rP, a.rS, a.hackOffsets, a.rIdx, a.hackSize,
                                   a.idxStride, a.maxNnz, a.rows, a.baseIndex, seen + 3, tag);
            if (prepared && said == 0 && hipStreamSynchronize(stream) == hipSuccess)
                said = spgpuFeedbackSaid(((volatile int*)seen)[3], tag); /* the answer a second call would have found */
            shape = said == 4 || said == 6 ? 4 : 0; /* 6: the blocks are the windows of an aligned order */
        }
        /* a matrix seen before has a plan (planned_spmv.hip): one launch, the deep sub-groups in workgroups of their own, no
         * list.  Same bits either way. */
        /* noDeepList: this stream of the handle has no deep list (every list belongs to a stream with work in flight, or the
         * allocation failed): the same kernel family without any state -- the matrix' plan if it is ready, else no plan at
         * all (every deep sub-group worked off by its own block).  Same bits in every case. */
        const bool tiledForm = form != SPGPU_SPMV_FORM_GATHER;
        if (noDeepList && !(shape == 4 || shape == 5))
            shape = 0;
        if (prepared) {
            if (!tiledForm || shape == 0 || shape == 4 || shape == 5)
                *prepared = launchPlanned<T, IS_HELL>(handle, stream, a, shape, tiledForm, false, freeze ? 2 : 1) ? 1 : 0;
            return;
        }
        if ((!tiledForm || shape == 0 || shape == 4 || shape == 5) && launchPlanned<T, IS_HELL>(handle, stream, a, shape, tiledForm, noDeepList, 0))
            return;
        const bool deepKernels = launchRagged<T, WIDE, IS_HELL, true>(stream, a, shape, form != SPGPU_SPMV_FORM_GATHER);
        if (deepPossible && deepKernels)
            launchDeep<T, WIDE, IS_HELL>(stream, a);
        if (list.idle) {
            /* complete = the list has no user (core.c: a list may change hands) -- unless this launch is being captured: a graph
             * carries the list's addresses and may be replayed at any time, so the list stays with this stream for good */
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing == hipStreamCaptureStatusNone)
                (void)hipEventRecord(list.idle, stream);
            else
                spgpuDeepListPin(handle);
        }
        return;
    }
    if (prepared) {
        /* (the forms below learn what they need from their own launches) -- Freeze of a matrix without a row order: the default
         * kernels' 16-bit index copy */
        if constexpr (WIDE > 1) {
            if (freeze && !a.rIdx && wideOk && !narrowVariant && !tiled && nt && (variant == 21 || variant == 22) && !deepSplit)
                *prepared = freezeSlab<T, IS_HELL>(handle, stream, a, variant == 22 ? (kWave / (2 * WIDE)) * WIDE : kWave * WIDE) ? 1 : 0;
        }
        return;
    }
#ifdef SPGPU_TUNING_VARIANTS
    if (deepSplit) {
        /* shapes in which a lane walks whole rows, for every type; the strip form does not apply to ordered rows */
        a.wideIO = alignedTo(a.z, 16) && alignedTo(a.y, 16);
        a.feedback = nullptr;
        spgpuNoteSpmvForm(handle, tiled ? SPGPU_SPMV_FORM_XTILE : SPGPU_SPMV_FORM_GATHER);
        if (tiled)
            launchTiled<T, WIDE, IS_HELL, true>(stream, a, tune->xTileShape);
        else
            launchShape<T, WIDE, 1, IS_HELL, (sizeof(T) == 8 ? 8 : 4), true, kBlockThreads, 0, true>(stream, a);
        if (deepPossible)
            launchDeep<T, WIDE, IS_HELL>(stream, a);
        return;
    }
#endif
    bool strips = false, autoTile = false, autoSweep = false, probeBehind = false;
    a.feedback = nullptr;
    a.tileSpanLimit = (long long)(32768 / sizeof(T)) * 5 / 4; /* 1.25 x the default tile (launchTiled, shape 0) */
    if (!narrowVariant && WIDE > 1 && !tiled) {
        if (form != SPGPU_SPMV_FORM_AUTO) {
            strips = form == SPGPU_SPMV_FORM_STRIPS;
        } else {
            int calls = 0, tag = 0;
            int* seen = spgpuFormFeedback(handle, a.rP, a.rows, &calls, &tag);
            a.feedbackTag = tag;
            int gathers = 0, local = 0, sweeps = 0;
            for (int q = 0; q < 3; ++q) {
                const int said = spgpuFeedbackSaid(((volatile int*)seen)[q], tag);
                gathers += said == 1 ? 1 : 0;
                local += said == 3 ? 1 : 0;
                sweeps += said == 4 ? 1 : 0;
            }
            /* two of three samples decide: scattered -> gathers; inside a window -> the LDS tile; otherwise (strips, or
             * nothing known yet) the strip-capable kernel */
            autoTile = local >= 2 && tune->spmvVariant < 1 && (variant == 21 || variant == 22);
            strips = gathers + local + sweeps < 2;
            /* scattered over all of x, ascending inside the rows, rows about equally long (only the probe says so: answer 4):
             * the SWEEP form -- same bits as the default kernel of the 8-byte types; it needs rows for a resident grid */
            autoSweep = sweeps >= 2 && !autoTile && sizeof(T) == 8 && variant == 21 && tune->spmvVariant < 1 && tune->autoSweep != 0 &&
                        !a.rIdx && a.rows >= kAutoSweepRows;
            a.feedback = seen; /* the strip-capable kernel's sample wavefronts report (it is what a new matrix runs first) */
            /* the other forms do not (see slabSpmvKernel): with every fourth call of theirs three wavefronts look at the
             * matrix again -- another one may live at this address by now -- and with the first of them (the samples know
             * strips, a window and "neither"; whether "neither" is a matrix for the SWEEP form only the probe finds out) */
            probeBehind = !strips && (calls % 4 == 0 || calls == 1);
        }
    }

    spgpuNoteSpmvForm(handle, (tiled || autoTile) ? SPGPU_SPMV_FORM_XTILE
                                                  : (autoSweep ? SPGPU_SPMV_FORM_SWEEP : (strips ? SPGPU_SPMV_FORM_STRIPS : SPGPU_SPMV_FORM_GATHER)));
    if (probeBehind)
        launchFormProbe<T, IS_HELL>(stream, a, wideOk); /* 3 wavefronts; its answer is for later calls */
    if (autoSweep) {
        if constexpr (WIDE > 1) {
            a.wideIO = alignedTo(a.z, 16) && alignedTo(a.y, 16);
            a.feedback = nullptr;
            launchSweep<T, WIDE, IS_HELL>(stream, a);
            return;
        }
    }
    if (!strips)
        a.feedback = nullptr;
    if (!narrowVariant) {
        a.wideIO = alignedTo(a.z, 16) && alignedTo(a.y, 16);
        if constexpr (WIDE > 1) {
            switch (variant) {
#ifdef SPGPU_TUNING_VARIANTS
            case 1: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 2>(stream, a, nt); break;
            case 2: launchSlab<T, WIDE, 1, IS_HELL, 4>(stream, a, nt); break;
            case 6: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 2, true>(stream, a, nt); break;
            case 12: launchSlab<T, WIDE, 1, IS_HELL, 8, true>(stream, a, nt); break;
            case 17: launchSlab<T, WIDE, 1, IS_HELL, 8, 1, true>(stream, a, nt); break;
            case 18: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 2, true, true>(stream, a, nt); break;
            case 5: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 4>(stream, a, nt); break;
            case 7: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 4, true>(stream, a, nt); break;
            case 8: launchSlab<T, WIDE, 1, IS_HELL, 4, true>(stream, a, nt); break;
            case 9: launchSlab<T, WIDE, 1, IS_HELL, 8>(stream, a, nt); break;
            case 10: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 1, true>(stream, a, nt); break;
            case 11: launchSlab<T, WIDE, 1, IS_HELL, 2, true>(stream, a, nt); break;
            case 23: launchSlab<T, WIDE, 1, IS_HELL, 4, 2, true>(stream, a, nt); break;
            case 24: launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 4, 2, true>(stream, a, nt); break;
            case 19: launchSlab<T, WIDE, 1, IS_HELL, 8, true, true, 1>(stream, a, nt); break; /* 17 + nt x gathers */
            case 20: launchSlab<T, WIDE, 1, IS_HELL, 8, true, true, 2>(stream, a, nt); break; /* 17 + sc1 x gathers */
#endif
            case 22:
                if (tiled || autoTile)
                    launchTiled<T, WIDE, IS_HELL, false>(stream, a, tune->xTileShape);
                else {
                    if (nt)
                        findFrozenSlab(handle, stream, a, (kWave / (2 * WIDE)) * WIDE);
                    if (a.planPacked && strips)
                        launchSlabPacked<T, WIDE, 2 * WIDE, IS_HELL, 2, true>(stream, a);
                    else if (a.planPacked)
                        launchSlabPacked<T, WIDE, 2 * WIDE, IS_HELL, 2, false>(stream, a);
                    else if (strips)
                        launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 2, 2, true, 0, true>(stream, a, nt);
                    else
                        launchSlab<T, WIDE, 2 * WIDE, IS_HELL, 2, 2, true>(stream, a, nt);
                }
                break;
            default: /* 21 */
                if (tiled || autoTile)
                    launchTiled<T, WIDE, IS_HELL, false>(stream, a, tune->xTileShape);
                else if (a.avgNnzPerRow > 0 && a.avgNnzPerRow <= 8 && form == SPGPU_SPMV_FORM_AUTO && (IS_HELL || a.maxNnz <= 16)) {
                    /* the caller says the rows are short (avgNnzPerRow: the reference's own tuning hint, which picks its
                     * threads-per-row shape, hell_spmv_base_template.cuh:306-325): such a row is one stage, and a kernel
                     * without the prefetch ring needs a third of the registers -- all wavefronts of a 1 M-row system are
                     * resident at once instead of queueing in three rounds (19.4 -> 17.4 us on configs[0]).  Same order of
                     * additions (the tail switch is considered every 8 columns, as in the default kernel:
                     * tests/test_gpu_spmv.py::test_short_row_hint_same_bits pins that on rows of 0 .. 300 entries).  ELL says how long
                     * its longest row is: beyond two stages the prefetching kernel stays, whatever the average; HELL has only the hint. */
                    spgpuNoteSpmvForm(handle, SPGPU_SPMV_FORM_GATHER);
                    a.feedback = nullptr;
                    launchLean<T, WIDE, IS_HELL>(stream, a);
                } else {
                    if (nt)
                        findFrozenSlab(handle, stream, a, kWave * WIDE);
                    if (a.planPacked && strips)
                        launchSlabPacked<T, WIDE, 1, IS_HELL, 8, true>(stream, a);
                    else if (a.planPacked)
                        launchSlabPacked<T, WIDE, 1, IS_HELL, 8, false>(stream, a);
                    else if (strips)
                        launchSlab<T, WIDE, 1, IS_HELL, 8, 2, true, 0, true>(stream, a, nt);
                    else
                        launchSlab<T, WIDE, 1, IS_HELL, 8, 2, true>(stream, a, nt);
                }
                break;
            }
            return;
        }
    }
    a.wideIO = 1; /* RPL == 1: element access is always aligned */
    switch (variant) {
#ifdef SPGPU_TUNING_VARIANTS
    case 3: launchSlab<T, 1, 2, IS_HELL, 4>(stream, a, nt); break;
    case 4: launchSlab<T, 1, 1, IS_HELL, 4>(stream, a, nt); break;
    case 14: launchSlab<T, 1, 1, IS_HELL, 8, true>(stream, a, nt); break;
    case 15: launchSlab<T, 1, 2, IS_HELL, 8>(stream, a, nt); break;
    case 16: launchSlab<T, 1, 4, IS_HELL, 2, true>(stream, a, nt); break;
#endif
    default: /* 13 */
        if (tiled)
            launchTiled<T, 1, IS_HELL, false>(stream, a, tune->xTileShape);
        else
            launchSlab<T, 1, 2, IS_HELL, 4, 2>(stream, a, nt);
        break;
    }
}

template <typename T, typename ApiT>
static void hellSpmv(spgpuHandle_t handle, ApiT* z, const ApiT* y, ApiT alpha, const ApiT* cM, const int* rP,
                     int hackSize, const int* hackOffsets, const int* rS, const int* rIdx, int rows,
                     const ApiT* x, ApiT beta, int baseIndex, int avgNnzPerRow = 0)
{
    static_assert(sizeof(T) == sizeof(ApiT), "ABI type and device type must have one layout");
    /* an ADOPTED matrix (spgpuHellSpmvAdopt, adopted_hell.hip): the call runs on the library's ordered copy and writes z through the
     * copy's row order -- z in the caller's row order, as ever */
    if (!rIdx) {
        const SpgpuAdopted* copy = spgpuAdoptedFind(handle, handle->currentStream, cM, rP, rS, hackOffsets, rows, hackSize, baseIndex, 0, 0);
        if (copy && spgpuSizeOf((spgpuType_t)copy->type) == sizeof(T)) { /* (adopted as another type: the call runs on the caller's arrays) */
            cM = static_cast<const ApiT*>(copy->values);
            rP = copy->indices;
            hackOffsets = copy->hackOffsetsOrdered;
            rS = copy->lengths;
            rIdx = copy->order;
        }
    }
    SlabArgs<T> a;
    a.z = reinterpret_cast<T*>(z);
    a.y = reinterpret_cast<const T*>(y);
    a.x = reinterpret_cast<const T*>(x);
    a.cM = reinterpret_cast<const T*>(cM);
    a.rP = rP;
    a.rS = rS;
    a.rIdx = rIdx;
    a.hackOffsets = hackOffsets;
    __builtin_memcpy(&a.alpha, &alpha, sizeof(T));
    __builtin_memcpy(&a.beta, &beta, sizeof(T));
    a.rows = rows;
    a.baseIndex = baseIndex;
    a.hackSize = hackSize;
    a.maxNnz = 0;
    a.valStride = hackSize;
    a.idxStride = hackSize;
    a.wideIO = 0;
    a.avgNnzPerRow = avgNnzPerRow;
    a.feedbackTag = 0;
    a.planBlocks = nullptr;
    a.planDeepSubs = nullptr;
    a.planFlags = nullptr;
    a.planDeep = a.planMainBlocks = a.planDeepPerBlock = a.planDeepStride = a.planDeepRuns = 0;
    a.planPacked = nullptr;
    a.packBases = nullptr;
    launchSlabFamily<T, true>(handle, a);
    spgpuDebugCheck(handle, "hellspmv");
}

template <typename T, typename ApiT>
static void ellSpmv(spgpuHandle_t handle, ApiT* z, const ApiT* y, ApiT alpha, const ApiT* cM, const int* rP,
                    int cMPitch, int rPPitch, const int* rS, const int* rIdx, int maxNnzPerRow, int rows,
                    const ApiT* x, ApiT beta, int baseIndex, int avgNnzPerRow = 0)
{
    static_assert(sizeof(T) == sizeof(ApiT), "ABI type and device type must have one layout");
    /* an ADOPTED ELL matrix (spgpuEllSpmvAdopt, adopted_hell.hip): the call runs on the library's ordered HELL copy */
    if (!rIdx && rS) {
        const SpgpuAdopted* copy = spgpuAdoptedFind(handle, handle->currentStream, cM, rP, rS, nullptr, rows, 0, baseIndex, cMPitch, rPPitch);
        if (copy && spgpuSizeOf((spgpuType_t)copy->type) == sizeof(T)) {
            hellSpmv<T, ApiT>(handle, z, y, alpha, static_cast<const ApiT*>(copy->values), copy->indices, 32, copy->hackOffsetsOrdered, copy->lengths,
                              copy->order, rows, x, beta, baseIndex, avgNnzPerRow);
            return;
        }
    }
    SlabArgs<T> a;
    a.z = reinterpret_cast<T*>(z);
    a.y = reinterpret_cast<const T*>(y);
    a.x = reinterpret_cast<const T*>(x);
    a.cM = reinterpret_cast<const T*>(cM);
    a.rP = rP;
    a.rS = rS;
    a.rIdx = rIdx;
    a.hackOffsets = nullptr;
    __builtin_memcpy(&a.alpha, &alpha, sizeof(T));
    __builtin_memcpy(&a.beta, &beta, sizeof(T));
    a.rows = rows;
    a.baseIndex = baseIndex;
    a.hackSize = 0;
    a.maxNnz = maxNnzPerRow;
    a.valStride = cMPitch;
    a.idxStride = rPPitch;
    a.wideIO = 0;
    a.avgNnzPerRow = avgNnzPerRow;
    a.feedbackTag = 0;
    a.planBlocks = nullptr;
    a.planDeepSubs = nullptr;
    a.planFlags = nullptr;
    a.planDeep = a.planMainBlocks = a.planDeepPerBlock = a.planDeepStride = a.planDeepRuns = 0;
    a.planPacked = nullptr;
    a.packBases = nullptr;
    launchSlabFamily<T, false>(handle, a);
    spgpuDebugCheck(handle, "ellspmv");
}

/* spgpu?SpmvPrepare (include/spgpu/tuning.h): the dispatch of an SpMV on these arrays, with nothing multiplied */
template <typename T, bool IS_HELL>
static int prepareSpmv(spgpuHandle_t handle, const void* cM, const int* rP, int hackSize, const int* hackOffsets, long long valStride, long long idxStride,
                       const int* rS, const int* rIdx, int maxNnz, int rows, int baseIndex, bool freeze)
{
    SlabArgs<T> a{};
    a.cM = static_cast<const T*>(cM);
    a.rP = rP;
    a.rS = rS;
    a.rIdx = rIdx;
    a.hackOffsets = hackOffsets;
    a.rows = rows;
    a.baseIndex = baseIndex;
    a.hackSize = hackSize;
    a.maxNnz = maxNnz;
    a.valStride = valStride;
    a.idxStride = idxStride;
    int prepared = freeze ? 2 : 0; /* in: what is asked for; out: 1 = done */
    launchSlabFamily<T, IS_HELL>(handle, a, &prepared);
    return prepared ? SPGPU_SUCCESS : SPGPU_UNSUPPORTED;
}

template <bool IS_HELL>
static int prepareSpmvOfType(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, long long valStride,
                             long long idxStride, const int* rS, const int* rIdx, int maxNnz, int rows, int baseIndex, bool freeze = false)
{
    if (!handle || !rP || !cM || rows < 0)
        return SPGPU_UNSPECIFIED;
    switch (type) {
    case SPGPU_TYPE_FLOAT: return prepareSpmv<float, IS_HELL>(handle, cM, rP, hackSize, hackOffsets, valStride, idxStride, rS, rIdx, maxNnz, rows, baseIndex, freeze);
    case SPGPU_TYPE_DOUBLE: return prepareSpmv<double, IS_HELL>(handle, cM, rP, hackSize, hackOffsets, valStride, idxStride, rS, rIdx, maxNnz, rows, baseIndex, freeze);
    case SPGPU_TYPE_COMPLEX_FLOAT: return prepareSpmv<cfloat, IS_HELL>(handle, cM, rP, hackSize, hackOffsets, valStride, idxStride, rS, rIdx, maxNnz, rows, baseIndex, freeze);
    case SPGPU_TYPE_COMPLEX_DOUBLE: return prepareSpmv<cdouble, IS_HELL>(handle, cM, rP, hackSize, hackOffsets, valStride, idxStride, rS, rIdx, maxNnz, rows, baseIndex, freeze);
    default: return SPGPU_UNSPECIFIED;
    }
}

/* ---- ELL coefficient update (include/spgpu/ell.h; reference ell_csput_base.cuh:33-75) ---- */
template <typename T>
__global__ __launch_bounds__(kBlockThreads) void ellCsputKernel(T* cM, const int* rP, long long cMPitch, long long rPPitch,
                                                               const int* rS, int nnz, const int* aI, const int* aJ,
                                                               const T* aVal, int baseIndex)
{
    const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= nnz)
        return;
    const int row = aI[i] - baseIndex;
    if (row < 0)
        return;
    const int column = aJ[i];
    int lower = 0, upper = rS[row] - 1;
    while (lower <= upper) { /* the row's stored indices ascend */
        const int mid = (lower + upper) / 2;
        const int stored = rP[row + mid * rPPitch];
        if (stored == column) {
            cM[row + mid * cMPitch] = aVal[i];
            return;
        }
        if (stored < column)
            lower = mid + 1;
        else
            upper = mid - 1;
    }
}

template <typename T, typename ApiT>
static void ellCsput(spgpuHandle_t handle, ApiT* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, int nnz,
                     const int* aI, const int* aJ, const ApiT* aVal, int baseIndex)
{
    if (nnz <= 0)
        return;
    const unsigned blocks = (unsigned)(((long long)nnz + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL((ellCsputKernel<T>), dim3(blocks), dim3(kBlockThreads), 0, handle->currentStream,
                       reinterpret_cast<T*>(cM), rP, (long long)cMPitch, (long long)rPPitch, rS, nnz, aI, aJ,
                       reinterpret_cast<const T*>(aVal), baseIndex);
    spgpuDebugCheck(handle, "ellcsput");
}


/* ---- spgpuHellSpmvForm / spgpuEllSpmvForm (include/spgpu/tuning.h): the probe, synchronously ---- */
template <typename T, bool IS_HELL>
static int analyseForm(spgpuHandle_t handle, const int* rP, int hackSize, const int* hackOffsets, long long idxStride, const int* rS,
                       int maxNnz, int rows, int baseIndex)
{
    if (rows <= 0)
        return SPGPU_SPMV_FORM_GATHER;
    constexpr int WIDE = 16 / (int)sizeof(T);
    SlabArgs<T> a{};
    a.rP = rP;
    a.rS = rS;
    a.hackOffsets = hackOffsets;
    a.rows = rows;
    a.baseIndex = baseIndex;
    a.hackSize = hackSize;
    a.maxNnz = maxNnz;
    a.idxStride = idxStride;
    a.valStride = idxStride;
    a.tileSpanLimit = (long long)(32768 / sizeof(T)) * 5 / 4;
    int* seen = spgpuAnalyseWords(handle);
    seen[0] = seen[1] = seen[2] = 0;
    a.feedback = seen;
    a.feedbackTag = 0;
    const bool wide = IS_HELL ? (hackSize > 0 && hackSize % WIDE == 0) : true;
    launchFormProbe<T, IS_HELL>(handle->currentStream, a, wide);
    if (hipStreamSynchronize(handle->currentStream) != hipSuccess)
        return SPGPU_SPMV_FORM_AUTO;
    int strips = 0, local = 0, sweeps = 0;
    for (int q = 0; q < 3; ++q) {
        strips += seen[q] == 2;
        local += seen[q] == 3;
        sweeps += seen[q] == 4;
    }
    if (sweeps >= 2 && sizeof(T) == 8 && rows >= kAutoSweepRows) /* where AUTO itself would take it */
        return SPGPU_SPMV_FORM_SWEEP;
    return strips >= 2 ? SPGPU_SPMV_FORM_STRIPS : (local >= 2 ? SPGPU_SPMV_FORM_XTILE : SPGPU_SPMV_FORM_GATHER);
}

template <bool IS_HELL>
static int analyseFormOfType(spgpuHandle_t handle, spgpuType_t type, const int* rP, int hackSize, const int* hackOffsets, long long idxStride,
                             const int* rS, int maxNnz, int rows, int baseIndex)
{
    switch (type) {
    case SPGPU_TYPE_FLOAT: return analyseForm<float, IS_HELL>(handle, rP, hackSize, hackOffsets, idxStride, rS, maxNnz, rows, baseIndex);
    case SPGPU_TYPE_DOUBLE: return analyseForm<double, IS_HELL>(handle, rP, hackSize, hackOffsets, idxStride, rS, maxNnz, rows, baseIndex);
    case SPGPU_TYPE_COMPLEX_FLOAT: return analyseForm<cfloat, IS_HELL>(handle, rP, hackSize, hackOffsets, idxStride, rS, maxNnz, rows, baseIndex);
    case SPGPU_TYPE_COMPLEX_DOUBLE: return analyseForm<cdouble, IS_HELL>(handle, rP, hackSize, hackOffsets, idxStride, rS, maxNnz, rows, baseIndex);
    default: return SPGPU_SPMV_FORM_AUTO;
    }
}

} // namespace spgpu

using namespace spgpu;

extern "C" {

int spgpuTuningVariantsBuilt(void)
{
#ifdef SPGPU_TUNING_VARIANTS
    return 1;
#else
    return 0;
#endif
}

int spgpuHellSpmvForm(spgpuHandle_t handle, spgpuType_t type, const int* rP, int hackSize, const int* hackOffsets, const int* rS, int rows,
                      int baseIndex)
{
    return analyseFormOfType<true>(handle, type, rP, hackSize, hackOffsets, hackSize, rS, 0, rows, baseIndex);
}

int spgpuHellSpmvPrepare(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                         const int* rIdx, int rows, int baseIndex)
{
    if (!hackOffsets || !rS || hackSize <= 0)
        return SPGPU_UNSPECIFIED;
    return spgpu::prepareSpmvOfType<true>(handle, type, cM, rP, hackSize, hackOffsets, hackSize, hackSize, rS, rIdx, 0, rows, baseIndex);
}

int spgpuEllSpmvPrepare(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, const int* rIdx,
                        int maxNnzPerRow, int rows, int baseIndex)
{
    return spgpu::prepareSpmvOfType<false>(handle, type, cM, rP, 0, nullptr, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, baseIndex);
}

int spgpuHellSpmvFreeze(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                        const int* rIdx, int rows, int baseIndex)
{
    if (!hackOffsets || !rS || hackSize <= 0)
        return SPGPU_UNSPECIFIED;
    return spgpu::prepareSpmvOfType<true>(handle, type, cM, rP, hackSize, hackOffsets, hackSize, hackSize, rS, rIdx, 0, rows, baseIndex, true);
}

int spgpuEllSpmvFreeze(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, const int* rIdx,
                       int maxNnzPerRow, int rows, int baseIndex)
{
    return spgpu::prepareSpmvOfType<false>(handle, type, cM, rP, 0, nullptr, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, baseIndex, true);
}

int spgpuEllSpmvForm(spgpuHandle_t handle, spgpuType_t type, const int* rP, int rPPitch, const int* rS, int maxNnzPerRow, int rows,
                     int baseIndex)
{
    return analyseFormOfType<false>(handle, type, rP, 0, nullptr, rPPitch, rS, maxNnzPerRow, rows, baseIndex);
}

#ifdef SPGPU_TRACE_BLOCKS
void spgpuPlannedSetTrace(unsigned long long* buffer);
void spgpuDebugSetTrace(unsigned long long* buffer)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(spgpu::spgpuTraceBuffer), &buffer, sizeof(buffer));
    spgpuPlannedSetTrace(buffer);
}
#endif

void spgpuDebugCheck(spgpuHandle_t h, const char* what)
{
#ifdef SPGPU_DEBUG
    hipError_t err = hipStreamSynchronize(h->currentStream);
    if (err == hipSuccess)
        err = hipGetLastError();
    if (err != hipSuccess) {
        fprintf(stderr, "spgpu: HIP error in %s: %s\n", what, hipGetErrorString(err));
        exit(1);
    }
#else
    (void)h;
    (void)what;
#endif
}

/* avgNnzPerRow is a tuning hint in the reference (threads-per-row choice, hell_spmv_base_template.cuh:306-325); here it
 * selects the kernel without a prefetch ring when it says 1 .. 8 (launchSlabFamily); any value gives the same bits. */

void spgpuShellspmv(spgpuHandle_t handle, float* z, const float* y, float alpha, const float* cM,
                    const int* rP, int hackSize, const int* hackOffsets, const int* rS, const int* rIdx,
                    int avgNnzPerRow, int rows, const float* x, float beta, int baseIndex)
{
    hellSpmv<float>(handle, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuDhellspmv(spgpuHandle_t handle, double* z, const double* y, double alpha, const double* cM,
                    const int* rP, int hackSize, const int* hackOffsets, const int* rS, const int* rIdx,
                    int avgNnzPerRow, int rows, const double* x, double beta, int baseIndex)
{
    hellSpmv<double>(handle, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuChellspmv(spgpuHandle_t handle, hipFloatComplex* z, const hipFloatComplex* y, hipFloatComplex alpha,
                    const hipFloatComplex* cM, const int* rP, int hackSize, const int* hackOffsets,
                    const int* rS, const int* rIdx, int avgNnzPerRow, int rows, const hipFloatComplex* x,
                    hipFloatComplex beta, int baseIndex)
{
    hellSpmv<cfloat>(handle, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuZhellspmv(spgpuHandle_t handle, hipDoubleComplex* z, const hipDoubleComplex* y,
                    hipDoubleComplex alpha, const hipDoubleComplex* cM, const int* rP, int hackSize,
                    const int* hackOffsets, const int* rS, const int* rIdx, int avgNnzPerRow, int rows,
                    const hipDoubleComplex* x, hipDoubleComplex beta, int baseIndex)
{
    hellSpmv<cdouble>(handle, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuSellspmv(spgpuHandle_t handle, float* z, const float* y, float alpha, const float* cM, const int* rP,
                   int cMPitch, int rPPitch, const int* rS, const int* rIdx, int avgNnzPerRow,
                   int maxNnzPerRow, int rows, const float* x, float beta, int baseIndex)
{
    ellSpmv<float>(handle, z, y, alpha, cM, rP, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuDellspmv(spgpuHandle_t handle, double* z, const double* y, double alpha, const double* cM,
                   const int* rP, int cMPitch, int rPPitch, const int* rS, const int* rIdx, int avgNnzPerRow,
                   int maxNnzPerRow, int rows, const double* x, double beta, int baseIndex)
{
    ellSpmv<double>(handle, z, y, alpha, cM, rP, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuCellspmv(spgpuHandle_t handle, hipFloatComplex* z, const hipFloatComplex* y, hipFloatComplex alpha,
                   const hipFloatComplex* cM, const int* rP, int cMPitch, int rPPitch, const int* rS,
                   const int* rIdx, int avgNnzPerRow, int maxNnzPerRow, int rows, const hipFloatComplex* x,
                   hipFloatComplex beta, int baseIndex)
{
    ellSpmv<cfloat>(handle, z, y, alpha, cM, rP, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, x, beta, baseIndex, avgNnzPerRow);
}

void spgpuZellspmv(spgpuHandle_t handle, hipDoubleComplex* z, const hipDoubleComplex* y, hipDoubleComplex alpha,
                   const hipDoubleComplex* cM, const int* rP, int cMPitch, int rPPitch, const int* rS,
                   const int* rIdx, int avgNnzPerRow, int maxNnzPerRow, int rows, const hipDoubleComplex* x,
                   hipDoubleComplex beta, int baseIndex)
{
    ellSpmv<cdouble>(handle, z, y, alpha, cM, rP, cMPitch, rPPitch, rS, rIdx, maxNnzPerRow, rows, x, beta, baseIndex, avgNnzPerRow);
}

/* alpha is accepted and not applied, as in the reference (ell_csput_base.cuh:35,44,66). */
void spgpuSellcsput(spgpuHandle_t handle, float alpha, float* cM, const int* rP, int cMPitch, int rPPitch, const int* rS,
                    int nnz, int* aI, int* aJ, float* aVal, int baseIndex)
{
    (void)alpha;
    ellCsput<float>(handle, cM, rP, cMPitch, rPPitch, rS, nnz, aI, aJ, aVal, baseIndex);
}
void spgpuDellcsput(spgpuHandle_t handle, double alpha, double* cM, const int* rP, int cMPitch, int rPPitch,
                    const int* rS, int nnz, int* aI, int* aJ, double* aVal, int baseIndex)
{
    (void)alpha;
    ellCsput<double>(handle, cM, rP, cMPitch, rPPitch, rS, nnz, aI, aJ, aVal, baseIndex);
}
void spgpuCellcsput(spgpuHandle_t handle, hipFloatComplex alpha, hipFloatComplex* cM, const int* rP, int cMPitch,
                    int rPPitch, const int* rS, int nnz, int* aI, int* aJ, hipFloatComplex* aVal, int baseIndex)
{
    (void)alpha;
    ellCsput<cfloat>(handle, cM, rP, cMPitch, rPPitch, rS, nnz, aI, aJ, aVal, baseIndex);
}
void spgpuZellcsput(spgpuHandle_t handle, hipDoubleComplex alpha, hipDoubleComplex* cM, const int* rP, int cMPitch,
                    int rPPitch, const int* rS, int nnz, int* aI, int* aJ, hipDoubleComplex* aVal, int baseIndex)
{
    (void)alpha;
    ellCsput<cdouble>(handle, cM, rP, cMPitch, rPPitch, rS, nnz, aI, aJ, aVal, baseIndex);
}

} // extern "C"
