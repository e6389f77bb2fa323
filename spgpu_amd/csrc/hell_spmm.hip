/*
 * HELL SpMM for gfx950 (MI355X):  Z = alpha*A*X + beta*Y, `count` right-hand
 * sides, interleaved multivectors (include/spgpu/spmm.h).  New operation (the
 * reference has none); A uses the HELL arguments of hell.h:45-59.
 *
 * ---- Wavefront design -------------------------------------------------------
 * A wavefront owns 64 consecutive rows.  Two lane roles alternate:
 *
 *  load role   lane l fetches coefficient and column index of (row l, slab
 *              column k): for hackSize 32 the wave reads two whole slab columns
 *              of two hacks -- fully coalesced, every byte of cM/rP is fetched
 *              exactly once, UNROLL columns ahead of their use.
 *  team role   KP lanes form a row team, G = 64/KP teams per wave; lane t of a
 *              team owns VEC consecutive right-hand sides (KP*VEC >= count, for
 *              16 rhs: 8 lanes x 2).  Team g owns rows g*KP .. g*KP+KP-1 of the
 *              group and keeps one running sum per owned row and rhs.  In step
 *              i every team takes the (coef, col) pair that lane g*KP+i loaded
 *              -- a lane shuffle inside the team (ds_bpermute / DPP, no LDS
 *              allocation, no barrier) -- and all its lanes read their slice of
 *              X row `col`: the KP lanes of a team read ONE contiguous 128-byte
 *              line (16 doubles), one wave-wide 16-byte load serves G nonzeros.
 *
 * Per (row, rhs) the products are added in ascending k.  More than 16
 * right-hand sides run as passes of 16 (the matrix is re-read per pass).
 *
 * This describes hellSpmmKernel (any hackSize, any rhs count).  The default for
 * hackSize % 32 == 0 and an even rhs count > 8 is hellSpmmStripKernel further
 * down: same teams and summation order, but 16-byte loads of whole half-columns,
 * the X window of a workgroup in LDS, and (offset, coefficient) handed from the
 * loader lanes to the teams through LDS instead of lane shuffles.
 *
 * Roofline: HBM.  Algorithmic bytes: the matrix once, nnz*(sizeof(T)+4) +
 * rows*4 + hacks*4, plus count * (cols + rows*(1+[beta!=0])) * sizeof(T).
 */
#include "numeric.hip.h"
#include <type_traits>
#include "spgpu_internal.h"

#include "spgpu/spmm.h"

#include <stdio.h>
#include <stdlib.h>

namespace spgpu {

template <typename T> struct SpmmArgs {
    T* Z;
    const T* Y;
    const T* X;
    const T* cM;
    const int* rP;
    const int* rS;
    const int* rIdx;
    const int* hackOffsets;
    T alpha, beta;
    int rows, baseIndex, hackSize;
    int count;      /* right-hand sides in this pass (<= KP*VEC) */
    int tileRows;   /* tiled kernel: X rows the LDS tile can hold */
    int directFill; /* strip kernel: every 16-byte piece of a tile row is 16 valid, aligned bytes of X (global_load_lds) */
    int bandMode;   /* strip kernel: wavefronts whose rows form a band take the sliding-window loop (SPGPU_SPMM_VARIANT=11: never) */
    long long ldX, ldYZ;
};

constexpr int kSpmmThreads = 256;
constexpr int kSpmmTileBytes = 43 * 1024; /* X tile; tile + padded record slots = 52 KiB, so three workgroups fit the 160 KiB LDS of a CU */

__device__ inline float laneFrom(float v, int src) { return __shfl(v, src, kWave); }
__device__ inline double laneFrom(double v, int src) { return __shfl(v, src, kWave); }
__device__ inline int laneFrom(int v, int src) { return __shfl(v, src, kWave); }

/* The accumulation loop shared by both kernels.
 * FROM_LDS == false: X rows are read from global memory (through L1/L2).
 * FROM_LDS == true : X rows come from the workgroup's LDS tile.  LDS reads retire on lgkmcnt, global loads on
 *                    vmcnt, and each counter retires in issue order -- so only in this form can the (coef, col)
 *                    pairs of the NEXT slab columns be requested from HBM at the top of an iteration and stay in
 *                    flight while the current columns are consumed (with global X reads a wait for them would
 *                    also wait for the older prefetch: measured, profiles/r01b_ab_spmm_pipelined.txt). */
/* A 16-byte LDS read is served in 16-lane groups over 64 banks (256 B): two teams whose records lie 128 B apart
 * hit the same banks.  One pad record after every 8 shifts the teams of a group onto different banks
 * (SQ_LDS_BANK_CONFLICT was 58 % of the LDS cycles without it). */
constexpr int kSpmmStage = 2; /* slab columns published to LDS and consumed at a time */
constexpr int kRecordPadEvery = 8;
constexpr int kRecordsPerColumn = kWave + kWave / kRecordPadEvery;

/* What a loader lane publishes for its row's entry of one slab column.  `at` is the byte offset of the X row inside
 * the LDS tile, computed once by the loader instead of by each of the KP consumer lanes; negative = no entry. */
template <typename T> struct alignas(16) SpmmRecord {
    T coef;
    int at;
};
/* moved as ONE 16-byte LDS access (the compiler would split a plain struct copy into b64 + b32) */
template <typename T> __device__ inline SpmmRecord<T> loadRecord(const SpmmRecord<T>* p)
{
    const Pack<uint32_t, 4> raw = loadPack<false, uint32_t, 4>(reinterpret_cast<const uint32_t*>(p));
    SpmmRecord<T> out;
    __builtin_memcpy(&out, &raw, sizeof(out));
    return out;
}
template <typename T> __device__ inline void storeRecord(SpmmRecord<T>* p, T coef, int at)
{
    SpmmRecord<T> rec = {};
    rec.coef = coef;
    rec.at = at;
    Pack<uint32_t, 4> raw;
    __builtin_memcpy(&raw, &rec, sizeof(raw));
    storePack<uint32_t, 4>(reinterpret_cast<uint32_t*>(p), raw);
}

template <typename T, int KP, int VEC, int UNROLL, bool FROM_LDS>
__device__ inline void spmmAccumulate(const SpmmArgs<T>& a, int lane, int myLen, int groupLongest,
                                      const T* __restrict__ vals, const int* __restrict__ idxs,
                                      const T* __restrict__ tile, int tileFirst, T (&sum)[KP][VEC],
                                      SpmmRecord<T>* records = nullptr)
{
    constexpr int TILE_LD = KP * VEC;
    /* CHUNK rows of the team at a time: CHUNK X-row reads in flight per lane */
    constexpr int CHUNK = KP < 4 ? KP : 4;
    const int team = lane / KP;
    const int rhs0 = (lane % KP) * VEC;
    const int rhsSafe = rhs0 < a.count ? rhs0 : 0; /* lanes beyond `count` read a valid slice, result discarded */

    auto fetch = [&](int kBase, T* coef, int* col) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u;
            if (k < myLen) {
                coef[u] = vals[(long long)k * a.hackSize];
                col[u] = idxs[(long long)k * a.hackSize] - a.baseIndex;
            } else {
                coef[u] = zeroOf<T>();
                col[u] = -1; /* no entry */
            }
        }
    };

    T coefMine[UNROLL];
    int colMine[UNROLL];
    if constexpr (FROM_LDS) {
        static_assert(KP == kRecordPadEvery, "record padding assumes one pad slot per team");
        static_assert(UNROLL % kSpmmStage == 0, "a trip is a whole number of stages");
        const unsigned char* const myTile = reinterpret_cast<const unsigned char*>(tile) + rhsSafe * sizeof(T);
        const SpmmRecord<T>* const teamRecords = records + team * (KP + 1);
        SpmmRecord<T>* const myRecord = records + lane + lane / kRecordPadEvery;
        /* ALL_PRESENT: every row of the wavefront has an entry in these slab columns (always, for uniform rows):
         * no per-entry test, 16 fused multiply-adds + 8 address adds per lane and column.  Otherwise absent
         * entries (at < 0) read tile row 0 and their product is discarded. */
        auto consume = [&](auto allPresent) {
            constexpr bool ALL_PRESENT = decltype(allPresent)::value;
            /* all KP rows of the team at once: LDS bounds the occupancy here (3 wavefronts per SIMD), so the
             * registers for KP reads in flight are free */
            constexpr int CHUNK = KP;
#pragma unroll
            for (int u = 0; u < kSpmmStage; ++u) {
#pragma unroll
                for (int i0 = 0; i0 < KP; i0 += CHUNK) {
                    SpmmRecord<T> rec[CHUNK];
                    Pack<T, VEC> xv[CHUNK];
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i) /* one 16-byte LDS read, the same address for the lanes of a team */
                        rec[i] = loadRecord(teamRecords + u * kRecordsPerColumn + i0 + i);
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i) {
                        const int at = ALL_PRESENT ? rec[i].at : (rec[i].at >= 0 ? rec[i].at : 0);
                        xv[i] = loadPack<false, T, VEC>(reinterpret_cast<const T*>(myTile + at));
                    }
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const T next = mulAdd(rec[i].coef, xv[i].v[e], sum[i0 + i][e]);
                            sum[i0 + i][e] = ALL_PRESENT ? next : pick(rec[i].at >= 0, next, sum[i0 + i][e]);
                        }
                    /* keep the scheduler from hoisting every chunk's reads to the top: that costs registers
                     * (occupancy), not latency -- other wavefronts cover it */
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        const int groupShortest = waveMin(myLen);
        T coefNext[UNROLL];
        int colNext[UNROLL];
        fetch(0, coefMine, colMine);
        for (int kBase = 0; kBase < groupLongest; kBase += UNROLL) {
            /* (coef, col) of the next UNROLL columns requested now: 2*UNROLL loads per lane stay in flight (vmcnt)
             * while this trip runs on LDS (lgkmcnt).  With the 3 wavefronts per SIMD the tile leaves room for,
             * this depth is what keeps enough bytes in flight to cover the HBM latency. */
            fetch(kBase + UNROLL, coefNext, colNext);
#pragma unroll
            for (int s0 = 0; s0 < UNROLL; s0 += kSpmmStage) {
                if (kBase + s0 < groupLongest) { /* wavefront-uniform */
                    /* publish kSpmmStage columns to the wavefront's own record slots; a wavefront's LDS operations
                     * execute in order, the barriers only stop the compiler from reordering across them */
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < kSpmmStage; ++u)
                        storeRecord(myRecord + u * kRecordsPerColumn, coefMine[s0 + u],
                                    colMine[s0 + u] >= 0 ? (colMine[s0 + u] - tileFirst) * (int)(TILE_LD * sizeof(T)) : -1);
                    __builtin_amdgcn_wave_barrier();
                    if (kBase + s0 + kSpmmStage <= groupShortest) /* wavefront-uniform */
                        consume(std::true_type{});
                    else
                        consume(std::false_type{});
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                coefMine[u] = coefNext[u];
                colMine[u] = colNext[u];
            }
        }
    } else {
        const T* __restrict__ Xsafe = a.X + rhsSafe;
        for (int kBase = 0; kBase < groupLongest; kBase += UNROLL) {
            fetch(kBase, coefMine, colMine);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
                for (int i0 = 0; i0 < KP; i0 += CHUNK) {
                    T coef[CHUNK];
                    int col[CHUNK];
                    Pack<T, VEC> xv[CHUNK];
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i) {
                        const int src = team * KP + i0 + i;
                        coef[i] = laneFrom(coefMine[u], src);
                        col[i] = laneFrom(colMine[u], src);
                    }
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i) /* no branch: absent entries read row 0 and are discarded below */
                        xv[i] = loadPack<false, T, VEC>(Xsafe + (long long)(col[i] >= 0 ? col[i] : 0) * a.ldX);
#pragma unroll
                    for (int i = 0; i < CHUNK; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            sum[i0 + i][e] = pick(col[i] >= 0, mulAdd(coef[i], xv[i].v[e], sum[i0 + i][e]), sum[i0 + i][e]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

/* Epilogue shared by both kernels: team g writes rows g*KP .. g*KP+KP-1, lane t the rhs t*VEC .. */
template <typename T, int KP, int VEC>
__device__ inline void spmmStore(const SpmmArgs<T>& a, int lane, long long groupRow0, T (&sum)[KP][VEC])
{
    const int team = lane / KP;
    const int rhs0 = (lane % KP) * VEC;
    if (rhs0 >= a.count)
        return;
    const bool hasBeta = isNotZero(a.beta);
    /* Z += alpha*A*X in place (Y == Z, beta == 1): rows of A without entries keep their Z, unread and unwritten.
     * This is what the "rest" product of a column-split row block is made of (spgpu_amd/sharded.py). */
    const bool inPlaceSum = hasBeta && a.Y == a.Z && a.beta == T(1);
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const long long r = groupRow0 + team * KP + i;
        if (r < a.rows && !(inPlaceSum && a.rS[r] == 0)) {
            const long long outRow = a.rIdx ? a.rIdx[r] : r;
            const long long at = outRow * a.ldYZ + rhs0;
            Pack<T, VEC> out;
            if (hasBeta) {
                const Pack<T, VEC> yv = loadPack<false, T, VEC>(a.Y + at);
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    out.v[e] = epilogue<true>(a.alpha, sum[i][e], a.beta, yv.v[e]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    out.v[e] = epilogue<false>(a.alpha, sum[i][e], a.beta, zeroOf<T>());
            }
            storePack<T, VEC>(a.Z + at, out);
        }
        __builtin_amdgcn_sched_barrier(0); /* one row's addresses and y values live at a time */
    }
}

/* TILED == false: plain kernel.  TILED == true: the workgroup first finds the window of X rows its 256 matrix rows
 * touch; if the window fits the LDS tile (banded / FEM-like matrices) it is copied into LDS once, coalesced, and the
 * accumulation reads X from there (LDS: 256 B/clk/CU, vector L1: 64); otherwise it accumulates from global memory. */
template <typename T, int KP, int VEC, int UNROLL, bool TILED>
__global__ __launch_bounds__(kSpmmThreads) void hellSpmmKernel(const SpmmArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char spmmLds[];
    const int lane = threadIdx.x & (kWave - 1);
    const long long group = (long long)blockIdx.x * (kSpmmThreads / kWave) + (threadIdx.x >> 6);
    const long long groupRow0 = group * kWave;
    if constexpr (!TILED) {
        if (groupRow0 >= a.rows)
            return; /* whole wavefront leaves together (the tiled form has workgroup barriers: everyone stays) */
    }

    /* ---- load role: this lane's row ---- */
    const long long myRow = groupRow0 + lane;
    int myLen = 0;
    long long slab = 0;
    if (myRow < a.rows) {
        const unsigned r = (unsigned)myRow, hs = (unsigned)a.hackSize;
        const unsigned hack = r / hs;
        slab = (long long)a.hackOffsets[hack] + (r - hack * hs);
        myLen = a.rS[myRow];
    }
    const int groupLongest = waveMax(myLen);
    const T* __restrict__ vals = a.cM + slab;
    const int* __restrict__ idxs = a.rP + slab;

    T sum[KP][VEC];
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            sum[i][e] = zeroOf<T>();

    if constexpr (TILED) {
        constexpr int TILE_LD = KP * VEC;
        T* const tile = reinterpret_cast<T*>(spmmLds);
        __shared__ int waveLo[kSpmmThreads / kWave], waveHi[kSpmmThreads / kWave];
        /* pass 1: column window of the workgroup (the indices are read again below, out of L2).
         * First a probe on slab column 0 only (one coalesced load): scattered matrices already span more than
         * the tile there and skip the full scan; then 8 independent loads per trip over all columns. */
        auto blockWindow = [&](int& lo, int& hi) {
#pragma unroll
            for (int m = 1; m < kWave; m <<= 1) {
                const int olo = laneXor(lo, m), ohi = laneXor(hi, m);
                lo = olo < lo ? olo : lo;
                hi = ohi > hi ? ohi : hi;
            }
            __syncthreads(); /* previous use of waveLo/waveHi is over */
            if (lane == 0) {
                waveLo[threadIdx.x >> 6] = lo;
                waveHi[threadIdx.x >> 6] = hi;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kSpmmThreads / kWave; ++w) {
                lo = waveLo[w] < lo ? waveLo[w] : lo;
                hi = waveHi[w] > hi ? waveHi[w] : hi;
            }
        };
        int lo = 0x7fffffff, hi = -1;
        if (myLen > 0) {
            const int c = idxs[0] - a.baseIndex;
            if (c >= 0)
                lo = hi = c;
        }
        blockWindow(lo, hi);
        const bool worthScanning = hi < lo || (long long)hi - lo < a.tileRows; /* workgroup-uniform */
        if (worthScanning) {
            constexpr int SCAN = 16; /* independent loads per trip: the scan is a chain of memory latencies */
            for (int k0 = 1; k0 < myLen; k0 += SCAN) {
                int c[SCAN];
#pragma unroll
                for (int u = 0; u < SCAN; ++u)
                    c[u] = k0 + u < myLen ? idxs[(long long)(k0 + u) * a.hackSize] - a.baseIndex : -1;
#pragma unroll
                for (int u = 0; u < SCAN; ++u) {
                    if (c[u] >= 0) {
                        lo = c[u] < lo ? c[u] : lo;
                        hi = c[u] > hi ? c[u] : hi;
                    }
                }
            }
            blockWindow(lo, hi);
        }
        const bool useTile = worthScanning && hi >= lo && (long long)hi - lo < a.tileRows; /* workgroup-uniform */
        if (useTile) {
            const int window = hi - lo + 1;
            /* KP lanes copy one X row, VEC elements (16 bytes) each; FILL loads per lane in flight */
            constexpr int FILL = 4;
            const int pieces = window * KP;
            for (int i0 = threadIdx.x; i0 < pieces; i0 += FILL * kSpmmThreads) {
                Pack<T, VEC> part[FILL];
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    const int i = i0 + f * kSpmmThreads;
                    const int r = i / KP, piece = i % KP;
                    if (i < pieces && piece * VEC < a.count)
                        part[f] = loadPack<false, T, VEC>(a.X + (long long)(lo + r) * a.ldX + piece * VEC);
                }
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    const int i = i0 + f * kSpmmThreads;
                    const int r = i / KP, piece = i % KP;
                    if (i < pieces && piece * VEC < a.count)
                        storePack<T, VEC>(tile + r * TILE_LD + piece * VEC, part[f]);
                }
            }
        }
        __syncthreads();
        /* per-wavefront record slots behind the tile */
        SpmmRecord<T>* records = reinterpret_cast<SpmmRecord<T>*>(spmmLds + kSpmmTileBytes) + (threadIdx.x >> 6) * (kSpmmStage * kRecordsPerColumn);
        if (useTile)
            spmmAccumulate<T, KP, VEC, UNROLL, true>(a, lane, myLen, groupLongest, vals, idxs, tile, lo, sum, records);
        else /* window too wide: X through L1/L2, the plain kernel's trip width */
            spmmAccumulate<T, KP, VEC, (UNROLL < 2 ? UNROLL : 2), false>(a, lane, myLen, groupLongest, vals, idxs, tile, 0, sum);
        if (groupRow0 >= a.rows)
            return;
    } else {
        spmmAccumulate<T, KP, VEC, UNROLL, false>(a, lane, myLen, groupLongest, vals, idxs, nullptr, 0, sum);
    }
    spmmStore<T, KP, VEC>(a, lane, groupRow0, sum);
}


/* ---------------------------------------------------------------------------------------------------------------
 * Strip-loading tiled kernel: hackSize a multiple of 32, up to 16 right-hand sides as 8 lanes x 2.
 *
 * The LDS tile leaves room for 3 wavefronts per SIMD only, so what bounds the kernel is the number of bytes each
 * wavefront keeps in flight.  One-row-per-lane loads move 4 (index) or 8 (coefficient) bytes per lane; here every
 * load is 16 bytes per lane: a wavefront's 64 rows are two halves of 32 rows, each inside one hack, and
 *   index role        lane l reads rows 4q..4q+3 (q = l%8) of half (l/8)%2 in slab column k0 + l/16: one
 *                     instruction covers 4 slab columns of the 64 rows;
 *   coefficient role  the same with 16/sizeof(T) rows per lane: 2 (double) or 1 (float) instructions per 4 columns.
 * A stage is 4 slab columns.  The loader lanes publish it to the wavefront's own LDS staging area -- the byte
 * offset of the X row inside the tile, computed once, or -1 for "no entry", and the coefficient -- and the teams
 * read their 8 rows' values back with 16-byte LDS reads (same address for the 8 lanes of a team).  A trip is
 * TRIP stages; the loads of the next trip are issued before the current one is consumed and stay in flight while
 * it runs on LDS.  The window scan of the prologue uses the same 16-byte index loads, 8 per lane in flight.
 * Per (row, rhs) the products are still added in ascending k.
 */
constexpr int kStageCols = 4;
constexpr int kStripTileBytes = 40 * 1024; /* + 4 staging areas of 3 KiB (double) = 52 KiB: three workgroups per CU */

template <typename T> struct alignas(16) SpmmStage {
    int at[kStageCols][kWave];
    T coef[kStageCols][kWave];
};

__device__ inline void waveSync()
{
    /* a wavefront's LDS operations execute in order; this only pins the compiler's order of the accesses */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* amdgpu_waves_per_eu(3): the LDS footprint admits 3 wavefronts per SIMD; tell the register allocator to stay
 * within the matching 168 VGPRs instead of trading occupancy for scheduling freedom */
template <typename T, int TRIP, int VEC>
__global__ __launch_bounds__(kSpmmThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void hellSpmmStripKernel(const SpmmArgs<T> a)
{
    constexpr int KP = 8, TILE_LD = KP * VEC; /* VEC right-hand sides per lane: 2 (up to 16 in all) or 1 (up to 8) */
    constexpr int ROW_BYTES = TILE_LD * (int)sizeof(T);
    constexpr int CR = 16 / (int)sizeof(T);                     /* rows per coefficient load */
    constexpr int COEF_LOADS = kStageCols * (int)sizeof(T) / 16; /* per stage */
    constexpr int COLS_PER_COEF_LOAD = kStageCols / COEF_LOADS;
    constexpr int LANES_PER_HALF_COL = 32 / CR;
    constexpr int WAVES = kSpmmThreads / kWave;
    /* Tile layout.  A team reads one whole X row (ROW_BYTES) per instruction, and the LDS serves 256 bytes (64 banks)
     * per pass: rows at a distance of 8 -- what neighbouring teams read in a banded matrix -- would share banks if
     * row r simply sat at r*ROW_BYTES.  Inside each 256-byte line the rows are therefore permuted by r>>3
     * (measured: SQ_LDS_BANK_CONFLICT was 32 % of the LDS cycles without it). */
    constexpr int ROWS_PER_LINE = 256 / ROW_BYTES;
    auto tileOffset = [](int r) { return (r / ROWS_PER_LINE) * 256 + ((r ^ (r >> 3)) & (ROWS_PER_LINE - 1)) * ROW_BYTES; };

    extern __shared__ __attribute__((aligned(16))) unsigned char spmmLds[];
    __shared__ int waveLo[WAVES], waveHi[WAVES], waveLongest[WAVES];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const long long groupRow0 = ((long long)blockIdx.x * WAVES + wave) * kWave;
    T* const tile = reinterpret_cast<T*>(spmmLds);
    SpmmStage<T>* const stage = reinterpret_cast<SpmmStage<T>*>(spmmLds + kStripTileBytes) + wave;
    const unsigned hs = (unsigned)a.hackSize;
    /* The tile fill: X rows first .. first + count - 1 into LDS.  Where whole 16-byte pieces line up (a.directFill, decided by
     * the host) the copy goes straight from global memory into LDS (global_load_lds_dwordx4): no registers, no ds_write, and
     * ALL of a lane's pieces in flight at once -- through registers (5 pieces per lane at a time beside the two trips
     * already in flight) a 288-row window was two dependent round trips.  One wave-wide instruction writes 1 KiB of LDS in
     * lane order from 64 per-lane addresses: lane -> LDS position is fixed, so the lane works out WHICH piece of X lands
     * there (the inverse of tileOffset); only whole wavefronts take it, the ragged end goes through registers. */
    auto fillTile = [&](int first, int count) {
        constexpr int PIECES_PER_ROW = ROW_BYTES >= 16 ? ROW_BYTES / 16 : 1;
        constexpr int PIECE_ELEMS = 16 / (int)sizeof(T);
        if (ROW_BYTES >= 16 && a.directFill) {
            const int lines = (count + ROWS_PER_LINE - 1) / ROWS_PER_LINE;
            const int slots = lines * 16; /* 16-byte slots of LDS, in address order */
            for (int s0 = wave * kWave; s0 < slots; s0 += kSpmmThreads) { /* wavefront-uniform */
                const int slot = s0 + lane;
                const int line = slot >> 4, q = (slot & 15) / PIECES_PER_ROW, piece = slot % PIECES_PER_ROW;
                const int r = line * ROWS_PER_LINE + ((q ^ ((line * ROWS_PER_LINE) >> 3)) & (ROWS_PER_LINE - 1));
                const bool live = slot < slots && r < count;
                const T* from = a.X + (long long)(first + (live ? r : 0)) * a.ldX + piece * PIECE_ELEMS;
                if (__ballot(live) == ~0ull) {
#if defined(__HIP_DEVICE_COMPILE__) /* the host pass of hipcc parses the kernel body too and has no such builtin */
                    __builtin_amdgcn_global_load_lds(from, reinterpret_cast<unsigned char*>(tile) + (size_t)slot * 16, 16, 0, 0);
#endif
                } else if (live) {
                    const Pack<T, PIECE_ELEMS> w = loadPack<false, T, PIECE_ELEMS>(from);
                    storePack<T, PIECE_ELEMS>(reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(tile) + (size_t)slot * 16), w);
                }
            }
            return;
        }
        /* KP lanes copy one X row, VEC elements each; FILL loads per lane in flight */
        constexpr int FILL = 5;
        const int pieces = count * KP;
        for (int i0 = threadIdx.x; i0 < pieces; i0 += FILL * kSpmmThreads) {
            Pack<T, VEC> part[FILL];
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                const int i = i0 + f * kSpmmThreads;
                const int r = i / KP, piece = i % KP;
                if (i < pieces && piece * VEC < a.count)
                    part[f] = loadPack<false, T, VEC>(a.X + (long long)(first + r) * a.ldX + piece * VEC);
            }
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                const int i = i0 + f * kSpmmThreads;
                const int r = i / KP, piece = i % KP;
                if (i < pieces && piece * VEC < a.count)
                    storePack<T, VEC>(reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(tile) + tileOffset(r)) + piece * VEC, part[f]);
            }
        }
    };


    /* ---- index role ---- */
    const int iCol = lane >> 4, iHalf = (lane >> 3) & 1, iQ = lane & 7;
    const long long iRow0 = groupRow0 + 32 * iHalf + 4 * iQ;
    int iLen[4] = {0, 0, 0, 0};
    const int* __restrict__ iBase = a.rP;
    if (iRow0 < a.rows) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (iRow0 + j < a.rows)
                iLen[j] = a.rS[iRow0 + j];
        const unsigned hack = (unsigned)iRow0 / hs;
        iBase += (long long)a.hackOffsets[hack] + ((unsigned)iRow0 - hack * hs);
    }
    int iLenMax = iLen[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        iLenMax = iLen[j] > iLenMax ? iLen[j] : iLenMax;
    /* ---- coefficient role ---- */
    const int cCol = lane / (2 * LANES_PER_HALF_COL), cHalf = (lane / LANES_PER_HALF_COL) & 1, cQ = lane % LANES_PER_HALF_COL;
    const long long cRow0 = groupRow0 + 32 * cHalf + CR * cQ;
    int cLenMax = 0;
    const T* __restrict__ cBase = a.cM;
    if (cRow0 < a.rows) {
#pragma unroll
        for (int j = 0; j < CR; ++j)
            if (cRow0 + j < a.rows) {
                const int len = a.rS[cRow0 + j];
                cLenMax = len > cLenMax ? len : cLenMax;
            }
        const unsigned hack = (unsigned)cRow0 / hs;
        cBase += (long long)a.hackOffsets[hack] + ((unsigned)cRow0 - hack * hs);
    }
    const int groupLongest = waveMax(iLenMax);

    /* ---- prologue: the window of X rows the workgroup's 256 matrix rows touch ---- */
    auto blockWindow = [&](int& lo, int& hi) {
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) {
            const int olo = laneXor(lo, m), ohi = laneXor(hi, m);
            lo = olo < lo ? olo : lo;
            hi = ohi > hi ? ohi : hi;
        }
        __syncthreads(); /* previous use of waveLo/waveHi is over */
        if (lane == 0) {
            waveLo[wave] = lo;
            waveHi[wave] = hi;
            waveLongest[wave] = groupLongest;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            lo = waveLo[w] < lo ? waveLo[w] : lo;
            hi = waveHi[w] > hi ? waveHi[w] : hi;
        }
    };
    auto widen = [&](const Pack<int, 4>& c4, int k, int& lo, int& hi) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4.v[j] - a.baseIndex;
            if (k < iLen[j] && c >= 0) {
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
        }
    };
    struct Trip {
        Pack<int, 4> idx[TRIP];
        Pack<T, CR> coef[TRIP * COEF_LOADS];
    };
    auto loadTripCoef = [&](int k0, Trip& t) {
#pragma unroll
        for (int s = 0; s < TRIP; ++s) {
#pragma unroll
            for (int j = 0; j < COEF_LOADS; ++j) {
                const int kc = k0 + kStageCols * s + COLS_PER_COEF_LOAD * j + cCol;
                if (kc < cLenMax) {
                    t.coef[s * COEF_LOADS + j] = loadPack<true, T, CR>(cBase + (long long)kc * hs);
                } else {
#pragma unroll
                    for (int e = 0; e < CR; ++e)
                        t.coef[s * COEF_LOADS + j].v[e] = zeroOf<T>();
                }
            }
        }
    };
    constexpr int STEP = kStageCols * TRIP;
    Trip cur, next;
    int lo = 0x7fffffff, hi = -1;
    /* The indices of the first HEAD*4 slab columns are requested at once and stay in registers: the accumulation
     * below takes them from there instead of reading them a second time. */
    constexpr int HEAD = 8;
    Pack<int, 4> head[HEAD];
#pragma unroll
    for (int u = 0; u < HEAD; ++u) {
        const int k = kStageCols * u + iCol;
        if (k < iLenMax)
            head[u] = loadPack<false, int, 4>(iBase + (long long)k * hs);
        else
            head[u] = Pack<int, 4>{{0, 0, 0, 0}};
    }
#pragma unroll
    for (int u = 0; u < HEAD; ++u)
        widen(head[u], kStageCols * u + iCol, lo, hi);
    blockWindow(lo, hi);
    int blockLongest = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
        blockLongest = waveLongest[w] > blockLongest ? waveLongest[w] : blockLongest;
    /* scattered matrices already span more than the tile here and skip the rest (workgroup-uniform) */
    const bool fitsSoFar = hi < lo || (long long)hi - lo < a.tileRows;
    if (fitsSoFar && blockLongest > kStageCols * HEAD) {
        constexpr int SCAN = 8; /* 16-byte loads per lane in flight: 32 slab columns per trip */
        for (int k0 = kStageCols * HEAD; k0 < groupLongest; k0 += kStageCols * SCAN) {
            Pack<int, 4> c4[SCAN];
#pragma unroll
            for (int u = 0; u < SCAN; ++u) {
                const int k = k0 + kStageCols * u + iCol;
                if (k < iLenMax)
                    c4[u] = loadPack<false, int, 4>(iBase + (long long)k * hs);
            }
#pragma unroll
            for (int u = 0; u < SCAN; ++u) {
                const int k = k0 + kStageCols * u + iCol;
                if (k < iLenMax)
                    widen(c4[u], k, lo, hi);
            }
        }
        blockWindow(lo, hi);
    }
    const bool useTile = fitsSoFar && hi >= lo && (long long)hi - lo < a.tileRows; /* workgroup-uniform */

    /* BAND wavefronts.  In a band or stencil matrix in natural order row r + 1 names the columns of row r shifted by one, and a
     * row's entries ascend by one: over the 8 rows of a team and 8 slab columns only 15 different X rows occur, each used up to 8
     * times.  A wavefront all of whose 64 rows have that shape through ALL their columns -- column of (row i, slab column k) =
     * bandBase + k + i, every row exactly groupLongest <= 32 entries long; checked here against the indices of the head, one
     * ballot -- takes a loop of its own below: a team keeps a sliding window of 8 X rows in registers and reads ONE new row per
     * slab column from the tile instead of 8, needs no offsets from the loader lanes (only the coefficients go through LDS), and
     * never looks at an index again (the head's 32 registers are dead in that loop: the window takes their place).  It is the
     * SpMM counterpart of the SpMV's strip x loads.  Same products, added in the same order: ascending k. */
    int bandBase = -1; /* relative to lo; wavefront-uniform */
    if (a.bandMode && useTile && groupLongest > 0 && groupLongest <= kStageCols * HEAD && groupLongest % (kStageCols * TRIP) == 0) {
        const int base = __builtin_amdgcn_readfirstlane(head[0].v[0] - a.baseIndex - lo);
        bool off = false;
#pragma unroll
        for (int u = 0; u < HEAD; ++u) {
            const int k = kStageCols * u + iCol;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                off |= k < groupLongest && (iLen[j] != groupLongest || head[u].v[j] - a.baseIndex - lo != base + k + 32 * iHalf + 4 * iQ + j);
        }
        bandBase = __ballot(off) == 0ull && groupRow0 + kWave <= a.rows ? base : -1;
    }

    auto loadTrip = [&](int k0, Trip& t) {
#pragma unroll
        for (int s = 0; s < TRIP; ++s) {
            const int ki = k0 + kStageCols * s + iCol;
            if (k0 + kStageCols * s < kStageCols * HEAD) { /* uniform: still in the registers of the prologue */
                t.idx[s] = head[0];
#pragma unroll
                for (int u = 0; u + 1 < HEAD; ++u)
                    head[u] = head[u + 1];
            } else if (ki < iLenMax) {
                t.idx[s] = loadPack<true, int, 4>(iBase + (long long)ki * hs);
            } else {
                t.idx[s] = Pack<int, 4>{{0, 0, 0, 0}};
            }
#pragma unroll
            for (int j = 0; j < COEF_LOADS; ++j) {
                const int kc = k0 + kStageCols * s + COLS_PER_COEF_LOAD * j + cCol;
                if (kc < cLenMax) {
                    t.coef[s * COEF_LOADS + j] = loadPack<true, T, CR>(cBase + (long long)kc * hs);
                } else {
#pragma unroll
                    for (int e = 0; e < CR; ++e)
                        t.coef[s * COEF_LOADS + j].v[e] = zeroOf<T>();
                }
            }
        }
    };
    auto loadTripIdx = [&](Trip& t) { /* of the first trips: from the registers of the prologue */
        static_assert(2 * TRIP <= HEAD, "the first two trips' indices are in the head");
#pragma unroll
        for (int s = 0; s < TRIP; ++s) {
            t.idx[s] = head[0];
#pragma unroll
            for (int u = 0; u + 1 < HEAD; ++u)
                head[u] = head[u + 1];
        }
    };
    if (useTile) {
        /* the first two trips' coefficients are requested before the tile is filled: one memory round trip for both */
        loadTripCoef(0, cur);
        loadTripCoef(STEP, next);
        loadTripIdx(cur);
        loadTripIdx(next);
        fillTile(lo, hi - lo + 1);
    }
    /* (global_load_lds retires on vmcnt like any load, and the barrier below is what hands the tile to the other wavefronts:
     * the wait is spelled out rather than left to whatever else happens to be waited for here) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    T sum[KP][VEC];
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            sum[i][e] = zeroOf<T>();

    if (bandBase >= 0) {
        static_assert(kStageCols * TRIP == KP, "the window's names come round once per trip");
        const int team = lane / KP;
        const int rhs0 = (lane % KP) * VEC;
        const int rhsSafe = rhs0 < a.count ? rhs0 : 0;
        const unsigned char* const myTile = reinterpret_cast<const unsigned char*>(tile) + rhsSafe * sizeof(T);
        auto tileRow = [&](int r) { return loadPack<false, T, VEC>(reinterpret_cast<const T*>(myTile + tileOffset(r))); };
        /* X rows first + C .. first + C + 7 of slab column C (counted from 0), by rotating name: row first + C + i sits in
         * window[(C + i) % 8]; a trip of 8 columns brings the names round once, so the window carries on from trip to trip */
        int newest = bandBase + KP * team + KP - 1; /* the row that enters with the next slab column */
        Pack<T, VEC> window[KP];
#pragma unroll
        for (int i = 0; i + 1 < KP; ++i)
            window[i] = tileRow(newest - (KP - 1) + i);
        Pack<T, CR> coefNow[TRIP * COEF_LOADS], coefNext[TRIP * COEF_LOADS];
#pragma unroll
        for (int q = 0; q < TRIP * COEF_LOADS; ++q) { /* requested with the tile fill, above */
            coefNow[q] = cur.coef[q];
            coefNext[q] = next.coef[q];
        }
        /* every row of the wavefront is groupLongest long: the bounds of the coefficient loads are wavefront-uniform, and a
         * lane's loads of one trip differ from the previous trip's by a uniform stride */
        const T* coefAt = cBase + ((long long)(2 * STEP) + cCol) * hs; /* this lane's first load of the trip after next */
        const long long tripStride = (long long)STEP * hs, loadStride = (long long)COLS_PER_COEF_LOAD * hs, stageStride = (long long)kStageCols * hs;
        const T* const stageCoefRead = &stage->coef[0][KP * team];
        T* const stageCoefWrite = &stage->coef[cCol][32 * cHalf + CR * cQ];
#pragma clang loop unroll(disable)
        for (int k0 = 0; k0 < groupLongest; k0 += STEP) {
            /* the trip after next: coefficients only, requested BEFORE this trip is consumed (a third set of registers: this
             * loop has them to spare), so that all of a 32-column row's matrix bytes are on their way within the first trip */
            Pack<T, CR> coefAfter[TRIP * COEF_LOADS];
            if (k0 + 2 * STEP < groupLongest) { /* wavefront-uniform */
#pragma unroll
                for (int s = 0; s < TRIP; ++s)
#pragma unroll
                    for (int j = 0; j < COEF_LOADS; ++j)
                        coefAfter[s * COEF_LOADS + j] = loadPack<true, T, CR>(coefAt + s * stageStride + j * loadStride);
            }
            coefAt += tripStride;
#pragma unroll
            for (int s = 0; s < TRIP; ++s) {
                waveSync();
#pragma unroll
                for (int j = 0; j < COEF_LOADS; ++j)
                    storePack<T, CR>(stageCoefWrite + COLS_PER_COEF_LOAD * j * kWave, coefNow[s * COEF_LOADS + j]);
                waveSync();
#pragma unroll
                for (int c = 0; c < kStageCols; ++c) {
                    const int C = kStageCols * s + c; /* compile-time after unrolling */
                    window[(C + KP - 1) % KP] = tileRow(newest);
                    newest += 1;
#pragma unroll
                    for (int i0 = 0; i0 < KP; i0 += 4) {
                        T coef[4];
#pragma unroll
                        for (int j0 = 0; j0 < 4; j0 += CR) {
                            const Pack<T, CR> part = loadPack<false, T, CR>(stageCoefRead + c * kWave + i0 + j0);
#pragma unroll
                            for (int j = 0; j < CR; ++j)
                                coef[j0 + j] = part.v[j];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                sum[i0 + i][e] = mulAdd(coef[i], window[(C + i0 + i) % KP].v[e], sum[i0 + i][e]);
                        /* The multiply-adds have no place of their own in the order of the block (nothing but the next
                         * iteration needs the sums): left alone, the compiler gathers all 128 of a trip behind all 40 LDS
                         * reads and spills what the reads delivered.  The empty statements tie each chunk's sums down here. */
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                asm volatile("" : "+v"(sum[i0 + i][e]));
                        __builtin_amdgcn_sched_barrier(0); /* keep the reads of later chunks from being hoisted: registers */
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < TRIP * COEF_LOADS; ++q) {
                coefNow[q] = coefNext[q];
                coefNext[q] = coefAfter[q];
            }
        }
    } else if (!useTile) {
        /* window too wide for the tile: one row per lane, X through L1/L2 (the plain kernel's loop) */
        const long long myRow = groupRow0 + lane;
        int myLen = 0;
        long long slab = 0;
        if (myRow < a.rows) {
            const unsigned hack = (unsigned)myRow / hs;
            slab = (long long)a.hackOffsets[hack] + ((unsigned)myRow - hack * hs);
            myLen = a.rS[myRow];
        }
        spmmAccumulate<T, KP, VEC, 2, false>(a, lane, myLen, groupLongest, a.cM + slab, a.rP + slab, nullptr, 0, sum);
    } else {
        const int team = lane / KP;
        const int rhs0 = (lane % KP) * VEC;
        const int rhsSafe = rhs0 < a.count ? rhs0 : 0; /* lanes beyond `count` read a valid slice, result discarded */
        const unsigned char* const myTile = reinterpret_cast<const unsigned char*>(tile) + rhsSafe * sizeof(T);

        /* wavefront-uniform: every row of the wavefront has an entry in every column of the trip */
        auto allPresent = [&](const Trip& t, int k0) {
            bool absent = false;
#pragma unroll
            for (int s = 0; s < TRIP; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    absent |= !(k0 + kStageCols * s + iCol < iLen[j] && t.idx[s].v[j] - a.baseIndex >= 0);
            return __ballot(absent) == 0ull;
        };
        auto publish = [&](const Trip& t, int s, int k0) {
            const int ki = k0 + kStageCols * s + iCol;
            Pack<int, 4> at4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = t.idx[s].v[j] - a.baseIndex;
                at4.v[j] = ki < iLen[j] && c >= 0 ? tileOffset(c - lo) : -1;
            }
            storePack<int, 4>(&stage->at[iCol][32 * iHalf + 4 * iQ], at4);
#pragma unroll
            for (int j = 0; j < COEF_LOADS; ++j)
                storePack<T, CR>(&stage->coef[COLS_PER_COEF_LOAD * j + cCol][32 * cHalf + CR * cQ], t.coef[s * COEF_LOADS + j]);
        };
        /* (Tried: the offsets -- and the coefficients -- of the next 4 rows read one step ahead, so that a wavefront does not go
         * through two dependent LDS round trips per 8 fused multiply-adds.  Offsets only: within the noise; both: +12 registers
         * at 168, spills inside this loop, 0.97 ms against 0.65.) */
        auto consume = [&](auto allPresent) {
            constexpr bool ALL_PRESENT = decltype(allPresent)::value;
#pragma unroll
            for (int c = 0; c < kStageCols; ++c) {
#pragma unroll
                for (int i0 = 0; i0 < KP; i0 += 4) { /* 4 rows of the team at a time */
                    T coef[4];
                    Pack<T, VEC> xv[4];
                    const Pack<int, 4> at = loadPack<false, int, 4>(&stage->at[c][KP * team + i0]);
#pragma unroll
                    for (int j0 = 0; j0 < 4; j0 += CR) {
                        const Pack<T, CR> part = loadPack<false, T, CR>(&stage->coef[c][KP * team + i0 + j0]);
#pragma unroll
                        for (int j = 0; j < CR; ++j)
                            coef[j0 + j] = part.v[j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        xv[i] = loadPack<false, T, VEC>(reinterpret_cast<const T*>(myTile + (ALL_PRESENT || at.v[i] >= 0 ? at.v[i] : 0)));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const T next = mulAdd(coef[i], xv[i].v[e], sum[i0 + i][e]);
                            sum[i0 + i][e] = ALL_PRESENT ? next : pick(at.v[i] >= 0, next, sum[i0 + i][e]);
                        }
                    __builtin_amdgcn_sched_barrier(0); /* keep the reads of later chunks from being hoisted: registers */
                }
            }
        };
        auto runTrip = [&](const Trip& t, int k0, auto mode) {
#pragma unroll
            for (int s = 0; s < TRIP; ++s) {
                if (k0 + kStageCols * s < groupLongest) { /* wavefront-uniform */
                    waveSync();
                    publish(t, s, k0);
                    waveSync();
                    consume(mode);
                }
            }
        };
        /* Two loops rather than a per-stage choice: absent entries only appear in the last columns of ragged rows,
         * and one loop body per mode keeps the 32 running sums in one set of registers. */
        int k0 = 0;
#pragma clang loop unroll(disable)
        while (k0 < groupLongest && allPresent(cur, k0)) {
            runTrip(cur, k0, std::true_type{});
            cur = next;
            k0 += STEP;
            loadTrip(k0 + STEP, next); /* in flight (vmcnt) while the next trip runs on LDS (lgkmcnt) */
        }
#pragma clang loop unroll(disable)
        while (k0 < groupLongest) {
            runTrip(cur, k0, std::false_type{});
            cur = next;
            k0 += STEP;
            loadTrip(k0 + STEP, next);
        }
    }
    if (groupRow0 >= a.rows)
        return;
    spmmStore<T, KP, VEC>(a, lane, groupRow0, sum);
}

template <typename T, int TRIP, int VEC = 2> static void launchSpmmStrips(hipStream_t stream, const SpmmArgs<T>& in)
{
    SpmmArgs<T> a = in;
    const long long groups = ((long long)a.rows + kWave - 1) / kWave;
    const unsigned blocks = (unsigned)((groups + kSpmmThreads / kWave - 1) / (kSpmmThreads / kWave));
    a.tileRows = kStripTileBytes / (8 * VEC * (int)sizeof(T));
    /* whole tile rows of valid bytes: all KP * VEC right-hand sides present, rows of X 16-byte aligned */
    a.directFill = a.count == 8 * VEC && (8 * VEC * sizeof(T)) % 16 == 0 && (uintptr_t)a.X % 16 == 0 &&
                   (a.ldX * (long long)sizeof(T)) % 16 == 0 && spgpuTuning()->spmmVariant != 10;
    a.bandMode = spgpuTuning()->spmmVariant != 11;
    const size_t lds = kStripTileBytes + (kSpmmThreads / kWave) * sizeof(SpmmStage<T>);
    hipLaunchKernelGGL((hellSpmmStripKernel<T, TRIP, VEC>), dim3(blocks), dim3(kSpmmThreads), lds, stream, a);
}

template <typename T, int KP, int VEC, int UNROLL, bool TILED = false>
static void launchSpmm(hipStream_t stream, const SpmmArgs<T>& in)
{
    SpmmArgs<T> a = in;
    const long long groups = ((long long)a.rows + kWave - 1) / kWave;
    const unsigned blocks = (unsigned)((groups + kSpmmThreads / kWave - 1) / (kSpmmThreads / kWave));
    a.tileRows = TILED ? kSpmmTileBytes / (KP * VEC * (int)sizeof(T)) : 0;
    const size_t lds = TILED ? kSpmmTileBytes + (kSpmmThreads / kWave) * kSpmmStage * kRecordsPerColumn * sizeof(SpmmRecord<T>) : 0;
    hipLaunchKernelGGL((hellSpmmKernel<T, KP, VEC, UNROLL, TILED>), dim3(blocks), dim3(kSpmmThreads), lds, stream, a);
}

template <typename T>
static void hellSpmm(spgpuHandle_t handle, T* Z, const T* Y, T alpha, const T* cM, const int* rP, int hackSize,
                     const int* hackOffsets, const int* rS, const int* rIdx, int rows, const T* X, T beta,
                     int baseIndex, int count, int ldX, int ldYZ)
{
    if (rows <= 0 || count <= 0 || hackSize <= 0)
        return;
    hipStream_t stream = handle->currentStream;
    /* two right-hand sides per lane need 2*sizeof(T)-aligned rows of X, Y and Z */
    const size_t pair = 2 * sizeof(T);
    const bool pairsOk = ldX % 2 == 0 && ldYZ % 2 == 0 && (uintptr_t)X % pair == 0 && (uintptr_t)Z % pair == 0 &&
                         (!Y || (uintptr_t)Y % pair == 0);

    for (int first = 0; first < count; first += 16) {
        SpmmArgs<T> a;
        a.Z = Z + first;
        a.Y = Y ? Y + first : nullptr;
        a.X = X + first;
        a.cM = cM;
        a.rP = rP;
        a.rS = rS;
        a.rIdx = rIdx;
        a.hackOffsets = hackOffsets;
        a.alpha = alpha;
        a.beta = beta;
        a.rows = rows;
        a.baseIndex = baseIndex;
        a.hackSize = hackSize;
        a.count = count - first < 16 ? count - first : 16;
        a.ldX = ldX;
        a.ldYZ = ldYZ;
        a.tileRows = 0;
        a.directFill = 0;
        a.bandMode = 0;
        const bool pairs = pairsOk && a.count % 2 == 0;
        /* 16-byte loads of whole 32-row half columns */
        const bool strips = pairs && hackSize % 32 == 0 && (uintptr_t)cM % 16 == 0 && (uintptr_t)rP % 16 == 0;
        const int variant = spgpuTuning()->spmmVariant; /* experiments, include/spgpu/tuning.h */
        if (a.count > 8) {
            if (pairs && variant == 1)
                launchSpmm<T, 8, 2, 2>(stream, a);          /* plain: X rows through L1 */
            else if (pairs && variant == 2)
                launchSpmm<T, 8, 2, 8, true>(stream, a);    /* tiled, 8 slab columns per trip */
            else if (pairs && variant == 3)
                launchSpmm<T, 8, 2, 4>(stream, a);
            else if (variant == 4)
                launchSpmm<T, 16, 1, 2>(stream, a);
            else if (pairs && variant == 5)
                launchSpmm<T, 8, 2, 2, true>(stream, a);    /* tiled, 2 slab columns per trip */
            else if (pairs && (variant == 6 || !strips))
                launchSpmm<T, 8, 2, 4, true>(stream, a);    /* tiled, one row per loader lane */
            else if (strips)
                launchSpmmStrips<T, 2>(stream, a);          /* default: X window in LDS when it fits, 16-byte loads */
            else
                launchSpmm<T, 16, 1, 2>(stream, a);
        } else if (a.count > 4) {
            /* 5 to 8 right-hand sides: the strip kernel with one per lane (64-byte X rows for fp64; no pairing, so odd
             * counts and odd leading dimensions too) */
            const bool strips1 = hackSize % 32 == 0 && (uintptr_t)cM % 16 == 0 && (uintptr_t)rP % 16 == 0;
            if (strips1 && variant == 9 && pairs)
                launchSpmmStrips<T, 2>(stream, a);          /* experiment: the 16-rhs form with half of each team idle */
            else if (strips1 && variant != 1)
                launchSpmmStrips<T, 2, 1>(stream, a);
            else if (pairs)
                launchSpmm<T, 4, 2, 4>(stream, a);
            else
                launchSpmm<T, 8, 1, 2>(stream, a);
        } else {
            /* 4: the strip kernel with half of each team idle still wins (banded 0.58 vs 0.63 ms, windowed 1.54 vs
             * 1.72 ms); 1-3: the small-team plain kernel is as fast or faster on scattered columns */
            if (a.count == 4 && variant != 1 && hackSize % 32 == 0 && (uintptr_t)cM % 16 == 0 && (uintptr_t)rP % 16 == 0)
                launchSpmmStrips<T, 2, 1>(stream, a);
            else
                launchSpmm<T, 4, 1, 4>(stream, a);
        }
    }
    spgpuDebugCheck(handle, "hellspmm");
}

/* Layout conversion through a 32x33 LDS tile so that both sides are coalesced. */
template <typename T, bool TO_INTERLEAVED>
__global__ __launch_bounds__(256) void mvTransposeKernel(T* dst, long long dstLd, const T* src, long long srcLd, int n,
                                                         int count)
{
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; /* 32 x 8 */
    /* "long" axis i (0..n), "short" axis j (0..count) */
    const long long i0 = (long long)blockIdx.x * 32;
    const int j0 = blockIdx.y * 32;
    if constexpr (TO_INTERLEAVED) {
        /* src[j*srcLd + i] -> dst[i*dstLd + j] */
        for (int jj = ty; jj < 32; jj += 8)
            if (i0 + tx < n && j0 + jj < count)
                tile[jj][tx] = src[(long long)(j0 + jj) * srcLd + i0 + tx];
        __syncthreads();
        for (int ii = ty; ii < 32; ii += 8)
            if (i0 + ii < n && j0 + tx < count)
                dst[(i0 + ii) * dstLd + j0 + tx] = tile[tx][ii];
    } else {
        /* src[i*srcLd + j] -> dst[j*dstLd + i] */
        for (int ii = ty; ii < 32; ii += 8)
            if (i0 + ii < n && j0 + tx < count)
                tile[ii][tx] = src[(i0 + ii) * srcLd + j0 + tx];
        __syncthreads();
        for (int jj = ty; jj < 32; jj += 8)
            if (i0 + tx < n && j0 + jj < count)
                dst[(long long)(j0 + jj) * dstLd + i0 + tx] = tile[tx][jj];
    }
}

template <typename T, bool TO_INTERLEAVED>
static void mvTranspose(spgpuHandle_t handle, T* dst, int dstLd, const T* src, int srcLd, int n, int count)
{
    if (n <= 0 || count <= 0)
        return;
    const dim3 grid((unsigned)(((long long)n + 31) / 32), (unsigned)((count + 31) / 32));
    hipLaunchKernelGGL((mvTransposeKernel<T, TO_INTERLEAVED>), grid, dim3(256), 0, handle->currentStream, dst,
                       (long long)dstLd, src, (long long)srcLd, n, count);
    spgpuDebugCheck(handle, "mvTranspose");
}

} // namespace spgpu

using namespace spgpu;

extern "C" {

void spgpuShellspmm(spgpuHandle_t handle, float* Z, const float* Y, float alpha, const float* cM, const int* rP,
                    int hackSize, const int* hackOffsets, const int* rS, const int* rIdx, int avgNnzPerRow, int rows,
                    const float* X, float beta, int baseIndex, int count, int ldX, int ldYZ)
{
    (void)avgNnzPerRow;
    hellSpmm<float>(handle, Z, Y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, X, beta, baseIndex, count, ldX, ldYZ);
}

void spgpuDhellspmm(spgpuHandle_t handle, double* Z, const double* Y, double alpha, const double* cM, const int* rP,
                    int hackSize, const int* hackOffsets, const int* rS, const int* rIdx, int avgNnzPerRow, int rows,
                    const double* X, double beta, int baseIndex, int count, int ldX, int ldYZ)
{
    (void)avgNnzPerRow;
    hellSpmm<double>(handle, Z, Y, alpha, cM, rP, hackSize, hackOffsets, rS, rIdx, rows, X, beta, baseIndex, count, ldX, ldYZ);
}

void spgpuSmvInterleave(spgpuHandle_t h, float* dst, int ld, const float* src, int pitch, int n, int count)
{ mvTranspose<float, true>(h, dst, ld, src, pitch, n, count); }
void spgpuDmvInterleave(spgpuHandle_t h, double* dst, int ld, const double* src, int pitch, int n, int count)
{ mvTranspose<double, true>(h, dst, ld, src, pitch, n, count); }
void spgpuSmvDeinterleave(spgpuHandle_t h, float* dst, int pitch, const float* src, int ld, int n, int count)
{ mvTranspose<float, false>(h, dst, pitch, src, ld, n, count); }
void spgpuDmvDeinterleave(spgpuHandle_t h, double* dst, int pitch, const double* src, int ld, int n, int count)
{ mvTranspose<double, false>(h, dst, pitch, src, ld, n, count); }

} // extern "C"
