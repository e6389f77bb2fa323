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
 * Roofline: HBM.  Algorithmic bytes: the matrix once, nnz*(sizeof(T)+4) +
 * rows*4 + hacks*4, plus count * (cols + rows*(1+[beta!=0])) * sizeof(T).
 */
#include "numeric.hip.h"
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
    long long ldX, ldYZ;
};

constexpr int kSpmmThreads = 256;

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
template <typename T, int KP, int VEC, int UNROLL, bool FROM_LDS>
__device__ inline void spmmAccumulate(const SpmmArgs<T>& a, int lane, int myLen, int groupLongest,
                                      const T* __restrict__ vals, const int* __restrict__ idxs,
                                      const T* __restrict__ tile, int tileFirst, T (&sum)[KP][VEC])
{
    constexpr int TILE_LD = KP * VEC;
    const int team = lane / KP;
    const int rhs0 = (lane % KP) * VEC;
    const int rhsSafe = rhs0 < a.count ? rhs0 : 0; /* lanes beyond `count` read a valid slice, result discarded */
    const T* __restrict__ Xsafe = a.X + rhsSafe;

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
    auto consume = [&](const T* coefMine, const int* colMine) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            /* CHUNK rows of the team at a time: CHUNK X-row reads in flight per lane */
            constexpr int CHUNK = KP < 4 ? KP : 4;
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
                for (int i = 0; i < CHUNK; ++i) {
                    /* no branch: inactive slots read a valid row and are discarded below */
                    if constexpr (FROM_LDS)
                        xv[i] = loadPack<false, T, VEC>(tile + (col[i] >= 0 ? col[i] - tileFirst : 0) * TILE_LD + rhsSafe);
                    else
                        xv[i] = loadPack<false, T, VEC>(Xsafe + (long long)(col[i] >= 0 ? col[i] : 0) * a.ldX);
                }
#pragma unroll
                for (int i = 0; i < CHUNK; ++i)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        sum[i0 + i][e] = pick(col[i] >= 0, mulAdd(coef[i], xv[i].v[e], sum[i0 + i][e]), sum[i0 + i][e]);
                /* keep the scheduler from hoisting every chunk's shuffles and reads to the top:
                 * that costs registers (occupancy), not latency -- other wavefronts cover it */
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    T coefMine[UNROLL];
    int colMine[UNROLL];
    if constexpr (FROM_LDS) {
        T coefNext[UNROLL];
        int colNext[UNROLL];
        fetch(0, coefMine, colMine);
        for (int kBase = 0; kBase < groupLongest; kBase += UNROLL) {
            fetch(kBase + UNROLL, coefNext, colNext);
            consume(coefMine, colMine);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                coefMine[u] = coefNext[u];
                colMine[u] = colNext[u];
            }
        }
    } else {
        for (int kBase = 0; kBase < groupLongest; kBase += UNROLL) {
            fetch(kBase, coefMine, colMine);
            consume(coefMine, colMine);
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
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const long long r = groupRow0 + team * KP + i;
        if (r < a.rows) {
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
            for (int k0 = 1; k0 < myLen; k0 += 8) {
                int c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    c[u] = k0 + u < myLen ? idxs[(long long)(k0 + u) * a.hackSize] - a.baseIndex : -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
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
            /* KP lanes copy one X row, VEC elements (16 bytes) each */
            for (int i = threadIdx.x; i < window * KP; i += kSpmmThreads) {
                const int r = i / KP, piece = i % KP;
                if (piece * VEC < a.count)
                    storePack<T, VEC>(tile + r * TILE_LD + piece * VEC,
                                      loadPack<false, T, VEC>(a.X + (long long)(lo + r) * a.ldX + piece * VEC));
            }
        }
        __syncthreads();
        if (useTile)
            spmmAccumulate<T, KP, VEC, UNROLL, true>(a, lane, myLen, groupLongest, vals, idxs, tile, lo, sum);
        else
            spmmAccumulate<T, KP, VEC, UNROLL, false>(a, lane, myLen, groupLongest, vals, idxs, tile, 0, sum);
        if (groupRow0 >= a.rows)
            return;
    } else {
        spmmAccumulate<T, KP, VEC, UNROLL, false>(a, lane, myLen, groupLongest, vals, idxs, nullptr, 0, sum);
    }
    spmmStore<T, KP, VEC>(a, lane, groupRow0, sum);
}

constexpr int kSpmmTileBytes = 48 * 1024; /* three workgroups per CU keep their tiles in the 160 KiB LDS */

template <typename T, int KP, int VEC, int UNROLL, bool TILED = false>
static void launchSpmm(hipStream_t stream, const SpmmArgs<T>& in)
{
    SpmmArgs<T> a = in;
    const long long groups = ((long long)a.rows + kWave - 1) / kWave;
    const unsigned blocks = (unsigned)((groups + kSpmmThreads / kWave - 1) / (kSpmmThreads / kWave));
    a.tileRows = TILED ? kSpmmTileBytes / (KP * VEC * (int)sizeof(T)) : 0;
    hipLaunchKernelGGL((hellSpmmKernel<T, KP, VEC, UNROLL, TILED>), dim3(blocks), dim3(kSpmmThreads),
                       TILED ? kSpmmTileBytes : 0, stream, a);
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
        const bool pairs = pairsOk && a.count % 2 == 0;
        const char* ev = getenv("SPGPU_SPMM_VARIANT"); /* experiments */
        const int variant = ev && *ev ? atoi(ev) : 0;
        if (a.count > 8) {
            if (pairs && variant == 1)
                launchSpmm<T, 8, 2, 2>(stream, a);          /* plain: X rows through L1 */
            else if (pairs && variant == 2)
                launchSpmm<T, 8, 2, 4, true>(stream, a);    /* tiled, 4 slab columns per stage */
            else if (pairs && variant == 3)
                launchSpmm<T, 8, 2, 4>(stream, a);
            else if (variant == 4)
                launchSpmm<T, 16, 1, 2>(stream, a);
            else if (pairs)
                launchSpmm<T, 8, 2, 2, true>(stream, a);    /* default: X window staged in LDS when it fits */
            else
                launchSpmm<T, 16, 1, 2>(stream, a);
        } else if (a.count > 4) {
            if (pairs)
                launchSpmm<T, 4, 2, 4>(stream, a);
            else
                launchSpmm<T, 8, 1, 2>(stream, a);
        } else {
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
