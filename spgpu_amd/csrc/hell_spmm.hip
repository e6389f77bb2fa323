/*
 * HELL SpMM for gfx950 (MI355X):  Z = alpha*A*X + beta*Y, `count` right-hand
 * sides, interleaved multivectors (include/spgpu/spmm.h).  New operation (the
 * reference has none); A uses the HELL arguments of hell.h:45-59.
 *
 * ---- Wavefront design -------------------------------------------------------
 * KP (4, 8 or 16) lanes form a "row team": lane j of the team owns right-hand
 * side j.  A wavefront has G = 64/KP teams and owns one 32-row group of a hack;
 * team g owns rows g, g+G, g+2G, ... of the group (32/G rows), one running sum
 * per owned row in registers.
 *   for every slab column k:  for every owned row i:
 *       (coef, col) of (row, k)    -- the same address for the KP lanes of a
 *                                     team: one broadcast load; the G teams of
 *                                     the wave read G adjacent elements, and
 *                                     the inner loop over i consumes the
 *                                     32-row slab column completely before k
 *                                     advances, so every 128-B line of cM/rP
 *                                     is fetched once and then hit in L1
 *       x = X[col*ld + j]          -- KP lanes read KP consecutive values: ONE
 *                                     128-B line per nonzero for 16 doubles
 *       sum[i] = fma(coef, x, sum[i])
 * No LDS, no cross-lane traffic; the order of additions per (row, rhs) is
 * ascending k.  More than 16 right-hand sides run as passes of 16.
 *
 * Roofline: HBM.  Algorithmic bytes: the matrix once, nnz*(sizeof(T)+4) +
 * rows*4 + hacks*4, plus count * (cols + rows*(1+[beta!=0])) * sizeof(T).
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/spmm.h"

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
    int count;      /* right-hand sides in this pass (<= KP) */
    long long ldX, ldYZ;
};

constexpr int kSpmmThreads = 256;
constexpr int kSpmmGroupRows = 32;

template <typename T, int KP, int UNROLL>
__global__ __launch_bounds__(kSpmmThreads) void hellSpmmKernel(const SpmmArgs<T> a)
{
    constexpr int G = kWave / KP;              /* row teams per wavefront */
    constexpr int OWN = kSpmmGroupRows / G;    /* rows per team */

    const int lane = threadIdx.x & (kWave - 1);
    const long long group = (long long)blockIdx.x * (kSpmmThreads / kWave) + (threadIdx.x >> 6);
    const long long groupRow0 = group * kSpmmGroupRows;
    if (groupRow0 >= a.rows)
        return;

    const int team = lane / KP;
    const int j = lane % KP;
    const bool rhsLive = j < a.count;

    /* hackSize is a multiple of 32 on this path, so the whole group is in one hack */
    const unsigned g0 = (unsigned)groupRow0, hs = (unsigned)a.hackSize;
    const unsigned hack = g0 / hs;
    const long long slab = (long long)a.hackOffsets[hack] + (g0 - hack * hs);

    int len[OWN];
    int longest = 0;
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        const long long r = groupRow0 + team + (long long)i * G;
        len[i] = r < a.rows ? a.rS[r] : 0;
        longest = len[i] > longest ? len[i] : longest;
    }
    const int groupLongest = waveMax(longest);

    T sum[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i)
        sum[i] = zeroOf<T>();

    const T* __restrict__ vals = a.cM + slab + team;
    const int* __restrict__ idxs = a.rP + slab + team;
    const T* __restrict__ X = a.X + j;

    for (int kBase = 0; kBase < groupLongest; kBase += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kBase + u;
            const long long column = (long long)k * a.hackSize;
            T coef[OWN];
            int col[OWN];
            bool use[OWN];
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                use[i] = k < len[i];
                coef[i] = use[i] ? vals[column + i * G] : zeroOf<T>();
                col[i] = use[i] ? idxs[column + i * G] - a.baseIndex : 0;
                use[i] = use[i] && col[i] >= 0 && rhsLive;
            }
            T xv[OWN];
#pragma unroll
            for (int i = 0; i < OWN; ++i)
                xv[i] = use[i] ? X[(long long)col[i] * a.ldX] : zeroOf<T>();
#pragma unroll
            for (int i = 0; i < OWN; ++i)
                sum[i] = use[i] ? mulAdd(coef[i], xv[i], sum[i]) : sum[i];
        }
    }

    if (!rhsLive)
        return;
    const bool hasBeta = isNotZero(a.beta);
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        const long long r = groupRow0 + team + (long long)i * G;
        if (r < a.rows) {
            const long long outRow = a.rIdx ? a.rIdx[r] : r;
            const long long at = outRow * a.ldYZ + j;
            a.Z[at] = hasBeta ? epilogue<true>(a.alpha, sum[i], a.beta, a.Y[at])
                              : epilogue<false>(a.alpha, sum[i], a.beta, zeroOf<T>());
        }
    }
}

template <typename T>
static void hellSpmm(spgpuHandle_t handle, T* Z, const T* Y, T alpha, const T* cM, const int* rP, int hackSize,
                     const int* hackOffsets, const int* rS, const int* rIdx, int rows, const T* X, T beta,
                     int baseIndex, int count, int ldX, int ldYZ)
{
    if (rows <= 0 || count <= 0)
        return;
    if (hackSize <= 0 || hackSize % kSpmmGroupRows != 0) {
        fprintf(stderr, "spgpu?hellspmm: hackSize must be a positive multiple of 32 (got %d)\n", hackSize);
        return;
    }
    hipStream_t stream = handle->currentStream;
    const long long groups = ((long long)rows + kSpmmGroupRows - 1) / kSpmmGroupRows;
    const unsigned blocks = (unsigned)((groups + kSpmmThreads / kWave - 1) / (kSpmmThreads / kWave));

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
        if (a.count > 8)
            hipLaunchKernelGGL((hellSpmmKernel<T, 16, 4>), dim3(blocks), dim3(kSpmmThreads), 0, stream, a);
        else if (a.count > 4)
            hipLaunchKernelGGL((hellSpmmKernel<T, 8, 4>), dim3(blocks), dim3(kSpmmThreads), 0, stream, a);
        else
            hipLaunchKernelGGL((hellSpmmKernel<T, 4, 2>), dim3(blocks), dim3(kSpmmThreads), 0, stream, a);
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
