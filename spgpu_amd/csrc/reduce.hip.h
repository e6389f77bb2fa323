#pragma once
/*
 * The shape of a two-stage reduction, shared by the Level-1 reductions (level1.hip) and the fused solver steps
 * (fused_solver.hip): 256-lane workgroups, 4 independent 16-byte accesses per lane, lane-xor tree inside a
 * wavefront, the 4 wavefront sums added in wavefront order, block partials combined in the fixed order of finalOrder.
 * Reference shape: kernels/ddot.cu:35-150 (per-block partials, host adds them).
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

namespace spgpu {

constexpr int kL1Threads = 256;
constexpr int kL1Unroll = 4; /* independent 16-byte accesses in flight per lane */
enum ReduceMode { kDot = 0, kNrm2 = 1, kAsum = 2, kAmax = 3 };



/* A coefficient given as num/den in device memory (NULL = 1). */
template <typename T> __device__ inline T quotientAt(const T* num, const T* den)
{
    if (num && den)
        return *num / *den;
    if (num)
        return *num;
    return den ? T(1) / *den : T(1);
}

template <int MODE, typename A> __device__ __host__ inline A combine(A x, A y)
{
    if constexpr (MODE == kAmax)
        return y > x ? y : x;
    else
        return add(x, y);
}

template <int MODE, typename A> __device__ inline A blockCombine(A v, A* lds)
{
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1)
        v = combine<MODE>(v, laneXor(v, m));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0)
        lds[wave] = v;
    __syncthreads();
    A total = lds[0];
#pragma unroll
    for (int w = 1; w < kL1Threads / kWave; ++w)
        total = combine<MODE>(total, lds[w]);
    return total;
}

/* Second stage: the block partials of one vector become one value.  The reference adds them on the host in block
 * order (ddot.cu:139-150) -- one dependent chain, 4.8 us of a 54 us CG iteration when it runs on the device.  Here
 * the order is a fixed two-level one, the same on the host (spgpu?dot, finalOrder) and on the device (spgpu?dotDevice,
 * reduceFinalKernel), so both return the same bits: partial k belongs to lane k / 16 of one wavefront (missing ones
 * count as +0); a lane adds its 16 in ascending k starting from +0, the 64 lane sums meet in the lane-xor tree
 * (1, 2, 4, ... 32).  22 dependent additions instead of up to 1 024.  (Finalising inside reduceKernel by the workgroup
 * that arrives last was measured slower: every workgroup then pays a device-scope fence, 16.6 us against 5.0 + 3 us
 * per dot of 2^20 doubles.) */
constexpr int kFinalPerLane = SPGPU_REDUCE_MAX_BLOCKS / kWave;

template <int MODE, typename Acc> static inline Acc finalOrder(const Acc* partials, long long blocks)
{
    Acc lane[kWave];
    for (int l = 0; l < kWave; ++l) {
        Acc sum = zeroOf<Acc>();
        for (int j = 0; j < kFinalPerLane; ++j) {
            const long long k = (long long)l * kFinalPerLane + j;
            sum = combine<MODE>(sum, k < blocks ? partials[k] : zeroOf<Acc>());
        }
        lane[l] = sum;
    }
    for (int m = 1; m < kWave; m <<= 1) {
        Acc next[kWave];
        for (int l = 0; l < kWave; ++l)
            next[l] = combine<MODE>(lane[l], lane[l ^ m]);
        for (int l = 0; l < kWave; ++l)
            lane[l] = next[l];
    }
    return lane[0];
}

__device__ inline float squareRoot(float v) { return __builtin_sqrtf(v); }
__device__ inline double squareRoot(double v) { return __builtin_sqrt(v); }

template <typename Acc, int MODE, bool ROOT = false>
__global__ __launch_bounds__(kWave) void reduceFinalKernel(Acc* result, const Acc* partials, int blocks)
{
    Acc mine[kFinalPerLane];
#pragma unroll
    for (int j = 0; j < kFinalPerLane; ++j) {
        const int k = (int)threadIdx.x * kFinalPerLane + j;
        mine[j] = k < blocks ? partials[k] : zeroOf<Acc>();
    }
    Acc sum = zeroOf<Acc>();
#pragma unroll
    for (int j = 0; j < kFinalPerLane; ++j)
        sum = combine<MODE>(sum, mine[j]);
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1)
        sum = combine<MODE>(sum, laneXor(sum, m));
    if (threadIdx.x == 0) {
        if constexpr (ROOT)
            *result = squareRoot(sum); /* nrm2: sqrt of the unscaled sum of squares (dnrm2.cu:146) */
        else
            *result = sum;
    }
}

} // namespace spgpu
