/*
 * Fused steps of a Krylov iteration on small systems, for gfx950 (MI355X).
 *
 * C ABI: spgpu{S,D}hellspmvDotDevice, spgpu{S,D}axpbyPairDotDevice (include/spgpu/device_scalars.h).  NEW: the
 * reference has no fused calls; a CG iteration written with it is hellspmv + dot + 2 axpby + dot + axpby, each
 * reduction a host round trip (vector.h:61-120, ddot.cu:120-150).  On the 1024 x 1024 Laplacian (BASELINE
 * configs[0]) every one of those kernels moves 8-60 MB that sit in the Infinity Cache: the iteration is bound by
 * the number of dependent launches, not by bytes.  These two calls take three launches and two re-reads of a vector
 * out of it.
 *
 * Both kernels ARE the first stage of the dot (reduce.hip.h: same grid, same element -> lane mapping, same order of
 * additions) with the second operand produced on the fly:
 *   hellspmvDot   element i of the second operand = row i of alpha*A*x + beta*y, computed by the lane that owns
 *                 element i of the dot and stored to z; the row sum runs over the row's entries in ascending
 *                 k (the reference's one-thread-per-row order, hell_spmv_base_template.cuh:104-215);
 *   axpbyPairDot  element i = y2[i] - a*x2[i] (stored to z2), next to z1 = y1 + a*x1.
 * So *result has the bits spgpu?dotDevice / spgpu?dot would return for the stored vectors, and z the bits of
 * orc_?hellspmv with one phase -- which are spgpu?hellspmv's own bits whenever its wavefronts do not switch to the
 * cooperative tail (rows of even length, e.g. every stencil).
 *
 * Roofline: HBM / Infinity Cache.  Algorithmic bytes: hellspmvDot = the SpMV's + n*sizeof(T) when w != x;
 * axpbyPairDot = 6*n*sizeof(T).
 */
#include "reduce.hip.h"

#include "spgpu/device_scalars.h"

#include <stdint.h>

namespace spgpu {

template <typename T> struct FusedSpmvArgs {
    T* partials;
    T* z;
    const T* y;
    const T* cM;
    const int* rP;
    const int* hackOffsets;
    const int* rS;
    const T* x;
    const T* w;
    T alpha, beta;
    int hackSize, rows, baseIndex;
};

/* Row sums of VEC consecutive rows (first row `row`, all inside one hack when PACKED) in ascending k. */
template <typename T, int VEC, bool PACKED>
__device__ inline void rowSums(const FusedSpmvArgs<T>& a, long long row, bool live, T (&sum)[VEC])
{
    int len[VEC];
    long long slot[VEC];
    int longest = 0;
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
        sum[t] = zeroOf<T>();
        len[t] = 0;
        slot[t] = 0;
    }
    if (live) {
        if constexpr (PACKED) {
            const Pack<int, VEC> l = loadPack<false, int, VEC>(a.rS + row);
            const long long first = (long long)a.hackOffsets[row / a.hackSize] + row % a.hackSize;
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                len[t] = l.v[t];
                slot[t] = first + t;
            }
        } else {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                const long long r = row + t;
                len[t] = a.rS[r];
                slot[t] = (long long)a.hackOffsets[r / a.hackSize] + r % a.hackSize;
            }
        }
#pragma unroll
        for (int t = 0; t < VEC; ++t)
            longest = len[t] > longest ? len[t] : longest;
    }
    for (int k = 0; k < longest; ++k) {
        T value[VEC];
        int column[VEC];
        if constexpr (PACKED) {
            const long long s = slot[0] + (long long)k * a.hackSize;
            const Pack<T, VEC> v = loadPack<false, T, VEC>(a.cM + s);
            const Pack<int, VEC> c = loadPack<false, int, VEC>(a.rP + s);
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                value[t] = v.v[t];
                column[t] = c.v[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                const long long s = slot[t] + (long long)k * a.hackSize;
                const bool in = k < len[t];
                value[t] = in ? a.cM[s] : zeroOf<T>();
                column[t] = in ? a.rP[s] : a.baseIndex;
            }
        }
        /* neighbouring rows of a stencil or a band name neighbouring columns: then the VEC values of x are ONE load
         * (wavefront-uniform choice; the same values either way) */
        bool strip = PACKED && VEC > 1;
        if constexpr (PACKED && VEC > 1) {
            bool mine = column[0] - a.baseIndex >= 0;
#pragma unroll
            for (int t = 0; t < VEC; ++t)
                mine = mine && k < len[t] && column[t] == column[0] + t;
            strip = __ballot(!mine && k < longest) == 0ull;
        }
        if (strip) {
            const Pack<T, VEC> xs = loadPackElementAligned<T, VEC>(a.x + (k < longest ? column[0] - a.baseIndex : 0));
#pragma unroll
            for (int t = 0; t < VEC; ++t)
                if (k < len[t])
                    sum[t] = mulAdd(value[t], xs.v[t], sum[t]);
        } else {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                const int col = column[t] - a.baseIndex;
                const bool use = k < len[t] && col >= 0;
                const T xv = a.x[use ? col : 0];
                if (use)
                    sum[t] = mulAdd(value[t], xv, sum[t]);
            }
        }
    }
}

template <typename T, int VEC, bool PACKED, bool HAS_BETA>
__global__ __launch_bounds__(kL1Threads) void hellSpmvDotKernel(FusedSpmvArgs<T> a)
{
    __shared__ T lds[kL1Threads / kWave];
    T acc = zeroOf<T>();
    const long long packs = a.rows / VEC;
    constexpr long long TILE = (long long)kL1Threads * kL1Unroll;
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        T sums[kL1Unroll][VEC];
        Pack<T, VEC> wv[kL1Unroll], yv[kL1Unroll];
        bool live[kL1Unroll];
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            live[u] = p < packs;
            if (live[u]) {
                wv[u] = loadPack<false, T, VEC>(a.w + p * VEC);
                if constexpr (HAS_BETA)
                    yv[u] = loadPackElementAligned<T, VEC>(a.y + p * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u)
            rowSums<T, VEC, PACKED>(a, (base + u * kL1Threads + threadIdx.x) * VEC, live[u], sums[u]);
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            if (live[u]) {
                Pack<T, VEC> out;
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    out.v[t] = epilogue<HAS_BETA>(a.alpha, sums[u][t], a.beta, HAS_BETA ? yv[u].v[t] : zeroOf<T>());
                    acc = mulAdd(wv[u].v[t], out.v[t], acc);
                }
                storePack<T, VEC>(a.z + (base + u * kL1Threads + threadIdx.x) * VEC, out);
            }
        }
    }
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < a.rows) {
        T sum[1];
        rowSums<T, 1, false>(a, tail, true, sum);
        const T out = epilogue<HAS_BETA>(a.alpha, sum[0], a.beta, HAS_BETA ? a.y[tail] : zeroOf<T>());
        acc = mulAdd(a.w[tail], out, acc);
        a.z[tail] = out;
    }
    const T total = blockCombine<kDot>(acc, lds);
    if (threadIdx.x == 0)
        a.partials[blockIdx.x] = total;
}

/* Grid of the dot over n elements (level1.hip dotToDevice): the fused kernels must use the same one. */
template <typename T> static long long dotBlocks(int n, bool wide)
{
    constexpr int WIDE = 16 / (int)sizeof(T);
    const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
    long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
    return blocks > SPGPU_REDUCE_MAX_BLOCKS ? SPGPU_REDUCE_MAX_BLOCKS : blocks;
}

static bool aligned(const void* p, size_t bytes) { return (uintptr_t)p % bytes == 0; }

template <typename T>
static void hellSpmvDot(spgpuHandle_t handle, T* result, const T* w, T* z, const T* y, T alpha, const T* cM, const int* rP,
                        int hackSize, const int* hackOffsets, const int* rS, int rows, const T* x, T beta, int baseIndex)
{
    constexpr int WIDE = 16 / (int)sizeof(T);
    hipStream_t s = handle->currentStream;
    FusedSpmvArgs<T> a;
    a.partials = static_cast<T*>(spgpuPrivate(handle)->reduceScratch);
    a.z = z;
    a.y = y;
    a.cM = cM;
    a.rP = rP;
    a.hackOffsets = hackOffsets;
    a.rS = rS;
    a.x = x;
    a.w = w ? w : x;
    a.alpha = alpha;
    a.beta = beta;
    a.hackSize = hackSize;
    a.rows = rows;
    a.baseIndex = baseIndex;
    long long blocks = 0;
    if (rows > 0) {
        const bool hasBeta = isNotZero(beta);
        /* the dot's own choice (its operands are w and z) */
        const bool wide = aligned(a.w, 16) && aligned(z, 16);
        const bool packed = wide && hackSize % WIDE == 0 && aligned(cM, 16) && aligned(rP, 4 * WIDE) && aligned(rS, 4 * WIDE);
        blocks = dotBlocks<T>(rows, wide);
        const dim3 grid((unsigned)blocks), block(kL1Threads);
#define SPGPU_FUSED_GO(VEC, PACKED)                                                                          \
    do {                                                                                                     \
        if (hasBeta)                                                                                         \
            hipLaunchKernelGGL((hellSpmvDotKernel<T, VEC, PACKED, true>), grid, block, 0, s, a);             \
        else                                                                                                 \
            hipLaunchKernelGGL((hellSpmvDotKernel<T, VEC, PACKED, false>), grid, block, 0, s, a);            \
    } while (0)
        if (packed)
            SPGPU_FUSED_GO(WIDE, true);
        else if (wide)
            SPGPU_FUSED_GO(WIDE, false);
        else
            SPGPU_FUSED_GO(1, false);
#undef SPGPU_FUSED_GO
    }
    hipLaunchKernelGGL((reduceFinalKernel<T, kDot>), dim3(1), dim3(kWave), 0, s, result, a.partials, (int)blocks);
    spgpuDebugCheck(handle, "hellspmvDotDevice");
}

/* z1 = y1 + a*x1, z2 = y2 - a*x2, partial sums of z2 . z2; a = *alphaNum / *alphaDen.
 * Arithmetic of axpbyDeviceKernel with beta = 1 (level1.hip): fma(a, x, 1*y) and fma(-a, x, 1*y). */
template <typename T, int VEC>
__global__ __launch_bounds__(kL1Threads) void axpbyPairDotKernel(T* partials, int n, T* z1, const T* y1, const T* x1, T* z2,
                                                                const T* y2, const T* x2, const T* alphaNum,
                                                                const T* alphaDen)
{
    __shared__ T lds[kL1Threads / kWave];
    const T up = quotientAt(alphaNum, alphaDen), down = -up, one = T(1);
    T acc = zeroOf<T>();
    const long long packs = n / VEC;
    constexpr long long TILE = (long long)kL1Threads * kL1Unroll;
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        Pack<T, VEC> a1[kL1Unroll], b1[kL1Unroll], a2[kL1Unroll], b2[kL1Unroll];
        bool live[kL1Unroll];
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            live[u] = p < packs;
            if (live[u]) {
                a2[u] = loadPackElementAligned<T, VEC>(x2 + p * VEC);
                b2[u] = loadPackElementAligned<T, VEC>(y2 + p * VEC);
                a1[u] = loadPackElementAligned<T, VEC>(x1 + p * VEC);
                b1[u] = loadPackElementAligned<T, VEC>(y1 + p * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            if (live[u]) {
                const long long p = base + u * kL1Threads + threadIdx.x;
                Pack<T, VEC> o1, o2;
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    o2.v[t] = mulAdd(down, a2[u].v[t], one * b2[u].v[t]);
                    o1.v[t] = mulAdd(up, a1[u].v[t], one * b1[u].v[t]);
                    acc = mulAdd(o2.v[t], o2.v[t], acc);
                }
                storePack<T, VEC>(z2 + p * VEC, o2);
                storePackElementAligned<T, VEC>(z1 + p * VEC, o1);
            }
        }
    }
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < n) {
        const T o2 = mulAdd(down, x2[tail], one * y2[tail]);
        z1[tail] = mulAdd(up, x1[tail], one * y1[tail]);
        z2[tail] = o2;
        acc = mulAdd(o2, o2, acc);
    }
    const T total = blockCombine<kDot>(acc, lds);
    if (threadIdx.x == 0)
        partials[blockIdx.x] = total;
}

template <typename T>
static void axpbyPairDot(spgpuHandle_t handle, T* result, int n, T* z1, const T* y1, const T* x1, T* z2, const T* y2,
                         const T* x2, const T* alphaNum, const T* alphaDen)
{
    constexpr int WIDE = 16 / (int)sizeof(T);
    hipStream_t s = handle->currentStream;
    T* partials = static_cast<T*>(spgpuPrivate(handle)->reduceScratch);
    long long blocks = 0;
    if (n > 0) {
        const bool wide = aligned(z2, 16); /* the dot's own choice: both of its operands are z2 */
        blocks = dotBlocks<T>(n, wide);
        if (wide)
            hipLaunchKernelGGL((axpbyPairDotKernel<T, WIDE>), dim3((unsigned)blocks), dim3(kL1Threads), 0, s, partials, n, z1, y1,
                               x1, z2, y2, x2, alphaNum, alphaDen);
        else
            hipLaunchKernelGGL((axpbyPairDotKernel<T, 1>), dim3((unsigned)blocks), dim3(kL1Threads), 0, s, partials, n, z1, y1, x1,
                               z2, y2, x2, alphaNum, alphaDen);
    }
    hipLaunchKernelGGL((reduceFinalKernel<T, kDot>), dim3(1), dim3(kWave), 0, s, result, partials, (int)blocks);
    spgpuDebugCheck(handle, "axpbyPairDotDevice");
}

} // namespace spgpu

using namespace spgpu;

extern "C" {

void spgpuShellspmvDotDevice(spgpuHandle_t h, float* result, const float* w, float* z, const float* y, float alpha,
                             const float* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS, int rows,
                             const float* x, float beta, int baseIndex)
{ hellSpmvDot<float>(h, result, w, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rows, x, beta, baseIndex); }

void spgpuDhellspmvDotDevice(spgpuHandle_t h, double* result, const double* w, double* z, const double* y, double alpha,
                             const double* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS, int rows,
                             const double* x, double beta, int baseIndex)
{ hellSpmvDot<double>(h, result, w, z, y, alpha, cM, rP, hackSize, hackOffsets, rS, rows, x, beta, baseIndex); }

void spgpuSaxpbyPairDotDevice(spgpuHandle_t h, float* result, int n, float* z1, const float* y1, const float* x1, float* z2,
                              const float* y2, const float* x2, const float* alphaNum, const float* alphaDen)
{ axpbyPairDot<float>(h, result, n, z1, y1, x1, z2, y2, x2, alphaNum, alphaDen); }

void spgpuDaxpbyPairDotDevice(spgpuHandle_t h, double* result, int n, double* z1, const double* y1, const double* x1,
                              double* z2, const double* y2, const double* x2, const double* alphaNum, const double* alphaDen)
{ axpbyPairDot<double>(h, result, n, z1, y1, x1, z2, y2, x2, alphaNum, alphaDen); }

} // extern "C"

