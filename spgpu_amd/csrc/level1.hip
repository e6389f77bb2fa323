/*
 * Level-1 operations next to SpMV on the hot path, for gfx950 (MI355X).
 *
 * C ABI: spgpu{S,D,C,Z}axpby / maxpby / dot / mdot / nrm2 / mnrm2 (hot path) and the rest of
 *        the reference's vector.h: scal, abs, axy, axypbz (+m forms), gath, scat, setscal (+I),
 *        asum, amax (+m forms)
 *        (include/spgpu/vector.h; reference vector.h, kernels/{s,d,c,z}axpby.cu, {s,d,c,z}dot.cu,
 *         {s,d,c,z}nrm2.cu, scal_base.cuh, abs_base.cuh, axy_base.cuh, gath_base.cuh, scat_base.cuh,
 *         setscal_base.cuh, asum_base.cuh, amax_base.cuh).
 *
 * axpby is a pure stream (2 reads + 1 write per element, 1 read when beta==0):
 * 16-byte accesses per lane, a capped grid with a grid-stride loop, one launch
 * for a whole multivector (grid.y = vector index).
 * dot / nrm2 are two-stage: every workgroup reduces a grid-stride slice with
 * lane-xor shuffles and one LDS hop between its 4 wavefronts, writes one
 * partial into scratch owned by the HANDLE, and the host combines the partials (reduce.hip.h:
 * finalOrder) after one async copy + stream sync (the reference copies from a
 * process-global __device__ array, kernels/ddot.cu:35,139).
 *
 * Roofline: HBM.  Algorithmic bytes: axpby n*sizeof(T)*(2 + [beta != 0]);
 * dot 2*n*sizeof(T); nrm2 n*sizeof(T).
 */
#include "reduce.hip.h"

#include "spgpu/vector.h"
#include "spgpu/device_scalars.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace spgpu {

constexpr int kL1MaxBlocksDefault = 16384; /* measured: 2 048 -> 64.7 %, 16 384 -> 71 % of 8 TB/s for axpby (tile-stride loop beyond) */
static int l1MaxBlocks()
{
    const int asked = spgpuTuning()->l1Blocks; /* experiments, include/spgpu/tuning.h */
    return asked >= 1 ? asked : kL1MaxBlocksDefault;
}

/* ---- axpby ---------------------------------------------------------------
 * Expression trees (reference): S/D  alpha*x + beta*y   (daxpby.cu:40-43)
 *                               C    fma(beta, y, alpha*x)  (caxpby.cu:44)
 *                               Z    fma(alpha, x, beta*y)  (zaxpby.cu:45) */
__device__ inline float axpbyOne(float alpha, float x, float beta, float y) { return mulAdd(alpha, x, beta * y); }
__device__ inline double axpbyOne(double alpha, double x, double beta, double y) { return mulAdd(alpha, x, beta * y); }
__device__ inline cfloat axpbyOne(cfloat alpha, cfloat x, cfloat beta, cfloat y) { return mulAdd(beta, y, mul(alpha, x)); }
__device__ inline cdouble axpbyOne(cdouble alpha, cdouble x, cdouble beta, cdouble y) { return mulAdd(alpha, x, mul(beta, y)); }


template <typename T, int VEC, bool HAS_BETA, bool NT = false>
__device__ inline void axpbyBody(T* z, int n, T beta, const T* y, T alpha, const T* x, long long pitch)
{
    const long long shift = (long long)blockIdx.y * pitch;
    z += shift;
    x += shift;
    if constexpr (HAS_BETA)
        y += shift;

    const long long packs = n / VEC;
    constexpr long long TILE = (long long)kL1Threads * kL1Unroll;
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        Pack<T, VEC> xv[kL1Unroll], yv[kL1Unroll];
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            if (p < packs) {
                xv[u] = loadPack<NT, T, VEC>(x + p * VEC);
                if constexpr (HAS_BETA)
                    yv[u] = loadPack<NT, T, VEC>(y + p * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            if (p < packs) {
                Pack<T, VEC> out;
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    if constexpr (HAS_BETA)
                        out.v[t] = axpbyOne(alpha, xv[u].v[t], beta, yv[u].v[t]);
                    else
                        out.v[t] = mul(alpha, xv[u].v[t]);
                }
                storePackMaybeNT<NT, T, VEC>(z + p * VEC, out);
            }
        }
    }
    /* tail (n % VEC elements) */
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < n) {
        if constexpr (HAS_BETA)
            z[tail] = axpbyOne(alpha, x[tail], beta, y[tail]);
        else
            z[tail] = mul(alpha, x[tail]);
    }
}

template <typename T, int VEC, bool HAS_BETA, bool NT = false>
__global__ __launch_bounds__(kL1Threads) void axpbyKernel(T* z, int n, T beta, const T* y, T alpha, const T* x,
                                                         long long pitch)
{
    axpbyBody<T, VEC, HAS_BETA, NT>(z, n, beta, y, alpha, x, pitch);
}

/* scalars from device memory (include/spgpu/device_scalars.h): the same two bodies, chosen by the value of beta.
 * A coefficient is num/den (NULL = 1); hasBeta == 0: no y at all. */

template <typename T, int VEC>
__global__ __launch_bounds__(kL1Threads) void axpbyDeviceKernel(T* z, int n, int hasBeta, const T* betaNum, const T* betaDen,
                                                               const T* y, const T* alphaNum, const T* alphaDen,
                                                               int negateAlpha, const T* x)
{
    T alpha = quotientAt(alphaNum, alphaDen);
    if (negateAlpha)
        alpha = -alpha;
    const T beta = hasBeta ? quotientAt(betaNum, betaDen) : zeroOf<T>();
    if (isNotZero(beta))
        axpbyBody<T, VEC, true>(z, n, beta, y, alpha, x, 0);
    else
        axpbyBody<T, VEC, false>(z, n, beta, y, alpha, x, 0);
}

template <typename T>
__global__ void divDeviceKernel(T* out, const T* num, const T* den, int negate)
{
    const T q = *num / *den;
    *out = negate ? -q : q;
}

template <typename T, typename ApiT>
static void axpbyLaunch(spgpuHandle_t handle, ApiT* zApi, int n, ApiT betaApi, ApiT* yApi, ApiT alphaApi,
                        ApiT* xApi, int count, int pitch)
{
    static_assert(sizeof(T) == sizeof(ApiT), "ABI type and device type must have one layout");
    if (n <= 0 || count <= 0)
        return;
    T *z = reinterpret_cast<T*>(zApi), *y = reinterpret_cast<T*>(yApi), *x = reinterpret_cast<T*>(xApi);
    T alpha, beta;
    __builtin_memcpy(&alpha, &alphaApi, sizeof(T));
    __builtin_memcpy(&beta, &betaApi, sizeof(T));
    const bool hasBeta = isNotZero(beta);

    constexpr int WIDE = 16 / (int)sizeof(T);
    const bool wide = WIDE > 1 && ((uintptr_t)z % 16 == 0) && ((uintptr_t)x % 16 == 0) &&
                      (!hasBeta || (uintptr_t)y % 16 == 0) && (count == 1 || pitch % WIDE == 0);
    const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
    long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
    const long long cap = l1MaxBlocks() / (count < l1MaxBlocks() ? count : l1MaxBlocks());
    if (blocks > (cap > 1 ? cap : 1))
        blocks = cap > 1 ? cap : 1;
    const dim3 grid((unsigned)blocks, (unsigned)count);
    hipStream_t s = handle->currentStream;
    /* Vectors larger than the 256 MiB Infinity Cache cannot be found there again by the next kernel: stream them
     * with the non-temporal hint (measured, n = 1e8: 70-71 % -> 75.5-77 % of the HBM peak, profiles/r01d_level1_nt.txt).
     * Smaller ones -- the vectors of a solver iteration -- stay cached.  SPGPU_L1_NT = 0 / 1 forces the choice. */
    const long long streamed = (long long)n * (long long)sizeof(T) * count * (2 + (hasBeta ? 1 : 0));
    const int ntKnob = spgpuTuning()->l1Nt;
    const bool nt = (ntKnob < 0 ? streamed >= (256ll << 20) : ntKnob != 0); /* exact aliasing of z is fine: a lane reads its elements before it writes them */

#define SPGPU_AXPBY_GO(VEC)                                                                               \
    do {                                                                                                  \
        if (hasBeta && nt)                                                                                \
            hipLaunchKernelGGL((axpbyKernel<T, VEC, true, true>), grid, dim3(kL1Threads), 0, s, z, n, beta, y,   \
                               alpha, x, (long long)pitch);                                               \
        else if (hasBeta)                                                                                 \
            hipLaunchKernelGGL((axpbyKernel<T, VEC, true>), grid, dim3(kL1Threads), 0, s, z, n, beta, y,   \
                               alpha, x, (long long)pitch);                                               \
        else if (nt)                                                                                      \
            hipLaunchKernelGGL((axpbyKernel<T, VEC, false, true>), grid, dim3(kL1Threads), 0, s, z, n, beta, y,  \
                               alpha, x, (long long)pitch);                                               \
        else                                                                                              \
            hipLaunchKernelGGL((axpbyKernel<T, VEC, false>), grid, dim3(kL1Threads), 0, s, z, n, beta, y,  \
                               alpha, x, (long long)pitch);                                               \
    } while (0)

    if (wide)
        SPGPU_AXPBY_GO(WIDE);
    else
        SPGPU_AXPBY_GO(1);
#undef SPGPU_AXPBY_GO
    spgpuDebugCheck(handle, "axpby");
}

/* ---- reductions ------------------------------------------------------------ */

/* DOT : a[i]*b[i] accumulated with the SpMV multiply-add (un-conjugated, zdot.cu:54)
 * NRM2: |a[i]|^2 accumulated in the real type (dnrm2.cu:52-53)
 * ASUM: |a[i]| added; AMAX: max |a[i]|  (|.| of a complex value as cuCabs) */

template <typename T> struct RealOf { using type = T; };
template <typename R> struct RealOf<Cx<R>> { using type = R; };

template <typename R> __device__ inline R absSqAdd(R v, R acc) { return mulAdd(v, v, acc); }
template <typename R> __device__ inline R absSqAdd(Cx<R> v, R acc) { return mulAdd(v.y, v.y, mulAdd(v.x, v.x, acc)); }

/* cuCabs / cuCabsf: v*sqrt(1 + (w/v)^2) with v = max(|re|,|im|), w = min; 1 + t*t is one fma. */
__device__ inline float magnitude(float v) { return __builtin_fabsf(v); }
__device__ inline double magnitude(double v) { return __builtin_fabs(v); }
template <typename R> __device__ inline R magnitude(Cx<R> z)
{
    const R a = magnitude(z.x), b = magnitude(z.y);
    const R v = a > b ? a : b, w = a > b ? b : a;
    R t = w / v;
    t = mulAdd(t, t, R(1));
    t = v * (sizeof(R) == 4 ? (R)__builtin_sqrtf((float)t) : (R)__builtin_sqrt((double)t));
    const R huge = sizeof(R) == 4 ? (R)3.402823466e38f : (R)1.79769313486231570e+308;
    return (v == R(0) || v > huge || w > huge) ? v + w : t;
}


/* What a reduction returns when the device could not deliver its partials. */
template <typename A> static inline A notANumber();
template <> inline float notANumber<float>() { return NAN; }
template <> inline double notANumber<double>() { return NAN; }
template <> inline cfloat notANumber<cfloat>() { return cfloat{NAN, NAN}; }
template <> inline cdouble notANumber<cdouble>() { return cdouble{NAN, NAN}; }

template <typename T, int MODE> struct AccOf { using type = typename RealOf<T>::type; };
template <typename T> struct AccOf<T, kDot> { using type = T; };

template <int MODE, typename T, typename Acc> __device__ inline Acc accumulate(T a, T b, Acc acc)
{
    if constexpr (MODE == kDot)
        return mulAdd(a, b, acc);
    else if constexpr (MODE == kNrm2)
        return absSqAdd(a, acc);
    else if constexpr (MODE == kAsum)
        return acc + magnitude(a);
    else {
        const Acc m = magnitude(a);
        return m > acc ? m : acc;
    }
}

template <typename T, int VEC, int MODE, bool NT = false>
__global__ __launch_bounds__(kL1Threads) void reduceKernel(typename AccOf<T, MODE>::type* partials, int n, const T* a,
                                                          const T* b, long long pitch)
{
    using Acc = typename AccOf<T, MODE>::type;
    __shared__ Acc lds[kL1Threads / kWave];

    const long long shift = (long long)blockIdx.y * pitch;
    a += shift;
    if constexpr (MODE == kDot)
        b += shift;

    Acc acc = zeroOf<Acc>();
    const long long packs = n / VEC;
    constexpr long long TILE = (long long)kL1Threads * kL1Unroll;
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        Pack<T, VEC> av[kL1Unroll], bv[kL1Unroll];
        bool live[kL1Unroll];
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            live[u] = p < packs;
            if (live[u]) {
                av[u] = loadPack<NT, T, VEC>(a + p * VEC);
                if constexpr (MODE == kDot)
                    bv[u] = loadPack<NT, T, VEC>(b + p * VEC);
                else
                    bv[u] = av[u];
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            if (live[u]) {
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    acc = accumulate<MODE>(av[u].v[t], bv[u].v[t], acc);
            }
        }
    }
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < n)
        acc = accumulate<MODE>(a[tail], MODE == kDot ? b[tail] : a[tail], acc);

    const Acc total = blockCombine<MODE>(acc, lds);
    if (threadIdx.x == 0)
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total;

}

/* Runs the two-stage reduction for `count` vectors and leaves one host value
 * per vector in out[].  Synchronises handle->currentStream. */
template <typename T, int MODE>
static void reduceVectors(spgpuHandle_t handle, typename AccOf<T, MODE>::type* out, int n, const T* a, const T* b,
                          int count, int pitch)
{
    using Acc = typename AccOf<T, MODE>::type;
    for (int j = 0; j < count; ++j)
        out[j] = zeroOf<Acc>();
    if (n <= 0 || count <= 0)
        return;

    SpgpuPrivateHandle* priv = spgpuPrivate(handle);
    hipStream_t s = handle->currentStream;
    Acc* dev = static_cast<Acc*>(priv->reduceScratch);
    Acc* host = static_cast<Acc*>(priv->reduceHost);

    constexpr int WIDE = 16 / (int)sizeof(T);
    const int maxVectorsPerPass = SPGPU_REDUCE_MAX_BLOCKS;

    for (int first = 0; first < count; first += maxVectorsPerPass) {
        const int vectors = count - first < maxVectorsPerPass ? count - first : maxVectorsPerPass;
        const T* a0 = a + (size_t)first * pitch;
        const T* b0 = MODE == kDot ? b + (size_t)first * pitch : nullptr;
        const bool wide = WIDE > 1 && ((uintptr_t)a0 % 16 == 0) && (MODE != kDot || (uintptr_t)b0 % 16 == 0) &&
                          (vectors == 1 || pitch % WIDE == 0);
        const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
        long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
        const long long cap = SPGPU_REDUCE_MAX_BLOCKS / vectors;
        if (blocks > cap)
            blocks = cap;
        const dim3 grid((unsigned)blocks, (unsigned)vectors);

        const int ntKnob = spgpuTuning()->l1Nt;
        const long long streamed = (long long)n * (long long)sizeof(T) * vectors * (MODE == kDot ? 2 : 1);
        if (wide && (ntKnob < 0 ? streamed >= (256ll << 20) : ntKnob != 0))
            hipLaunchKernelGGL((reduceKernel<T, WIDE, MODE, true>), grid, dim3(kL1Threads), 0, s, dev, n, a0, b0,
                               (long long)pitch);
        else if (wide)
            hipLaunchKernelGGL((reduceKernel<T, WIDE, MODE>), grid, dim3(kL1Threads), 0, s, dev, n, a0, b0,
                               (long long)pitch);
        else
            hipLaunchKernelGGL((reduceKernel<T, 1, MODE>), grid, dim3(kL1Threads), 0, s, dev, n, a0, b0,
                               (long long)pitch);
        hipError_t status = hipMemcpyAsync(host, dev, sizeof(Acc) * (size_t)blocks * vectors, hipMemcpyDeviceToHost, s);
        if (status == hipSuccess)
            status = hipStreamSynchronize(s);
        if (status != hipSuccess) {
            /* the pinned mirror holds whatever an earlier call left there: do not hand that out as a result */
            for (int j = first; j < count; ++j)
                out[j] = notANumber<Acc>();
            fprintf(stderr, "spgpu: reduction failed: %s\n", hipGetErrorString(status));
            return;
        }

        for (int j = 0; j < vectors; ++j) {
            out[first + j] = finalOrder<MODE>(host + (size_t)j * blocks, blocks);
        }
    }
    spgpuDebugCheck(handle, "reduction");
}


/* dot with the result left in device memory: same first stage and same grid as reduceVectors for one vector;
 * no copy, no synchronisation (capturable in a graph). */
template <typename T>
static void dotToDevice(spgpuHandle_t handle, T* result, int n, const T* a, const T* b)
{
    hipStream_t s = handle->currentStream;
    T* dev = static_cast<T*>(spgpuPrivate(handle)->reduceScratch);
    long long blocks = 0;
    if (n > 0) {
        constexpr int WIDE = 16 / (int)sizeof(T);
        const bool wide = WIDE > 1 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
        const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
        blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
        if (blocks > SPGPU_REDUCE_MAX_BLOCKS)
            blocks = SPGPU_REDUCE_MAX_BLOCKS;
        const dim3 grid((unsigned)blocks, 1);
        if (wide)
            hipLaunchKernelGGL((reduceKernel<T, WIDE, kDot>), grid, dim3(kL1Threads), 0, s, dev, n, a, b, 0ll);
        else
            hipLaunchKernelGGL((reduceKernel<T, 1, kDot>), grid, dim3(kL1Threads), 0, s, dev, n, a, b, 0ll);
    }
    hipLaunchKernelGGL((reduceFinalKernel<T, kDot>), dim3(1), dim3(kWave), 0, s, result, dev, (int)blocks);
    spgpuDebugCheck(handle, "dotDevice");
}

/* nrm2 with the result left in device memory: first stage and grid of spgpu?nrm2, the square root taken by the final
 * stage's lane. */
template <typename T>
static void nrm2ToDevice(spgpuHandle_t handle, T* result, int n, const T* a)
{
    hipStream_t s = handle->currentStream;
    T* dev = static_cast<T*>(spgpuPrivate(handle)->reduceScratch);
    long long blocks = 0;
    if (n > 0) {
        constexpr int WIDE = 16 / (int)sizeof(T);
        const bool wide = WIDE > 1 && ((uintptr_t)a % 16 == 0);
        const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
        blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
        if (blocks > SPGPU_REDUCE_MAX_BLOCKS)
            blocks = SPGPU_REDUCE_MAX_BLOCKS;
        const dim3 grid((unsigned)blocks, 1);
        if (wide)
            hipLaunchKernelGGL((reduceKernel<T, WIDE, kNrm2>), grid, dim3(kL1Threads), 0, s, dev, n, a, (const T*)nullptr, 0ll);
        else
            hipLaunchKernelGGL((reduceKernel<T, 1, kNrm2>), grid, dim3(kL1Threads), 0, s, dev, n, a, (const T*)nullptr, 0ll);
    }
    hipLaunchKernelGGL((reduceFinalKernel<T, kNrm2, true>), dim3(1), dim3(kWave), 0, s, result, dev, (int)blocks);
    spgpuDebugCheck(handle, "nrm2Device");
}

template <typename T>
static void axpbyFromDevice(spgpuHandle_t handle, T* z, int n, int hasBeta, const T* betaNum, const T* betaDen, const T* y,
                            const T* alphaNum, const T* alphaDen, int negateAlpha, const T* x)
{
    if (n <= 0)
        return;
    constexpr int WIDE = 16 / (int)sizeof(T);
    /* y may be ignored at run time (beta == 0); its alignment is required only when it is given */
    const bool wide = ((uintptr_t)z % 16 == 0) && ((uintptr_t)x % 16 == 0) && (!hasBeta || !y || (uintptr_t)y % 16 == 0);
    const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
    long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
    if (blocks > l1MaxBlocks())
        blocks = l1MaxBlocks();
    hipStream_t s = handle->currentStream;
    if (wide)
        hipLaunchKernelGGL((axpbyDeviceKernel<T, WIDE>), dim3((unsigned)blocks), dim3(kL1Threads), 0, s, z, n, hasBeta, betaNum,
                           betaDen, y, alphaNum, alphaDen, negateAlpha, x);
    else
        hipLaunchKernelGGL((axpbyDeviceKernel<T, 1>), dim3((unsigned)blocks), dim3(kL1Threads), 0, s, z, n, hasBeta, betaNum,
                           betaDen, y, alphaNum, alphaDen, negateAlpha, x);
    spgpuDebugCheck(handle, "axpbyDevice");
}

/* ---- element-wise maps: scal, abs, axy, axypbz -------------------------------
 * One kernel shape (16-byte accesses, kL1Unroll in flight per lane, tile-stride loop,
 * grid.y = vector of a multivector); the operation is a compile-time tag.
 *   kScal   : out = alpha * x                               (scal_base.cuh:34-45)
 *   kAbs    : out = alpha * |x|   (complex: alpha * (|x|,0); alpha == 1 -> |x|)   (abs_base.cuh:43-70)
 *   kAxy    : out = alpha * (x*y)                           (axy_base.cuh:37-47)
 *   kAxypbz : out = fma(alpha, x*y, beta*z)                 (axy_base.cuh:95-108) */
enum MapOp { kScal = 0, kAbs = 1, kAxy = 2, kAxypbz = 3 };

__device__ inline float fromMagnitude(float m, float) { return m; }
__device__ inline double fromMagnitude(double m, double) { return m; }
template <typename R> __device__ inline Cx<R> fromMagnitude(R m, Cx<R>) { return Cx<R>{m, R(0)}; }

template <int OP, typename T> __device__ inline T mapOne(T alpha, T beta, T x, T y, T z, bool alphaIsOne)
{
    if constexpr (OP == kScal)
        return mul(alpha, x);
    else if constexpr (OP == kAbs) {
        const T m = fromMagnitude(magnitude(x), x);
        return alphaIsOne ? m : mul(alpha, m);
    } else if constexpr (OP == kAxy)
        return mul(alpha, mul(x, y));
    else
        return mulAdd(alpha, mul(x, y), mul(beta, z));
}

template <typename T, int VEC, int OP, bool NT = false>
__global__ __launch_bounds__(kL1Threads) void mapKernel(T* out, int n, T alpha, T beta, const T* x, const T* y, const T* z,
                                                       long long pitch, int alphaIsOne)
{
    const long long shift = (long long)blockIdx.y * pitch;
    out += shift;
    x += shift;
    if constexpr (OP >= kAxy)
        y += shift;
    if constexpr (OP == kAxypbz)
        z += shift;

    const long long packs = n / VEC;
    constexpr long long TILE = (long long)kL1Threads * kL1Unroll;
    for (long long base = (long long)blockIdx.x * TILE; base < packs; base += (long long)gridDim.x * TILE) {
        Pack<T, VEC> xv[kL1Unroll], yv[kL1Unroll], zv[kL1Unroll];
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            if (p < packs) {
                xv[u] = loadPack<NT, T, VEC>(x + p * VEC);
                if constexpr (OP >= kAxy)
                    yv[u] = loadPack<NT, T, VEC>(y + p * VEC);
                else
                    yv[u] = xv[u];
                if constexpr (OP == kAxypbz)
                    zv[u] = loadPack<NT, T, VEC>(z + p * VEC);
                else
                    zv[u] = xv[u];
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            const long long p = base + u * kL1Threads + threadIdx.x;
            if (p < packs) {
                Pack<T, VEC> o;
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    o.v[t] = mapOne<OP>(alpha, beta, xv[u].v[t], yv[u].v[t], zv[u].v[t], alphaIsOne != 0);
                storePackMaybeNT<NT, T, VEC>(out + p * VEC, o);
            }
        }
    }
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < n)
        out[tail] = mapOne<OP>(alpha, beta, x[tail], OP >= kAxy ? y[tail] : x[tail], OP == kAxypbz ? z[tail] : x[tail],
                               alphaIsOne != 0);
}

template <typename T, int OP, typename ApiT>
static void mapLaunch(spgpuHandle_t handle, ApiT* outApi, int n, ApiT alphaApi, ApiT betaApi, const ApiT* xApi,
                      const ApiT* yApi, const ApiT* zApi, int count, int pitch)
{
    static_assert(sizeof(T) == sizeof(ApiT), "ABI type and device type must have one layout");
    if (n <= 0 || count <= 0)
        return;
    T* out = reinterpret_cast<T*>(outApi);
    const T *x = reinterpret_cast<const T*>(xApi), *y = reinterpret_cast<const T*>(yApi), *z = reinterpret_cast<const T*>(zApi);
    T alpha, beta, one = zeroOf<T>();
    __builtin_memcpy(&alpha, &alphaApi, sizeof(T));
    __builtin_memcpy(&beta, &betaApi, sizeof(T));
    using R = typename RealOf<T>::type;
    const R unit = R(1);
    __builtin_memcpy(&one, &unit, sizeof(R)); /* (1) or (1, 0) */
    const int alphaIsOne = __builtin_memcmp(&alpha, &one, sizeof(T)) == 0;

    constexpr int WIDE = 16 / (int)sizeof(T);
    const bool wide = WIDE > 1 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)x % 16 == 0) &&
                      (OP < kAxy || (uintptr_t)y % 16 == 0) && (OP != kAxypbz || (uintptr_t)z % 16 == 0) &&
                      (count == 1 || pitch % WIDE == 0);
    const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
    long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
    const long long cap = l1MaxBlocks() / (count < l1MaxBlocks() ? count : l1MaxBlocks());
    if (blocks > (cap > 1 ? cap : 1))
        blocks = cap > 1 ? cap : 1;
    const dim3 grid((unsigned)blocks, (unsigned)count);
    hipStream_t s = handle->currentStream;
    /* same rule as axpby: streams beyond the Infinity Cache go non-temporal */
    const int ntKnob = spgpuTuning()->l1Nt;
    const long long streamed = (long long)n * (long long)sizeof(T) * count * (OP == kAxypbz ? 4 : OP == kAxy ? 3 : 2);
    if (wide && (ntKnob < 0 ? streamed >= (256ll << 20) : ntKnob != 0))
        hipLaunchKernelGGL((mapKernel<T, WIDE, OP, true>), grid, dim3(kL1Threads), 0, s, out, n, alpha, beta, x, y, z,
                           (long long)pitch, alphaIsOne);
    else if (wide)
        hipLaunchKernelGGL((mapKernel<T, WIDE, OP>), grid, dim3(kL1Threads), 0, s, out, n, alpha, beta, x, y, z,
                           (long long)pitch, alphaIsOne);
    else
        hipLaunchKernelGGL((mapKernel<T, 1, OP>), grid, dim3(kL1Threads), 0, s, out, n, alpha, beta, x, y, z,
                           (long long)pitch, alphaIsOne);
    spgpuDebugCheck(handle, "level-1 map");
}

/* axypbz dispatch of the reference (axy_base.cuh:139-176): alpha == 0 -> scal(beta, z); beta == 0 -> axy. */
template <typename T, typename ApiT>
static void axypbz(spgpuHandle_t h, ApiT* w, int n, ApiT beta, ApiT* z, ApiT alpha, ApiT* x, ApiT* y, int count, int pitch)
{
    T a, b;
    __builtin_memcpy(&a, &alpha, sizeof(T));
    __builtin_memcpy(&b, &beta, sizeof(T));
    if (!isNotZero(a))
        mapLaunch<T, kScal>(h, w, n, beta, beta, z, z, z, count, pitch);
    else if (!isNotZero(b))
        mapLaunch<T, kAxy>(h, w, n, alpha, alpha, x, y, y, count, pitch);
    else
        mapLaunch<T, kAxypbz>(h, w, n, alpha, beta, x, y, z, count, pitch);
}

/* ---- gather / scatter / fill ---------------------------------------------------- */
template <typename T> __device__ inline T scatCombine(T beta, T y, T v) { return mulAdd(beta, y, v); }
template <> __device__ inline int scatCombine<int>(int beta, int y, int v) { return beta * y + v; }
__device__ inline bool isNotZero(int a) { return a != 0; }

template <typename T>
__global__ __launch_bounds__(kL1Threads) void gathKernel(T* values, int count, const int* indices, int firstIndex, const T* vector)
{
    const long long stride = (long long)gridDim.x * kL1Threads;
    for (long long i = (long long)blockIdx.x * kL1Threads + threadIdx.x; i < count; i += stride) {
        const int pos = indices[i] - firstIndex;
        if (pos >= 0)
            values[i] = vector[pos];
    }
}

template <typename T>
__global__ __launch_bounds__(kL1Threads) void scatKernel(T* vector, int count, const int* indices, const T* values, int firstIndex, T beta)
{
    const bool hasBeta = isNotZero(beta);
    const long long stride = (long long)gridDim.x * kL1Threads;
    for (long long i = (long long)blockIdx.x * kL1Threads + threadIdx.x; i < count; i += stride) {
        const int pos = indices[i] - firstIndex;
        if (pos >= 0)
            vector[pos] = hasBeta ? scatCombine(beta, vector[pos], values[i]) : values[i];
    }
}

template <typename T> __global__ __launch_bounds__(kL1Threads) void fillKernel(T* vector, long long count, T value)
{
    const long long stride = (long long)gridDim.x * kL1Threads;
    for (long long i = (long long)blockIdx.x * kL1Threads + threadIdx.x; i < count; i += stride)
        vector[i] = value;
}

static unsigned sparseGrid(long long count)
{
    long long blocks = (count + kL1Threads - 1) / kL1Threads;
    return (unsigned)(blocks > 4 * l1MaxBlocks() ? 4 * l1MaxBlocks() : (blocks < 1 ? 1 : blocks));
}

template <typename T, typename ApiT>
static void gath(spgpuHandle_t h, ApiT* xValues, int xNnz, const int* xIndices, int xBaseIndex, const ApiT* y)
{
    if (xNnz <= 0)
        return;
    hipLaunchKernelGGL((gathKernel<T>), dim3(sparseGrid(xNnz)), dim3(kL1Threads), 0, h->currentStream,
                       reinterpret_cast<T*>(xValues), xNnz, xIndices, xBaseIndex, reinterpret_cast<const T*>(y));
    spgpuDebugCheck(h, "gath");
}

template <typename T, typename ApiT>
static void scat(spgpuHandle_t h, ApiT* y, int xNnz, const ApiT* xValues, const int* xIndices, int xBaseIndex, ApiT betaApi)
{
    if (xNnz <= 0)
        return;
    T beta;
    __builtin_memcpy(&beta, &betaApi, sizeof(T));
    hipLaunchKernelGGL((scatKernel<T>), dim3(sparseGrid(xNnz)), dim3(kL1Threads), 0, h->currentStream,
                       reinterpret_cast<T*>(y), xNnz, xIndices, reinterpret_cast<const T*>(xValues), xBaseIndex, beta);
    spgpuDebugCheck(h, "scat");
}

template <typename T, typename ApiT>
static void setscal(spgpuHandle_t h, int first, int last, int baseIndex, ApiT valApi, ApiT* y)
{
    const long long n = (long long)last - first + 1;
    if (n <= 0)
        return;
    T val;
    __builtin_memcpy(&val, &valApi, sizeof(T));
    hipLaunchKernelGGL((fillKernel<T>), dim3(sparseGrid(n)), dim3(kL1Threads), 0, h->currentStream,
                       reinterpret_cast<T*>(y) + (first - baseIndex), n, val);
    spgpuDebugCheck(h, "setscal");
}

} // namespace spgpu

using namespace spgpu;

#define SPGPU_CF(p) reinterpret_cast<cfloat*>(p)
#define SPGPU_CD(p) reinterpret_cast<cdouble*>(p)

extern "C" {

/* ---- axpby ---- */
void spgpuSaxpby(spgpuHandle_t h, float* z, int n, float beta, float* y, float alpha, float* x)
{ axpbyLaunch<float>(h, z, n, beta, y, alpha, x, 1, 0); }
void spgpuDaxpby(spgpuHandle_t h, double* z, int n, double beta, double* y, double alpha, double* x)
{ axpbyLaunch<double>(h, z, n, beta, y, alpha, x, 1, 0); }
void spgpuCaxpby(spgpuHandle_t h, hipFloatComplex* z, int n, hipFloatComplex beta, hipFloatComplex* y,
                 hipFloatComplex alpha, hipFloatComplex* x)
{ axpbyLaunch<cfloat>(h, z, n, beta, y, alpha, x, 1, 0); }
void spgpuZaxpby(spgpuHandle_t h, hipDoubleComplex* z, int n, hipDoubleComplex beta, hipDoubleComplex* y,
                 hipDoubleComplex alpha, hipDoubleComplex* x)
{ axpbyLaunch<cdouble>(h, z, n, beta, y, alpha, x, 1, 0); }

void spgpuSmaxpby(spgpuHandle_t h, float* z, int n, float beta, float* y, float alpha, float* x, int count, int pitch)
{ axpbyLaunch<float>(h, z, n, beta, y, alpha, x, count, pitch); }
void spgpuDmaxpby(spgpuHandle_t h, double* z, int n, double beta, double* y, double alpha, double* x, int count, int pitch)
{ axpbyLaunch<double>(h, z, n, beta, y, alpha, x, count, pitch); }
void spgpuCmaxpby(spgpuHandle_t h, hipFloatComplex* z, int n, hipFloatComplex beta, hipFloatComplex* y,
                  hipFloatComplex alpha, hipFloatComplex* x, int count, int pitch)
{ axpbyLaunch<cfloat>(h, z, n, beta, y, alpha, x, count, pitch); }
void spgpuZmaxpby(spgpuHandle_t h, hipDoubleComplex* z, int n, hipDoubleComplex beta, hipDoubleComplex* y,
                  hipDoubleComplex alpha, hipDoubleComplex* x, int count, int pitch)
{ axpbyLaunch<cdouble>(h, z, n, beta, y, alpha, x, count, pitch); }

/* ---- dot ---- */
float spgpuSdot(spgpuHandle_t h, int n, float* a, float* b)
{ float r; reduceVectors<float, kDot>(h, &r, n, a, b, 1, 0); return r; }
double spgpuDdot(spgpuHandle_t h, int n, double* a, double* b)
{ double r; reduceVectors<double, kDot>(h, &r, n, a, b, 1, 0); return r; }
hipFloatComplex spgpuCdot(spgpuHandle_t h, int n, hipFloatComplex* a, hipFloatComplex* b)
{
    cfloat r;
    reduceVectors<cfloat, kDot>(h, &r, n, SPGPU_CF(a), SPGPU_CF(b), 1, 0);
    return make_hipFloatComplex(r.x, r.y);
}
hipDoubleComplex spgpuZdot(spgpuHandle_t h, int n, hipDoubleComplex* a, hipDoubleComplex* b)
{
    cdouble r;
    reduceVectors<cdouble, kDot>(h, &r, n, SPGPU_CD(a), SPGPU_CD(b), 1, 0);
    return make_hipDoubleComplex(r.x, r.y);
}
void spgpuSmdot(spgpuHandle_t h, float* y, int n, float* a, float* b, int count, int pitch)
{ reduceVectors<float, kDot>(h, y, n, a, b, count, pitch); }
void spgpuDmdot(spgpuHandle_t h, double* y, int n, double* a, double* b, int count, int pitch)
{ reduceVectors<double, kDot>(h, y, n, a, b, count, pitch); }
void spgpuCmdot(spgpuHandle_t h, hipFloatComplex* y, int n, hipFloatComplex* a, hipFloatComplex* b, int count, int pitch)
{ reduceVectors<cfloat, kDot>(h, SPGPU_CF(y), n, SPGPU_CF(a), SPGPU_CF(b), count, pitch); }
void spgpuZmdot(spgpuHandle_t h, hipDoubleComplex* y, int n, hipDoubleComplex* a, hipDoubleComplex* b, int count, int pitch)
{ reduceVectors<cdouble, kDot>(h, SPGPU_CD(y), n, SPGPU_CD(a), SPGPU_CD(b), count, pitch); }

/* ---- nrm2 / asum / amax ---- */
#define SPGPU_REAL_REDUCTION(NAME, MODE, FINISH_F, FINISH_D)                                                              \
    float spgpuS##NAME(spgpuHandle_t h, int n, float* x)                                                                  \
    { float r; reduceVectors<float, MODE>(h, &r, n, x, (const float*)nullptr, 1, 0); return FINISH_F(r); }                \
    double spgpuD##NAME(spgpuHandle_t h, int n, double* x)                                                                \
    { double r; reduceVectors<double, MODE>(h, &r, n, x, (const double*)nullptr, 1, 0); return FINISH_D(r); }             \
    float spgpuC##NAME(spgpuHandle_t h, int n, hipFloatComplex* x)                                                        \
    { float r; reduceVectors<cfloat, MODE>(h, &r, n, SPGPU_CF(x), (const cfloat*)nullptr, 1, 0); return FINISH_F(r); }    \
    double spgpuZ##NAME(spgpuHandle_t h, int n, hipDoubleComplex* x)                                                      \
    { double r; reduceVectors<cdouble, MODE>(h, &r, n, SPGPU_CD(x), (const cdouble*)nullptr, 1, 0); return FINISH_D(r); } \
    void spgpuSm##NAME(spgpuHandle_t h, float* y, int n, float* x, int count, int pitch)                                  \
    { reduceVectors<float, MODE>(h, y, n, x, (const float*)nullptr, count, pitch); for (int j = 0; j < count; ++j) y[j] = FINISH_F(y[j]); } \
    void spgpuDm##NAME(spgpuHandle_t h, double* y, int n, double* x, int count, int pitch)                                \
    { reduceVectors<double, MODE>(h, y, n, x, (const double*)nullptr, count, pitch); for (int j = 0; j < count; ++j) y[j] = FINISH_D(y[j]); } \
    void spgpuCm##NAME(spgpuHandle_t h, float* y, int n, hipFloatComplex* x, int count, int pitch)                        \
    { reduceVectors<cfloat, MODE>(h, y, n, SPGPU_CF(x), (const cfloat*)nullptr, count, pitch); for (int j = 0; j < count; ++j) y[j] = FINISH_F(y[j]); } \
    void spgpuZm##NAME(spgpuHandle_t h, double* y, int n, hipDoubleComplex* x, int count, int pitch)                      \
    { reduceVectors<cdouble, MODE>(h, y, n, SPGPU_CD(x), (const cdouble*)nullptr, count, pitch); for (int j = 0; j < count; ++j) y[j] = FINISH_D(y[j]); }

#define SPGPU_SAME(v) (v)
SPGPU_REAL_REDUCTION(nrm2, kNrm2, sqrtf, sqrt)
SPGPU_REAL_REDUCTION(asum, kAsum, SPGPU_SAME, SPGPU_SAME)
SPGPU_REAL_REDUCTION(amax, kAmax, SPGPU_SAME, SPGPU_SAME)

/* ---- scal / abs / axy / axypbz ---- */
#define SPGPU_MAPS(L, T, ApiT)                                                                                            \
    void spgpu##L##scal(spgpuHandle_t h, ApiT* y, int n, ApiT alpha, ApiT* x)                                             \
    { mapLaunch<T, kScal>(h, y, n, alpha, alpha, x, x, x, 1, 0); }                                                        \
    void spgpu##L##abs(spgpuHandle_t h, ApiT* y, int n, ApiT alpha, ApiT* x)                                              \
    { mapLaunch<T, kAbs>(h, y, n, alpha, alpha, x, x, x, 1, 0); }                                                         \
    void spgpu##L##axy(spgpuHandle_t h, ApiT* z, int n, ApiT alpha, ApiT* x, ApiT* y)                                     \
    { mapLaunch<T, kAxy>(h, z, n, alpha, alpha, x, y, y, 1, 0); }                                                         \
    void spgpu##L##maxy(spgpuHandle_t h, ApiT* z, int n, ApiT alpha, ApiT* x, ApiT* y, int count, int pitch)              \
    { mapLaunch<T, kAxy>(h, z, n, alpha, alpha, x, y, y, count, pitch); }                                                 \
    void spgpu##L##axypbz(spgpuHandle_t h, ApiT* w, int n, ApiT beta, ApiT* z, ApiT alpha, ApiT* x, ApiT* y)              \
    { axypbz<T>(h, w, n, beta, z, alpha, x, y, 1, 0); }                                                                   \
    void spgpu##L##maxypbz(spgpuHandle_t h, ApiT* w, int n, ApiT beta, ApiT* z, ApiT alpha, ApiT* x, ApiT* y, int count,  \
                           int pitch)                                                                                     \
    { axypbz<T>(h, w, n, beta, z, alpha, x, y, count, pitch); }                                                           \
    void spgpu##L##gath(spgpuHandle_t h, ApiT* xValues, int xNnz, const int* xIndices, int xBaseIndex, const ApiT* y)     \
    { gath<T>(h, xValues, xNnz, xIndices, xBaseIndex, y); }                                                               \
    void spgpu##L##scat(spgpuHandle_t h, ApiT* y, int xNnz, const ApiT* xValues, const int* xIndices, int xBaseIndex,     \
                        ApiT beta)                                                                                        \
    { scat<T>(h, y, xNnz, xValues, xIndices, xBaseIndex, beta); }                                                         \
    void spgpu##L##setscal(spgpuHandle_t h, int first, int last, int baseIndex, ApiT val, ApiT* y)                        \
    { setscal<T>(h, first, last, baseIndex, val, y); }

SPGPU_MAPS(S, float, float)
SPGPU_MAPS(D, double, double)
SPGPU_MAPS(C, cfloat, hipFloatComplex)
SPGPU_MAPS(Z, cdouble, hipDoubleComplex)

void spgpuIgath(spgpuHandle_t h, int* xValues, int xNnz, const int* xIndices, int xBaseIndex, const int* y)
{ gath<int>(h, xValues, xNnz, xIndices, xBaseIndex, y); }
void spgpuIscat(spgpuHandle_t h, int* y, int xNnz, const int* xValues, const int* xIndices, int xBaseIndex, int beta)
{ scat<int>(h, y, xNnz, xValues, xIndices, xBaseIndex, beta); }
void spgpuIsetscal(spgpuHandle_t h, int first, int last, int baseIndex, int val, int* y)
{ setscal<int>(h, first, last, baseIndex, val, y); }


/* ---- include/spgpu/device_scalars.h ---- */
void spgpuSdotDevice(spgpuHandle_t h, float* result, int n, const float* a, const float* b) { dotToDevice<float>(h, result, n, a, b); }
void spgpuDdotDevice(spgpuHandle_t h, double* result, int n, const double* a, const double* b) { dotToDevice<double>(h, result, n, a, b); }
void spgpuSnrm2Device(spgpuHandle_t h, float* result, int n, const float* a) { nrm2ToDevice<float>(h, result, n, a); }
void spgpuDnrm2Device(spgpuHandle_t h, double* result, int n, const double* a) { nrm2ToDevice<double>(h, result, n, a); }
void spgpuSaxpbyDevice(spgpuHandle_t h, float* z, int n, const float* beta, const float* y, const float* alpha, const float* x)
{ axpbyFromDevice<float>(h, z, n, beta != nullptr, beta, nullptr, y, alpha, nullptr, 0, x); }
void spgpuSaxpbyQuotDevice(spgpuHandle_t h, float* z, int n, const float* betaNum, const float* betaDen, const float* y,
                           const float* alphaNum, const float* alphaDen, int negateAlpha, const float* x)
{ axpbyFromDevice<float>(h, z, n, 1, betaNum, betaDen, y, alphaNum, alphaDen, negateAlpha, x); }
void spgpuDaxpbyDevice(spgpuHandle_t h, double* z, int n, const double* beta, const double* y, const double* alpha, const double* x)
{ axpbyFromDevice<double>(h, z, n, beta != nullptr, beta, nullptr, y, alpha, nullptr, 0, x); }
void spgpuDaxpbyQuotDevice(spgpuHandle_t h, double* z, int n, const double* betaNum, const double* betaDen, const double* y,
                           const double* alphaNum, const double* alphaDen, int negateAlpha, const double* x)
{ axpbyFromDevice<double>(h, z, n, 1, betaNum, betaDen, y, alphaNum, alphaDen, negateAlpha, x); }
void spgpuSdivDevice(spgpuHandle_t h, float* out, const float* num, const float* den, int negate)
{ hipLaunchKernelGGL(divDeviceKernel<float>, dim3(1), dim3(1), 0, h->currentStream, out, num, den, negate); }
void spgpuDdivDevice(spgpuHandle_t h, double* out, const double* num, const double* den, int negate)
{ hipLaunchKernelGGL(divDeviceKernel<double>, dim3(1), dim3(1), 0, h->currentStream, out, num, den, negate); }
} // extern "C"
