/*
 * Level-1 operations next to SpMV on the hot path, for gfx950 (MI355X).
 *
 * C ABI: spgpu{S,D,C,Z}axpby / maxpby / dot / mdot / nrm2 / mnrm2
 *        (include/spgpu/vector.h; reference vector.h, kernels/{s,d,c,z}axpby.cu,
 *         kernels/{s,d,c,z}dot.cu, kernels/{s,d,c,z}nrm2.cu).
 *
 * axpby is a pure stream (2 reads + 1 write per element, 1 read when beta==0):
 * 16-byte accesses per lane, a capped grid with a grid-stride loop, one launch
 * for a whole multivector (grid.y = vector index).
 * dot / nrm2 are two-stage: every workgroup reduces a grid-stride slice with
 * lane-xor shuffles and one LDS hop between its 4 wavefronts, writes one
 * partial into scratch owned by the HANDLE, and the host adds the partials in
 * block order after one async copy + stream sync (the reference copies from a
 * process-global __device__ array, kernels/ddot.cu:35,139).
 *
 * Roofline: HBM.  Algorithmic bytes: axpby n*sizeof(T)*(2 + [beta != 0]);
 * dot 2*n*sizeof(T); nrm2 n*sizeof(T).
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/vector.h"

#include <math.h>
#include <type_traits>

namespace spgpu {

constexpr int kL1Threads = 256;
constexpr int kL1MaxBlocks = 2048; /* 256 CUs x 8 resident workgroups */

/* ---- axpby ---------------------------------------------------------------
 * Expression trees (reference): S/D  alpha*x + beta*y   (daxpby.cu:40-43)
 *                               C    fma(beta, y, alpha*x)  (caxpby.cu:44)
 *                               Z    fma(alpha, x, beta*y)  (zaxpby.cu:45) */
__device__ inline float axpbyOne(float alpha, float x, float beta, float y) { return mulAdd(alpha, x, beta * y); }
__device__ inline double axpbyOne(double alpha, double x, double beta, double y) { return mulAdd(alpha, x, beta * y); }
__device__ inline cfloat axpbyOne(cfloat alpha, cfloat x, cfloat beta, cfloat y) { return mulAdd(beta, y, mul(alpha, x)); }
__device__ inline cdouble axpbyOne(cdouble alpha, cdouble x, cdouble beta, cdouble y) { return mulAdd(alpha, x, mul(beta, y)); }

constexpr int kL1Unroll = 4; /* independent 16-byte accesses in flight per lane */

template <typename T, int VEC, bool HAS_BETA>
__global__ __launch_bounds__(kL1Threads) void axpbyKernel(T* z, int n, T beta, const T* y, T alpha, const T* x,
                                                         long long pitch)
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
                xv[u] = loadPack<false, T, VEC>(x + p * VEC);
                if constexpr (HAS_BETA)
                    yv[u] = loadPack<false, T, VEC>(y + p * VEC);
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
                storePack<T, VEC>(z + p * VEC, out);
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
    const long long cap = kL1MaxBlocks / (count < kL1MaxBlocks ? count : kL1MaxBlocks);
    if (blocks > (cap > 1 ? cap : 1))
        blocks = cap > 1 ? cap : 1;
    const dim3 grid((unsigned)blocks, (unsigned)count);
    hipStream_t s = handle->currentStream;

#define SPGPU_AXPBY_GO(VEC)                                                                               \
    do {                                                                                                  \
        if (hasBeta)                                                                                      \
            hipLaunchKernelGGL((axpbyKernel<T, VEC, true>), grid, dim3(kL1Threads), 0, s, z, n, beta, y,   \
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

/* DOT: a[i]*b[i] accumulated with the SpMV multiply-add (un-conjugated);
 * NRM2: |a[i]|^2 accumulated in the real type. */
template <typename T> struct RealOf { using type = T; };
template <typename R> struct RealOf<Cx<R>> { using type = R; };

template <typename R> __device__ inline R absSqAdd(R v, R acc) { return mulAdd(v, v, acc); }
template <typename R> __device__ inline R absSqAdd(Cx<R> v, R acc) { return mulAdd(v.y, v.y, mulAdd(v.x, v.x, acc)); }

template <typename A> __device__ inline A blockSum(A v, A* lds)
{
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1)
        v = add(v, laneXor(v, m));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0)
        lds[wave] = v;
    __syncthreads();
    A total = lds[0];
#pragma unroll
    for (int w = 1; w < kL1Threads / kWave; ++w)
        total = add(total, lds[w]);
    return total;
}

template <typename T, int VEC, bool NRM2>
__global__ __launch_bounds__(kL1Threads) void reduceKernel(typename std::conditional<NRM2, typename RealOf<T>::type, T>::type* partials,
                                                          int n, const T* a, const T* b, long long pitch)
{
    using Acc = typename std::conditional<NRM2, typename RealOf<T>::type, T>::type;
    __shared__ Acc lds[kL1Threads / kWave];

    const long long shift = (long long)blockIdx.y * pitch;
    a += shift;
    if constexpr (!NRM2)
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
                av[u] = loadPack<false, T, VEC>(a + p * VEC);
                if constexpr (!NRM2)
                    bv[u] = loadPack<false, T, VEC>(b + p * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < kL1Unroll; ++u) {
            if (live[u]) {
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    if constexpr (NRM2)
                        acc = absSqAdd(av[u].v[t], acc);
                    else
                        acc = mulAdd(av[u].v[t], bv[u].v[t], acc);
                }
            }
        }
    }
    const long long tail = packs * VEC + (long long)blockIdx.x * kL1Threads + threadIdx.x;
    if (tail < n) {
        if constexpr (NRM2)
            acc = absSqAdd(a[tail], acc);
        else
            acc = mulAdd(a[tail], b[tail], acc);
    }

    const Acc total = blockSum(acc, lds);
    if (threadIdx.x == 0)
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
}

/* Runs the two-stage reduction for `count` vectors and leaves one host value
 * per vector in out[].  Synchronises handle->currentStream. */
template <typename T, bool NRM2, typename Acc>
static void reduceVectors(spgpuHandle_t handle, Acc* out, int n, const T* a, const T* b, int count, int pitch)
{
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
        const T* b0 = NRM2 ? nullptr : b + (size_t)first * pitch;
        const bool wide = WIDE > 1 && ((uintptr_t)a0 % 16 == 0) && (NRM2 || (uintptr_t)b0 % 16 == 0) &&
                          (vectors == 1 || pitch % WIDE == 0);
        const long long work = wide ? ((long long)n + WIDE - 1) / WIDE : n;
        long long blocks = (work + kL1Threads * kL1Unroll - 1) / (kL1Threads * kL1Unroll);
        const long long cap = SPGPU_REDUCE_MAX_BLOCKS / vectors;
        if (blocks > cap)
            blocks = cap;
        const dim3 grid((unsigned)blocks, (unsigned)vectors);

        if (wide)
            hipLaunchKernelGGL((reduceKernel<T, WIDE, NRM2>), grid, dim3(kL1Threads), 0, s, dev, n, a0, b0,
                               (long long)pitch);
        else
            hipLaunchKernelGGL((reduceKernel<T, 1, NRM2>), grid, dim3(kL1Threads), 0, s, dev, n, a0, b0,
                               (long long)pitch);
        (void)hipMemcpyAsync(host, dev, sizeof(Acc) * (size_t)blocks * vectors, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);

        for (int j = 0; j < vectors; ++j) {
            Acc total = zeroOf<Acc>();
            for (long long k = 0; k < blocks; ++k) {
                total = add(total, host[(size_t)j * blocks + k]);
            }
            out[first + j] = total;
        }
    }
    spgpuDebugCheck(handle, NRM2 ? "nrm2" : "dot");
}

} // namespace spgpu

using namespace spgpu;

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
{ float r; reduceVectors<float, false>(h, &r, n, a, b, 1, 0); return r; }
double spgpuDdot(spgpuHandle_t h, int n, double* a, double* b)
{ double r; reduceVectors<double, false>(h, &r, n, a, b, 1, 0); return r; }
hipFloatComplex spgpuCdot(spgpuHandle_t h, int n, hipFloatComplex* a, hipFloatComplex* b)
{
    cfloat r;
    reduceVectors<cfloat, false>(h, &r, n, reinterpret_cast<cfloat*>(a), reinterpret_cast<cfloat*>(b), 1, 0);
    return make_hipFloatComplex(r.x, r.y);
}
hipDoubleComplex spgpuZdot(spgpuHandle_t h, int n, hipDoubleComplex* a, hipDoubleComplex* b)
{
    cdouble r;
    reduceVectors<cdouble, false>(h, &r, n, reinterpret_cast<cdouble*>(a), reinterpret_cast<cdouble*>(b), 1, 0);
    return make_hipDoubleComplex(r.x, r.y);
}

void spgpuSmdot(spgpuHandle_t h, float* y, int n, float* a, float* b, int count, int pitch)
{ reduceVectors<float, false>(h, y, n, a, b, count, pitch); }
void spgpuDmdot(spgpuHandle_t h, double* y, int n, double* a, double* b, int count, int pitch)
{ reduceVectors<double, false>(h, y, n, a, b, count, pitch); }
void spgpuCmdot(spgpuHandle_t h, hipFloatComplex* y, int n, hipFloatComplex* a, hipFloatComplex* b, int count, int pitch)
{
    reduceVectors<cfloat, false>(h, reinterpret_cast<cfloat*>(y), n, reinterpret_cast<cfloat*>(a),
                                 reinterpret_cast<cfloat*>(b), count, pitch);
}
void spgpuZmdot(spgpuHandle_t h, hipDoubleComplex* y, int n, hipDoubleComplex* a, hipDoubleComplex* b, int count, int pitch)
{
    reduceVectors<cdouble, false>(h, reinterpret_cast<cdouble*>(y), n, reinterpret_cast<cdouble*>(a),
                                  reinterpret_cast<cdouble*>(b), count, pitch);
}

/* ---- nrm2 ---- */
float spgpuSnrm2(spgpuHandle_t h, int n, float* x)
{ float r; reduceVectors<float, true>(h, &r, n, x, (const float*)nullptr, 1, 0); return sqrtf(r); }
double spgpuDnrm2(spgpuHandle_t h, int n, double* x)
{ double r; reduceVectors<double, true>(h, &r, n, x, (const double*)nullptr, 1, 0); return sqrt(r); }
float spgpuCnrm2(spgpuHandle_t h, int n, hipFloatComplex* x)
{ float r; reduceVectors<cfloat, true>(h, &r, n, reinterpret_cast<cfloat*>(x), (const cfloat*)nullptr, 1, 0); return sqrtf(r); }
double spgpuZnrm2(spgpuHandle_t h, int n, hipDoubleComplex* x)
{ double r; reduceVectors<cdouble, true>(h, &r, n, reinterpret_cast<cdouble*>(x), (const cdouble*)nullptr, 1, 0); return sqrt(r); }

void spgpuSmnrm2(spgpuHandle_t h, float* y, int n, float* x, int count, int pitch)
{
    reduceVectors<float, true>(h, y, n, x, (const float*)nullptr, count, pitch);
    for (int j = 0; j < count; ++j) y[j] = sqrtf(y[j]);
}
void spgpuDmnrm2(spgpuHandle_t h, double* y, int n, double* x, int count, int pitch)
{
    reduceVectors<double, true>(h, y, n, x, (const double*)nullptr, count, pitch);
    for (int j = 0; j < count; ++j) y[j] = sqrt(y[j]);
}
void spgpuCmnrm2(spgpuHandle_t h, float* y, int n, hipFloatComplex* x, int count, int pitch)
{
    reduceVectors<cfloat, true>(h, y, n, reinterpret_cast<cfloat*>(x), (const cfloat*)nullptr, count, pitch);
    for (int j = 0; j < count; ++j) y[j] = sqrtf(y[j]);
}
void spgpuZmnrm2(spgpuHandle_t h, double* y, int n, hipDoubleComplex* x, int count, int pitch)
{
    reduceVectors<cdouble, true>(h, y, n, reinterpret_cast<cdouble*>(x), (const cdouble*)nullptr, count, pitch);
    for (int j = 0; j < count; ++j) y[j] = sqrt(y[j]);
}

} // extern "C"
