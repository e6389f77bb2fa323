/*
 * HDIA (sliced diagonal) SpMV for gfx950 (MI355X):  z = alpha*A*x + beta*y.
 *
 * C ABI: spgpu{S,D,C,Z}hdiaspmv (include/spgpu/hdia.h, reference hdia.h:37-142).
 * Behaviour follows the reference dispatcher/kernel
 * (kernels/hdia_spmv_base.cuh:99-145, hdia_spmv_base_template.cuh:19-252):
 * a stored slot (hack-local diagonal d, row r) contributes iff
 * 0 <= offsets[d] + r < cols; products of a row are added in ascending d.
 * The kernel design below is new.
 *
 * ---- Wavefront design -------------------------------------------------------
 * A hack's coefficients are one contiguous block, diagonal after diagonal,
 * hackSize rows each, and consecutive hacks follow each other in memory.  A
 * lane owns a strip of RPL = 16 B / sizeof(T) consecutive rows (one 16-byte
 * load per stored diagonal), a wavefront owns 64*RPL consecutive rows (for
 * double and hackSize 32: four whole hacks, i.e. one contiguous piece of dM).
 * Diagonal counts per hack are small (7 for a 3-D Laplacian), so a lane walks
 * all diagonals of its strip itself: no cross-lane reduction, no LDS, no
 * barrier, and the summation order per row equals the reference's.
 * The diagonal offsets of a hack are read with one load per diagonal that is
 * the same address for all lanes of the strip's hack (a broadcast out of L1);
 * x is read at offsets[d] + row, i.e. contiguously across the lanes of a hack:
 * the RPL values of a strip are one 16-byte load (element-aligned: an odd
 * offset shifts it by one element) whenever no strip of the wavefront crosses
 * an edge of the matrix, element loads otherwise.
 * UNROLL diagonals are in flight per lane before the first multiply-add.
 *
 * Roofline: HBM bandwidth.  Algorithmic bytes: sizeof(T) per stored in-range
 * slot, 4 per stored diagonal, 4 per hack (+4), sizeof(T) per column (x once)
 * and per row (z) [+ y when beta != 0].
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/hdia.h"

#include <stdlib.h>

namespace spgpu {

template <typename T> struct HdiaArgs {
    T* z;
    const T* y;
    const T* x;
    const T* dM;
    const int* offsets;
    const int* hackOffsets; /* NULL: plain DIA -- one hack holding all rows, `flatDiags` diagonals */
    T alpha, beta;
    int rows, cols, hackSize;
    int flatDiags;
    int wideIO;
    int xcdOrder; /* 0: hardware order; 1: XCD-contiguous eighths; n > 1: runs of n blocks per XCD */
};


template <typename T, int RPL, bool NT, int UNROLL, int kHdiaThreads>
__global__ __launch_bounds__(kHdiaThreads) void hdiaSpmvKernel(const HdiaArgs<T> a)
{
    const unsigned block = a.xcdOrder == 0 ? blockIdx.x
                         : a.xcdOrder == 1 ? xcdContiguous(blockIdx.x, gridDim.x)
                                           : xcdRuns(blockIdx.x, gridDim.x, (unsigned)a.xcdOrder);
    const long long strip = (long long)block * kHdiaThreads + threadIdx.x;
    const long long waveRow0 = (strip - (threadIdx.x & (kWave - 1))) * RPL;
    if (waveRow0 >= a.rows)
        return; /* whole wavefront leaves together */

    const long long row0 = strip * RPL;
    const bool live = row0 < a.rows;

    int firstDiag = 0, diags = 0;
    long long slab = 0;
    if (live) {
        if (a.hackOffsets) {
            const unsigned r0 = (unsigned)row0, hs = (unsigned)a.hackSize;
            const unsigned hack = r0 / hs;
            firstDiag = a.hackOffsets[hack];
            diags = a.hackOffsets[hack + 1] - firstDiag;
            slab = (long long)firstDiag * hs + (r0 - hack * hs);
        } else { /* DIA: dM[row + d*pitch], every row sees every stored diagonal */
            diags = a.flatDiags;
            slab = row0;
        }
    }
    const int waveDiags = waveMax(diags); /* wave-uniform trip count */
    const bool stripInside = row0 + RPL <= a.rows;

    T sum[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t)
        sum[t] = zeroOf<T>();

    const T* __restrict__ vals = a.dM + slab;
    const int* __restrict__ offs = a.offsets + firstDiag;
    const T* __restrict__ x = a.x;

    /* coefficients and offsets of a stage are requested one stage ahead: they depend on nothing the stage before
     * computes, so they travel while its x values are fetched and used */
    Pack<T, RPL> v[UNROLL], vNext[UNROLL];
    int off[UNROLL], offNext[UNROLL];
    auto fetch = [&](int dBase, Pack<T, RPL>* vv, int* oo) {
        /* a full stage everywhere in the wavefront: its UNROLL offsets are consecutive ints, one load instead of
         * UNROLL (element-aligned, like the x strips) */
        const bool whole = UNROLL == 4 && __ballot(dBase + UNROLL > diags) == 0ull;
        if (whole) {
            const Pack<int, 4> o4 = loadPackElementAligned<int, 4>(offs + dBase);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                vv[u] = loadPack<NT, T, RPL>(vals + (long long)(dBase + u) * a.hackSize);
                oo[u] = o4.v[u & 3];
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (dBase + u < diags) {
                vv[u] = loadPack<NT, T, RPL>(vals + (long long)(dBase + u) * a.hackSize);
                oo[u] = offs[dBase + u];
            } else {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    vv[u].v[t] = zeroOf<T>();
                oo[u] = 0;
            }
        }
    };
    fetch(0, v, off);
    for (int dBase = 0; dBase < waveDiags; dBase += UNROLL) {
        if (dBase + UNROLL < waveDiags) /* wave-uniform */
            fetch(dBase + UNROLL, vNext, offNext);
        Pack<T, RPL> xv[UNROLL];
        bool use[UNROLL][RPL];
        bool ragged = false; /* a live diagonal whose strip crosses an edge of the matrix */
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long col0 = row0 + off[u];
            const bool dLive = dBase + u < diags;
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const long long col = col0 + t;
                use[u][t] = dLive && row0 + t < a.rows && col >= 0 && col < a.cols;
            }
            ragged |= dLive && !(stripInside && col0 >= 0 && col0 + RPL <= a.cols);
        }
        /* Wavefront-uniform choice (a per-lane one is turned back into element loads by the compiler): when no strip
         * of the wavefront crosses an edge, the RPL consecutive columns of a strip are ONE 16-byte load -- aligned to
         * the element size only, an odd offset shifts it by one element. */
        if (RPL > 1 && a.cols >= RPL && __ballot(ragged) == 0ull) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                xv[u] = loadPackElementAligned<T, RPL>(x + (dBase + u < diags ? row0 + off[u] : 0));
        } else {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    xv[u].v[t] = x[use[u][t] ? row0 + off[u] + t : 0];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                sum[t] = pick(use[u][t], mulAdd(v[u].v[t], xv[u].v[t], sum[t]), sum[t]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = vNext[u];
            off[u] = offNext[u];
        }
    }

    if (!live)
        return;

    const bool hasBeta = isNotZero(a.beta);
    if (a.wideIO && stripInside) {
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
        storePackMaybeNT<NT, T, RPL>(a.z + row0, out); /* z is written once and not read again by this call */
    } else {
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            if (row0 + t < a.rows)
                a.z[row0 + t] = hasBeta ? epilogue<true>(a.alpha, sum[t], a.beta, a.y[row0 + t])
                                        : epilogue<false>(a.alpha, sum[t], a.beta, zeroOf<T>());
        }
    }
}

template <typename T, int RPL, int UNROLL, int kHdiaThreads>
static void launchHdiaSized(hipStream_t stream, const HdiaArgs<T>& a, bool nt)
{
    const long long strips = ((long long)a.rows + RPL - 1) / RPL;
    const unsigned blocks = (unsigned)((strips + kHdiaThreads - 1) / kHdiaThreads);
    if (nt)
        hipLaunchKernelGGL((hdiaSpmvKernel<T, RPL, true, UNROLL, kHdiaThreads>), dim3(blocks), dim3(kHdiaThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((hdiaSpmvKernel<T, RPL, false, UNROLL, kHdiaThreads>), dim3(blocks), dim3(kHdiaThreads), 0, stream, a);
}

template <typename T, int RPL, int UNROLL>
static void launchHdia(hipStream_t stream, const HdiaArgs<T>& a, bool nt)
{
    /* SPGPU_HDIA_BLOCK: workgroup size 256 / 512 (default: +3-4 % over 256 on 512^3 with the current kernel) / 1024 */
    const int block = spgpuTuning()->hdiaBlock;
    if (block == 1024)
        launchHdiaSized<T, RPL, UNROLL, 1024>(stream, a, nt);
    else if (block == 256)
        launchHdiaSized<T, RPL, UNROLL, 256>(stream, a, nt);
    else
        launchHdiaSized<T, RPL, UNROLL, 512>(stream, a, nt);
}

template <typename T, typename ApiT>
static void hdiaSpmv(spgpuHandle_t handle, ApiT* z, const ApiT* y, ApiT alpha, const ApiT* dM, const int* offsets,
                     int hackSize, const int* hackOffsets, int rows, int cols, const ApiT* x, ApiT beta,
                     int flatDiags = 0)
{
    static_assert(sizeof(T) == sizeof(ApiT), "ABI type and device type must have one layout");
    if (rows <= 0 || hackSize <= 0)
        return;
    HdiaArgs<T> a;
    a.z = reinterpret_cast<T*>(z);
    a.y = reinterpret_cast<const T*>(y);
    a.x = reinterpret_cast<const T*>(x);
    a.dM = reinterpret_cast<const T*>(dM);
    a.offsets = offsets;
    a.hackOffsets = hackOffsets;
    __builtin_memcpy(&a.alpha, &alpha, sizeof(T));
    __builtin_memcpy(&a.beta, &beta, sizeof(T));
    a.rows = rows;
    a.cols = cols;
    a.hackSize = hackSize;
    a.flatDiags = flatDiags;

    const SpgpuTuning* tune = spgpuTuning();
    a.xcdOrder = tune->xcdOrder;
    constexpr int WIDE = 16 / (int)sizeof(T);
    const bool nt = tune->ntLoads != 0;
    const bool wideOk = WIDE > 1 && hackSize % WIDE == 0 && ((uintptr_t)dM % 16 == 0) && !tune->hdiaNarrow;

    hipStream_t stream = handle->currentStream;
    if constexpr (WIDE > 1) {
        if (wideOk) {
            a.wideIO = ((uintptr_t)z % 16 == 0) && ((uintptr_t)y % 16 == 0);
            /* 4 diagonals per stage by default: with the strip's x values fetched as one 16-byte load the 8-per-stage
             * form needs 97 VGPRs (5 wavefronts per SIMD) against 56 (8 wavefronts) and is 15 % slower on 512^3
             * (tools/ab_hdia.py, profiles/r01d_ab_hdia_wide_x.txt); SPGPU_HDIA_VARIANT=2 selects it.  XCD-contiguous
             * block orders and workgroups of 512/1024 lanes are 2-13 % slower than the hardware order with 256. */
            if (tune->hdiaVariant == 2)
                launchHdia<T, WIDE, 8>(stream, a, nt);
            else
                launchHdia<T, WIDE, 4>(stream, a, nt);
            spgpuDebugCheck(handle, "hdiaspmv");
            return;
        }
    }
    a.wideIO = 1;
    launchHdia<T, 1, 4>(stream, a, nt);
    spgpuDebugCheck(handle, "hdiaspmv");
}

} // namespace spgpu

using namespace spgpu;

extern "C" {

void spgpuShdiaspmv(spgpuHandle_t handle, float* z, const float* y, float alpha, const float* dM,
                    const int* offsets, int hackSize, const int* hackOffsets, int rows, int cols,
                    const float* x, float beta)
{
    hdiaSpmv<float>(handle, z, y, alpha, dM, offsets, hackSize, hackOffsets, rows, cols, x, beta);
}

void spgpuDhdiaspmv(spgpuHandle_t handle, double* z, const double* y, double alpha, const double* dM,
                    const int* offsets, int hackSize, const int* hackOffsets, int rows, int cols,
                    const double* x, double beta)
{
    hdiaSpmv<double>(handle, z, y, alpha, dM, offsets, hackSize, hackOffsets, rows, cols, x, beta);
}

void spgpuChdiaspmv(spgpuHandle_t handle, hipFloatComplex* z, const hipFloatComplex* y, hipFloatComplex alpha,
                    const hipFloatComplex* dM, const int* offsets, int hackSize, const int* hackOffsets,
                    int rows, int cols, const hipFloatComplex* x, hipFloatComplex beta)
{
    hdiaSpmv<cfloat>(handle, z, y, alpha, dM, offsets, hackSize, hackOffsets, rows, cols, x, beta);
}

void spgpuZhdiaspmv(spgpuHandle_t handle, hipDoubleComplex* z, const hipDoubleComplex* y, hipDoubleComplex alpha,
                    const hipDoubleComplex* dM, const int* offsets, int hackSize, const int* hackOffsets,
                    int rows, int cols, const hipDoubleComplex* x, hipDoubleComplex beta)
{
    hdiaSpmv<cdouble>(handle, z, y, alpha, dM, offsets, hackSize, hackOffsets, rows, cols, x, beta);
}

/* ---- DIA (include/spgpu/dia.h; reference dia.h:42-143, dia_spmv_base_template.cuh:20-216): the same
 * kernel over one all-rows hack: hackSize = dMPitch, hackOffsets = NULL. ---- */
#include "spgpu/dia.h"

void spgpuSdiaspmv(spgpuHandle_t handle, float* z, const float* y, float alpha, const float* dM, const int* offsets,
                   int dMPitch, int rows, int cols, int diags, const float* x, float beta)
{
    hdiaSpmv<float>(handle, z, y, alpha, dM, offsets, dMPitch, nullptr, rows, cols, x, beta, diags);
}
void spgpuDdiaspmv(spgpuHandle_t handle, double* z, const double* y, double alpha, const double* dM, const int* offsets,
                   int dMPitch, int rows, int cols, int diags, const double* x, double beta)
{
    hdiaSpmv<double>(handle, z, y, alpha, dM, offsets, dMPitch, nullptr, rows, cols, x, beta, diags);
}
void spgpuCdiaspmv(spgpuHandle_t handle, hipFloatComplex* z, const hipFloatComplex* y, hipFloatComplex alpha,
                   const hipFloatComplex* dM, const int* offsets, int dMPitch, int rows, int cols, int diags,
                   const hipFloatComplex* x, hipFloatComplex beta)
{
    hdiaSpmv<cfloat>(handle, z, y, alpha, dM, offsets, dMPitch, nullptr, rows, cols, x, beta, diags);
}
void spgpuZdiaspmv(spgpuHandle_t handle, hipDoubleComplex* z, const hipDoubleComplex* y, hipDoubleComplex alpha,
                   const hipDoubleComplex* dM, const int* offsets, int dMPitch, int rows, int cols, int diags,
                   const hipDoubleComplex* x, hipDoubleComplex beta)
{
    hdiaSpmv<cdouble>(handle, z, y, alpha, dM, offsets, dMPitch, nullptr, rows, cols, x, beta, diags);
}

} // extern "C"
