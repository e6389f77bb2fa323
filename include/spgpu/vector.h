#pragma once
/*
 * Level-1 dense and sparse-vector operations.  The first half (axpby, dot,
 * nrm2 and their multivector forms) sits next to SpMV on the hot path; the
 * second half (scal, abs, axy, axypbz, gath, scat, setscal, asum, amax) is the
 * rest of the reference's vector.h, the operations a Krylov solver runs
 * between two SpMVs.
 *
 * Multivector convention (reference: vector.h:75-91): vector j of a
 * multivector starts at base + j*pitch, pitch in elements.
 * axpby-type calls are asynchronous on handle->currentStream; reductions
 * (dot, nrm2) return a host scalar and therefore synchronise that stream.
 * Reductions use scratch owned by the handle (the reference uses one
 * process-global device array, kernels/ddot.cu:35), so two handles may run
 * them concurrently; a single handle must not be shared between host threads.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- z = beta*y + alpha*x.  beta == 0 never reads y.  z may be x or y.
 * reference: vector.h:165-171, 447-453, 726-732, 1005-1011;
 * kernels/daxpby.cu:31-86, caxpby.cu:32-47, zaxpby.cu:33-48. */
void spgpuSaxpby(spgpuHandle_t handle, __device float* z, int n, float beta, __device float* y,
                 float alpha, __device float* x);
void spgpuDaxpby(spgpuHandle_t handle, __device double* z, int n, double beta, __device double* y,
                 double alpha, __device double* x);
void spgpuCaxpby(spgpuHandle_t handle, __device hipFloatComplex* z, int n, hipFloatComplex beta,
                 __device hipFloatComplex* y, hipFloatComplex alpha, __device hipFloatComplex* x);
void spgpuZaxpby(spgpuHandle_t handle, __device hipDoubleComplex* z, int n, hipDoubleComplex beta,
                 __device hipDoubleComplex* y, hipDoubleComplex alpha, __device hipDoubleComplex* x);

/* ---- the same over `count` vectors at stride `pitch`.
 * reference: vector.h:187-194, 469-476, 748-755, 1027-1034; kernels/daxpby.cu:88-101. */
void spgpuSmaxpby(spgpuHandle_t handle, __device float* z, int n, float beta, __device float* y,
                  float alpha, __device float* x, int count, int pitch);
void spgpuDmaxpby(spgpuHandle_t handle, __device double* z, int n, double beta, __device double* y,
                  double alpha, __device double* x, int count, int pitch);
void spgpuCmaxpby(spgpuHandle_t handle, __device hipFloatComplex* z, int n, hipFloatComplex beta,
                  __device hipFloatComplex* y, hipFloatComplex alpha, __device hipFloatComplex* x,
                  int count, int pitch);
void spgpuZmaxpby(spgpuHandle_t handle, __device hipDoubleComplex* z, int n, hipDoubleComplex beta,
                  __device hipDoubleComplex* y, hipDoubleComplex alpha, __device hipDoubleComplex* x,
                  int count, int pitch);

/* ---- sum_i a[i]*b[i], NOT conjugated for C/Z (reference: kernels/zdot.cu:54).
 * reference: vector.h:69-72, 366-369, 630-633, 925-928; kernels/ddot.cu:37-150. */
float spgpuSdot(spgpuHandle_t handle, int n, __device float* a, __device float* b);
double spgpuDdot(spgpuHandle_t handle, int n, __device double* a, __device double* b);
hipFloatComplex spgpuCdot(spgpuHandle_t handle, int n, __device hipFloatComplex* a,
                          __device hipFloatComplex* b);
hipDoubleComplex spgpuZdot(spgpuHandle_t handle, int n, __device hipDoubleComplex* a,
                           __device hipDoubleComplex* b);

/* ---- y[j] = dot(a_j, b_j) for j < count; y is a HOST array.
 * reference: vector.h:85-91, 397-403, 646-652, 941-947; kernels/ddot.cu:152-160. */
void spgpuSmdot(spgpuHandle_t handle, float* y, int n, __device float* a, __device float* b,
                int count, int pitch);
void spgpuDmdot(spgpuHandle_t handle, double* y, int n, __device double* a, __device double* b,
                int count, int pitch);
void spgpuCmdot(spgpuHandle_t handle, hipFloatComplex* y, int n, __device hipFloatComplex* a,
                __device hipFloatComplex* b, int count, int pitch);
void spgpuZmdot(spgpuHandle_t handle, hipDoubleComplex* y, int n, __device hipDoubleComplex* a,
                __device hipDoubleComplex* b, int count, int pitch);

/* ---- sqrt(sum |x_i|^2), unscaled (reference: kernels/dnrm2.cu:52-53,146).
 * reference: vector.h:117-119, 414-416, 678-680, 972-974. */
float spgpuSnrm2(spgpuHandle_t handle, int n, __device float* x);
double spgpuDnrm2(spgpuHandle_t handle, int n, __device double* x);
float spgpuCnrm2(spgpuHandle_t handle, int n, __device hipFloatComplex* x);
double spgpuZnrm2(spgpuHandle_t handle, int n, __device hipDoubleComplex* x);

/* ---- y[j] = nrm2(x_j); y is a HOST array.
 * reference: vector.h:131-136, 429-434, 692-697, 987-992. */
void spgpuSmnrm2(spgpuHandle_t handle, float* y, int n, __device float* x, int count, int pitch);
void spgpuDmnrm2(spgpuHandle_t handle, double* y, int n, __device double* x, int count, int pitch);
void spgpuCmnrm2(spgpuHandle_t handle, float* y, int n, __device hipFloatComplex* x, int count, int pitch);
void spgpuZmnrm2(spgpuHandle_t handle, double* y, int n, __device hipDoubleComplex* x, int count, int pitch);

/* ======================================================================== */
/* Rest of the reference's vector.h.                        */
/* Element-wise calls are asynchronous on handle->currentStream and may run  */
/* in place (output == an input, no offset).                                 */
/* ======================================================================== */

/* ---- y = alpha * x.  reference: vector.h:140-152 (S), scal_base.cuh:34-83. */
void spgpuSscal(spgpuHandle_t handle, __device float* y, int n, float alpha, __device float* x);
void spgpuDscal(spgpuHandle_t handle, __device double* y, int n, double alpha, __device double* x);
void spgpuCscal(spgpuHandle_t handle, __device hipFloatComplex* y, int n, hipFloatComplex alpha, __device hipFloatComplex* x);
void spgpuZscal(spgpuHandle_t handle, __device hipDoubleComplex* y, int n, hipDoubleComplex alpha, __device hipDoubleComplex* x);

/* ---- y = alpha * |x|.  For C/Z the result is complex: alpha * (|x| + 0i), |x| computed like
 * cuCabs (scaled hypot).  reference: vector.h:95-107, 664-668; abs_base.cuh:43-110. */
void spgpuSabs(spgpuHandle_t handle, __device float* y, int n, float alpha, __device float* x);
void spgpuDabs(spgpuHandle_t handle, __device double* y, int n, double alpha, __device double* x);
void spgpuCabs(spgpuHandle_t handle, __device hipFloatComplex* y, int n, hipFloatComplex alpha, __device hipFloatComplex* x);
void spgpuZabs(spgpuHandle_t handle, __device hipDoubleComplex* y, int n, hipDoubleComplex alpha, __device hipDoubleComplex* x);

/* ---- z = alpha * (x .* y).  reference: vector.h:197-211, axy_base.cuh:37-92. */
void spgpuSaxy(spgpuHandle_t handle, __device float* z, int n, float alpha, __device float* x, __device float* y);
void spgpuDaxy(spgpuHandle_t handle, __device double* z, int n, double alpha, __device double* x, __device double* y);
void spgpuCaxy(spgpuHandle_t handle, __device hipFloatComplex* z, int n, hipFloatComplex alpha, __device hipFloatComplex* x, __device hipFloatComplex* y);
void spgpuZaxy(spgpuHandle_t handle, __device hipDoubleComplex* z, int n, hipDoubleComplex alpha, __device hipDoubleComplex* x, __device hipDoubleComplex* y);
void spgpuSmaxy(spgpuHandle_t handle, __device float* z, int n, float alpha, __device float* x, __device float* y, int count, int pitch);
void spgpuDmaxy(spgpuHandle_t handle, __device double* z, int n, double alpha, __device double* x, __device double* y, int count, int pitch);
void spgpuCmaxy(spgpuHandle_t handle, __device hipFloatComplex* z, int n, hipFloatComplex alpha, __device hipFloatComplex* x, __device hipFloatComplex* y, int count, int pitch);
void spgpuZmaxy(spgpuHandle_t handle, __device hipDoubleComplex* z, int n, hipDoubleComplex alpha, __device hipDoubleComplex* x, __device hipDoubleComplex* y, int count, int pitch);

/* ---- w = beta*z + alpha*(x .* y); alpha == 0 -> w = beta*z, beta == 0 -> w = alpha*(x.*y).
 * reference: vector.h:214-232, axy_base.cuh:95-190 (the D and Z single-vector forms are defined in
 * daxy.cu / zaxy.cu but missing from the reference's header; declared here). */
void spgpuSaxypbz(spgpuHandle_t handle, __device float* w, int n, float beta, __device float* z, float alpha, __device float* x, __device float* y);
void spgpuDaxypbz(spgpuHandle_t handle, __device double* w, int n, double beta, __device double* z, double alpha, __device double* x, __device double* y);
void spgpuCaxypbz(spgpuHandle_t handle, __device hipFloatComplex* w, int n, hipFloatComplex beta, __device hipFloatComplex* z, hipFloatComplex alpha, __device hipFloatComplex* x, __device hipFloatComplex* y);
void spgpuZaxypbz(spgpuHandle_t handle, __device hipDoubleComplex* w, int n, hipDoubleComplex beta, __device hipDoubleComplex* z, hipDoubleComplex alpha, __device hipDoubleComplex* x, __device hipDoubleComplex* y);
void spgpuSmaxypbz(spgpuHandle_t handle, __device float* w, int n, float beta, __device float* z, float alpha, __device float* x, __device float* y, int count, int pitch);
void spgpuDmaxypbz(spgpuHandle_t handle, __device double* w, int n, double beta, __device double* z, double alpha, __device double* x, __device double* y, int count, int pitch);
void spgpuCmaxypbz(spgpuHandle_t handle, __device hipFloatComplex* w, int n, hipFloatComplex beta, __device hipFloatComplex* z, hipFloatComplex alpha, __device hipFloatComplex* x, __device hipFloatComplex* y, int count, int pitch);
void spgpuZmaxypbz(spgpuHandle_t handle, __device hipDoubleComplex* w, int n, hipDoubleComplex beta, __device hipDoubleComplex* z, hipDoubleComplex alpha, __device hipDoubleComplex* x, __device hipDoubleComplex* y, int count, int pitch);

/* ---- gather: xValues[i] = y[xIndices[i] - xBaseIndex] (entries with a negative position are skipped).
 * reference: vector.h:30-35, 282-296; gath_base.cuh:32-86. */
void spgpuIgath(spgpuHandle_t handle, __device int* xValues, int xNnz, const __device int* xIndices, int xBaseIndex, const __device int* y);
void spgpuSgath(spgpuHandle_t handle, __device float* xValues, int xNnz, const __device int* xIndices, int xBaseIndex, const __device float* y);
void spgpuDgath(spgpuHandle_t handle, __device double* xValues, int xNnz, const __device int* xIndices, int xBaseIndex, const __device double* y);
void spgpuCgath(spgpuHandle_t handle, __device hipFloatComplex* xValues, int xNnz, const __device int* xIndices, int xBaseIndex, const __device hipFloatComplex* y);
void spgpuZgath(spgpuHandle_t handle, __device hipDoubleComplex* xValues, int xNnz, const __device int* xIndices, int xBaseIndex, const __device hipDoubleComplex* y);

/* ---- scatter: y[p] = beta*y[p] + xValues[i], p = xIndices[i] - xBaseIndex; beta == 0 -> y[p] = xValues[i].
 * Repeated indices are a race, as in the reference.  reference: vector.h:50-56, 299-316; scat_base.cuh:32-89. */
void spgpuIscat(spgpuHandle_t handle, __device int* y, int xNnz, const __device int* xValues, const __device int* xIndices, int xBaseIndex, int beta);
void spgpuSscat(spgpuHandle_t handle, __device float* y, int xNnz, const __device float* xValues, const __device int* xIndices, int xBaseIndex, float beta);
void spgpuDscat(spgpuHandle_t handle, __device double* y, int xNnz, const __device double* xValues, const __device int* xIndices, int xBaseIndex, double beta);
void spgpuCscat(spgpuHandle_t handle, __device hipFloatComplex* y, int xNnz, const __device hipFloatComplex* xValues, const __device int* xIndices, int xBaseIndex, hipFloatComplex beta);
void spgpuZscat(spgpuHandle_t handle, __device hipDoubleComplex* y, int xNnz, const __device hipDoubleComplex* xValues, const __device int* xIndices, int xBaseIndex, hipDoubleComplex beta);

/* ---- y[first-baseIndex .. last-baseIndex] = val.  reference: vector.h:1182-1215; setscal_base.cuh:32-82. */
void spgpuIsetscal(spgpuHandle_t handle, int first, int last, int baseIndex, int val, __device int* y);
void spgpuSsetscal(spgpuHandle_t handle, int first, int last, int baseIndex, float val, __device float* y);
void spgpuDsetscal(spgpuHandle_t handle, int first, int last, int baseIndex, double val, __device double* y);
void spgpuCsetscal(spgpuHandle_t handle, int first, int last, int baseIndex, hipFloatComplex val, __device hipFloatComplex* y);
void spgpuZsetscal(spgpuHandle_t handle, int first, int last, int baseIndex, hipDoubleComplex val, __device hipDoubleComplex* y);

/* ---- sum |x_i| and max |x_i| (|.| of a complex value as cuCabs); host results, stream synchronised.
 * These implement the DOCUMENTED semantics; the reference's own asum/amax kernels drop all but two
 * lanes' contributions per block (asum_base.cuh:167-184, amax_base.cuh:156-173).
 * reference: vector.h:319-339 (+D/C/Z). */
float spgpuSasum(spgpuHandle_t handle, int n, float* x);
double spgpuDasum(spgpuHandle_t handle, int n, double* x);
float spgpuCasum(spgpuHandle_t handle, int n, hipFloatComplex* x);
double spgpuZasum(spgpuHandle_t handle, int n, hipDoubleComplex* x);
float spgpuSamax(spgpuHandle_t handle, int n, float* x);
double spgpuDamax(spgpuHandle_t handle, int n, double* x);
float spgpuCamax(spgpuHandle_t handle, int n, hipFloatComplex* x);
double spgpuZamax(spgpuHandle_t handle, int n, hipDoubleComplex* x);
void spgpuSmasum(spgpuHandle_t handle, float* y, int n, float* x, int count, int pitch);
void spgpuDmasum(spgpuHandle_t handle, double* y, int n, double* x, int count, int pitch);
void spgpuCmasum(spgpuHandle_t handle, float* y, int n, hipFloatComplex* x, int count, int pitch);
void spgpuZmasum(spgpuHandle_t handle, double* y, int n, hipDoubleComplex* x, int count, int pitch);
void spgpuSmamax(spgpuHandle_t handle, float* y, int n, float* x, int count, int pitch);
void spgpuDmamax(spgpuHandle_t handle, double* y, int n, double* x, int count, int pitch);
void spgpuCmamax(spgpuHandle_t handle, float* y, int n, hipFloatComplex* x, int count, int pitch);
void spgpuZmamax(spgpuHandle_t handle, double* y, int n, hipDoubleComplex* x, int count, int pitch);

#ifdef __cplusplus
}
#endif
