#pragma once
/*
 * Level-1 dense vector operations that sit next to SpMV on the hot path.
 * Replaces the matching entry points of the reference's vector.h; the rest
 * of that header (abs, amax, asum, axy, scal, gath, scat, setscal) is a later
 * scope row and is not declared here.
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

#ifdef __cplusplus
}
#endif
