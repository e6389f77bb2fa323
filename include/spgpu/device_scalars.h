#pragma once
/*
 * Level-1 calls whose scalars live in DEVICE memory (NEW; SURVEY.md section 8,
 * row f4).  The reference's reductions return a host value and therefore
 * synchronise the stream (vector.h:61-120, ddot.cu:120-150); a Krylov
 * iteration written with them cannot be captured in a graph and pays two host
 * round trips per iteration.  These variants leave results in device memory
 * and take coefficients from device memory: no synchronisation, no allocation,
 * capturable in a HIP graph on handle->currentStream.
 *
 * Values are the SAME BITS as the host-scalar calls produce:
 *   spgpu?dotDevice     *result == what spgpu?dot returns (block partials added in block order);
 *   spgpu?axpbyDevice   z == what spgpu?axpby writes for alpha = *alpha, beta = *beta
 *                       (beta == NULL or *beta == 0: y is not read, z = alpha*x);
 *   spgpu?axpbyQuotDevice  the same with each coefficient given as a quotient of two device scalars,
 *                       beta = betaNum/betaDen, alpha = (negateAlpha ? -1 : 1) * alphaNum/alphaDen (a NULL operand
 *                       stands for 1; one IEEE division each, as the host would compute rr / pAp): the
 *                       x += alpha p, r -= alpha Ap, p = r + beta p of CG without a kernel for the division;
 *   spgpu?divDevice     *out = (negate ? -1 : 1) * (*num / *den).
 * A reduction is two kernels (block partials; one workgroup that adds them in block order).
 * tools/cg_amd.c runs CG both ways (eager with host scalars, and one captured
 * graph per iteration with these) and compares the iterates.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

void spgpuSdotDevice(spgpuHandle_t handle, __device float* result, int n, const __device float* a, const __device float* b);
void spgpuDdotDevice(spgpuHandle_t handle, __device double* result, int n, const __device double* a, const __device double* b);

void spgpuSaxpbyDevice(spgpuHandle_t handle, __device float* z, int n, const __device float* beta, const __device float* y,
                       const __device float* alpha, const __device float* x);
void spgpuDaxpbyDevice(spgpuHandle_t handle, __device double* z, int n, const __device double* beta, const __device double* y,
                       const __device double* alpha, const __device double* x);

void spgpuSaxpbyQuotDevice(spgpuHandle_t handle, __device float* z, int n, const __device float* betaNum,
                           const __device float* betaDen, const __device float* y, const __device float* alphaNum,
                           const __device float* alphaDen, int negateAlpha, const __device float* x);
void spgpuDaxpbyQuotDevice(spgpuHandle_t handle, __device double* z, int n, const __device double* betaNum,
                           const __device double* betaDen, const __device double* y, const __device double* alphaNum,
                           const __device double* alphaDen, int negateAlpha, const __device double* x);

void spgpuSdivDevice(spgpuHandle_t handle, __device float* out, const __device float* num, const __device float* den, int negate);
void spgpuDdivDevice(spgpuHandle_t handle, __device double* out, const __device double* num, const __device double* den, int negate);

#ifdef __cplusplus
}
#endif
