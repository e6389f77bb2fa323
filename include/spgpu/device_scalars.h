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
 *   spgpu?dotDevice     *result == what spgpu?dot returns (the same block partials, combined in the same fixed order);
 *   spgpu?nrm2Device    *result == what spgpu?nrm2 returns;
 *   spgpu?axpbyDevice   z == what spgpu?axpby writes for alpha = *alpha, beta = *beta
 *                       (beta == NULL or *beta == 0: y is not read, z = alpha*x);
 *   spgpu?axpbyQuotDevice  the same with each coefficient given as a quotient of two device scalars,
 *                       beta = betaNum/betaDen, alpha = (negateAlpha ? -1 : 1) * alphaNum/alphaDen (a NULL operand
 *                       stands for 1; one IEEE division each, as the host would compute rr / pAp): the
 *                       x += alpha p, r -= alpha Ap, p = r + beta p of CG without a kernel for the division;
 *   spgpu?divDevice     *out = (negate ? -1 : 1) * (*num / *den).
 * A reduction is two kernels (block partials; one wavefront that combines them: 16 per lane, then a lane-xor tree).
 * tools/cg_amd.c runs CG both ways (eager with host scalars, and one captured
 * graph per iteration with these) and compares the iterates.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

void spgpuSdotDevice(spgpuHandle_t handle, __device float* result, int n, const __device float* a, const __device float* b);
void spgpuDdotDevice(spgpuHandle_t handle, __device double* result, int n, const __device double* a, const __device double* b);

/* *result == what spgpu?nrm2 returns (the residual norm of a solver without a host round trip). */
void spgpuSnrm2Device(spgpuHandle_t handle, __device float* result, int n, const __device float* a);
void spgpuDnrm2Device(spgpuHandle_t handle, __device double* result, int n, const __device double* a);

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

/* ---- fused steps of a Krylov iteration (NEW; csrc/fused_solver.hip) --------------------------------------------
 * On a small system (BASELINE configs[0], 1 M rows) an iteration is bound by its chain of dependent launches, not by
 * bytes.  These calls are the first stage of spgpu?dotDevice with the second operand produced on the fly, so *result
 * has the bits spgpu?dotDevice(result, n, w, z) would leave for the vectors they store.
 *
 * spgpu?hellspmvDotDevice: z = alpha*A*x + beta*y for a HELL matrix in natural row order (arguments as
 *   spgpu?hellspmv, hell.h; no rIdx), *result = w . z (w == NULL: w = x, the p.Ap of CG).  Row sums run over a
 *   row's entries in ascending k -- the reference's one-thread-per-row order (hell_spmv_base_template.cuh:104-215);
 *   spgpu?hellspmv's default kernels add in the same order unless their cooperative tail engages (rows of uneven
 *   length).  z may alias y; z must not alias x or w.
 * spgpu?axpbyPairDotDevice: with a = *alphaNum / *alphaDen (a NULL operand stands for 1): z1 = y1 + a*x1,
 *   z2 = y2 - a*x2, *result = z2 . z2 -- the x += alpha p, r -= alpha Ap, |r|^2 of CG; z1, z2 hold the bits of
 *   spgpu?axpbyQuotDevice(z1, n, NULL, NULL, y1, alphaNum, alphaDen, 0, x1) and (..., 1, x2).  z1 may alias y1,
 *   z2 may alias y2. */
void spgpuShellspmvDotDevice(spgpuHandle_t handle, __device float* result, const __device float* w, __device float* z,
                             const __device float* y, float alpha, const __device float* cM, const __device int* rP,
                             int hackSize, const __device int* hackOffsets, const __device int* rS, int rows,
                             const __device float* x, float beta, int baseIndex);
void spgpuDhellspmvDotDevice(spgpuHandle_t handle, __device double* result, const __device double* w, __device double* z,
                             const __device double* y, double alpha, const __device double* cM, const __device int* rP,
                             int hackSize, const __device int* hackOffsets, const __device int* rS, int rows,
                             const __device double* x, double beta, int baseIndex);
void spgpuSaxpbyPairDotDevice(spgpuHandle_t handle, __device float* result, int n, __device float* z1,
                              const __device float* y1, const __device float* x1, __device float* z2,
                              const __device float* y2, const __device float* x2, const __device float* alphaNum,
                              const __device float* alphaDen);
void spgpuDaxpbyPairDotDevice(spgpuHandle_t handle, __device double* result, int n, __device double* z1,
                              const __device double* y1, const __device double* x1, __device double* z2,
                              const __device double* y2, const __device double* x2, const __device double* alphaNum,
                              const __device double* alphaDen);

#ifdef __cplusplus
}
#endif
