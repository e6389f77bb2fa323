#pragma once
/*
 * HELL multi-vector product (SpMM):  Z = alpha*A*X + beta*Y  for `count`
 * right-hand sides at once.  NEW: the reference has no SpMM; this is the
 * operation BASELINE.json's north_star row-partitions across the GPUs of a
 * node.  A keeps the exact HELL arguments of
 * spgpu?hellspmv (hell.h:45-59).
 *
 * Multivector layout: INTERLEAVED -- element j of row i is M[i*ld + j],
 * ld >= count.  (The reference's Level-1 multivectors put vector j at
 * base + j*pitch, vector.h:75-91; for a sparse product that layout turns one
 * gather per nonzero into `count` gathers from `count` cache lines, while the
 * interleaved rows make it ONE 128-byte line for 16 doubles, and they make a
 * rank's row block of X one contiguous buffer for the all-gather.
 * spgpu?mvInterleave / spgpu?mvDeinterleave convert between the two.)
 *
 * All arrays are device pointers; calls are asynchronous on
 * handle->currentStream.  Z may alias Y exactly.  count <= 0 or rows <= 0 is a
 * no-op.  Per (row, rhs) the products are added in ascending k, i.e. in the
 * order of the reference's one-thread-per-row kernel.
 *
 * In-place sum: with Z == Y and beta == 1 (Z += alpha*A*X) the rows of A that
 * have no entries (rS[i] == 0) are left untouched -- neither read nor written.
 * The second product of a column-split row block (own columns first, the other
 * ranks' columns after the all-gather) is almost entirely such rows.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

void spgpuShellspmm(spgpuHandle_t handle, __device float* Z, const __device float* Y, float alpha,
                    const __device float* cM, const __device int* rP, int hackSize,
                    const __device int* hackOffsets, const __device int* rS, const __device int* rIdx,
                    int avgNnzPerRow, int rows, const __device float* X, float beta, int baseIndex,
                    int count, int ldX, int ldYZ);

void spgpuDhellspmm(spgpuHandle_t handle, __device double* Z, const __device double* Y, double alpha,
                    const __device double* cM, const __device int* rP, int hackSize,
                    const __device int* hackOffsets, const __device int* rS, const __device int* rIdx,
                    int avgNnzPerRow, int rows, const __device double* X, double beta, int baseIndex,
                    int count, int ldX, int ldYZ);

/* dst[i*ld + j] = src[j*pitch + i]  (reference multivector -> interleaved), i < n, j < count */
void spgpuSmvInterleave(spgpuHandle_t handle, __device float* dst, int ld, const __device float* src,
                        int pitch, int n, int count);
void spgpuDmvInterleave(spgpuHandle_t handle, __device double* dst, int ld, const __device double* src,
                        int pitch, int n, int count);
/* dst[j*pitch + i] = src[i*ld + j]  (interleaved -> reference multivector) */
void spgpuSmvDeinterleave(spgpuHandle_t handle, __device float* dst, int pitch, const __device float* src,
                          int ld, int n, int count);
void spgpuDmvDeinterleave(spgpuHandle_t handle, __device double* dst, int pitch, const __device double* src,
                          int ld, int n, int count);

#ifdef __cplusplus
}
#endif
