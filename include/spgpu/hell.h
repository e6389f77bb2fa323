#pragma once
/*
 * HELL (sliced / "hacked" ELLpack) SpMV:  z = alpha*A*x + beta*y.
 * Replaces spgpu{S,D,C,Z}hellspmv of the reference (hell.h:45-169, dispatcher
 * kernels/hell_spmv_base.cuh:103-157).  Argument order and meaning are the
 * reference's; all array arguments are device pointers owned by the caller.
 *
 * Storage (reference: hell.c:46-104):
 *   hacks          = ceil(rows / hackSize), hackSize a multiple of 32
 *   hackOffsets[h] = first slot of hack h (hacks entries, no trailing total)
 *   slot of (row r, k-th entry) = hackOffsets[r/hackSize] + r%hackSize + k*hackSize
 *   cM[slot] coefficient, rP[slot] column index (+baseIndex), rS[r] row length
 *   rIdx (optional): row r of the storage is row rIdx[r] of y and z
 * Padding slots (k >= rS[r]) are never dereferenced as column indices.
 * z may alias y exactly.  Calls are asynchronous on handle->currentStream.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: hell.h:24 */
#define HELL_PITCH_ALIGN_BYTE 128

/* reference: hell.h:45-59 */
void spgpuShellspmv(spgpuHandle_t handle, __device float* z, const __device float* y, float alpha,
                    const __device float* cM, const __device int* rP, int hackSize,
                    const __device int* hackOffsets, const __device int* rS, const __device int* rIdx,
                    int avgNnzPerRow, int rows, const __device float* x, float beta, int baseIndex);

/* reference: hell.h:82-96 */
void spgpuDhellspmv(spgpuHandle_t handle, __device double* z, const __device double* y, double alpha,
                    const __device double* cM, const __device int* rP, int hackSize,
                    const __device int* hackOffsets, const __device int* rS, const __device int* rIdx,
                    int avgNnzPerRow, int rows, const __device double* x, double beta, int baseIndex);

/* reference: hell.h:118-132 */
void spgpuChellspmv(spgpuHandle_t handle, __device hipFloatComplex* z, const __device hipFloatComplex* y,
                    hipFloatComplex alpha, const __device hipFloatComplex* cM, const __device int* rP,
                    int hackSize, const __device int* hackOffsets, const __device int* rS,
                    const __device int* rIdx, int avgNnzPerRow, int rows,
                    const __device hipFloatComplex* x, hipFloatComplex beta, int baseIndex);

/* reference: hell.h:155-169 */
void spgpuZhellspmv(spgpuHandle_t handle, __device hipDoubleComplex* z, const __device hipDoubleComplex* y,
                    hipDoubleComplex alpha, const __device hipDoubleComplex* cM, const __device int* rP,
                    int hackSize, const __device int* hackOffsets, const __device int* rS,
                    const __device int* rIdx, int avgNnzPerRow, int rows,
                    const __device hipDoubleComplex* x, hipDoubleComplex beta, int baseIndex);

#ifdef __cplusplus
}
#endif
