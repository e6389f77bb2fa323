#pragma once
/*
 * HDIA (sliced / "hacked" diagonal) SpMV:  z = alpha*A*x + beta*y.
 * Replaces spgpu{S,D,C,Z}hdiaspmv of the reference (hdia.h:37-142,
 * dispatcher kernels/hdia_spmv_base.cuh:99-145, kernel
 * kernels/hdia_spmv_base_template.cuh:19-206).
 *
 * Storage (reference: hdia.cpp:161-324):
 *   hacks = ceil(rows/hackSize); hackOffsets has hacks+1 entries, the last
 *   one being the total number of stored diagonals H.
 *   For hack h and its p-th diagonal (d = hackOffsets[h] + p):
 *     offsets[d]                    = column - row of that diagonal
 *     dM[d*hackSize + r%hackSize]   = coefficient of row r on it
 *   A stored slot contributes iff 0 <= offsets[d] + r < cols.
 * z may alias y exactly.  Calls are asynchronous on handle->currentStream.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: hdia.h:37-49 */
void spgpuShdiaspmv(spgpuHandle_t handle, float* z, const float* y, float alpha, const float* dM,
                    const int* offsets, int hackSize, const int* hackOffsets, int rows, int cols,
                    const float* x, float beta);

/* reference: hdia.h:68-80 */
void spgpuDhdiaspmv(spgpuHandle_t handle, double* z, const double* y, double alpha, const double* dM,
                    const int* offsets, int hackSize, const int* hackOffsets, int rows, int cols,
                    const double* x, double beta);

/* reference: hdia.h:99-111 */
void spgpuChdiaspmv(spgpuHandle_t handle, hipFloatComplex* z, const hipFloatComplex* y,
                    hipFloatComplex alpha, const hipFloatComplex* dM, const int* offsets, int hackSize,
                    const int* hackOffsets, int rows, int cols, const hipFloatComplex* x,
                    hipFloatComplex beta);

/* reference: hdia.h:130-142 */
void spgpuZhdiaspmv(spgpuHandle_t handle, hipDoubleComplex* z, const hipDoubleComplex* y,
                    hipDoubleComplex alpha, const hipDoubleComplex* dM, const int* offsets, int hackSize,
                    const int* hackOffsets, int rows, int cols, const hipDoubleComplex* x,
                    hipDoubleComplex beta);

#ifdef __cplusplus
}
#endif
