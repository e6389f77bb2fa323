#pragma once
/*
 * DIA (diagonal) SpMV:  z = alpha*A*x + beta*y.   SURVEY.md section 8, row f3.
 * Replaces spgpu{S,D,C,Z}diaspmv of the reference (dia.h:42-160, dispatcher
 * kernels/dia_spmv_base.cuh:99-143, kernel kernels/dia_spmv_base_template.cuh:20-216).
 *
 * Storage (reference: dia.c:5-104): `diags` stored diagonals in ascending
 * (column - row); offsets[d] = column - row; value of (row r, diagonal d) at
 * dM[r + d*dMPitch], dMPitch in elements.  A slot contributes iff
 * 0 <= offsets[d] + r < cols.  z may alias y exactly; calls are asynchronous on
 * handle->currentStream.  On MI355X this is the HDIA kernel run over one
 * all-rows hack (the addressing is identical with hackSize = dMPitch).
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: dia.h:24 */
#define DIA_PITCH_ALIGN_BYTE 128

/* reference: dia.h:42-53 */
void spgpuSdiaspmv(spgpuHandle_t handle, __device float* z, const __device float* y, float alpha,
                   const __device float* dM, const __device int* offsets, int dMPitch, int rows, int cols, int diags,
                   const __device float* x, float beta);
/* reference: dia.h:72-83 */
void spgpuDdiaspmv(spgpuHandle_t handle, __device double* z, const __device double* y, double alpha,
                   const __device double* dM, const __device int* offsets, int dMPitch, int rows, int cols, int diags,
                   const __device double* x, double beta);
/* reference: dia.h:102-113 */
void spgpuCdiaspmv(spgpuHandle_t handle, __device hipFloatComplex* z, const __device hipFloatComplex* y,
                   hipFloatComplex alpha, const __device hipFloatComplex* dM, const __device int* offsets, int dMPitch,
                   int rows, int cols, int diags, const __device hipFloatComplex* x, hipFloatComplex beta);
/* reference: dia.h:132-143 */
void spgpuZdiaspmv(spgpuHandle_t handle, __device hipDoubleComplex* z, const __device hipDoubleComplex* y,
                   hipDoubleComplex alpha, const __device hipDoubleComplex* dM, const __device int* offsets, int dMPitch,
                   int rows, int cols, int diags, const __device hipDoubleComplex* x, hipDoubleComplex beta);

#ifdef __cplusplus
}
#endif
