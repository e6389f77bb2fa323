#pragma once
/*
 * ELLpack SpMV:  z = alpha*A*x + beta*y.
 * Replaces spgpu{S,D,C,Z}ellspmv of the reference (ell.h:46-173, dispatcher
 * kernels/ell_spmv_base.cuh:99-146, launch rules
 * kernels/ell_spmv_base_template.cuh:350-425).
 *
 * Storage (reference: ell.c:33-80): column-major, slot of (row r, k-th entry)
 * = r + k*pitch, pitch in ELEMENTS and given separately for cM and rP.
 *   rS   == NULL : every row iterates maxNnzPerRow slots; padding slots must
 *                  hold coefficient 0.  A padding slot whose (index-baseIndex)
 *                  is negative is skipped instead of reading x[-1] (the
 *                  reference reads out of bounds there,
 *                  ell_spmv_base_nors.cuh:235).
 *   rIdx != NULL : row r of the storage is row rIdx[r] of y and z.
 * z may alias y exactly.  Calls are asynchronous on handle->currentStream.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: ell.h:24 */
#define ELL_PITCH_ALIGN_BYTE 128

/* reference: ell.h:46-61 */
void spgpuSellspmv(spgpuHandle_t handle, __device float* z, const __device float* y, float alpha,
                   const __device float* cM, const __device int* rP, int cMPitch, int rPPitch,
                   const __device int* rS, const __device int* rIdx, int avgNnzPerRow,
                   int maxNnzPerRow, int rows, const __device float* x, float beta, int baseIndex);

/* reference: ell.h:83-98 */
void spgpuDellspmv(spgpuHandle_t handle, __device double* z, const __device double* y, double alpha,
                   const __device double* cM, const __device int* rP, int cMPitch, int rPPitch,
                   const __device int* rS, const __device int* rIdx, int avgNnzPerRow,
                   int maxNnzPerRow, int rows, const __device double* x, double beta, int baseIndex);

/* reference: ell.h:121-136 */
void spgpuCellspmv(spgpuHandle_t handle, __device hipFloatComplex* z, const __device hipFloatComplex* y,
                   hipFloatComplex alpha, const __device hipFloatComplex* cM, const __device int* rP,
                   int cMPitch, int rPPitch, const __device int* rS, const __device int* rIdx,
                   int avgNnzPerRow, int maxNnzPerRow, int rows, const __device hipFloatComplex* x,
                   hipFloatComplex beta, int baseIndex);

/* reference: ell.h:158-173 */
void spgpuZellspmv(spgpuHandle_t handle, __device hipDoubleComplex* z, const __device hipDoubleComplex* y,
                   hipDoubleComplex alpha, const __device hipDoubleComplex* cM, const __device int* rP,
                   int cMPitch, int rPPitch, const __device int* rS, const __device int* rIdx,
                   int avgNnzPerRow, int maxNnzPerRow, int rows, const __device hipDoubleComplex* x,
                   hipDoubleComplex beta, int baseIndex);

/*
 * ELL coefficient update ("csput"): for every i < nnz, find column aJ[i] in row aI[i] - baseIndex by
 * binary search over the row's first rS[row] stored indices (which must ascend) and overwrite that
 * coefficient with aVal[i]; entries that are not found, or whose row is negative, are ignored.
 * As in the reference, aJ is compared with the STORED indices as they are (same base as rP) and
 * `alpha` is accepted but not applied (ell_csput_base.cuh:33-75: SURVEY.md A.3 item 3).
 * reference: ell.h:194-302, kernels/ell_csput_base.cuh:33-125.
 */
void spgpuSellcsput(spgpuHandle_t handle, float alpha, __device float* cM, __device const int* rP, int cMPitch,
                    int rPPitch, __device const int* rS, int nnz, __device int* aI, __device int* aJ,
                    __device float* aVal, int baseIndex);
void spgpuDellcsput(spgpuHandle_t handle, double alpha, __device double* cM, __device const int* rP, int cMPitch,
                    int rPPitch, __device const int* rS, int nnz, __device int* aI, __device int* aJ,
                    __device double* aVal, int baseIndex);
void spgpuCellcsput(spgpuHandle_t handle, hipFloatComplex alpha, __device hipFloatComplex* cM, __device const int* rP,
                    int cMPitch, int rPPitch, __device const int* rS, int nnz, __device int* aI, __device int* aJ,
                    __device hipFloatComplex* aVal, int baseIndex);
void spgpuZellcsput(spgpuHandle_t handle, hipDoubleComplex alpha, __device hipDoubleComplex* cM, __device const int* rP,
                    int cMPitch, int rPPitch, __device const int* rS, int nnz, __device int* aI, __device int* aJ,
                    __device hipDoubleComplex* aVal, int baseIndex);

#ifdef __cplusplus
}
#endif
