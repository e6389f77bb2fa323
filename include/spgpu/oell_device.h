#pragma once
/*
 * Rows ordered by length, on the device: the step between a ragged matrix and a HELL matrix without padding.
 *
 * The reference does this on one host thread (ellToOell, ell.c:85-202, driven by hellPerf.cpp:324-378 and
 * cusparsePerf.cpp:440-459): rows sorted by descending length, the permutation handed to the SpMV kernels as
 * rIdx.  A HELL hack is as deep as its longest row, so with power-law row lengths (BASELINE configs[2], the
 * north_star target) the plain format stores 5 slots per nonzero; after the sort neighbouring rows have equal
 * lengths and the format stores 1.0-1.1.  These calls compute the same order in HBM -- plus a windowed form that
 * keeps every row within `window` rows of its original place, so that x and z keep their locality (a global
 * sort turns a banded matrix into one with scattered columns) -- and apply it to ELL arrays or to the row
 * indices of a COO matrix that then goes through spgpuCooToHellDevice (convert_device.h).
 *
 * All array arguments are DEVICE pointers; calls run on handle->currentStream and do not synchronise.
 * The order is the one oellOrder (ell_conv.h) defines; for window <= 0 and longRows <= 0 it is exactly
 * ellToOell's, rIdx and dstRs byte for byte.  The sort itself is rocPRIM's radix sort (format construction,
 * not the SpMV path).
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch spgpuOellOrderDevice / spgpuEllToOellDevice need (0 if no GPU can be asked). */
size_t spgpuOellOrderWorkBytes(int rowsCount);

/* rIdx[i] = original row at position i, dstRs[i] = srcRs[rIdx[i]]; window / longRows as oellOrder (ell_conv.h). */
spgpuStatus_t spgpuOellOrderDevice(spgpuHandle_t handle, __device int* rIdx, __device int* dstRs,
                                   const __device int* srcRs, int rowsCount, int window, int longRows,
                                   __device void* work);

/* The aligned form (oellOrderAligned, ell_conv.h): same arguments and scratch; the windows of the rows that are not set
 * aside are runs of `window` of them, every one but the first starting at a multiple of `window` in the new order. */
spgpuStatus_t spgpuOellOrderAlignedDevice(spgpuHandle_t handle, __device int* rIdx, __device int* dstRs,
                                          const __device int* srcRs, int rowsCount, int window, int longRows,
                                          __device void* work);

/* Device counterpart of ellToOell (ell_conv.h; reference ell.c:161-202) with the two extra parameters of oellOrder:
 * order, then copy of the real entries (destination arrays zeroed by the caller).  Byte-identical to the host call
 * for window <= 0, longRows <= 0. */
spgpuStatus_t spgpuEllToOellDevice(spgpuHandle_t handle, __device int* rIdx, __device void* dstEllValues,
                                   __device int* dstEllIndices, __device int* dstRs, const __device void* srcEllValues,
                                   const __device int* srcEllIndices, const __device int* srcRs, int ellValuesPitch,
                                   int ellIndicesPitch, int rowsCount, spgpuType_t valuesType, int window, int longRows,
                                   __device void* work);

/* The COO route: dstCooRowIndices[e] = new position of row srcCooRowIndices[e] (same base), i.e. the inverse of rIdx
 * applied to every entry; `inverse` is rowsCount ints of scratch (left holding the inverse permutation).  The result,
 * with the untouched column and value arrays, is the COO form of the ordered matrix: feed it to
 * spgpuCooRowLengthsDevice / spgpuHellPlanDevice / spgpuCooToHellDevice.  Entries keep their order inside a row.
 * dst may alias src. */
spgpuStatus_t spgpuCooPermuteRowsDevice(spgpuHandle_t handle, __device int* dstCooRowIndices,
                                        const __device int* srcCooRowIndices, int nonZerosCount,
                                        const __device int* rIdx, int rowsCount, int cooBaseIndex, __device int* inverse);

#ifdef __cplusplus
}
#endif
