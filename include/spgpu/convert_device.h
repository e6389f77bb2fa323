#pragma once
/*
 * Device-side format construction: COO -> ELL, COO -> HELL, COO -> HDIA and COO -> DIA entirely in HBM.
 * NEW: the reference converts on ONE host thread
 * (ell.c:39-80, hell.c:46-104) and uploads; for the 320 M nonzeros of BASELINE
 * configs[1] that is seconds of CPU time and a 3.8 GB PCIe copy per matrix.
 * These routines produce arrays that are BYTE-IDENTICAL to what
 * computeEllRowLenghts / cooToEll / computeHellAllocSize / ellToHell produce for
 * the same COO input (any entry order, duplicates kept, k-th entry of a row =
 * k-th occurrence in COO order), given destination arrays zeroed by the caller.
 *
 * All array arguments are DEVICE pointers unless marked host.  The calls run on
 * handle->currentStream; the two "plan" calls return host scalars and therefore
 * synchronise that stream.  `work` is caller-provided scratch of
 * spgpuCooConvertWorkBytes(rows, nnz) bytes.
 *
 * Method: ONE stable radix sort of (row, entry id) pairs groups the entries by row in encounter order (rocPRIM;
 * format construction, not the SpMV path); row starts by binary search in the sorted rows; an entry's position k
 * inside its row is its sorted position minus the row's start.  No atomics: the arrays do not depend on scheduling,
 * and a row of any length costs what its entries cost.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of scratch the conversions below need (about 16 bytes per nonzero; 0 if no GPU can be asked for rocPRIM's share). */
size_t spgpuCooConvertWorkBytes(int rowsCount, int nonZerosCount);

/* Pass 1 (device counterpart of computeEllRowLenghts, ell.c:5-31): fills rowLengths[rowsCount] on the device,
 * returns the longest row in *maxRowSize (host).  Leaves in `work` the per-row buckets the fill calls use.
 * Returns SPGPU_UNSUPPORTED if a COO row index lies outside [cooBaseIndex, cooBaseIndex + rowsCount). */
spgpuStatus_t spgpuCooRowLengthsDevice(spgpuHandle_t handle, __device int* rowLengths, __host int* maxRowSize,
                                       int rowsCount, int nonZerosCount, const __device int* cooRowIndices,
                                       int cooBaseIndex, __device void* work);

/* Pass 2a (device counterpart of cooToEll, ell.c:39-80).  Needs `work` as left by spgpuCooRowLengthsDevice for the
 * same COO arrays.  elementSize is spgpuSizeOf(type): 4, 8 or 16. */
spgpuStatus_t spgpuCooToEllDevice(spgpuHandle_t handle, __device void* ellValues, __device int* ellIndices,
                                  int ellValuesPitch, int ellIndicesPitch, int ellBaseIndex, int rowsCount,
                                  int nonZerosCount, const __device int* cooRowIndices,
                                  const __device int* cooColsIndices, const __device void* cooValues,
                                  int cooBaseIndex, spgpuType_t valuesType, const __device int* rowLengths,
                                  __device void* work);

/* HELL plan (device counterpart of computeHellAllocSize + the hackOffsets part of ellToHell, hell.c:4-44,64-100):
 * fills hackOffsets[ceil(rows/hackSize)] on the device, returns allocationHeight (slots = hackSize * height) on the host. */
spgpuStatus_t spgpuHellPlanDevice(spgpuHandle_t handle, __host int* allocationHeight, __device int* hackOffsets,
                                  int hackSize, int rowsCount, const __device int* rowLengths, __device void* work);

/* Pass 2b: COO -> HELL directly (equals cooToEll followed by ellToHell).  Needs `work` as left by
 * spgpuCooRowLengthsDevice and hackOffsets from spgpuHellPlanDevice. */
spgpuStatus_t spgpuCooToHellDevice(spgpuHandle_t handle, __device void* hellValues, __device int* hellIndices,
                                   const __device int* hackOffsets, int hackSize, int hellBaseIndex, int rowsCount,
                                   int nonZerosCount, const __device int* cooRowIndices,
                                   const __device int* cooColsIndices, const __device void* cooValues,
                                   int cooBaseIndex, spgpuType_t valuesType, const __device int* rowLengths,
                                   __device void* work);

/* ---- COO -> HDIA ---------------------------------------------------------------------------------------------
 * Device counterparts of computeHdiaHackOffsetsFromCoo and cooToHdia (hdia.cpp:161-228, 230-349): same arrays,
 * byte for byte, for the same COO input (any entry order; of several entries with one (row, column) the LAST in
 * COO order is stored, as in the reference's in-order memcpy), hdiaValues zeroed by the caller.
 * The plan sorts one 64-bit (hack, diagonal) key per entry (rocPRIM radix sort) and keeps the distinct ones. */

/* Bytes of device scratch for spgpuCooHdiaPlanDevice (0 if no GPU can be asked for rocPRIM's share). */
size_t spgpuCooHdiaPlanWorkBytes(int rowsCount, int nonZerosCount);

/* Fills hackOffsets[hacks + 1] on the device (hacks = getHdiaHacksCount(hackSize, rowsCount)) and returns the number
 * of stored diagonals in *allocationHeight (host); synchronises the stream.  Leaves the diagonal list in `work` for
 * spgpuCooToHdiaDevice.  SPGPU_UNSUPPORTED if an entry lies outside the rowsCount x columnsCount matrix. */
spgpuStatus_t spgpuCooHdiaPlanDevice(spgpuHandle_t handle, __host int* allocationHeight, __device int* hackOffsets,
                                     int hackSize, int rowsCount, int columnsCount, int nonZerosCount,
                                     const __device int* cooRowIndices, const __device int* cooColsIndices,
                                     int cooBaseIndex, __device void* work);

/* Bytes of the second scratch area of spgpuCooToHdiaDevice: one int per HDIA slot (hackSize * allocationHeight). */
size_t spgpuCooToHdiaScratchBytes(int hackSize, int allocationHeight);

/* Fills hdiaValues[hackSize * allocationHeight] (zeroed by the caller) and hdiaOffsets[allocationHeight].
 * `work` as left by spgpuCooHdiaPlanDevice for the same COO arrays; `scratch` of spgpuCooToHdiaScratchBytes bytes. */
spgpuStatus_t spgpuCooToHdiaDevice(spgpuHandle_t handle, __device void* hdiaValues, __device int* hdiaOffsets,
                                   const __device int* hackOffsets, int hackSize, int rowsCount, int columnsCount,
                                   int nonZerosCount, const __device int* cooRowIndices,
                                   const __device int* cooColsIndices, const __device void* cooValues,
                                   int cooBaseIndex, spgpuType_t valuesType, int allocationHeight,
                                   const __device void* work, __device void* scratch);

/* ---- COO -> DIA ---------------------------------------------------------------------------------------------
 * Device counterparts of computeDiaDiagonalsCount and coo2dia (dia.c:11-104): offsets = the distinct (column - row)
 * values ascending, value of an entry at (row - base) + position * valuesPitch, last duplicate wins, `values` zeroed
 * by the caller.  Unlike the host call the plan needs cooBaseIndex only to validate the entries. */
size_t spgpuCooDiaWorkBytes(int rowsCount, int columnsCount);

/* Returns the number of stored diagonals in *diagonals (host); synchronises the stream; leaves the diagonal
 * numbering in `work`.  SPGPU_UNSUPPORTED if an entry lies outside the matrix. */
spgpuStatus_t spgpuCooDiaPlanDevice(spgpuHandle_t handle, __host int* diagonals, int rowsCount, int columnsCount,
                                    int nonZerosCount, const __device int* cooRowIndices,
                                    const __device int* cooColsIndices, int cooBaseIndex, __device void* work);

/* One int per DIA slot (valuesPitch * diagonals). */
size_t spgpuCooToDiaScratchBytes(int valuesPitch, int diagonals);

spgpuStatus_t spgpuCooToDiaDevice(spgpuHandle_t handle, __device void* values, __device int* offsets, int valuesPitch,
                                  int diagonals, int rowsCount, int columnsCount, int nonZerosCount,
                                  const __device int* cooRowIndices, const __device int* cooColsIndices,
                                  const __device void* cooValues, int cooBaseIndex, spgpuType_t valuesType,
                                  const __device void* work, __device void* scratch);

#ifdef __cplusplus
}
#endif
