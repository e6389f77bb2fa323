#pragma once
/*
 * Host-side COO -> HDIA and DIA -> HDIA conversion (CPU, single thread, like
 * the reference).  Replaces hdia_conv.h:20-70 / hdia.cpp:8-349 of the
 * reference, bit for bit.  The blocked (BCOO/BHDIA) converter of that header
 * has no consumer on the path and is not provided.
 * All pointers are HOST pointers.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: hdia_conv.h:20 / hdia.cpp:8-11.  ceil(rowsCount/hackSize). */
int getHdiaHacksCount(int hackSize, int rowsCount);

/* reference: hdia_conv.h:22-30 / hdia.cpp:13-57.  DIA -> HDIA, pass 1: a DIA diagonal is kept in a
 * hack iff any BYTE of its values in the hack's rows is non-zero; hackOffsets gets hacks+1 entries. */
void computeHdiaHackOffsets(int* allocationHeight, int* hackOffsets, int hackSize, const void* diaValues,
                            int diaValuesPitch, int diagonals, int rowsCount, spgpuType_t valuesType);

/* reference: hdia_conv.h:32-43 / hdia.cpp:59-153.  DIA -> HDIA, pass 2: kept diagonals in DIA order,
 * hdiaOffsets = their DIA offsets, values copied for the hack's existing rows (rows past rowsCount in
 * the last hack stay untouched). */
void diaToHdia(void* hdiaValues, int* hdiaOffsets, const int* hackOffsets, int hackSize, const void* diaValues,
               const int* diaOffsets, int diaValuesPitch, int diagonals, int rowsCount, spgpuType_t valuesType);

/* reference: hdia_conv.h:45-55 / hdia.cpp:161-228.
 * hackOffsets gets hacks+1 entries: running count of distinct diagonals per
 * hack; *allocationHeight = hackOffsets[hacks]. */
void computeHdiaHackOffsetsFromCoo(int* allocationHeight, int* hackOffsets, int hackSize, int rowsCount,
                                   int columnsCount, int nonZerosCount, const int* cooRowIndices,
                                   const int* cooColsIndices, int cooBaseIndex);

/* reference: hdia_conv.h:57-70 / hdia.cpp:230-349.
 * Per hack the diagonals are emitted in ascending (column - row) order;
 * hdiaOffsets[d] = column - row; the value of (row, diagonal d) goes to
 * hdiaValues[d*hackSize + row%hackSize].  Only real entries are written:
 * the caller must zero hdiaValues first.  Duplicate (row, col) entries: the
 * last one in COO order wins. */
void cooToHdia(void* hdiaValues, int* hdiaOffsets, const int* hackOffsets, int hackSize, int rowsCount,
               int columnsCount, int nonZerosCount, const int* cooRowIndices, const int* cooColsIndices,
               const void* cooValues, int cooBaseIndex, spgpuType_t valuesType);

#ifdef __cplusplus
}
#endif
