#pragma once
/*
 * Host-side COO -> HDIA conversion (CPU, single thread, like the reference).
 * Replaces the COO entry points of hdia_conv.h:20,45-70 / hdia.cpp:8-11,
 * 161-349 of the reference, bit for bit.  The DIA -> HDIA and blocked
 * (BCOO/BHDIA) converters of that header are a later scope row.
 * All pointers are HOST pointers.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: hdia_conv.h:20 / hdia.cpp:8-11.  ceil(rowsCount/hackSize). */
int getHdiaHacksCount(int hackSize, int rowsCount);

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
