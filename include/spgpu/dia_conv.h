#pragma once
/*
 * Host-side COO -> DIA conversion (CPU, single thread, like the reference).
 * Replaces dia_conv.h:20-49 / dia.c:5-104 of the reference, bit for bit.
 * All pointers are HOST pointers.
 */
#include "dia.h"
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: dia_conv.h:20-25 / dia.c:11-39.  Number of distinct (column - row) values. */
int computeDiaDiagonalsCount(int rowsCount, int columnsCount, int nonZerosCount, const int* cooRowIndices,
                             const int* cooColsIndices);

/* reference: dia_conv.h:28-40 / dia.c:41-104.  offsets[] gets the distinct (column - row) values in
 * ascending order; the value of an entry goes to values[(row - cooBaseIndex) + position*valuesPitch].
 * Only real entries are written (callers zero `values` first); duplicates: the last one wins. */
void coo2dia(void* values, int* offsets, int valuesPitch, int diagonals, int rowsCount, int columnsCount,
             int nonZerosCount, const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
             int cooBaseIndex, spgpuType_t valuesType);

/* reference: dia_conv.h:43 / dia.c:5-9.  rowsCount rounded up to 32. */
int computeDiaAllocPitch(int rowsCount);

#ifdef __cplusplus
}
#endif
