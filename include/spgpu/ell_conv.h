#pragma once
/*
 * Host-side COO -> ELL conversion (CPU, single thread, like the reference).
 * Replaces ell_conv.h:30-62 / ell.c:5-80 of the reference, bit for bit.
 * All pointers are HOST pointers.
 */
#include "ell.h"
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: ell_conv.h:30-37 / ell.c:5-31 (the misspelling is the ABI).
 * ellRowLengths[r] = number of COO entries of row r; *ellMaxRowSize = max. */
void computeEllRowLenghts(int* ellRowLengths, int* ellMaxRowSize, int rowsCount, int nonZerosCount,
                          const int* cooRowIndices, int cooBaseIndex);

/* reference: ell_conv.h:39 / ell.c:33-37.  rowsCount rounded up to 32. */
int computeEllAllocPitch(int rowsCount);

/* reference: ell_conv.h:42-56 / ell.c:39-80.  Entry k of a row is the k-th
 * COO entry of that row in encounter order; stored index = col - cooBaseIndex
 * + ellBaseIndex.  Slots that receive no entry are left untouched (callers
 * zero the arrays first). */
void cooToEll(void* ellValues, int* ellIndices, int ellValuesPitch, int ellIndicesPitch,
              int ellMaxRowSize, int ellBaseIndex, int rowsCount, int nonZerosCount,
              const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
              int cooBaseIndex, spgpuType_t valuesType);

#ifdef __cplusplus
}
#endif
