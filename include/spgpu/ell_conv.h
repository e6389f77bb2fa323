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

/* reference: ell_conv.h:64-76 / ell.c:161-202 (+ its merge sort :85-157).  ELL -> "ordered ELL":
 * rows sorted by DESCENDING length; rows of equal length come out in DESCENDING original index
 * (that is what the reference's merge, which takes the right run on ties, produces for every
 * input size except rowsCount == 2, which the reference leaves unsorted -- reproduced).  rIdx[i] = original row stored at position i, dstRs[i] its length; only real
 * entries are copied (callers zero the destination first).  Pass rIdx to spgpu?ellspmv. */
void ellToOell(int* rIdx, void* dstEllValues, int* dstEllIndices, int* dstRs, const void* srcEllValues,
               const int* srcEllIndices, const int* srcRs, int ellValuesPitch, int ellIndicesPitch, int rowsCount,
               spgpuType_t valuesType);

/* NEW (no counterpart in the reference): the order ellToOell computes, on its own and generalised, for callers that
 * build the ordered matrix themselves (e.g. straight from COO into HELL).  rIdx[i] = original row at position i,
 * dstRs[i] = its length.
 *   window <= 0 (or >= rowsCount), longRows <= 0: exactly ellToOell's order.
 *   window  > 0: rows are sorted inside consecutive windows of `window` rows only, so that position i stays within
 *                `window` rows of its original place (x and z keep their locality); windows alternate between
 *                descending and ascending (length, row) order so that rows of similar length meet where two windows meet.
 *   longRows > 0: rows LONGER than longRows are taken out of their windows and come first, sorted inside windows of
 *                their own, SPGPU_OELL_LONG_WINDOW_FACTOR times as large (one window when window <= 0): a handful of very
 *                long rows otherwise sets the depth of one hack per window, and sorted over the whole matrix the 32 rows of
 *                one of their hacks would come from everywhere (every x gather a cache line from memory). */
#define SPGPU_OELL_LONG_WINDOW_FACTOR 32
void oellOrder(int* rIdx, int* dstRs, const int* srcRs, int rowsCount, int window, int longRows);
/* The same with the windows of the shorter rows ALIGNED: when rows are set aside (longRows > 0, window > 0) the others are cut
 * into runs of `window` of them -- counted among themselves, not by original row number -- placed so that every window but the
 * first starts at a multiple of `window` in the new order.  A kernel whose workgroups own `window` consecutive ordered rows
 * then owns exactly one window: its slice of x is one window wide and it writes whole lines of z (with oellOrder's windows a
 * workgroup straddles two).  The SpMV recognises such an order by itself (three sampled blocks of rIdx) and takes its 2 048-row
 * shape for it.  Measured on the north_star target with window 2 048: +6 % on band columns, +7 % on columns
 * spread over +-2 048 of the row, against oellOrder's windows in the same process.  Identical to oellOrder when longRows <= 0
 * or window <= 0; kept as a call of its own because oellOrder's rule (window = original row number / window) is the simpler
 * contract and the one the tests of earlier rounds pin. */
void oellOrderAligned(int* rIdx, int* dstRs, const int* srcRs, int rowsCount, int window, int longRows);

#ifdef __cplusplus
}
#endif
