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

/* How many distinct (column - row) values the entries have (dia_conv.h:20-25 / dia.c:11-39). */
int computeDiaDiagonalsCount(int rows, int cols, int nnz, const int* cooRows, const int* cooCols);

/* offsets[] receives those values in ascending order; an entry's value goes to
 * values[(row - cooBase) + position*pitch] (dia_conv.h:28-40 / dia.c:41-104).  Only real entries are written
 * (callers zero `values` first); of duplicates the last one wins. */
void coo2dia(void* values, int* offsets, int pitch, int diagonals, int rows, int cols, int nnz, const int* cooRows,
             const int* cooCols, const void* cooValues, int cooBase, spgpuType_t type);

/* rows rounded up to a multiple of 32 (dia_conv.h:43 / dia.c:5-9). */
int computeDiaAllocPitch(int rows);

#ifdef __cplusplus
}
#endif
