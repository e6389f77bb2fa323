#pragma once
/*
 * Host-side ELL -> HELL conversion (CPU, single thread, like the reference).
 * Replaces hell_conv.h:29-63 / hell.c:4-104 of the reference, bit for bit.
 * All pointers are HOST pointers.
 */
#include "hell.h"
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: hell_conv.h:29-34 / hell.c:4-44.
 * *allocationHeight = sum over hacks of the longest row in the hack;
 * the HELL arrays hold hackSize * allocationHeight slots. */
void computeHellAllocSize(int* allocationHeight, int hackSize, int rowsCount, const int* ellRowLengths);

/* reference: hell_conv.h:36-48 / hell.c:46-104.
 * hackOffsets gets ceil(rowsCount/hackSize) entries (no trailing total).
 * Only real entries (k < ellRowLengths[row]) are written; padding slots keep
 * whatever the caller's buffers held. */
void ellToHell(void* hellValues, int* hellIndices, int* hackOffsets, int hackSize,
               const void* ellValues, const int* ellIndices, int ellValuesPitch, int ellIndicesPitch,
               int* ellRowLengths, int rowsCount, spgpuType_t valuesType);

#ifdef __cplusplus
}
#endif
