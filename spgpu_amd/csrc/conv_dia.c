/*
 * COO -> DIA on the host.  Own implementation of the behaviour specified by the reference's
 * src/core/dia.c:5-104 (see include/spgpu/dia_conv.h); output arrays are byte-identical to the
 * reference's for the same input.
 */
#include "spgpu/dia_conv.h"

#include <stdint.h>
#include <stdlib.h>

int computeDiaAllocPitch(int rowsCount)
{
    return (rowsCount + 31) & ~31;
}

/* One flag per possible diagonal, indexed by (column - row) + rowsCount - 1. */
static unsigned char* markDiagonals(int rowsCount, int columnsCount, int nonZerosCount, const int* cooRows,
                                    const int* cooCols)
{
    const size_t span = (size_t)rowsCount + (size_t)columnsCount - 1;
    unsigned char* present = (unsigned char*)calloc(span ? span : 1, 1);
    if (!present)
        return NULL;
    for (int e = 0; e < nonZerosCount; ++e)
        present[(size_t)(rowsCount - 1 + cooCols[e] - cooRows[e])] = 1;
    return present;
}

int computeDiaDiagonalsCount(int rowsCount, int columnsCount, int nonZerosCount, const int* cooRowIndices,
                             const int* cooColsIndices)
{
    unsigned char* present = markDiagonals(rowsCount, columnsCount, nonZerosCount, cooRowIndices, cooColsIndices);
    if (!present)
        return 0;
    const size_t span = (size_t)rowsCount + (size_t)columnsCount - 1;
    int count = 0;
    for (size_t d = 0; d < span; ++d)
        count += present[d];
    free(present);
    return count;
}

void coo2dia(void* values, int* offsets, int valuesPitch, int diagonals, int rowsCount, int columnsCount,
             int nonZerosCount, const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
             int cooBaseIndex, spgpuType_t valuesType)
{
    (void)diagonals;
    const size_t elem = spgpuSizeOf(valuesType);
    const size_t span = (size_t)rowsCount + (size_t)columnsCount - 1;
    unsigned char* present = markDiagonals(rowsCount, columnsCount, nonZerosCount, cooRowIndices, cooColsIndices);
    int* slotOf = (int*)malloc((span ? span : 1) * sizeof(int));
    if (!present || !slotOf) {
        free(present);
        free(slotOf);
        return;
    }
    int next = 0;
    for (size_t d = 0; d < span; ++d) {
        if (present[d]) {
            offsets[next] = (int)d - (rowsCount - 1);
            slotOf[d] = next++;
        }
    }
    for (int e = 0; e < nonZerosCount; ++e) {
        const size_t d = (size_t)(rowsCount - 1 + cooColsIndices[e] - cooRowIndices[e]);
        const size_t at = (size_t)(cooRowIndices[e] - cooBaseIndex) + (size_t)slotOf[d] * (size_t)valuesPitch;
        memcpy((char*)values + at * elem, (const char*)cooValues + (size_t)e * elem, elem);
    }
    free(present);
    free(slotOf);
}
