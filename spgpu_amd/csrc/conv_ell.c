/*
 * COO -> ELL on the host.  Own implementation of the behaviour specified by
 * the reference's src/core/ell.c:5-80 (see include/spgpu/ell_conv.h); output
 * arrays are byte-identical to the reference's for the same input.
 */
#include "spgpu/ell_conv.h"

#include <stdint.h>
#include <stdlib.h>

void computeEllRowLenghts(int* ellRowLengths, int* ellMaxRowSize, int rowsCount, int nonZerosCount,
                          const int* cooRowIndices, int cooBaseIndex)
{
    memset(ellRowLengths, 0, (size_t)(rowsCount > 0 ? rowsCount : 0) * sizeof(int));
    for (int e = 0; e < nonZerosCount; ++e)
        ellRowLengths[cooRowIndices[e] - cooBaseIndex] += 1;

    int longest = 0;
    for (int r = 0; r < rowsCount; ++r)
        if (ellRowLengths[r] > longest)
            longest = ellRowLengths[r];
    *ellMaxRowSize = longest;
}

int computeEllAllocPitch(int rowsCount)
{
    return (rowsCount + 31) & ~31;
}

/* One scatter loop per element width keeps the inner loop free of memcpy calls. */
#define SPGPU_COO_TO_ELL_LOOP(ELEM_T)                                                    \
    do {                                                                                 \
        ELEM_T* dst = (ELEM_T*)ellValues;                                                \
        const ELEM_T* src = (const ELEM_T*)cooValues;                                    \
        for (int e = 0; e < nonZerosCount; ++e) {                                        \
            const int r = cooRowIndices[e] - cooBaseIndex;                               \
            const size_t k = (size_t)fill[r]++;                                          \
            ellIndices[(size_t)r + k * (size_t)ellIndicesPitch] = cooColsIndices[e] + shift; \
            dst[(size_t)r + k * (size_t)ellValuesPitch] = src[e];                        \
        }                                                                                \
    } while (0)

typedef struct { uint64_t lo, hi; } spgpu_bits128;

void cooToEll(void* ellValues, int* ellIndices, int ellValuesPitch, int ellIndicesPitch,
              int ellMaxRowSize, int ellBaseIndex, int rowsCount, int nonZerosCount,
              const int* cooRowIndices, const int* cooColsIndices, const void* cooValues,
              int cooBaseIndex, spgpuType_t valuesType)
{
    (void)ellMaxRowSize;
    const size_t elem = spgpuSizeOf(valuesType);
    const int shift = ellBaseIndex - cooBaseIndex;
    int* fill = (int*)calloc((size_t)(rowsCount > 0 ? rowsCount : 1), sizeof(int));
    if (!fill)
        return;

    switch (elem) {
    case 4:  SPGPU_COO_TO_ELL_LOOP(uint32_t); break;
    case 8:  SPGPU_COO_TO_ELL_LOOP(uint64_t); break;
    case 16: SPGPU_COO_TO_ELL_LOOP(spgpu_bits128); break;
    default: break; /* unknown type code: nothing is written */
    }
    free(fill);
}
