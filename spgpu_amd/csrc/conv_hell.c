/*
 * ELL -> HELL on the host.  Own implementation of the behaviour specified by
 * the reference's src/core/hell.c:4-104 (see include/spgpu/hell_conv.h);
 * output arrays are byte-identical to the reference's for the same input.
 */
#include "spgpu/hell_conv.h"

#include <stdint.h>

static int longestRowInHack(const int* rowLengths, int firstRow, int endRow)
{
    int longest = 0;
    for (int r = firstRow; r < endRow; ++r)
        if (rowLengths[r] > longest)
            longest = rowLengths[r];
    return longest;
}

void computeHellAllocSize(int* allocationHeight, int hackSize, int rowsCount, const int* ellRowLengths)
{
    int height = 0;
    for (int first = 0; first < rowsCount; first += hackSize) {
        const int end = first + hackSize < rowsCount ? first + hackSize : rowsCount;
        height += longestRowInHack(ellRowLengths, first, end);
    }
    *allocationHeight = height;
}

/* Copy the real entries of one hack, slab column by slab column, so that the
 * HELL side is written front to back. */
#define SPGPU_ELL_TO_HELL_HACK(ELEM_T)                                                      \
    do {                                                                                    \
        const ELEM_T* src = (const ELEM_T*)ellValues;                                       \
        ELEM_T* dst = (ELEM_T*)hellValues + slab;                                           \
        for (int k = 0; k < depth; ++k) {                                                   \
            const size_t srcVal = (size_t)k * (size_t)ellValuesPitch;                       \
            const size_t srcIdx = (size_t)k * (size_t)ellIndicesPitch;                      \
            const size_t out = (size_t)k * (size_t)hackSize;                                \
            for (int r = first; r < end; ++r) {                                             \
                if (k < ellRowLengths[r]) {                                                 \
                    dst[out + (size_t)(r - first)] = src[srcVal + (size_t)r];               \
                    hellIndices[slab + out + (size_t)(r - first)] = ellIndices[srcIdx + (size_t)r]; \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    } while (0)

typedef struct { uint64_t lo, hi; } spgpu_bits128;

void ellToHell(void* hellValues, int* hellIndices, int* hackOffsets, int hackSize,
               const void* ellValues, const int* ellIndices, int ellValuesPitch, int ellIndicesPitch,
               int* ellRowLengths, int rowsCount, spgpuType_t valuesType)
{
    const size_t elem = spgpuSizeOf(valuesType);
    size_t slab = 0; /* first slot of the current hack */
    int hack = 0;
    for (int first = 0; first < rowsCount; first += hackSize, ++hack) {
        const int end = first + hackSize < rowsCount ? first + hackSize : rowsCount;
        const int depth = longestRowInHack(ellRowLengths, first, end);
        hackOffsets[hack] = (int)slab;
        switch (elem) {
        case 4:  SPGPU_ELL_TO_HELL_HACK(uint32_t); break;
        case 8:  SPGPU_ELL_TO_HELL_HACK(uint64_t); break;
        case 16: SPGPU_ELL_TO_HELL_HACK(spgpu_bits128); break;
        default: break;
        }
        slab += (size_t)hackSize * (size_t)depth;
    }
}
