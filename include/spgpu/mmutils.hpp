#pragma once
/*
 * Unfolding of symmetric Matrix Market matrices (only one triangle is stored in the file).
 * Same templates as the reference's src/utils/mmutils.hpp:10-62: entries whose value is exactly zero are
 * dropped, diagonal entries are kept once, off-diagonal entries are emitted as (r,c) followed by (c,r).
 */

/* reference: mmutils.hpp:10-26.  ADDS to *unfoldedNonZerosCount (callers initialise it to 0, hellPerf.cpp:97-99). */
template <typename T>
void getUnfoldedMmSymmetricSize(int* unfoldedNonZerosCount, T* value, int* rows, int* cols, int nonZerosCount)
{
    int extra = 0;
    for (int e = 0; e < nonZerosCount; ++e)
        if (value[e] != 0)
            extra += rows[e] == cols[e] ? 1 : 2;
    *unfoldedNonZerosCount += extra;
}

/* reference: mmutils.hpp:28-62 */
template <typename T>
void unfoldMmSymmetricReal(int* unfoldedRows, int* unfoldedCols, T* unfoldedValues, int* rows, int* cols, T* values,
                           int nonZerosCount)
{
    int out = 0;
    for (int e = 0; e < nonZerosCount; ++e) {
        if (!(values[e] != 0))
            continue;
        unfoldedRows[out] = rows[e];
        unfoldedCols[out] = cols[e];
        unfoldedValues[out++] = values[e];
        if (rows[e] != cols[e]) {
            unfoldedRows[out] = cols[e];
            unfoldedCols[out] = rows[e];
            unfoldedValues[out++] = values[e];
        }
    }
}
