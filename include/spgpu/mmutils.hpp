#pragma once
/*
 * Unfolding of symmetric Matrix Market matrices (only one triangle is stored in the file).
 * Same templates as the reference's src/utils/mmutils.hpp:10-62: entries whose value is exactly zero are
 * dropped, diagonal entries are kept once, off-diagonal entries are emitted as (r,c) followed by (c,r).
 */

/* Entries the unfolded matrix will have (mmutils.hpp:10-26).  ADDS to *total: callers start it at 0
 * (hellPerf.cpp:97-99). */
template <typename T> void getUnfoldedMmSymmetricSize(int* total, T* vals, int* ri, int* ci, int nnz)
{
    int more = 0;
    for (int e = 0; e < nnz; ++e) {
        if (vals[e] == 0)
            continue;
        more += ri[e] == ci[e] ? 1 : 2;
    }
    *total += more;
}

/* The unfolding itself (mmutils.hpp:28-62): (r,c) then, off the diagonal, (c,r) with the same value. */
template <typename T> void unfoldMmSymmetricReal(int* outRi, int* outCi, T* outVals, int* ri, int* ci, T* vals, int nnz)
{
    int at = 0;
    for (int e = 0; e < nnz; ++e) {
        const T v = vals[e];
        if (v == 0)
            continue;
        outRi[at] = ri[e], outCi[at] = ci[e], outVals[at] = v;
        ++at;
        if (ri[e] == ci[e])
            continue;
        outRi[at] = ci[e], outCi[at] = ri[e], outVals[at] = v;
        ++at;
    }
}
