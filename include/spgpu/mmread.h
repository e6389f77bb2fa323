#pragma once
/*
 * C entry points of the Matrix Market reader (the C++ overloads with the
 * reference's own signatures are in mmread.hpp / mmutils.hpp; these wrap them
 * for C programs and FFI).  Indices come back ZERO-based, as from the
 * reference's loadMmMatrixToCoo (mmread.cpp:118-222).
 */
#ifdef __cplusplus
extern "C" {
#endif

/* Opens `path`, reads banner + size line; returns 1 on success.
 * out = {rows, columns, nonZeros, isStoredSparse, matrixStorage, matrixType} (codes of mmread.hpp). */
int spgpuMmProperties(const char* path, int out[6]);

/* Reads the whole file into caller arrays of `nonZeros` entries; valueKind 'f' (float), 'd' (double), 'i' (int) or
 * 'p' (pattern: values unused).  Returns the MATRIX_READ_* code of loadMmMatrixToCoo, -1 if the file cannot be opened. */
int spgpuMmReadCoo(const char* path, char valueKind, void* values, int* rows, int* cols);

/* Symmetric storage -> general: number of entries after mirroring the off-diagonal ones, and the mirroring itself
 * (getUnfoldedMmSymmetricSize / unfoldMmSymmetricReal, mmutils.hpp:10-62). */
int spgpuMmUnfoldedSizeD(double* values, int* rows, int* cols, int nonZeros);
void spgpuMmUnfoldD(int* outRows, int* outCols, double* outValues, int* rows, int* cols, double* values, int nonZeros);

#ifdef __cplusplus
}
#endif
