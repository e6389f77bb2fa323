#pragma once
/*
 * Matrix Market front end (C++), the step in front of the COO -> ELL/HELL/HDIA converters.
 * Same names, overloads, argument order and return codes as the reference's src/utils/mmread.hpp:15-111
 * (implemented there on top of NIST mmio, src/external/mmio.c); this is an independent parser.
 * SURVEY.md section 8, row f1.
 *
 * loadMmProperties reads the banner and the size line; loadMmMatrixToCoo then reads `nonZerosCount`
 * coordinate entries, turning the file's 1-based indices into 0-based ones (mmread.cpp:84-92).
 * The float overload accepts real and integer files, the double overload real files only
 * (mmread.cpp:154-187), the int overload integer files, the value-less overload pattern files.
 */
#include <stdio.h>

#define MATRIX_READ_SUCCESS       0
#define MATRIX_READ_UNSUPPORTED   1
#define MATRIX_READ_INVALID_INPUT 2

#define MATRIX_STORAGE_INTEGER 0
#define MATRIX_STORAGE_REAL    1
#define MATRIX_STORAGE_COMPLEX 2
#define MATRIX_STORAGE_PATTERN 3

#define MATRIX_TYPE_GENERAL   0
#define MATRIX_TYPE_SYMMETRIC 1
#define MATRIX_TYPE_SKEW      2
#define MATRIX_TYPE_HERMITIAN 3

/* reference: mmread.hpp:44-50 / mmread.cpp:15-58.  false for a missing/invalid banner, a non-matrix object or a
 * combination Matrix Market forbids (array+pattern, real+hermitian, pattern+hermitian/skew: mmio.c mm_is_valid). */
bool loadMmProperties(int* rowsCount, int* columnsCount, int* nonZerosCount, bool* isStoredSparse, int* matrixStorage,
                      int* matrixType, FILE* file);

/* reference: mmread.hpp:52-94 / mmread.cpp:144-222 */
int loadMmMatrixToCoo(float* values, int* rowIndices, int* columnIndices, int rowsCount, int columnsCount,
                      int nonZerosCount, bool isStoredSparse, int matrixStorage, FILE* file);
int loadMmMatrixToCoo(double* values, int* rowIndices, int* columnIndices, int rowsCount, int columnsCount,
                      int nonZerosCount, bool isStoredSparse, int matrixStorage, FILE* file);
int loadMmMatrixToCoo(int* values, int* rowIndices, int* columnIndices, int rowsCount, int columnsCount,
                      int nonZerosCount, bool isStoredSparse, int matrixStorage, FILE* file);
int loadMmMatrixToCoo(int* rowIndices, int* columnIndices, int rowsCount, int columnsCount, int nonZerosCount,
                      bool isStoredSparse, int matrixStorage, FILE* file);

/* reference: mmread.hpp:96-111 / mmread.cpp:225-277: `vectorSize` whitespace-separated values. */
int loadMmVectorToDenseVector(float* values, int vectorSize, int matrixStorage, FILE* file);
int loadMmVectorToDenseVector(double* values, int vectorSize, int matrixStorage, FILE* file);
int loadMmVectorToDenseVector(int* values, int vectorSize, int matrixStorage, FILE* file);
