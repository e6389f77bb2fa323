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

/* return codes of the loaders, value kinds and symmetry classes: the reference's names and numbers (mmread.hpp:15-42) */
enum { MATRIX_READ_SUCCESS = 0, MATRIX_READ_UNSUPPORTED = 1, MATRIX_READ_INVALID_INPUT = 2 };
enum { MATRIX_STORAGE_INTEGER = 0, MATRIX_STORAGE_REAL = 1, MATRIX_STORAGE_COMPLEX = 2, MATRIX_STORAGE_PATTERN = 3 };
enum { MATRIX_TYPE_GENERAL = 0, MATRIX_TYPE_SYMMETRIC = 1, MATRIX_TYPE_SKEW = 2, MATRIX_TYPE_HERMITIAN = 3 };

/* Banner + size line (mmread.hpp:44-50 / mmread.cpp:15-58).  false for a missing or invalid banner, an object that
 * is not a matrix, or a combination Matrix Market forbids (array+pattern, real+hermitian, pattern+hermitian/skew:
 * mmio.c mm_is_valid).  All outputs are written on success. */
bool loadMmProperties(int* rows, int* cols, int* nnz, bool* sparse, int* storage, int* symmetry, FILE* in);

/* `nnz` coordinate entries, zero-based on return (mmread.hpp:52-94 / mmread.cpp:144-222).  One overload per value
 * kind; the last one reads pattern files. */
int loadMmMatrixToCoo(float* vals, int* ri, int* ci, int rows, int cols, int nnz, bool sparse, int storage, FILE* in);
int loadMmMatrixToCoo(double* vals, int* ri, int* ci, int rows, int cols, int nnz, bool sparse, int storage, FILE* in);
int loadMmMatrixToCoo(int* vals, int* ri, int* ci, int rows, int cols, int nnz, bool sparse, int storage, FILE* in);
int loadMmMatrixToCoo(int* ri, int* ci, int rows, int cols, int nnz, bool sparse, int storage, FILE* in);

/* `n` whitespace-separated values of a dense vector file (mmread.hpp:96-111 / mmread.cpp:225-277). */
int loadMmVectorToDenseVector(float* vals, int n, int storage, FILE* in);
int loadMmVectorToDenseVector(double* vals, int n, int storage, FILE* in);
int loadMmVectorToDenseVector(int* vals, int n, int storage, FILE* in);
