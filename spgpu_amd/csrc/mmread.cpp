/*
 * Matrix Market reader (include/spgpu/mmread.hpp).  Own parser with the observable behaviour of the reference's
 * src/utils/mmread.cpp:15-277 on top of NIST mmio (src/external/mmio.c: mm_read_banner, mm_is_valid,
 * mm_read_mtx_crd_size); checked against that code, compiled unmodified, in tests/test_mmread.py.
 * C-callable wrappers (spgpuMm*) at the end serve bindings that cannot call C++ overloads.
 */
#include "spgpu/mmread.hpp"

#include <ctype.h>
#include <string.h>

namespace {

constexpr int kMaxLine = 1025; /* MM_MAX_LINE_LENGTH */

void lower(char* s)
{
    for (; *s; ++s)
        *s = (char)tolower((unsigned char)*s);
}

struct Banner {
    bool sparse;
    int storage, symmetry;
};

/* "%%MatrixMarket matrix <coordinate|array> <real|complex|pattern|integer> <general|symmetric|hermitian|skew-symmetric>" */
bool readBanner(FILE* f, Banner* b)
{
    char line[kMaxLine], tag[64], object[64], format[64], field[64], symmetry[64];
    if (!fgets(line, kMaxLine, f))
        return false;
    if (sscanf(line, "%63s %63s %63s %63s %63s", tag, object, format, field, symmetry) != 5)
        return false;
    lower(object);
    lower(format);
    lower(field);
    lower(symmetry);
    if (strncmp(tag, "%%MatrixMarket", 14) != 0 || strcmp(object, "matrix") != 0)
        return false;
    if (strcmp(format, "coordinate") == 0)
        b->sparse = true;
    else if (strcmp(format, "array") == 0)
        b->sparse = false;
    else
        return false;
    if (strcmp(field, "real") == 0)
        b->storage = MATRIX_STORAGE_REAL;
    else if (strcmp(field, "complex") == 0)
        b->storage = MATRIX_STORAGE_COMPLEX;
    else if (strcmp(field, "pattern") == 0)
        b->storage = MATRIX_STORAGE_PATTERN;
    else if (strcmp(field, "integer") == 0)
        b->storage = MATRIX_STORAGE_INTEGER;
    else
        return false;
    if (strcmp(symmetry, "general") == 0)
        b->symmetry = MATRIX_TYPE_GENERAL;
    else if (strcmp(symmetry, "symmetric") == 0)
        b->symmetry = MATRIX_TYPE_SYMMETRIC;
    else if (strcmp(symmetry, "hermitian") == 0)
        b->symmetry = MATRIX_TYPE_HERMITIAN;
    else if (strcmp(symmetry, "skew-symmetric") == 0)
        b->symmetry = MATRIX_TYPE_SKEW;
    else
        return false;
    /* combinations the format forbids (mmio.c mm_is_valid) */
    if (!b->sparse && b->storage == MATRIX_STORAGE_PATTERN)
        return false;
    if (b->storage == MATRIX_STORAGE_REAL && b->symmetry == MATRIX_TYPE_HERMITIAN)
        return false;
    if (b->storage == MATRIX_STORAGE_PATTERN && (b->symmetry == MATRIX_TYPE_HERMITIAN || b->symmetry == MATRIX_TYPE_SKEW))
        return false;
    return true;
}

/* Skip '%' comment lines, then "rows cols nnz" -- on that line or, after blank lines, on a later one.
 * (The reference's mm_read_mtx_crd_size retries fscanf("%d %d %d") forever when the next token is not an integer,
 * e.g. on the two-number size line of an `array` file; here such a file is reported as unreadable instead.) */
bool readCoordinateSize(FILE* f, int* rows, int* cols, int* nnz)
{
    char line[kMaxLine];
    *rows = *cols = *nnz = 0;
    do {
        if (!fgets(line, kMaxLine, f))
            return false;
    } while (line[0] == '%');
    for (;;) {
        if (sscanf(line, "%d %d %d", rows, cols, nnz) == 3)
            return true;
        const char* c = line;
        while (*c == ' ' || *c == '\t' || *c == '\r' || *c == '\n')
            ++c;
        if (*c != '\0') { /* a non-blank line that is not "M N nz" */
            *rows = *cols = *nnz = 0;
            return false;
        }
        if (!fgets(line, kMaxLine, f))
            return false;
    }
}

template <typename T> void readRealEntries(T* values, int* rows, int* cols, int nnz, FILE* f)
{
    for (int e = 0; e < nnz; ++e) {
        int r, c;
        double v;
        if (fscanf(f, "%d %d %lg\n", &r, &c, &v) < 3) {
            printf("Error, file has not %i but just %i elements.\n", nnz, e);
            return;
        }
        values[e] = (T)v;
        rows[e] = r - 1;
        cols[e] = c - 1;
    }
}

} // namespace

bool loadMmProperties(int* rowsCount, int* columnsCount, int* nonZerosCount, bool* isStoredSparse, int* matrixStorage,
                      int* matrixType, FILE* file)
{
    Banner b;
    if (!readBanner(file, &b))
        return false;
    if (!readCoordinateSize(file, rowsCount, columnsCount, nonZerosCount))
        return false;
    *isStoredSparse = b.sparse;
    *matrixStorage = b.storage;
    *matrixType = b.symmetry;
    return true;
}

int loadMmMatrixToCoo(float* values, int* rowIndices, int* columnIndices, int, int, int nonZerosCount, bool isStoredSparse,
                      int matrixStorage, FILE* file)
{
    if (!isStoredSparse)
        return MATRIX_READ_INVALID_INPUT;
    if (matrixStorage != MATRIX_STORAGE_REAL && matrixStorage != MATRIX_STORAGE_INTEGER)
        return MATRIX_READ_UNSUPPORTED;
    readRealEntries(values, rowIndices, columnIndices, nonZerosCount, file);
    return MATRIX_READ_SUCCESS;
}

int loadMmMatrixToCoo(double* values, int* rowIndices, int* columnIndices, int, int, int nonZerosCount, bool isStoredSparse,
                      int matrixStorage, FILE* file)
{
    if (!isStoredSparse)
        return MATRIX_READ_INVALID_INPUT;
    if (matrixStorage != MATRIX_STORAGE_REAL)
        return MATRIX_READ_UNSUPPORTED;
    readRealEntries(values, rowIndices, columnIndices, nonZerosCount, file);
    return MATRIX_READ_SUCCESS;
}

int loadMmMatrixToCoo(int* values, int* rowIndices, int* columnIndices, int, int, int nonZerosCount, bool isStoredSparse,
                      int matrixStorage, FILE* file)
{
    if (!isStoredSparse)
        return MATRIX_READ_INVALID_INPUT;
    if (matrixStorage != MATRIX_STORAGE_INTEGER)
        return MATRIX_READ_UNSUPPORTED;
    for (int e = 0; e < nonZerosCount; ++e) {
        int r = 0, c = 0, v = 0;
        if (fscanf(file, "%d %d %d\n", &r, &c, &v) < 3)
            break;
        values[e] = v;
        rowIndices[e] = r - 1;
        columnIndices[e] = c - 1;
    }
    return MATRIX_READ_SUCCESS;
}

int loadMmMatrixToCoo(int* rowIndices, int* columnIndices, int, int, int nonZerosCount, bool isStoredSparse,
                      int matrixStorage, FILE* file)
{
    if (!isStoredSparse)
        return MATRIX_READ_INVALID_INPUT;
    if (matrixStorage != MATRIX_STORAGE_PATTERN)
        return MATRIX_READ_UNSUPPORTED;
    for (int e = 0; e < nonZerosCount; ++e) {
        int r = 0, c = 0;
        if (fscanf(file, "%d %d\n", &r, &c) < 2)
            break;
        rowIndices[e] = r - 1;
        columnIndices[e] = c - 1;
    }
    return MATRIX_READ_SUCCESS;
}

int loadMmVectorToDenseVector(float* values, int vectorSize, int matrixStorage, FILE* file)
{
    if (matrixStorage != MATRIX_STORAGE_REAL)
        return MATRIX_READ_INVALID_INPUT;
    for (int i = 0; i < vectorSize; ++i) {
        double v = 0;
        if (fscanf(file, "%lg\n", &v) < 1)
            break;
        values[i] = (float)v;
    }
    return MATRIX_READ_SUCCESS;
}

int loadMmVectorToDenseVector(double* values, int vectorSize, int matrixStorage, FILE* file)
{
    if (matrixStorage != MATRIX_STORAGE_REAL)
        return MATRIX_READ_INVALID_INPUT;
    for (int i = 0; i < vectorSize; ++i) {
        double v = 0;
        if (fscanf(file, "%lg\n", &v) < 1)
            break;
        values[i] = v;
    }
    return MATRIX_READ_SUCCESS;
}

int loadMmVectorToDenseVector(int* values, int vectorSize, int matrixStorage, FILE* file)
{
    if (matrixStorage != MATRIX_STORAGE_INTEGER)
        return MATRIX_READ_INVALID_INPUT;
    for (int i = 0; i < vectorSize; ++i) {
        int v = 0;
        if (fscanf(file, "%i\n", &v) < 1)
            break;
        values[i] = v;
    }
    return MATRIX_READ_SUCCESS;
}

/* ---- C-callable wrappers (ctypes, C programs) ------------------------------------------------------------ */
#include "spgpu/mmutils.hpp"

extern "C" {

/* Opens `path`, reads banner + size; returns 1 on success.  Writes {rows, cols, nnz, sparse, storage, symmetry}. */
int spgpuMmProperties(const char* path, int out[6])
{
    FILE* f = fopen(path, "r");
    if (!f)
        return 0;
    bool sparse = false;
    const bool ok = loadMmProperties(&out[0], &out[1], &out[2], &sparse, &out[4], &out[5], f);
    out[3] = sparse ? 1 : 0;
    fclose(f);
    return ok ? 1 : 0;
}

/* Reads the whole file into caller arrays of nnz entries; valueKind 'f','d','i' or 'p' (pattern: values unused).
 * Returns the loadMmMatrixToCoo code, or -1 if the file cannot be opened / has no valid header. */
int spgpuMmReadCoo(const char* path, char valueKind, void* values, int* rows, int* cols)
{
    FILE* f = fopen(path, "r");
    if (!f)
        return -1;
    int m, n, nnz, storage, symmetry;
    bool sparse;
    int code = -1;
    if (loadMmProperties(&m, &n, &nnz, &sparse, &storage, &symmetry, f)) {
        switch (valueKind) {
        case 'f': code = loadMmMatrixToCoo(static_cast<float*>(values), rows, cols, m, n, nnz, sparse, storage, f); break;
        case 'd': code = loadMmMatrixToCoo(static_cast<double*>(values), rows, cols, m, n, nnz, sparse, storage, f); break;
        case 'i': code = loadMmMatrixToCoo(static_cast<int*>(values), rows, cols, m, n, nnz, sparse, storage, f); break;
        case 'p': code = loadMmMatrixToCoo(rows, cols, m, n, nnz, sparse, storage, f); break;
        default: break;
        }
    }
    fclose(f);
    return code;
}

int spgpuMmUnfoldedSizeD(double* values, int* rows, int* cols, int nnz)
{
    int total = 0;
    getUnfoldedMmSymmetricSize(&total, values, rows, cols, nnz);
    return total;
}

void spgpuMmUnfoldD(int* outRows, int* outCols, double* outValues, int* rows, int* cols, double* values, int nnz)
{
    unfoldMmSymmetricReal(outRows, outCols, outValues, rows, cols, values, nnz);
}

} // extern "C"
