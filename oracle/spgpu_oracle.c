/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the spgpu-amd hot path.
 *
 * A plain-C restatement of what the reference (davidebarbieri/spgpu) computes
 * on this path, each function citing the reference lines it follows (paths
 * relative to the reference's src/core/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's library; the product
 * (spgpu_amd/lib/libspgpu.so) never links, loads or calls it.
 *
 * PINNING
 *  - Format conversions (orc_cooToEll, orc_ellToHell, orc_cooToHdia, ...):
 *    PINNED bit-exact against the reference's own converters, compiled from
 *    /root/reference into oracle/_ref (see oracle/Makefile) and compared on
 *    randomised inputs (tests/test_oracle_vs_reference.py), against fixtures that
 *    build produced (tests/golden/ (npz files), oracle/make_golden.py) and against the
 *    structural known answers of SURVEY.md section 8(a) (pitch, heights, hack
 *    offsets, diagonal sets; tests/test_oracle_golden.py).
 *  - SpMV / axpby / dot values: the reference has NO CPU implementation of
 *    these kernels and NO golden output values (its tests print dot(z,z) and a
 *    human compares formats).  Absolute values are therefore "parity
 *    unpinned" by the reference; this oracle is pinned instead by the one
 *    analytic identity the reference's ctest.c implies (A = 2I, alpha = 2,
 *    beta = -3  =>  z = 4x - 3y), by the cross-format equalities its perf
 *    tests rely on (ELL == HELL == HDIA results), and by an independent
 *    extended-precision product computed from the COO triplets
 *    (oracle/make_golden.py -> tests/golden/ (npz files), tests/test_oracle_golden.py).
 *
 * Every routine exists in four flavours generated from one macro body:
 * s (float), d (double), c (float complex), z (double complex).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* Type codes and sizes: core.h:51-57, core.c:82-99                          */
/* ======================================================================== */
size_t orc_sizeOf(int typeCode)
{
    static const size_t bytes[5] = {sizeof(int), sizeof(float), sizeof(double), 2 * sizeof(float), 2 * sizeof(double)};
    return (typeCode >= 0 && typeCode < 5) ? bytes[typeCode] : 0;
}

/* FNV-1a 64-bit over raw bytes: the checksum SURVEY.md 8(a) quotes. */
uint64_t orc_fnv1a64(const void* data, size_t bytes)
{
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

int orc_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================== */
/* COO -> ELL: ell.c:5-80                                                    */
/* ======================================================================== */

/* ell.c:5-31: histogram of row indices, then the maximum. */
void orc_computeEllRowLenghts(int* rowLen, int* maxRow, int rows, int nnz, const int* cooRows, int base)
{
    for (int r = 0; r < rows; ++r)
        rowLen[r] = 0;
    for (int e = 0; e < nnz; ++e)
        rowLen[cooRows[e] - base]++;
    int m = 0;
    for (int r = 0; r < rows; ++r)
        if (rowLen[r] > m)
            m = rowLen[r];
    *maxRow = m;
}

/* ell.c:33-37 */
int orc_computeEllAllocPitch(int rows)
{
    return ((rows + 31) / 32) * 32;
}

/* ell.c:39-80: entry e goes to slot (row, next free k of that row); stored
 * index = col - cooBase + ellBase; value copied verbatim. */
void orc_cooToEll(void* ellVals, int* ellIdx, int valPitch, int idxPitch, int maxRow, int ellBase, int rows,
                  int nnz, const int* cooRows, const int* cooCols, const void* cooVals, int cooBase, int type)
{
    (void)maxRow;
    const size_t es = orc_sizeOf(type);
    int* next = (int*)calloc(rows > 0 ? (size_t)rows : 1, sizeof(int));
    for (int e = 0; e < nnz; ++e) {
        const int r = cooRows[e] - cooBase;
        const size_t k = (size_t)next[r];
        ellIdx[(size_t)r + k * (size_t)idxPitch] = cooCols[e] - cooBase + ellBase;
        memcpy((char*)ellVals + ((size_t)r + k * (size_t)valPitch) * es, (const char*)cooVals + (size_t)e * es, es);
        next[r]++;
    }
    free(next);
}

/* ======================================================================== */
/* ELL -> HELL: hell.c:4-104                                                 */
/* ======================================================================== */

/* hell.c:4-44: sum over hacks (last one possibly partial) of the longest row. */
void orc_computeHellAllocSize(int* height, int hackSize, int rows, const int* rowLen)
{
    int total = 0;
    const int hacks = (rows + hackSize - 1) / hackSize;
    for (int h = 0; h < hacks; ++h) {
        int longest = 0;
        for (int j = 0; j < hackSize && h * hackSize + j < rows; ++j)
            if (rowLen[h * hackSize + j] > longest)
                longest = rowLen[h * hackSize + j];
        total += longest;
    }
    *height = total;
}

/* hell.c:46-104: hackOffsets[h] = hackSize * (sum of longest rows of earlier
 * hacks); slot of (row, k) = hackOffsets[h] + row%hackSize + k*hackSize; only
 * k < rowLen[row] is written. */
void orc_ellToHell(void* hellVals, int* hellIdx, int* hackOffsets, int hackSize, const void* ellVals,
                   const int* ellIdx, int valPitch, int idxPitch, const int* rowLen, int rows, int type)
{
    const size_t es = orc_sizeOf(type);
    const int hacks = (rows + hackSize - 1) / hackSize;
    size_t offset = 0;
    for (int h = 0; h < hacks; ++h) {
        int longest = 0;
        hackOffsets[h] = (int)offset;
        for (int j = 0; j < hackSize; ++j) {
            const int row = h * hackSize + j;
            if (row >= rows)
                break;
            if (rowLen[row] > longest)
                longest = rowLen[row];
            for (int k = 0; k < rowLen[row]; ++k) {
                const size_t dst = offset + (size_t)j + (size_t)k * (size_t)hackSize;
                memcpy((char*)hellVals + dst * es,
                       (const char*)ellVals + ((size_t)k * (size_t)valPitch + (size_t)row) * es, es);
                hellIdx[dst] = ellIdx[(size_t)k * (size_t)idxPitch + (size_t)row];
            }
        }
        offset += (size_t)hackSize * (size_t)longest;
    }
}

/* ======================================================================== */
/* COO -> HDIA: hdia.cpp:8-11, 161-349                                       */
/* ======================================================================== */

int orc_getHdiaHacksCount(int hackSize, int rows)
{
    return (rows + hackSize - 1) / hackSize;
}

/* The reference keeps, per hack, an ordered map keyed by
 *   diagPos = hackSize - 1 + (col0 - row0 % hackSize)         (hdia.cpp:210-211)
 * Here the ordered map is a small sorted array with insertion. */
typedef struct { int* key; int count, cap; } orc_keyset;

static int orc_keyset_find(const orc_keyset* s, int key) /* index or -1 */
{
    int lo = 0, hi = s->count - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (s->key[mid] == key)
            return mid;
        if (s->key[mid] < key)
            lo = mid + 1;
        else
            hi = mid - 1;
    }
    return -1;
}

static void orc_keyset_insert(orc_keyset* s, int key)
{
    if (orc_keyset_find(s, key) >= 0)
        return;
    if (s->count == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 16;
        s->key = (int*)realloc(s->key, (size_t)s->cap * sizeof(int));
    }
    int i = s->count++;
    while (i > 0 && s->key[i - 1] > key) {
        s->key[i] = s->key[i - 1];
        --i;
    }
    s->key[i] = key;
}

/* Entries grouped by hack in COO order (hdia.cpp:180-191 push_back per hack). */
static void orc_groupByHack(int** startOut, int** listOut, int hacks, int hackSize, int nnz, const int* cooRows, int base)
{
    int* start = (int*)calloc((size_t)hacks + 1, sizeof(int));
    int* list = (int*)malloc((nnz > 0 ? (size_t)nnz : 1) * sizeof(int));
    for (int e = 0; e < nnz; ++e)
        start[(cooRows[e] - base) / hackSize + 1]++;
    for (int h = 0; h < hacks; ++h)
        start[h + 1] += start[h];
    int* fill = (int*)malloc(((size_t)hacks + 1) * sizeof(int));
    memcpy(fill, start, ((size_t)hacks + 1) * sizeof(int));
    for (int e = 0; e < nnz; ++e)
        list[fill[(cooRows[e] - base) / hackSize]++] = e;
    free(fill);
    *startOut = start;
    *listOut = list;
}

/* hdia.cpp:161-228: hackOffsets[h+1] = hackOffsets[h] + #distinct diagPos. */
void orc_computeHdiaHackOffsetsFromCoo(int* height, int* hackOffsets, int hackSize, int rows, int cols, int nnz,
                                       const int* cooRows, const int* cooCols, int base)
{
    (void)cols;
    const int hacks = orc_getHdiaHacksCount(hackSize, rows);
    int *start, *list;
    orc_groupByHack(&start, &list, hacks, hackSize, nnz, cooRows, base);
    orc_keyset set = {0, 0, 0};
    hackOffsets[0] = 0;
    for (int h = 0; h < hacks; ++h) {
        set.count = 0;
        for (int p = start[h]; p < start[h + 1]; ++p) {
            const int e = list[p];
            const int diagId = (cooCols[e] - base) - ((cooRows[e] - base) % hackSize);
            orc_keyset_insert(&set, hackSize - 1 + diagId);
        }
        hackOffsets[h + 1] = hackOffsets[h] + set.count;
    }
    *height = hackOffsets[hacks];
    free(set.key);
    free(start);
    free(list);
}

/* hdia.cpp:230-349: per hack, diagonals in ascending diagPos; offsets entry =
 * col - row (hdia.cpp:279,300); value slot = row%hackSize + hackSize *
 * (hackOffsets[h] + position) (hdia.cpp:315-319); later duplicates overwrite. */
void orc_cooToHdia(void* hdiaVals, int* hdiaOffsets, const int* hackOffsets, int hackSize, int rows, int cols,
                   int nnz, const int* cooRows, const int* cooCols, const void* cooVals, int base, int type)
{
    (void)cols;
    const size_t es = orc_sizeOf(type);
    const int hacks = orc_getHdiaHacksCount(hackSize, rows);
    int *start, *list;
    orc_groupByHack(&start, &list, hacks, hackSize, nnz, cooRows, base);
    orc_keyset set = {0, 0, 0};
    int* global = NULL; /* col - row of each key, first occurrence */
    int globalCap = 0;
    for (int h = 0; h < hacks; ++h) {
        set.count = 0;
        for (int p = start[h]; p < start[h + 1]; ++p) {
            const int e = list[p];
            orc_keyset_insert(&set, hackSize - 1 + (cooCols[e] - base) - ((cooRows[e] - base) % hackSize));
        }
        if (set.count > globalCap) {
            globalCap = set.count;
            global = (int*)realloc(global, (size_t)globalCap * sizeof(int));
        }
        for (int i = 0; i < set.count; ++i)
            global[i] = 0;
        /* first entry seen on a diagonal defines its global offset */
        {
            char* seen = (char*)calloc(set.count > 0 ? (size_t)set.count : 1, 1);
            for (int p = start[h]; p < start[h + 1]; ++p) {
                const int e = list[p];
                const int pos = orc_keyset_find(&set, hackSize - 1 + (cooCols[e] - base) - ((cooRows[e] - base) % hackSize));
                if (!seen[pos]) {
                    seen[pos] = 1;
                    global[pos] = cooCols[e] - cooRows[e];
                }
            }
            free(seen);
        }
        for (int i = 0; i < set.count; ++i)
            hdiaOffsets[hackOffsets[h] + i] = global[i];
        for (int p = start[h]; p < start[h + 1]; ++p) {
            const int e = list[p];
            const int lane = (cooRows[e] - base) % hackSize;
            const int pos = orc_keyset_find(&set, hackSize - 1 + (cooCols[e] - base) - lane);
            const size_t slot = (size_t)lane + (size_t)hackSize * ((size_t)hackOffsets[h] + (size_t)pos);
            memcpy((char*)hdiaVals + slot * es, (const char*)cooVals + (size_t)e * es, es);
        }
    }
    free(global);
    free(set.key);
    free(start);
    free(list);
}

/* ======================================================================== */
/* COO -> DIA: dia.c:5-104                                                   */
/* ======================================================================== */
int orc_computeDiaAllocPitch(int rows)
{
    return ((rows + 31) / 32) * 32;
}

/* dia.c:11-39: count distinct diagPos = rows - 1 + col - row */
int orc_computeDiaDiagonalsCount(int rows, int cols, int nnz, const int* cooRows, const int* cooCols)
{
    const int span = rows + cols - 1;
    int* id = (int*)malloc((span > 0 ? (size_t)span : 1) * sizeof(int));
    int count = 0;
    for (int i = 0; i < span; ++i)
        id[i] = -1;
    for (int e = 0; e < nnz; ++e) {
        const int pos = rows - 1 + cooCols[e] - cooRows[e];
        if (id[pos] < 0)
            id[pos] = count++;
    }
    free(id);
    return count;
}

/* dia.c:41-104: present diagonals numbered in ascending diagPos, offsets = diagPos - rows + 1,
 * value to values[(row - base) + position*pitch]. */
void orc_coo2dia(void* values, int* offsets, int pitch, int diagonals, int rows, int cols, int nnz, const int* cooRows,
                 const int* cooCols, const void* cooVals, int base, int type)
{
    (void)diagonals;
    const size_t es = orc_sizeOf(type);
    const int span = rows + cols - 1;
    int* toPos = (int*)malloc((span > 0 ? (size_t)span : 1) * sizeof(int));
    int count = 0;
    for (int i = 0; i < span; ++i)
        toPos[i] = -1;
    for (int e = 0; e < nnz; ++e)
        toPos[rows - 1 + cooCols[e] - cooRows[e]] = 1;
    for (int i = 0; i < span; ++i)
        if (toPos[i] == 1) {
            toPos[i] = count;
            offsets[count++] = i - rows + 1;
        }
    for (int e = 0; e < nnz; ++e) {
        const int pos = toPos[rows - 1 + cooCols[e] - cooRows[e]];
        memcpy((char*)values + ((size_t)(cooRows[e] - base) + (size_t)pos * (size_t)pitch) * es,
               (const char*)cooVals + (size_t)e * es, es);
    }
    free(toPos);
}

/* hdia.cpp:13-57 / 59-153: DIA -> HDIA; a diagonal is kept in a hack iff any byte of its values in the
 * hack's rows is non-zero. */
static int orc_diagInHack(const char* dia, size_t es, int d, int pitch, int hack, int hackSize, int rows)
{
    for (int r = 0; r < hackSize; ++r) {
        const int row = hack * hackSize + r;
        if (row >= rows)
            break;
        const char* v = dia + es * ((size_t)row + (size_t)d * (size_t)pitch);
        for (size_t s = 0; s < es; ++s)
            if (v[s] != 0)
                return 1;
    }
    return 0;
}

void orc_computeHdiaHackOffsets(int* height, int* hackOffsets, int hackSize, const void* diaValues, int pitch,
                                int diagonals, int rows, int type)
{
    const size_t es = orc_sizeOf(type);
    const int hacks = orc_getHdiaHacksCount(hackSize, rows);
    int total = 0;
    hackOffsets[0] = 0;
    for (int h = 0; h < hacks; ++h) {
        for (int d = 0; d < diagonals; ++d)
            total += orc_diagInHack((const char*)diaValues, es, d, pitch, h, hackSize, rows);
        hackOffsets[h + 1] = total;
    }
    *height = hackOffsets[hacks];
}

void orc_diaToHdia(void* hdiaValues, int* hdiaOffsets, const int* hackOffsets, int hackSize, const void* diaValues,
                   const int* diaOffsets, int pitch, int diagonals, int rows, int type)
{
    const size_t es = orc_sizeOf(type);
    const int hacks = orc_getHdiaHacksCount(hackSize, rows);
    for (int h = 0; h < hacks; ++h) {
        int pos = hackOffsets[h];
        for (int d = 0; d < diagonals; ++d) {
            if (!orc_diagInHack((const char*)diaValues, es, d, pitch, h, hackSize, rows))
                continue;
            hdiaOffsets[pos] = diaOffsets[d];
            for (int r = 0; r < hackSize && h * hackSize + r < rows; ++r)
                memcpy((char*)hdiaValues + es * ((size_t)pos * (size_t)hackSize + (size_t)r),
                       (const char*)diaValues + es * ((size_t)(h * hackSize + r) + (size_t)d * (size_t)pitch), es);
            ++pos;
        }
    }
}

/* ======================================================================== */
/* ELL -> ordered ELL: ell.c:85-202.  A merge sort on (length, row) whose    */
/* merge takes the right run when lengths are equal (ell.c:94-104).          */
/* ======================================================================== */
static void orc_oell_sort(int* len, int* idx, int* tmpLen, int* tmpIdx, int lo, int hi) /* [lo, hi) */
{
    if (hi - lo < 2)
        return;
    const int mid = lo + (hi - lo) / 2;
    orc_oell_sort(len, idx, tmpLen, tmpIdx, lo, mid);
    orc_oell_sort(len, idx, tmpLen, tmpIdx, mid, hi);
    int i = lo, j = mid, k = lo;
    while (i < mid && j < hi) {
        if (len[i] > len[j]) { tmpLen[k] = len[i]; tmpIdx[k++] = idx[i++]; }
        else                 { tmpLen[k] = len[j]; tmpIdx[k++] = idx[j++]; }
    }
    while (i < mid) { tmpLen[k] = len[i]; tmpIdx[k++] = idx[i++]; }
    while (j < hi)  { tmpLen[k] = len[j]; tmpIdx[k++] = idx[j++]; }
    for (k = lo; k < hi; ++k) { len[k] = tmpLen[k]; idx[k] = tmpIdx[k]; }
}

void orc_ellToOell(int* rIdx, void* dstVals, int* dstIdx, int* dstRs, const void* srcVals, const int* srcIdx,
                   const int* srcRs, int valPitch, int idxPitch, int rows, int type)
{
    const size_t es = orc_sizeOf(type);
    int* t1 = (int*)malloc((rows > 0 ? (size_t)rows : 1) * sizeof(int));
    int* t2 = (int*)malloc((rows > 0 ? (size_t)rows : 1) * sizeof(int));
    for (int i = 0; i < rows; ++i) {
        rIdx[i] = i;
        dstRs[i] = srcRs[i];
    }
    /* ell.c:131-157: for exactly two rows the loop `while (n < sizetomerge*2)` and the final merge are both
     * skipped -- two rows are left unsorted.  All other sizes sort. */
    if (rows != 2)
        orc_oell_sort(dstRs, rIdx, t1, t2, 0, rows);
    free(t1);
    free(t2);
    for (int i = 0; i < rows; ++i) {
        const int src = rIdx[i];
        for (int k = 0; k < srcRs[src]; ++k) {
            memcpy((char*)dstVals + ((size_t)i + (size_t)k * (size_t)valPitch) * es,
                   (const char*)srcVals + ((size_t)src + (size_t)k * (size_t)valPitch) * es, es);
            dstIdx[(size_t)i + (size_t)k * (size_t)idxPitch] = srcIdx[(size_t)src + (size_t)k * (size_t)idxPitch];
        }
    }
}

/* ======================================================================== */
/* Arithmetic: hell_spmv_base.cuh:29-51 (real: (a*b)+c contracted to one fma  */
/* by nvcc's default -fmad; complex: cuCfma / cuCmul).                        */
/* ======================================================================== */
typedef struct { float x, y; } orc_cfloat;
typedef struct { double x, y; } orc_cdouble;

static inline float s_zero(void) { return 0.0f; }
static inline double d_zero(void) { return 0.0; }
static inline orc_cfloat c_zero(void) { orc_cfloat r = {0.0f, 0.0f}; return r; }
static inline orc_cdouble z_zero(void) { orc_cdouble r = {0.0, 0.0}; return r; }

static inline int s_nz(float a) { return a != 0.0f; }
static inline int d_nz(double a) { return a != 0.0; }
static inline int c_nz(orc_cfloat a) { return a.x != 0.0f || a.y != 0.0f; }
static inline int z_nz(orc_cdouble a) { return a.x != 0.0 || a.y != 0.0; }

static inline float s_fma(float a, float b, float c) { return fmaf(a, b, c); }
static inline double d_fma(double a, double b, double c) { return fma(a, b, c); }
static inline float s_mul(float a, float b) { return a * b; }
static inline double d_mul(double a, double b) { return a * b; }
static inline float s_add(float a, float b) { return a + b; }
static inline double d_add(double a, double b) { return a + b; }

/* cuCfma: re = (p.x*q.x + r.x) - p.y*q.y ; im = (q.x*p.y + r.y) + p.x*q.y */
static inline orc_cfloat c_fma(orc_cfloat p, orc_cfloat q, orc_cfloat r)
{
    float re = fmaf(p.x, q.x, r.x), im = fmaf(q.x, p.y, r.y);
    orc_cfloat o = {fmaf(-p.y, q.y, re), fmaf(p.x, q.y, im)};
    return o;
}
static inline orc_cdouble z_fma(orc_cdouble p, orc_cdouble q, orc_cdouble r)
{
    double re = fma(p.x, q.x, r.x), im = fma(q.x, p.y, r.y);
    orc_cdouble o = {fma(-p.y, q.y, re), fma(p.x, q.y, im)};
    return o;
}
/* cuCmul: re = a.x*b.x - a.y*b.y ; im = a.x*b.y + a.y*b.x */
static inline orc_cfloat c_mul(orc_cfloat a, orc_cfloat b)
{
    orc_cfloat o = {fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
    return o;
}
static inline orc_cdouble z_mul(orc_cdouble a, orc_cdouble b)
{
    orc_cdouble o = {fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
    return o;
}
static inline orc_cfloat c_add(orc_cfloat a, orc_cfloat b) { orc_cfloat o = {a.x + b.x, a.y + b.y}; return o; }
static inline orc_cdouble z_add(orc_cdouble a, orc_cdouble b) { orc_cdouble o = {a.x + b.x, a.y + b.y}; return o; }

static inline float s_abs2acc(float v, float acc) { return fmaf(v, v, acc); }
static inline double d_abs2acc(double v, double acc) { return fma(v, v, acc); }
static inline float c_abs2acc(orc_cfloat v, float acc) { return fmaf(v.y, v.y, fmaf(v.x, v.x, acc)); }
static inline double z_abs2acc(orc_cdouble v, double acc) { return fma(v.y, v.y, fma(v.x, v.x, acc)); }

#define ORC_MAX_PHASES 64

/*
 * One body for the four value types.
 *
 * `phases` selects the order in which the products of one row are added:
 *   1  ascending k                  -- the reference's one-thread-per-row
 *                                      kernels (hell_spmv_base_template.cuh:197-214,
 *                                      ell_spmv_base_template.cuh:178-266)
 *   2  (even k) + (odd k)           -- the reference's two-threads-per-row
 *                                      kernels (hell_spmv_base_template.cuh:59-101)
 *   4,8  k mod phases partial sums, combined pairwise p with p^1, p^2, p^4
 *                                      -- the MI355X slab kernel's order
 * Epilogue (hell_spmv_base_template.cuh:219-222):
 *   beta != 0 : z = fma(beta, y, alpha*sum) ;  beta == 0 : z = alpha*sum, y unread.
 */
#define ORC_DEFINE_TYPE(P, T, R)                                                                              \
    static T P##_combine(T* part, int phases)                                                                 \
    {                                                                                                         \
        for (int m = 1; m < phases; m <<= 1) {                                                                \
            T next[ORC_MAX_PHASES];                                                                           \
            for (int p = 0; p < phases; ++p)                                                                  \
                next[p] = P##_add(part[p], part[p ^ m]);                                                      \
            for (int p = 0; p < phases; ++p)                                                                  \
                part[p] = next[p];                                                                            \
        }                                                                                                     \
        return part[0];                                                                                       \
    }                                                                                                         \
                                                                                                              \
    static void P##_store(T* z, const T* y, int outRow, T alpha, T sum, T beta)                               \
    {                                                                                                         \
        if (P##_nz(beta))                                                                                     \
            z[outRow] = P##_fma(beta, y[outRow], P##_mul(alpha, sum));                                        \
        else                                                                                                  \
            z[outRow] = P##_mul(alpha, sum);                                                                  \
    }                                                                                                         \
                                                                                                              \
    /* hell_spmv_base_template.cuh:112-252; rIdx: :227-252 */                                                 \
    void orc_##P##hellspmv(T* z, const T* y, T alpha, const T* cM, const int* rP, int hackSize,               \
                           const int* hackOffsets, const int* rS, const int* rIdx, int rows, const T* x,      \
                           T beta, int baseIndex, int phases)                                                 \
    {                                                                                                         \
        _Pragma("omp parallel for schedule(static)")                                                          \
        for (int i = 0; i < rows; ++i) {                                                                      \
            const size_t slot0 = (size_t)hackOffsets[i / hackSize] + (size_t)(i % hackSize);                  \
            T part[ORC_MAX_PHASES];                                                                           \
            for (int p = 0; p < phases; ++p)                                                                  \
                part[p] = P##_zero();                                                                         \
            for (int k = 0; k < rS[i]; ++k) {                                                                 \
                const size_t s = slot0 + (size_t)k * (size_t)hackSize;                                        \
                part[k % phases] = P##_fma(cM[s], x[rP[s] - baseIndex], part[k % phases]);                    \
            }                                                                                                 \
            P##_store(z, y, rIdx ? rIdx[i] : i, alpha, P##_combine(part, phases), beta);                      \
        }                                                                                                     \
    }                                                                                                         \
                                                                                                              \
    /* ell_spmv_base_template.cuh:102-266, ell_spmv_base_nors.cuh:97-258 (rS == NULL) */                     \
    void orc_##P##ellspmv(T* z, const T* y, T alpha, const T* cM, const int* rP, int cMPitch, int rPPitch,    \
                          const int* rS, const int* rIdx, int maxNnzPerRow, int rows, const T* x, T beta,     \
                          int baseIndex, int phases)                                                          \
    {                                                                                                         \
        _Pragma("omp parallel for schedule(static)")                                                          \
        for (int i = 0; i < rows; ++i) {                                                                      \
            const int len = rS ? rS[i] : maxNnzPerRow;                                                        \
            T part[ORC_MAX_PHASES];                                                                           \
            for (int p = 0; p < phases; ++p)                                                                  \
                part[p] = P##_zero();                                                                         \
            for (int k = 0; k < len; ++k) {                                                                   \
                const int col = rP[(size_t)i + (size_t)k * (size_t)rPPitch] - baseIndex;                      \
                if (col < 0)                                                                                  \
                    continue; /* zero padding with baseIndex 1: the reference reads x[-1] here */            \
                part[k % phases] = P##_fma(cM[(size_t)i + (size_t)k * (size_t)cMPitch], x[col], part[k % phases]); \
            }                                                                                                 \
            P##_store(z, y, rIdx ? rIdx[i] : i, alpha, P##_combine(part, phases), beta);                      \
        }                                                                                                     \
    }                                                                                                         \
                                                                                                              \
    /* hdia_spmv_base_template.cuh:19-206: diagonals of the row's hack in stored order, */                    \
    /* a slot counts iff 0 <= offsets[d] + i < cols (:111-118). */                                            \
    void orc_##P##hdiaspmv(T* z, const T* y, T alpha, const T* dM, const int* offsets, int hackSize,          \
                           const int* hackOffsets, int rows, int cols, const T* x, T beta)                    \
    {                                                                                                         \
        _Pragma("omp parallel for schedule(static)")                                                          \
        for (int i = 0; i < rows; ++i) {                                                                      \
            const int h = i / hackSize, lane = i % hackSize;                                                  \
            T sum = P##_zero();                                                                               \
            for (int d = hackOffsets[h]; d < hackOffsets[h + 1]; ++d) {                                       \
                const long long col = (long long)offsets[d] + i;                                              \
                if (col >= 0 && col < cols)                                                                   \
                    sum = P##_fma(dM[(size_t)d * (size_t)hackSize + (size_t)lane], x[col], sum);              \
            }                                                                                                 \
            P##_store(z, y, i, alpha, sum, beta);                                                             \
        }                                                                                                     \
    }                                                                                                         \
                                                                                                              \
    /* ddot.cu:37-150 (plain, un-conjugated sum; zdot.cu:54).  The reference's order of */                    \
    /* addition depends on its launch geometry; this oracle adds in ascending i. */                           \
    void orc_##P##dot(T* out, int n, const T* a, const T* b)                                                  \
    {                                                                                                         \
        T acc = P##_zero();                                                                                   \
        for (int i = 0; i < n; ++i)                                                                           \
            acc = P##_fma(a[i], b[i], acc);                                                                   \
        *out = acc;                                                                                           \
    }                                                                                                         \
                                                                                                              \
    /* dnrm2.cu:52-53,146: sqrt of the unscaled sum of squares. */                                            \
    void orc_##P##nrm2(R* out, int n, const T* a)                                                             \
    {                                                                                                         \
        R acc = 0;                                                                                            \
        for (int i = 0; i < n; ++i)                                                                           \
            acc = P##_abs2acc(a[i], acc);                                                                     \
        *out = (R)sqrt((double)acc);                                                                          \
    }

ORC_DEFINE_TYPE(s, float, float)
ORC_DEFINE_TYPE(d, double, double)
ORC_DEFINE_TYPE(c, orc_cfloat, float)
ORC_DEFINE_TYPE(z, orc_cdouble, double)

/* The MI355X slab kernel's "tail" order (spgpu_amd/csrc/ellpack_spmv.hip, TAIL): a wavefront owns
 * groupRows consecutive rows, rowsPerLane per lane.  It walks slab columns `step` at a time while more than
 * tailLanes lanes still have entries (a strip of rowsPerLane rows keeps `phases` lanes busy); the first
 * column block at which <= tailLanes lanes are busy is tailFrom.  A row's sum is then: `phases` partial sums
 * over entries k < tailFrom by k mod phases, each ascending; entries k >= tailFrom split over 64 partial sums
 * by (k - tailFrom) mod 64, each ascending, combined pairwise (p with p^1, ... p^32) and added to partial 0;
 * finally the `phases` partials combined pairwise.
 * Same arithmetic and epilogue as orc_?hellspmv / orc_?ellspmv; hackOffsets == NULL selects ELL addressing. */
#define ORC_DEFINE_TAIL(P, T)                                                                                 \
    /* length of row i as the slab kernel walks it: with the deep split (deepCap > 0) the rows of a 32-row */   \
    /* sub-group whose longest row exceeds deepCap stop at deepCap there */                                    \
    /* depth of row i's 32-row sub-group if the sub-group is deep, else 0 */                                  \
    static int P##_deep_group(const int* rS, int maxNnz, int rows, int i, int deepCap)                        \
    {                                                                                                         \
        if (deepCap <= 0) return 0;                                                                           \
        const int s0 = i / 32 * 32;                                                                           \
        int depth = 0;                                                                                        \
        for (int r = s0; r < s0 + 32 && r < rows; ++r)                                                        \
            if ((rS ? rS[r] : maxNnz) > depth) depth = rS ? rS[r] : maxNnz;                                   \
        return depth > deepCap ? depth : 0;                                                                   \
    }                                                                                                         \
    /* deepKeep <= deepCap: of a deep sub-group the main kernel walks the first deepKeep columns only, the rest are the deep kernels' */  \
    static int P##_walked_keep(const int* rS, int maxNnz, int rows, int i, int deepCap, int deepKeep)         \
    {                                                                                                         \
        const int l = rS ? rS[i] : maxNnz;                                                                    \
        return (P##_deep_group(rS, maxNnz, rows, i, deepCap) && l > deepKeep) ? deepKeep : l;                 \
    }                                                                                                         \
    /* mainChunk > 0 (the queue kernel for ordered rows, csrc/ragged_spmv.hip.h SPLIT): a 32-row sub-group whose walked */  \
    /* depth exceeds mainChunk is cut into chunks of mainChunk columns; a chunk's `phases` phase sums (by k mod phases, */   \
    /* each ascending) are combined pairwise, and the chunk sums are added in chunk order -- every chunk of the sub-group */ \
    /* for every one of its rows, a shorter row's later chunks being +0.  (No tail rows in that kernel: tailLanes 0.) */     \
    void orc_##P##spmv_split(T* z, const T* y, T alpha, const T* cM, const int* rP, int hackSize,            \
                            const int* hackOffsets, int cMPitch, int rPPitch, const int* rS, int maxNnz,      \
                            const int* rIdx, int rows, const T* x, T beta, int baseIndex, int groupRows,      \
                            int rowsPerLane, int step, int tailLanes, int phases, int deepCap, int deepPhases,\
                            int deepChunk, int mainChunk, int deepKeep)                                       \
    {                                                                                                         \
        for (int g0 = 0; g0 < rows; g0 += groupRows) {                                                        \
            const int gEnd = g0 + groupRows < rows ? g0 + groupRows : rows;                                   \
            int longest = 0;                                                                                  \
            for (int i = g0; i < gEnd; ++i) {                                                                 \
                const int l = P##_walked_keep(rS, maxNnz, rows, i, deepCap, deepKeep);                                       \
                if (l > longest) longest = l;                                                                 \
            }                                                                                                 \
            int tailFrom = longest;                                                                           \
            for (int kBase = 0; kBase < longest; kBase += step) {                                             \
                int busy = 0;                                                                                 \
                for (int s0 = g0; s0 < gEnd; s0 += rowsPerLane) {                                             \
                    int laneLongest = 0;                                                                      \
                    for (int i = s0; i < s0 + rowsPerLane && i < gEnd; ++i) {                                 \
                        const int l = P##_walked_keep(rS, maxNnz, rows, i, deepCap, deepKeep);                               \
                        if (l > laneLongest) laneLongest = l;                                                 \
                    }                                                                                         \
                    busy += kBase < laneLongest;                                                              \
                }                                                                                             \
                if (busy * phases <= tailLanes) { tailFrom = kBase; break; } /* every strip has `phases` lanes */ \
            }                                                                                                 \
            for (int i = g0; i < gEnd; ++i) {                                                                 \
                const int fullLen = rS ? rS[i] : maxNnz;                                                      \
                const int len = P##_walked_keep(rS, maxNnz, rows, i, deepCap, deepKeep);                                     \
                const size_t slot0 = hackOffsets ? (size_t)hackOffsets[i / hackSize] + (size_t)(i % hackSize) : (size_t)i; \
                const size_t vs = hackOffsets ? (size_t)hackSize : (size_t)cMPitch;                           \
                const size_t is = hackOffsets ? (size_t)hackSize : (size_t)rPPitch;                           \
                T head[ORC_MAX_PHASES]; /* k < tailFrom: `phases` partial sums by k mod phases */             \
                for (int p = 0; p < phases; ++p) head[p] = P##_zero();                                        \
                for (int k = 0; k < len && k < tailFrom; ++k) {                                               \
                    const int col = rP[slot0 + (size_t)k * is] - baseIndex;                                   \
                    if (col >= 0) head[k % phases] = P##_fma(cM[slot0 + (size_t)k * vs], x[col], head[k % phases]); \
                }                                                                                             \
                if (len > tailFrom) { /* the 64-way tail sum joins the phase-0 partial before the phases combine */ \
                    T part[ORC_MAX_PHASES];                                                                   \
                    for (int p = 0; p < 64; ++p) part[p] = P##_zero();                                        \
                    for (int k = tailFrom; k < len; ++k) {                                                    \
                        const int col = rP[slot0 + (size_t)k * is] - baseIndex;                               \
                        if (col >= 0)                                                                         \
                            part[(k - tailFrom) % 64] = P##_fma(cM[slot0 + (size_t)k * vs], x[col], part[(k - tailFrom) % 64]); \
                    }                                                                                         \
                    head[0] = P##_add(head[0], P##_combine(part, 64));                                        \
                }                                                                                             \
                T total = P##_combine(head, phases);                                                          \
                if (mainChunk > 0) {                                                                          \
                    int walkedDepth = 0; /* of row i's 32-row sub-group */                                     \
                    for (int r = i / 32 * 32; r < i / 32 * 32 + 32 && r < rows; ++r) {                         \
                        const int l = P##_walked_keep(rS, maxNnz, rows, r, deepCap, deepKeep);                               \
                        if (l > walkedDepth) walkedDepth = l;                                                 \
                    }                                                                                         \
                    if (walkedDepth > mainChunk) {                                                            \
                        for (int c0 = 0; c0 < walkedDepth; c0 += mainChunk) {                                 \
                            T part[ORC_MAX_PHASES];                                                           \
                            for (int p = 0; p < phases; ++p) part[p] = P##_zero();                            \
                            for (int k = c0; k < len && k < c0 + mainChunk; ++k) {                            \
                                const int col = rP[slot0 + (size_t)k * is] - baseIndex;                       \
                                if (col >= 0) part[k % phases] = P##_fma(cM[slot0 + (size_t)k * vs], x[col], part[k % phases]); \
                            }                                                                                 \
                            total = c0 == 0 ? P##_combine(part, phases) : P##_add(total, P##_combine(part, phases)); \
                        }                                                                                     \
                    }                                                                                         \
                }                                                                                             \
                const int subDepth = P##_deep_group(rS, maxNnz, rows, i, deepCap);                            \
                if (subDepth) {                                                                               \
                    /* deepSpmvKernel: columns >= deepCap in chunks of deepChunk; a chunk's deepPhases phase sums */ \
                    /* (each over ascending k) are combined pairwise; the chunk sums join the slab kernel's sum in */ \
                    /* chunk order */                                                                         \
                    /* chunk order; a row shorter than its sub-group adds the later chunks' +0 as the kernel does */ \
                    for (int c0 = deepKeep; c0 < subDepth; c0 += deepChunk) {                                  \
                        T part[ORC_MAX_PHASES];                                                               \
                        for (int p = 0; p < deepPhases; ++p) part[p] = P##_zero();                            \
                        for (int k = c0; k < fullLen && k < c0 + deepChunk; ++k) {                            \
                            const int col = rP[slot0 + (size_t)k * is] - baseIndex;                           \
                            if (col >= 0)                                                                     \
                                part[(k - deepKeep) % deepPhases] = P##_fma(cM[slot0 + (size_t)k * vs], x[col], part[(k - deepKeep) % deepPhases]); \
                        }                                                                                     \
                        total = P##_add(total, P##_combine(part, deepPhases));                                \
                    }                                                                                         \
                }                                                                                             \
                P##_store(z, y, rIdx ? rIdx[i] : i, alpha, total, beta);                                      \
            }                                                                                                 \
        }                                                                                                     \
    }                                                                                                         \
    void orc_##P##spmv_deep(T* z, const T* y, T alpha, const T* cM, const int* rP, int hackSize,             \
                            const int* hackOffsets, int cMPitch, int rPPitch, const int* rS, int maxNnz,      \
                            const int* rIdx, int rows, const T* x, T beta, int baseIndex, int groupRows,      \
                            int rowsPerLane, int step, int tailLanes, int phases, int deepCap, int deepPhases,\
                            int deepChunk)                                                                    \
    {                                                                                                         \
        orc_##P##spmv_split(z, y, alpha, cM, rP, hackSize, hackOffsets, cMPitch, rPPitch, rS, maxNnz, rIdx, rows, x, beta, \
                            baseIndex, groupRows, rowsPerLane, step, tailLanes, phases, deepCap, deepPhases, deepChunk, 0, deepCap); \
    }                                                                                                         \
    void orc_##P##spmv_tail(T* z, const T* y, T alpha, const T* cM, const int* rP, int hackSize,             \
                            const int* hackOffsets, int cMPitch, int rPPitch, const int* rS, int maxNnz,      \
                            const int* rIdx, int rows, const T* x, T beta, int baseIndex, int groupRows,      \
                            int rowsPerLane, int step, int tailLanes, int phases)                             \
    {                                                                                                         \
        orc_##P##spmv_deep(z, y, alpha, cM, rP, hackSize, hackOffsets, cMPitch, rPPitch, rS, maxNnz, rIdx, rows, x, beta, \
                           baseIndex, groupRows, rowsPerLane, step, tailLanes, phases, 0, 1, 1);              \
    }
ORC_DEFINE_TAIL(s, float)
ORC_DEFINE_TAIL(d, double)
ORC_DEFINE_TAIL(c, orc_cfloat)
ORC_DEFINE_TAIL(z, orc_cdouble)

/* DIA SpMV (dia_spmv_base_template.cuh:20-216): diagonals in stored order, slot counts iff 0 <= offsets[d]+i < cols.
 * ELL csput (ell_csput_base.cuh:33-75): binary search of aJ among the row's stored indices, overwrite on a hit;
 * alpha unused, aJ compared with the stored index as is. */
#define ORC_DEFINE_DIA_CSPUT(P, T)                                                                            \
    void orc_##P##diaspmv(T* z, const T* y, T alpha, const T* dM, const int* offsets, int pitch, int rows,    \
                          int cols, int diags, const T* x, T beta)                                            \
    {                                                                                                         \
        for (int i = 0; i < rows; ++i) {                                                                      \
            T sum = P##_zero();                                                                               \
            for (int d = 0; d < diags; ++d) {                                                                 \
                const long long col = (long long)offsets[d] + i;                                              \
                if (col >= 0 && col < cols)                                                                   \
                    sum = P##_fma(dM[(size_t)i + (size_t)d * (size_t)pitch], x[col], sum);                    \
            }                                                                                                 \
            P##_store(z, y, i, alpha, sum, beta);                                                             \
        }                                                                                                     \
    }                                                                                                         \
    void orc_##P##ellcsput(T* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, int nnz,            \
                           const int* aI, const int* aJ, const T* aVal, int baseIndex)                        \
    {                                                                                                         \
        for (int i = 0; i < nnz; ++i) {                                                                       \
            const int row = aI[i] - baseIndex;                                                                \
            if (row < 0)                                                                                      \
                continue;                                                                                     \
            int lower = 0, upper = rS[row] - 1;                                                               \
            while (lower <= upper) {                                                                          \
                const int mid = (lower + upper) / 2;                                                          \
                const int cur = rP[(size_t)row + (size_t)mid * (size_t)rPPitch];                              \
                if (cur == aJ[i]) {                                                                           \
                    cM[(size_t)row + (size_t)mid * (size_t)cMPitch] = aVal[i];                                \
                    break;                                                                                    \
                }                                                                                             \
                if (cur < aJ[i]) lower = mid + 1; else upper = mid - 1;                                       \
            }                                                                                                 \
        }                                                                                                     \
    }
ORC_DEFINE_DIA_CSPUT(s, float)
ORC_DEFINE_DIA_CSPUT(d, double)
ORC_DEFINE_DIA_CSPUT(c, orc_cfloat)
ORC_DEFINE_DIA_CSPUT(z, orc_cdouble)

/* axpby: daxpby.cu:31-45 (alpha*x + beta*y, contracted as fma(alpha, x, beta*y));
 * caxpby.cu:41-44 (fma(beta, y, alpha*x)); zaxpby.cu:42-45 (fma(alpha, x, beta*y)).
 * beta == 0: z = alpha*x, y unread. */
void orc_saxpby(float* z, int n, float beta, const float* y, float alpha, const float* x)
{
    for (int i = 0; i < n; ++i)
        z[i] = beta == 0.0f ? alpha * x[i] : fmaf(alpha, x[i], beta * y[i]);
}
void orc_daxpby(double* z, int n, double beta, const double* y, double alpha, const double* x)
{
    for (int i = 0; i < n; ++i)
        z[i] = beta == 0.0 ? alpha * x[i] : fma(alpha, x[i], beta * y[i]);
}
void orc_caxpby(orc_cfloat* z, int n, orc_cfloat beta, const orc_cfloat* y, orc_cfloat alpha, const orc_cfloat* x)
{
    for (int i = 0; i < n; ++i)
        z[i] = c_nz(beta) ? c_fma(beta, y[i], c_mul(alpha, x[i])) : c_mul(alpha, x[i]);
}
void orc_zaxpby(orc_cdouble* z, int n, orc_cdouble beta, const orc_cdouble* y, orc_cdouble alpha, const orc_cdouble* x)
{
    for (int i = 0; i < n; ++i)
        z[i] = z_nz(beta) ? z_fma(alpha, x[i], z_mul(beta, y[i])) : z_mul(alpha, x[i]);
}

/* ======================================================================== */
/* Rest of Level-1: scal_base.cuh:34-45, abs_base.cuh:43-70,    */
/* axy_base.cuh:37-47,95-176, gath_base.cuh:32-45, scat_base.cuh:32-48,       */
/* setscal_base.cuh:32-82; asum/amax follow the DOCUMENTED semantics          */
/* (vector.h:319-339), not the defective final reduction of asum_base.cuh.    */
/* ======================================================================== */

/* cuCabs / cuCabsf (CUDA cuComplex.h): v*sqrt(1 + (w/v)^2), v = max(|re|,|im|), w = min. */
static inline float s_mag(float v) { return fabsf(v); }
static inline double d_mag(double v) { return fabs(v); }
static inline float c_mag(orc_cfloat z)
{
    const float a = fabsf(z.x), b = fabsf(z.y), v = a > b ? a : b, w = a > b ? b : a;
    float t = w / v;
    t = fmaf(t, t, 1.0f);
    t = v * sqrtf(t);
    return (v == 0.0f || v > 3.402823466e38f || w > 3.402823466e38f) ? v + w : t;
}
static inline double z_mag(orc_cdouble z)
{
    const double a = fabs(z.x), b = fabs(z.y), v = a > b ? a : b, w = a > b ? b : a;
    double t = w / v;
    t = fma(t, t, 1.0);
    t = v * sqrt(t);
    return (v == 0.0 || v > 1.79769313486231570e+308 || w > 1.79769313486231570e+308) ? v + w : t;
}
static inline float s_fromMag(float m) { return m; }
static inline double d_fromMag(double m) { return m; }
static inline orc_cfloat c_fromMag(float m) { orc_cfloat r = {m, 0.0f}; return r; }
static inline orc_cdouble z_fromMag(double m) { orc_cdouble r = {m, 0.0}; return r; }
static inline int s_isOne(float a) { return a == 1.0f; }
static inline int d_isOne(double a) { return a == 1.0; }
static inline int c_isOne(orc_cfloat a) { return a.x == 1.0f && a.y == 0.0f; }
static inline int z_isOne(orc_cdouble a) { return a.x == 1.0 && a.y == 0.0; }

#define ORC_DEFINE_LEVEL1(P, T, R)                                                                            \
    void orc_##P##scal(T* y, int n, T alpha, const T* x)                                                      \
    {                                                                                                         \
        for (int i = 0; i < n; ++i)                                                                           \
            y[i] = P##_mul(alpha, x[i]);                                                                      \
    }                                                                                                         \
    void orc_##P##abs(T* y, int n, T alpha, const T* x)                                                       \
    {                                                                                                         \
        for (int i = 0; i < n; ++i) {                                                                         \
            const T m = P##_fromMag(P##_mag(x[i]));                                                           \
            y[i] = P##_isOne(alpha) ? m : P##_mul(alpha, m);                                                  \
        }                                                                                                     \
    }                                                                                                         \
    void orc_##P##axy(T* z, int n, T alpha, const T* x, const T* y)                                           \
    {                                                                                                         \
        for (int i = 0; i < n; ++i)                                                                           \
            z[i] = P##_mul(alpha, P##_mul(x[i], y[i]));                                                       \
    }                                                                                                         \
    void orc_##P##axypbz(T* w, int n, T beta, const T* z, T alpha, const T* x, const T* y)                    \
    {                                                                                                         \
        for (int i = 0; i < n; ++i) {                                                                         \
            if (!P##_nz(alpha))                                                                               \
                w[i] = P##_mul(beta, z[i]);                                                                   \
            else if (!P##_nz(beta))                                                                           \
                w[i] = P##_mul(alpha, P##_mul(x[i], y[i]));                                                   \
            else                                                                                              \
                w[i] = P##_fma(alpha, P##_mul(x[i], y[i]), P##_mul(beta, z[i]));                              \
        }                                                                                                     \
    }                                                                                                         \
    void orc_##P##gath(T* xValues, int xNnz, const int* xIndices, int base, const T* y)                       \
    {                                                                                                         \
        for (int i = 0; i < xNnz; ++i)                                                                        \
            if (xIndices[i] - base >= 0)                                                                      \
                xValues[i] = y[xIndices[i] - base];                                                           \
    }                                                                                                         \
    void orc_##P##scat(T* y, int xNnz, const T* xValues, const int* xIndices, int base, T beta)               \
    {                                                                                                         \
        for (int i = 0; i < xNnz; ++i) {                                                                      \
            const int pos = xIndices[i] - base;                                                               \
            if (pos < 0)                                                                                      \
                continue;                                                                                     \
            y[pos] = P##_nz(beta) ? P##_fma(beta, y[pos], xValues[i]) : xValues[i];                           \
        }                                                                                                     \
    }                                                                                                         \
    void orc_##P##setscal(int first, int last, int base, T val, T* y)                                         \
    {                                                                                                         \
        for (int i = first - base; i <= last - base; ++i)                                                     \
            y[i] = val;                                                                                       \
    }                                                                                                         \
    void orc_##P##asum(R* out, int n, const T* x)                                                             \
    {                                                                                                         \
        R acc = 0;                                                                                            \
        for (int i = 0; i < n; ++i)                                                                           \
            acc += P##_mag(x[i]);                                                                             \
        *out = acc;                                                                                           \
    }                                                                                                         \
    void orc_##P##amax(R* out, int n, const T* x)                                                             \
    {                                                                                                         \
        R acc = 0;                                                                                            \
        for (int i = 0; i < n; ++i)                                                                           \
            if (P##_mag(x[i]) > acc)                                                                          \
                acc = P##_mag(x[i]);                                                                          \
        *out = acc;                                                                                           \
    }

ORC_DEFINE_LEVEL1(s, float, float)
ORC_DEFINE_LEVEL1(d, double, double)
ORC_DEFINE_LEVEL1(c, orc_cfloat, float)
ORC_DEFINE_LEVEL1(z, orc_cdouble, double)

/* integer flavours: gath_base.cuh / scat_base.cuh / setscal_base.cuh with VALUE_TYPE int (igath.cu, iscat.cu, isetscal.cu) */
void orc_igath(int* xValues, int xNnz, const int* xIndices, int base, const int* y)
{
    for (int i = 0; i < xNnz; ++i)
        if (xIndices[i] - base >= 0)
            xValues[i] = y[xIndices[i] - base];
}
void orc_iscat(int* y, int xNnz, const int* xValues, const int* xIndices, int base, int beta)
{
    for (int i = 0; i < xNnz; ++i) {
        const int pos = xIndices[i] - base;
        if (pos >= 0)
            y[pos] = beta != 0 ? beta * y[pos] + xValues[i] : xValues[i];
    }
}
void orc_isetscal(int first, int last, int base, int val, int* y)
{
    for (int i = first - base; i <= last - base; ++i)
        y[i] = val;
}

/* Multi-vector SpMM oracle for the row-sharded path (NEW operation, not in the
 * reference; include/spgpu/spmm.h).  Interleaved multivectors: element j of row i
 * at M[i*ld + j].  Per (row, rhs): products added in ascending k -- the order of
 * the reference's one-thread-per-row HELL kernel (hell_spmv_base_template.cuh:197-214)
 * -- then the usual epilogue (:219-222). */
#define ORC_DEFINE_SPMM(P, T)                                                                                 \
    void orc_##P##hellspmm(T* Z, const T* Y, T alpha, const T* cM, const int* rP, int hackSize,               \
                           const int* hackOffsets, const int* rS, const int* rIdx, int rows, const T* X,      \
                           T beta, int baseIndex, int count, int ldX, int ldYZ)                               \
    {                                                                                                         \
        _Pragma("omp parallel for schedule(static)")                                                          \
        for (int i = 0; i < rows; ++i) {                                                                      \
            const size_t slot0 = (size_t)hackOffsets[i / hackSize] + (size_t)(i % hackSize);                  \
            const size_t out = (size_t)(rIdx ? rIdx[i] : i) * (size_t)ldYZ;                                   \
            if (Y == Z && beta == (T)1 && rS[i] == 0)                                                         \
                continue; /* in-place sum: rows without entries are left untouched (spmm.h) */                \
            for (int j = 0; j < count; ++j) {                                                                 \
                T sum = P##_zero();                                                                           \
                for (int k = 0; k < rS[i]; ++k) {                                                             \
                    const size_t s = slot0 + (size_t)k * (size_t)hackSize;                                    \
                    sum = P##_fma(cM[s], X[(size_t)(rP[s] - baseIndex) * (size_t)ldX + j], sum);              \
                }                                                                                             \
                if (P##_nz(beta))                                                                             \
                    Z[out + j] = P##_fma(beta, Y[out + j], P##_mul(alpha, sum));                              \
                else                                                                                          \
                    Z[out + j] = P##_mul(alpha, sum);                                                         \
            }                                                                                                 \
        }                                                                                                     \
    }
ORC_DEFINE_SPMM(s, float)
ORC_DEFINE_SPMM(d, double)

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}
