/*
 * Plain-C performance harness with the flow of the reference's src/tests/hellPerf.cpp:127-317
 * (COO -> computeEllRowLenghts/cooToEll -> ELL run; computeHellAllocSize/ellToHell -> HELL run; ellToOell -> ordered ELL run;
 * per format: 1 warm-up, dot(z,z) printed as checksum, N timed launches, GFlop/s), on a Matrix Market
 * file as the reference's harness (symmetric storage unfolded, hellPerf.cpp:93-113) or on a synthetic
 * matrix (no .mtx ships with the reference), and with HIP in place of the CUDA runtime.  alpha = 1, beta = 0 as hellPerf.cpp:27-28.  Adds what the reference's harness
 * lacks: the two dot(z,z) are compared, and achieved HBM GB/s is printed next to GFlop/s.
 *
 *   usage: hellperf_amd [rows=1000000] [nnzPerRow=32] [banded|random] [reps=200] [s|d] [norowsize]
 *          hellperf_amd matrix.mtx [reps=200] [s|d] [norowsize]
 * norowsize: the ELL run gets rS == NULL -- the reference builds that as separate executables (hellperf_norowsize_s/_d:
 * -DNO_ROW_SIZE, src/CMakeLists.txt:186-188, hellPerf.cpp:200-204); every row is then walked to maxRowSize over cooToEll's
 * zero padding, and the result must equal the run with row sizes.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spgpu/core.h"
#include "spgpu/ell.h"
#include "spgpu/ell_conv.h"
#include "spgpu/hell.h"
#include "spgpu/hell_conv.h"
#include "spgpu/mmread.h"
#include "spgpu/tuning.h"
#include "spgpu/vector.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_));   \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

static uint64_t splitmix(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double unit(uint64_t* s) { return (double)(splitmix(s) >> 11) / 9007199254740992.0; }

/* Matrix Market file -> zero-based COO in the requested precision; symmetric storage is mirrored. */
static int loadMtx(const char* path, int dbl, int* rows, int* cols, long long* nnz, int** cooR, int** cooC, void** cooV)
{
    int prop[6];
    if (!spgpuMmProperties(path, prop)) { fprintf(stderr, "%s: not a readable Matrix Market file\n", path); return 0; }
    int n = prop[2];
    int *r = (int*)malloc((size_t)n * sizeof(int)), *c = (int*)malloc((size_t)n * sizeof(int));
    double* v = (double*)malloc((size_t)n * sizeof(double));
    const int code = spgpuMmReadCoo(path, 'd', v, r, c);
    if (code != 0) { fprintf(stderr, "%s: read failed with code %d\n", path, code); return 0; }
    if (prop[5] == 1 /* MATRIX_TYPE_SYMMETRIC */) {
        const int total = spgpuMmUnfoldedSizeD(v, r, c, n);
        int *ur = (int*)malloc((size_t)total * sizeof(int)), *uc = (int*)malloc((size_t)total * sizeof(int));
        double* uv = (double*)malloc((size_t)total * sizeof(double));
        spgpuMmUnfoldD(ur, uc, uv, r, c, v, n);
        free(r); free(c); free(v);
        r = ur; c = uc; v = uv; n = total;
        printf("symmetric storage unfolded: %d entries\n", n);
    }
    *rows = prop[0]; *cols = prop[1]; *nnz = n; *cooR = r; *cooC = c;
    if (dbl) {
        *cooV = v;
    } else {
        float* f = (float*)malloc((size_t)n * sizeof(float));
        for (int e = 0; e < n; ++e) f[e] = (float)v[e];
        free(v);
        *cooV = f;
    }
    return 1;
}

int main(int argc, char** argv)
{
    const int fromFile = argc > 1 && strstr(argv[1], ".mtx") != NULL;
    int rows = !fromFile && argc > 1 ? atoi(argv[1]) : 1000000;
    int cols = rows;
    int perRow = !fromFile && argc > 2 ? atoi(argv[2]) : 32;
    const int randomCols = !fromFile && argc > 3 && strcmp(argv[3], "random") == 0;
    const int reps = fromFile ? (argc > 2 ? atoi(argv[2]) : 200) : (argc > 4 ? atoi(argv[4]) : 200);
    const char* prec = fromFile ? (argc > 3 ? argv[3] : "d") : (argc > 5 ? argv[5] : "d");
    const int dbl = prec[0] != 's';
    const char* last = argc > 1 ? argv[argc - 1] : "";
    const int noRowSize = strcmp(last, "norowsize") == 0; /* hellPerf.cpp:200-204 */
    const size_t es = dbl ? sizeof(double) : sizeof(float);
    const spgpuType_t type = dbl ? SPGPU_TYPE_DOUBLE : SPGPU_TYPE_FLOAT;
    const int hackSize = 32; /* hellPerf.cpp:254 */
    long long nnz = (long long)rows * perRow;
    if (nnz > 2000000000LL) { fprintf(stderr, "too many nonzeros for int indices\n"); return 2; }

    int *cooR, *cooC;
    void* cooV;
    uint64_t seed = 1;
    if (fromFile) {
        if (!loadMtx(argv[1], dbl, &rows, &cols, &nnz, &cooR, &cooC, &cooV)) return 2;
        perRow = rows ? (int)(nnz / rows) : 0; /* ellAvgRowSize, hellPerf.cpp:131 */
    } else {
        /* ---- synthetic COO ---- */
        cooR = (int*)malloc(nnz * sizeof(int));
        cooC = (int*)malloc(nnz * sizeof(int));
        cooV = malloc(nnz * es);
        for (long long e = 0; e < nnz; ++e) {
            const int r = (int)(e / perRow), k = (int)(e % perRow);
            cooR[e] = r;
            cooC[e] = randomCols ? (int)(splitmix(&seed) % (uint64_t)rows)
                                 : (int)(((long long)r + k - perRow / 2 + rows) % rows);
            if (dbl) ((double*)cooV)[e] = unit(&seed); else ((float*)cooV)[e] = (float)unit(&seed);
        }
    }
    void *x = malloc((size_t)cols * es), *y = malloc((size_t)rows * es);
    for (int i = 0; i < cols; ++i) {
        if (dbl) ((double*)x)[i] = unit(&seed); else ((float*)x)[i] = (float)unit(&seed);
    }
    for (int i = 0; i < rows; ++i) {
        if (dbl) ((double*)y)[i] = unit(&seed); else ((float*)y)[i] = (float)unit(&seed);
    }

    /* ---- COO -> ELL -> HELL on the host (hellPerf.cpp:136-152, 254-264) ---- */
    int maxRow = 0, height = 0;
    int* rowLen = (int*)malloc(rows * sizeof(int));
    computeEllRowLenghts(rowLen, &maxRow, rows, (int)nnz, cooR, 0);
    const int pitch = computeEllAllocPitch(rows);
    void* ellV = calloc((size_t)maxRow * pitch, es);
    int* ellI = (int*)calloc((size_t)maxRow * pitch, sizeof(int));
    cooToEll(ellV, ellI, pitch, pitch, maxRow, 0, rows, (int)nnz, cooR, cooC, cooV, 0, type);
    computeHellAllocSize(&height, hackSize, rows, rowLen);
    const int hacks = (rows + hackSize - 1) / hackSize;
    void* hellV = calloc((size_t)hackSize * height, es);
    int* hellI = (int*)calloc((size_t)hackSize * height, sizeof(int));
    int* hackOff = (int*)calloc(hacks, sizeof(int));
    ellToHell(hellV, hellI, hackOff, hackSize, ellV, ellI, pitch, pitch, rowLen, rows, type);
    printf("%d rows, %d columns, %lld nnz, %s, %s | ELL %d x %d (%.1f MB) | HELL height %d (%.1f MB)\n", rows, cols, nnz,
           fromFile ? argv[1] : (randomCols ? "random columns" : "banded"), dbl ? "double" : "float", maxRow, pitch,
           (double)maxRow * pitch * (es + 4) * 1e-6, height, (double)hackSize * height * (es + 4) * 1e-6);

    /* ---- upload (hellPerf.cpp:176-190, 274-280) ---- */
    void *dX, *dY, *dZ, *dEllV, *dHellV;
    int *dRs, *dEllI, *dHellI, *dHack;
    CHECK(hipMalloc(&dX, (size_t)cols * es)); CHECK(hipMalloc(&dY, rows * es)); CHECK(hipMalloc(&dZ, rows * es));
    CHECK(hipMalloc((void**)&dRs, rows * sizeof(int)));
    CHECK(hipMalloc(&dEllV, (size_t)maxRow * pitch * es)); CHECK(hipMalloc((void**)&dEllI, (size_t)maxRow * pitch * sizeof(int)));
    CHECK(hipMalloc(&dHellV, (size_t)hackSize * height * es)); CHECK(hipMalloc((void**)&dHellI, (size_t)hackSize * height * sizeof(int)));
    CHECK(hipMalloc((void**)&dHack, hacks * sizeof(int)));
    CHECK(hipMemcpy(dX, x, (size_t)cols * es, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dY, y, rows * es, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRs, rowLen, rows * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dEllV, ellV, (size_t)maxRow * pitch * es, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dEllI, ellI, (size_t)maxRow * pitch * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHellV, hellV, (size_t)hackSize * height * es, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHellI, hellI, (size_t)hackSize * height * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHack, hackOff, hacks * sizeof(int), hipMemcpyHostToDevice));

    spgpuHandle_t h;
    if (spgpuCreate(&h, 0) != SPGPU_SUCCESS) return 2;
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const double bytes = (double)nnz * (es + 4) + (double)rows * (4 + es) + (double)cols * es;
    double dots[4];
    int frozen = 0, adopted = 0;

    /* format 2 (not in the reference): the HELL run once more after spgpuHellSpmvFreeze -- this loop never touches the index
     * arrays, which is all the call asks the caller to promise; a matrix with scattered columns is not frozen (the line is left out) */
    /* format 3 (not in the reference either): the HELL run after spgpuHellSpmvAdopt -- a matrix with very unequal row lengths of which
     * the library keeps an ordered copy (what the third leg below does by hand with ellToOell); refused for rows about equally long */
    for (int format = 0; format < 4; ++format) {
        if (format == 2) {
            frozen = spgpuHellSpmvFreeze(h, type, dHellV, dHellI, hackSize, dHack, dRs, NULL, rows, 0) == SPGPU_SUCCESS;
            if (!frozen)
                continue;
        }
        if (format == 3) {
            if (frozen)
                spgpuSpmvThaw(h, dHellI);
            adopted = spgpuHellSpmvAdopt(h, type, dHellV, dHellI, hackSize, dHack, dRs, rows, 0) == SPGPU_SUCCESS;
            if (!adopted)
                break;
        }
#define RUN()                                                                                                   \
        do {                                                                                                    \
            if (format == 0 && dbl)  spgpuDellspmv(h, dZ, dY, 1.0, dEllV, dEllI, pitch, pitch, noRowSize ? NULL : dRs, NULL, perRow, maxRow, rows, dX, 0.0, 0); \
            if (format == 0 && !dbl) spgpuSellspmv(h, dZ, dY, 1.0f, dEllV, dEllI, pitch, pitch, noRowSize ? NULL : dRs, NULL, perRow, maxRow, rows, dX, 0.0f, 0); \
            if (format >= 1 && dbl)  spgpuDhellspmv(h, dZ, dY, 1.0, dHellV, dHellI, hackSize, dHack, dRs, NULL, maxRow, rows, dX, 0.0, 0); \
            if (format >= 1 && !dbl) spgpuShellspmv(h, dZ, dY, 1.0f, dHellV, dHellI, hackSize, dHack, dRs, NULL, maxRow, rows, dX, 0.0f, 0); \
        } while (0)
        RUN(); /* warm-up */
        dots[format] = dbl ? spgpuDdot(h, rows, dZ, dZ) : (double)spgpuSdot(h, rows, dZ, dZ);
        hipStream_t s = spgpuGetStream(h);
        CHECK(hipEventRecord(t0, s));
        for (int i = 0; i < reps; ++i) RUN();
        CHECK(hipEventRecord(t1, s));
        CHECK(hipEventSynchronize(t1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        const double t = ms * 1e-3 / reps;
        printf("%s dot res: %.10e | %.4f ms | %.1f GFlop/s | %.1f GB/s (%.1f%% of 8 TB/s)\n",
               format == 3 ? "HELL adopted" : format == 2 ? "HELL frozen" : format ? "HELL" : (noRowSize ? "ELL (rS == NULL)" : "ELL "),
               dots[format], t * 1e3, 2.0 * nnz / t * 1e-9, bytes / t * 1e-9, bytes / t * 1e-9 / 80.0);
    }
    if (frozen) {
        printf(dots[2] == dots[1] ? "frozen HELL checksum identical: PASSED\n" : "frozen HELL checksum differs: FAILED\n");
        if (dots[2] != dots[1])
            return 1;
    }
    if (adopted) { /* another order of additions (the ordered kernel's): equal within rounding */
        const int closeAdopted = fabs(dots[3] - dots[1]) <= (dbl ? 1e-10 : 1e-4) * fabs(dots[1]);
        printf(closeAdopted ? "adopted HELL checksum equal within rounding: PASSED\n" : "adopted HELL checksum differs: FAILED\n");
        if (!closeAdopted)
            return 1;
    }
    spgpuSpmvThaw(h, dHellI);
    /* ---- third format of the reference's harness: ordered ELL (hellPerf.cpp:320-378).  ellToOell on the host, the
     * row order handed to spgpu?ellspmv as rIdx.  (The reference's harness uploads the ordered row lengths to devRs but
     * passes devEllRs -- the unordered ones -- to the kernel, hellPerf.cpp:342,352; here the ordered lengths are used.) ---- */
    int* oellIdx = (int*)malloc(rows * sizeof(int));
    int* oellLen = (int*)malloc(rows * sizeof(int));
    void* oellV = calloc((size_t)maxRow * pitch, es);
    int* oellI = (int*)calloc((size_t)maxRow * pitch, sizeof(int));
    ellToOell(oellIdx, oellV, oellI, oellLen, ellV, ellI, rowLen, pitch, pitch, rows, type);
    int* dRidx;
    CHECK(hipMalloc((void**)&dRidx, rows * sizeof(int)));
    CHECK(hipMemcpy(dRidx, oellIdx, rows * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRs, oellLen, rows * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dEllV, oellV, (size_t)maxRow * pitch * es, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dEllI, oellI, (size_t)maxRow * pitch * sizeof(int), hipMemcpyHostToDevice));
#define RUN_OELL()                                                                                                  \
    do {                                                                                                            \
        if (dbl) spgpuDellspmv(h, dZ, dY, 1.0, dEllV, dEllI, pitch, pitch, dRs, dRidx, perRow, maxRow, rows, dX, 0.0, 0); \
        else     spgpuSellspmv(h, dZ, dY, 1.0f, dEllV, dEllI, pitch, pitch, dRs, dRidx, perRow, maxRow, rows, dX, 0.0f, 0); \
    } while (0)
    RUN_OELL();
    const double dotOell = dbl ? spgpuDdot(h, rows, dZ, dZ) : (double)spgpuSdot(h, rows, dZ, dZ);
    {
        hipStream_t s = spgpuGetStream(h);
        CHECK(hipEventRecord(t0, s));
        for (int i = 0; i < reps; ++i) RUN_OELL();
        CHECK(hipEventRecord(t1, s));
        CHECK(hipEventSynchronize(t1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        const double t = ms * 1e-3 / reps;
        printf("OELL dot res: %.10e | %.4f ms | %.1f GFlop/s | %.1f GB/s (%.1f%% of 8 TB/s)\n", dotOell, t * 1e3, 2.0 * nnz / t * 1e-9,
               (bytes + 4.0 * rows) / t * 1e-9, (bytes + 4.0 * rows) / t * 1e-9 / 80.0);
    }
    spgpuDestroy(h);
    CHECK(hipGetLastError());
    const int same = dots[0] == dots[1];
    /* the ordered run adds a row's products in another order (the kernel for ordered rows): equal within rounding */
    const int close = fabs(dotOell - dots[0]) <= (dbl ? 1e-10 : 1e-4) * fabs(dots[0]);
    printf(same ? "ELL and HELL checksums identical: PASSED\n" : "ELL and HELL checksums differ: FAILED\n");
    printf(close ? "OELL checksum equal within rounding: PASSED\n" : "OELL checksum differs: FAILED\n");
    return same && close ? 0 : 1;
}
