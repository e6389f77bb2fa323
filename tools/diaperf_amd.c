/*
 * Plain-C performance harness with the flow of the reference's src/tests/diaPerf.cpp:127-340
 * (COO -> computeDiaDiagonalsCount/coo2dia -> DIA run; computeHdiaHackOffsetsFromCoo/cooToHdia -> HDIA run;
 * per format: 1 warm-up, dot(z,z) printed as checksum, N timed launches, GFlop/s) on a Matrix Market file as the
 * reference's harness, or on a synthetic stencil (no .mtx ships with the reference): the 7-point Laplacian on an
 * m x m x m grid (BASELINE configs[3]'s matrix) or the 5-point one on m x m.  alpha = 1, beta = 0 as diaPerf.cpp:27-28.
 * Adds what the reference's harness lacks: the two checksums are compared, and achieved HBM GB/s is printed.
 *
 *   usage: diaperf_amd [m=128] [7|5] [reps=100] [s|d]
 *          diaperf_amd matrix.mtx [reps=100] [s|d]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spgpu/core.h"
#include "spgpu/dia.h"
#include "spgpu/dia_conv.h"
#include "spgpu/hdia.h"
#include "spgpu/hdia_conv.h"
#include "spgpu/mmread.h"
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

int main(int argc, char** argv)
{
    const int fromFile = argc > 1 && strstr(argv[1], ".mtx") != NULL;
    const int m = !fromFile && argc > 1 ? atoi(argv[1]) : 128;
    const int points = !fromFile && argc > 2 ? atoi(argv[2]) : 7;
    const int reps = fromFile ? (argc > 2 ? atoi(argv[2]) : 100) : (argc > 3 ? atoi(argv[3]) : 100);
    const char* prec = fromFile ? (argc > 3 ? argv[3] : "d") : (argc > 4 ? argv[4] : "d");
    const int dbl = prec[0] != 's';
    const size_t es = dbl ? sizeof(double) : sizeof(float);
    const spgpuType_t type = dbl ? SPGPU_TYPE_DOUBLE : SPGPU_TYPE_FLOAT;
    const int hackSize = 32; /* diaPerf.cpp:256 */

    int rows = 0, cols = 0, nnz = 0;
    int *cooR = NULL, *cooC = NULL;
    double* val = NULL;
    if (fromFile) {
        int prop[6];
        if (!spgpuMmProperties(argv[1], prop)) { fprintf(stderr, "%s: not a readable Matrix Market file\n", argv[1]); return 2; }
        int n = prop[2];
        cooR = (int*)malloc((size_t)n * sizeof(int));
        cooC = (int*)malloc((size_t)n * sizeof(int));
        val = (double*)malloc((size_t)n * sizeof(double));
        if (spgpuMmReadCoo(argv[1], 'd', val, cooR, cooC) != 0) { fprintf(stderr, "%s: read failed\n", argv[1]); return 2; }
        if (prop[5] == 1 /* MATRIX_TYPE_SYMMETRIC: unfolded as diaPerf.cpp:93-113 does */) {
            const int total = spgpuMmUnfoldedSizeD(val, cooR, cooC, n);
            int *ur = (int*)malloc((size_t)total * sizeof(int)), *uc = (int*)malloc((size_t)total * sizeof(int));
            double* uv = (double*)malloc((size_t)total * sizeof(double));
            spgpuMmUnfoldD(ur, uc, uv, cooR, cooC, val, n);
            free(cooR); free(cooC); free(val);
            cooR = ur; cooC = uc; val = uv; n = total;
            printf("symmetric storage unfolded: %d entries\n", n);
        }
        rows = prop[0]; cols = prop[1]; nnz = n;
    } else {
        /* natural order; per row ascending column: -M^2, -M, -1, 0, +1, +M, +M^2 (7-point) */
        const long long total = points == 7 ? (long long)m * m * m : (long long)m * m;
        if (total > 300000000LL) { fprintf(stderr, "grid too large for this harness\n"); return 2; }
        rows = cols = (int)total;
        const long long plane = (long long)m * m;
        cooR = (int*)malloc((size_t)total * points * sizeof(int));
        cooC = (int*)malloc((size_t)total * points * sizeof(int));
        val = (double*)malloc((size_t)total * points * sizeof(double));
        long long e = 0;
        for (long long i = 0; i < total; ++i) {
            const int ix = (int)(i % m), iy = (int)((i / m) % m), iz = (int)(i / plane);
#define PUT(cond, col, v) do { if (cond) { cooR[e] = (int)i; cooC[e] = (int)(col); val[e] = (v); ++e; } } while (0)
            if (points == 7) PUT(iz > 0, i - plane, -1.0);
            PUT(iy > 0, i - m, -1.0);
            PUT(ix > 0, i - 1, -1.0);
            PUT(1, i, points == 7 ? 6.0 : 4.0);
            PUT(ix < m - 1, i + 1, -1.0);
            PUT(iy < m - 1, i + m, -1.0);
            if (points == 7) PUT(iz < m - 1, i + plane, -1.0);
#undef PUT
        }
        nnz = (int)e;
    }
    void* cooV = val;
    if (!dbl) {
        float* f = (float*)malloc((size_t)nnz * sizeof(float));
        for (int e = 0; e < nnz; ++e) f[e] = (float)val[e];
        cooV = f;
    }
    uint64_t seed = 3;
    void *x = malloc((size_t)cols * es), *y = malloc((size_t)rows * es);
    for (int i = 0; i < cols; ++i) { if (dbl) ((double*)x)[i] = unit(&seed); else ((float*)x)[i] = (float)unit(&seed); }
    for (int i = 0; i < rows; ++i) { if (dbl) ((double*)y)[i] = unit(&seed); else ((float*)y)[i] = (float)unit(&seed); }

    /* ---- COO -> DIA (diaPerf.cpp:160-201), COO -> HDIA (diaPerf.cpp:254-295) on the host ---- */
    const int diags = computeDiaDiagonalsCount(rows, cols, nnz, cooR, cooC);
    const int pitch = computeDiaAllocPitch(rows);
    const double diaBytes = (double)diags * pitch * es + diags * 4.0;
    const int diaFits = diaBytes < 64e9;
    void* diaV = NULL;
    int* diaO = NULL;
    if (diaFits) {
        diaV = calloc((size_t)diags * pitch, es);
        diaO = (int*)calloc((size_t)diags, sizeof(int));
        coo2dia(diaV, diaO, pitch, diags, rows, cols, nnz, cooR, cooC, cooV, 0, type);
    }
    const int hacks = getHdiaHacksCount(hackSize, rows);
    int height = 0;
    int* hackOff = (int*)calloc((size_t)hacks + 1, sizeof(int));
    computeHdiaHackOffsetsFromCoo(&height, hackOff, hackSize, rows, cols, nnz, cooR, cooC, 0);
    void* hdiaV = calloc((size_t)hackSize * height, es); /* the caller zeroes it (hdia.cpp writes real entries only) */
    int* hdiaO = (int*)calloc((size_t)height, sizeof(int));
    cooToHdia(hdiaV, hdiaO, hackOff, hackSize, rows, cols, nnz, cooR, cooC, cooV, 0, type);
    printf("%d rows, %d columns, %d nnz, %s, %s | DIA %d diagonals x %d (%.1f MB)%s | HDIA %d stored diagonals of %d rows (%.1f MB)\n",
           rows, cols, nnz, fromFile ? argv[1] : (points == 7 ? "7-point Laplacian" : "5-point Laplacian"), dbl ? "double" : "float",
           diags, pitch, diaBytes * 1e-6, diaFits ? "" : " -- skipped, too large", height, hackSize,
           ((double)hackSize * height * es + height * 4.0) * 1e-6);

    void *dX, *dY, *dZ, *dDiaV = NULL, *dHdiaV;
    int *dDiaO = NULL, *dHdiaO, *dHack;
    CHECK(hipMalloc(&dX, (size_t)cols * es)); CHECK(hipMalloc(&dY, (size_t)rows * es)); CHECK(hipMalloc(&dZ, (size_t)rows * es));
    CHECK(hipMemcpy(dX, x, (size_t)cols * es, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dY, y, (size_t)rows * es, hipMemcpyHostToDevice));
    if (diaFits) {
        CHECK(hipMalloc(&dDiaV, (size_t)diags * pitch * es)); CHECK(hipMalloc((void**)&dDiaO, (size_t)diags * sizeof(int)));
        CHECK(hipMemcpy(dDiaV, diaV, (size_t)diags * pitch * es, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dDiaO, diaO, (size_t)diags * sizeof(int), hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&dHdiaV, (size_t)hackSize * height * es)); CHECK(hipMalloc((void**)&dHdiaO, (size_t)height * sizeof(int)));
    CHECK(hipMalloc((void**)&dHack, ((size_t)hacks + 1) * sizeof(int)));
    CHECK(hipMemcpy(dHdiaV, hdiaV, (size_t)hackSize * height * es, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHdiaO, hdiaO, (size_t)height * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHack, hackOff, ((size_t)hacks + 1) * sizeof(int), hipMemcpyHostToDevice));

    spgpuHandle_t h;
    if (spgpuCreate(&h, 0) != SPGPU_SUCCESS) return 2;
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    double dots[2] = {0, 0};
    for (int format = diaFits ? 0 : 1; format < 2; ++format) {
#define RUN()                                                                                                   \
        do {                                                                                                    \
            if (format == 0 && dbl)  spgpuDdiaspmv(h, dZ, dY, 1.0, dDiaV, dDiaO, pitch, rows, cols, diags, dX, 0.0); \
            if (format == 0 && !dbl) spgpuSdiaspmv(h, dZ, dY, 1.0f, dDiaV, dDiaO, pitch, rows, cols, diags, dX, 0.0f); \
            if (format == 1 && dbl)  spgpuDhdiaspmv(h, dZ, dY, 1.0, dHdiaV, dHdiaO, hackSize, dHack, rows, cols, dX, 0.0); \
            if (format == 1 && !dbl) spgpuShdiaspmv(h, dZ, dY, 1.0f, dHdiaV, dHdiaO, hackSize, dHack, rows, cols, dX, 0.0f); \
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
        /* algorithmic bytes: the stored slots, the offsets, x once, z once */
        const double slots = format ? (double)hackSize * height : (double)diags * pitch;
        const double bytes = slots * es + (format ? height + hacks + 1.0 : diags) * 4.0 + (double)cols * es + (double)rows * es;
        printf("%s dot res: %.10e | %.4f ms | %.1f GFlop/s | %.1f GB/s (%.1f%% of 8 TB/s)\n", format ? "HDIA" : "DIA ", dots[format],
               t * 1e3, 2.0 * nnz / t * 1e-9, bytes / t * 1e-9, bytes / t * 1e-9 / 80.0);
    }
    spgpuDestroy(h);
    CHECK(hipGetLastError());
    if (!diaFits) {
        printf("HDIA only: PASSED\n");
        return 0;
    }
    /* DIA and HDIA add a row's products in the same (ascending diagonal) order: the same bits */
    const int same = dots[0] == dots[1];
    printf(same ? "DIA and HDIA checksums identical: PASSED\n" : "checksums differ: FAILED\n");
    return same ? 0 : 1;
}
