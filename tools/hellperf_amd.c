/*
 * Plain-C performance harness with the flow of the reference's src/tests/hellPerf.cpp:127-317
 * (COO -> computeEllRowLenghts/cooToEll -> ELL run; computeHellAllocSize/ellToHell -> HELL run;
 * per format: 1 warm-up, dot(z,z) printed as checksum, N timed launches, GFlop/s), with a synthetic
 * matrix instead of a Matrix Market file (none ships with the reference) and HIP in place of the
 * CUDA runtime.  alpha = 1, beta = 0 as hellPerf.cpp:27-28.  Adds what the reference's harness
 * lacks: the two dot(z,z) are compared, and achieved HBM GB/s is printed next to GFlop/s.
 *
 *   usage: hellperf_amd [rows=1000000] [nnzPerRow=32] [banded|random] [reps=200] [s|d]
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
    const int rows = argc > 1 ? atoi(argv[1]) : 1000000;
    const int perRow = argc > 2 ? atoi(argv[2]) : 32;
    const int randomCols = argc > 3 && strcmp(argv[3], "random") == 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 200;
    const int dbl = !(argc > 5 && argv[5][0] == 's');
    const size_t es = dbl ? sizeof(double) : sizeof(float);
    const spgpuType_t type = dbl ? SPGPU_TYPE_DOUBLE : SPGPU_TYPE_FLOAT;
    const int hackSize = 32; /* hellPerf.cpp:254 */
    const long long nnz = (long long)rows * perRow;
    if (nnz > 2000000000LL) { fprintf(stderr, "too many nonzeros for int indices\n"); return 2; }

    /* ---- synthetic COO ---- */
    int* cooR = (int*)malloc(nnz * sizeof(int));
    int* cooC = (int*)malloc(nnz * sizeof(int));
    void* cooV = malloc(nnz * es);
    uint64_t seed = 1;
    for (long long e = 0; e < nnz; ++e) {
        const int r = (int)(e / perRow), k = (int)(e % perRow);
        cooR[e] = r;
        cooC[e] = randomCols ? (int)(splitmix(&seed) % (uint64_t)rows)
                             : (int)(((long long)r + k - perRow / 2 + rows) % rows);
        if (dbl) ((double*)cooV)[e] = unit(&seed); else ((float*)cooV)[e] = (float)unit(&seed);
    }
    void *x = malloc(rows * es), *y = malloc(rows * es);
    for (int i = 0; i < rows; ++i) {
        if (dbl) { ((double*)x)[i] = unit(&seed); ((double*)y)[i] = unit(&seed); }
        else     { ((float*)x)[i] = (float)unit(&seed); ((float*)y)[i] = (float)unit(&seed); }
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
    printf("%d rows, %lld nnz, %s columns, %s | ELL %d x %d (%.1f MB) | HELL height %d (%.1f MB)\n", rows, nnz,
           randomCols ? "random" : "banded", dbl ? "double" : "float", maxRow, pitch,
           (double)maxRow * pitch * (es + 4) * 1e-6, height, (double)hackSize * height * (es + 4) * 1e-6);

    /* ---- upload (hellPerf.cpp:176-190, 274-280) ---- */
    void *dX, *dY, *dZ, *dEllV, *dHellV;
    int *dRs, *dEllI, *dHellI, *dHack;
    CHECK(hipMalloc(&dX, rows * es)); CHECK(hipMalloc(&dY, rows * es)); CHECK(hipMalloc(&dZ, rows * es));
    CHECK(hipMalloc((void**)&dRs, rows * sizeof(int)));
    CHECK(hipMalloc(&dEllV, (size_t)maxRow * pitch * es)); CHECK(hipMalloc((void**)&dEllI, (size_t)maxRow * pitch * sizeof(int)));
    CHECK(hipMalloc(&dHellV, (size_t)hackSize * height * es)); CHECK(hipMalloc((void**)&dHellI, (size_t)hackSize * height * sizeof(int)));
    CHECK(hipMalloc((void**)&dHack, hacks * sizeof(int)));
    CHECK(hipMemcpy(dX, x, rows * es, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dY, y, rows * es, hipMemcpyHostToDevice));
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
    const double bytes = (double)nnz * (es + 4) + (double)rows * (4 + es) + (double)rows * es;
    double dots[2];

    for (int format = 0; format < 2; ++format) {
#define RUN()                                                                                                   \
        do {                                                                                                    \
            if (format == 0 && dbl)  spgpuDellspmv(h, dZ, dY, 1.0, dEllV, dEllI, pitch, pitch, dRs, NULL, perRow, maxRow, rows, dX, 0.0, 0); \
            if (format == 0 && !dbl) spgpuSellspmv(h, dZ, dY, 1.0f, dEllV, dEllI, pitch, pitch, dRs, NULL, perRow, maxRow, rows, dX, 0.0f, 0); \
            if (format == 1 && dbl)  spgpuDhellspmv(h, dZ, dY, 1.0, dHellV, dHellI, hackSize, dHack, dRs, NULL, maxRow, rows, dX, 0.0, 0); \
            if (format == 1 && !dbl) spgpuShellspmv(h, dZ, dY, 1.0f, dHellV, dHellI, hackSize, dHack, dRs, NULL, maxRow, rows, dX, 0.0f, 0); \
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
        printf("%s dot res: %.10e | %.4f ms | %.1f GFlop/s | %.1f GB/s (%.1f%% of 8 TB/s)\n", format ? "HELL" : "ELL ",
               dots[format], t * 1e3, 2.0 * nnz / t * 1e-9, bytes / t * 1e-9, bytes / t * 1e-9 / 80.0);
    }
    spgpuDestroy(h);
    CHECK(hipGetLastError());
    const int same = dots[0] == dots[1];
    printf(same ? "ELL and HELL checksums identical: PASSED\n" : "checksums differ: FAILED\n");
    return same ? 0 : 1;
}
