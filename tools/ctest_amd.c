/*
 * Plain-C caller of the C ABI, following the call sequence of the reference's
 * src/tests/ctest.c:25-149 (COO 100x100 with 200 unit entries -> ELL -> HELL,
 * alpha = 2, beta = -3, dot(z,z) printed for both formats), with the CUDA
 * runtime calls replaced by their HIP twins.  It additionally CHECKS what the
 * reference only prints: A = 2I, so z = 4x - 3y.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/ctest_amd.c \
 *       -Lspgpu_amd/lib -lspgpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/spgpu_amd/lib -o build/ctest_amd
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spgpu/core.h"
#include "spgpu/ell.h"
#include "spgpu/ell_conv.h"
#include "spgpu/hell.h"
#include "spgpu/hell_conv.h"
#include "spgpu/vector.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

int main(void)
{
    const int rows = 100, cols = 100, nnz = 200, hackSize = 32;
    float* values = (float*)malloc(nnz * sizeof(float));
    int* cooRows = (int*)malloc(nnz * sizeof(int));
    int* cooCols = (int*)malloc(nnz * sizeof(int));
    for (int i = 0; i < nnz; ++i) {
        cooRows[i] = i % rows;
        cooCols[i] = i % cols;
        values[i] = 1.0f;
    }

    int maxRow = 0;
    int* rowLen = (int*)malloc(rows * sizeof(int));
    computeEllRowLenghts(rowLen, &maxRow, rows, nnz, cooRows, 0);
    const int pitch = computeEllAllocPitch(rows);
    float* ellVal = (float*)calloc((size_t)maxRow * pitch, sizeof(float));
    int* ellIdx = (int*)calloc((size_t)maxRow * pitch, sizeof(int));
    cooToEll(ellVal, ellIdx, pitch, pitch, maxRow, 0, rows, nnz, cooRows, cooCols, values, 0, SPGPU_TYPE_FLOAT);

    int height = 0;
    computeHellAllocSize(&height, hackSize, rows, rowLen);
    const int hacks = (rows + hackSize - 1) / hackSize;
    float* hellVal = (float*)calloc((size_t)hackSize * height, sizeof(float));
    int* hellIdx = (int*)calloc((size_t)hackSize * height, sizeof(int));
    int* hackOffsets = (int*)calloc(hacks, sizeof(int));
    ellToHell(hellVal, hellIdx, hackOffsets, hackSize, ellVal, ellIdx, pitch, pitch, rowLen, rows, SPGPU_TYPE_FLOAT);
    printf("ELL maxRow %d pitch %d | HELL height %d hackOffsets %d %d %d %d\n", maxRow, pitch, height,
           hackOffsets[0], hackOffsets[1], hackOffsets[2], hackOffsets[3]);

    float *x = (float*)malloc(rows * sizeof(float)), *y = (float*)malloc(rows * sizeof(float));
    float* z = (float*)malloc(rows * sizeof(float));
    srand(1);
    for (int i = 0; i < rows; ++i) {
        x[i] = rand() / (float)RAND_MAX;
        y[i] = rand() / (float)RAND_MAX;
    }

    float *dX, *dY, *dZ, *dCm, *dHellCm;
    int *dRp, *dRs, *dHellRp, *dHack;
    CHECK(hipMalloc((void**)&dX, rows * sizeof(float)));
    CHECK(hipMalloc((void**)&dY, rows * sizeof(float)));
    CHECK(hipMalloc((void**)&dZ, rows * sizeof(float)));
    CHECK(hipMalloc((void**)&dRs, rows * sizeof(int)));
    CHECK(hipMalloc((void**)&dCm, (size_t)maxRow * pitch * sizeof(float)));
    CHECK(hipMalloc((void**)&dRp, (size_t)maxRow * pitch * sizeof(int)));
    CHECK(hipMalloc((void**)&dHellCm, (size_t)hackSize * height * sizeof(float)));
    CHECK(hipMalloc((void**)&dHellRp, (size_t)hackSize * height * sizeof(int)));
    CHECK(hipMalloc((void**)&dHack, hacks * sizeof(int)));
    CHECK(hipMemcpy(dX, x, rows * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dY, y, rows * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRs, rowLen, rows * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dCm, ellVal, (size_t)maxRow * pitch * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRp, ellIdx, (size_t)maxRow * pitch * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHellCm, hellVal, (size_t)hackSize * height * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHellRp, hellIdx, (size_t)hackSize * height * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHack, hackOffsets, hacks * sizeof(int), hipMemcpyHostToDevice));

    spgpuHandle_t handle;
    if (spgpuCreate(&handle, 0) != SPGPU_SUCCESS) {
        fprintf(stderr, "spgpuCreate failed\n");
        return 2;
    }
    printf("device %d: warpSize %d, %d CUs\n", handle->device, handle->warpSize, handle->multiProcessorCount);

    int bad = 0;
    for (int format = 0; format < 2; ++format) {
        if (format == 0)
            spgpuSellspmv(handle, dZ, dY, 2.0f, dCm, dRp, pitch, pitch, dRs, NULL, maxRow, maxRow, rows, dX, -3.0f, 0);
        else
            spgpuShellspmv(handle, dZ, dY, 2.0f, dHellCm, dHellRp, hackSize, dHack, dRs, NULL, maxRow, rows, dX, -3.0f, 0);
        const float dotRes = spgpuSdot(handle, rows, dZ, dZ);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(z, dZ, rows * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int i = 0; i < rows; ++i) {
            const double want = 4.0 * x[i] - 3.0 * y[i];
            if (fabs(z[i] - want) > worst)
                worst = fabs(z[i] - want);
        }
        printf("%s dot res: %e   max |z - (4x-3y)| = %.3g\n", format ? "HELL" : "ELL ", dotRes, worst);
        if (worst > 1e-5)
            bad = 1;
    }
    spgpuDestroy(handle);
    CHECK(hipGetLastError());
    printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
