/*
 * Conjugate gradient on the 5-point Laplacian (BASELINE config 1: 1024 x 1024 grid) written against the
 * C ABI only: the pattern the SpMV path lives in inside a Krylov solver.
 * Per iteration: 1 spgpuDhellspmv, 2 spgpuDdot (host scalars), 3 spgpuDaxpby -- all on the handle's stream.
 * Then the same iterations again with the scalars kept on the device (spgpu/device_scalars.h): one iteration is
 * captured into a HIP graph (6 kernels, no host round trip; two copies that alternate the |r|^2 cell) and replayed; the iterate must come out bit for bit
 * the same as in the eager run.
 *
 *   usage: cg_amd [grid=1024] [maxIter=200] [tol=1e-8] [timing]
 *   timing: the caller only wants the per-iteration times of a fixed number of iterations (bench.py: 60 of them): the
 *   convergence criterion of the self-check is waived, the graph runs must still repeat the eager run bit for bit
 * Prints the residual history and time per iteration; exits non-zero if the residual does not fall or the two
 * runs differ.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "spgpu/core.h"
#include "spgpu/ell_conv.h"
#include "spgpu/hell.h"
#include "spgpu/hell_conv.h"
#include "spgpu/vector.h"
#include "spgpu/device_scalars.h"
#include <string.h>

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            exit(2);                                                                                \
        }                                                                                           \
    } while (0)

int main(int argc, char** argv)
{
    const int g = argc > 1 ? atoi(argv[1]) : 1024;
    const int maxIter = argc > 2 ? atoi(argv[2]) : 200;
    const double tol = argc > 3 ? atof(argv[3]) : 1e-8;
    const int timingOnly = argc > 4 && strcmp(argv[4], "timing") == 0;
    const int n = g * g, hackSize = 32;

    /* 5-point Laplacian in COO, natural order */
    int nnz = 0;
    int* cr = (int*)malloc((size_t)5 * n * sizeof(int));
    int* cc = (int*)malloc((size_t)5 * n * sizeof(int));
    double* cv = (double*)malloc((size_t)5 * n * sizeof(double));
    for (int i = 0; i < n; ++i) {
        const int gx = i % g, gy = i / g;
        if (gy > 0)     { cr[nnz] = i; cc[nnz] = i - g; cv[nnz++] = -1.0; }
        if (gx > 0)     { cr[nnz] = i; cc[nnz] = i - 1; cv[nnz++] = -1.0; }
        cr[nnz] = i; cc[nnz] = i; cv[nnz++] = 4.0;
        if (gx < g - 1) { cr[nnz] = i; cc[nnz] = i + 1; cv[nnz++] = -1.0; }
        if (gy < g - 1) { cr[nnz] = i; cc[nnz] = i + g; cv[nnz++] = -1.0; }
    }
    int maxRow = 0, height = 0;
    int* rowLen = (int*)malloc((size_t)n * sizeof(int));
    computeEllRowLenghts(rowLen, &maxRow, n, nnz, cr, 0);
    const int pitch = computeEllAllocPitch(n);
    double* ev = (double*)calloc((size_t)maxRow * pitch, sizeof(double));
    int* ei = (int*)calloc((size_t)maxRow * pitch, sizeof(int));
    cooToEll(ev, ei, pitch, pitch, maxRow, 0, n, nnz, cr, cc, cv, 0, SPGPU_TYPE_DOUBLE);
    computeHellAllocSize(&height, hackSize, n, rowLen);
    const int hacks = (n + hackSize - 1) / hackSize;
    double* hv = (double*)calloc((size_t)hackSize * height, sizeof(double));
    int* hi = (int*)calloc((size_t)hackSize * height, sizeof(int));
    int* ho = (int*)calloc(hacks, sizeof(int));
    ellToHell(hv, hi, ho, hackSize, ev, ei, pitch, pitch, rowLen, n, SPGPU_TYPE_DOUBLE);

    double *dV, *dX, *dR, *dP, *dAp;
    int *dI, *dHo, *dRs;
    CHECK(hipMalloc((void**)&dV, (size_t)hackSize * height * sizeof(double)));
    CHECK(hipMalloc((void**)&dI, (size_t)hackSize * height * sizeof(int)));
    CHECK(hipMalloc((void**)&dHo, hacks * sizeof(int)));
    CHECK(hipMalloc((void**)&dRs, (size_t)n * sizeof(int)));
    CHECK(hipMalloc((void**)&dX, (size_t)n * sizeof(double)));
    CHECK(hipMalloc((void**)&dR, (size_t)n * sizeof(double)));
    CHECK(hipMalloc((void**)&dP, (size_t)n * sizeof(double)));
    CHECK(hipMalloc((void**)&dAp, (size_t)n * sizeof(double)));
    CHECK(hipMemcpy(dV, hv, (size_t)hackSize * height * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dI, hi, (size_t)hackSize * height * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHo, ho, hacks * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRs, rowLen, (size_t)n * sizeof(int), hipMemcpyHostToDevice));

    /* b = A * ones is known only through r0 = b - A*0 = b: start from x = 0, exact solution = ones */
    double* b = (double*)calloc(n, sizeof(double));
    for (int e = 0; e < nnz; ++e)
        b[cr[e]] += cv[e];
    CHECK(hipMemcpy(dR, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dP, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemset(dX, 0, (size_t)n * sizeof(double)));

    spgpuHandle_t h;
    if (spgpuCreate(&h, 0) != SPGPU_SUCCESS) return 2;
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));

    double rr = spgpuDdot(h, n, dR, dR);
    const double rr0 = rr;
    printf("CG on the %d x %d 5-point Laplacian (%d rows, %d nnz), HELL hackSize %d\niter 0  |r| = %.6e\n", g, g, n, nnz, hackSize, sqrt(rr));
    CHECK(hipEventRecord(t0, spgpuGetStream(h)));
    int it = 0;
    while (it < maxIter && sqrt(rr / rr0) > tol) {
        spgpuDhellspmv(h, dAp, dAp, 1.0, dV, dI, hackSize, dHo, dRs, NULL, maxRow, n, dP, 0.0, 0); /* Ap = A p        */
        const double pAp = spgpuDdot(h, n, dP, dAp);
        const double alpha = rr / pAp;
        spgpuDaxpby(h, dX, n, 1.0, dX, alpha, dP);                                               /* x += alpha p    */
        spgpuDaxpby(h, dR, n, 1.0, dR, -alpha, dAp);                                             /* r -= alpha Ap   */
        const double rrNew = spgpuDdot(h, n, dR, dR);
        spgpuDaxpby(h, dP, n, rrNew / rr, dP, 1.0, dR);                                          /* p = r + beta p  */
        rr = rrNew;
        ++it;
        if (it % 25 == 0 || sqrt(rr / rr0) <= tol)
            printf("iter %d  |r| = %.6e\n", it, sqrt(rr));
    }
    CHECK(hipEventRecord(t1, spgpuGetStream(h)));
    CHECK(hipEventSynchronize(t1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t0, t1));

    double* x = (double*)malloc((size_t)n * sizeof(double));
    CHECK(hipMemcpy(x, dX, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < n; ++i)
        if (fabs(x[i] - 1.0) > err)
            err = fabs(x[i] - 1.0);
    printf("%d iterations, %.3f ms total, %.1f us per iteration, relative residual %.3e, max |x - 1| = %.3e\n", it, ms,
           it ? ms * 1e3 / it : 0.0, sqrt(rr / rr0), err);
    /* ---- the same iterations as one captured graph per iteration, scalars on the device ---- */
    enum { RR_A, RR_B, PAP, SCALARS }; /* |r|^2 alternates between two cells: no copy, no pointer swap in the graph */
    double* dS;
    CHECK(hipMalloc((void**)&dS, SCALARS * sizeof(double)));
    CHECK(hipMemcpy(dR, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dP, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemset(dX, 0, (size_t)n * sizeof(double)));
    hipStream_t stream = spgpuGetStream(h);
    spgpuDdotDevice(h, dS + RR_A, n, dR, dR);
    hipGraph_t graph[2];
    hipGraphExec_t step[2];
    for (int parity = 0; parity < 2; ++parity) {
        double* rrOld = dS + (parity ? RR_B : RR_A);
        double* rrNew = dS + (parity ? RR_A : RR_B);
        CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
        spgpuDhellspmv(h, dAp, dAp, 1.0, dV, dI, hackSize, dHo, dRs, NULL, maxRow, n, dP, 0.0, 0); /* Ap = A p               */
        spgpuDdotDevice(h, dS + PAP, n, dP, dAp);
        spgpuDaxpbyQuotDevice(h, dX, n, NULL, NULL, dX, rrOld, dS + PAP, 0, dP);                  /* x += (rr/pAp) p        */
        spgpuDaxpbyQuotDevice(h, dR, n, NULL, NULL, dR, rrOld, dS + PAP, 1, dAp);                 /* r -= (rr/pAp) Ap       */
        spgpuDdotDevice(h, rrNew, n, dR, dR);
        spgpuDaxpbyQuotDevice(h, dP, n, rrNew, rrOld, dP, NULL, NULL, 0, dR);                     /* p = r + (rr'/rr) p     */
        CHECK(hipStreamEndCapture(stream, &graph[parity]));
        CHECK(hipGraphInstantiate(&step[parity], graph[parity], NULL, NULL, 0));
    }
    CHECK(hipEventRecord(t0, stream));
    for (int i = 0; i < it; ++i)
        CHECK(hipGraphLaunch(step[i & 1], stream));
    CHECK(hipEventRecord(t1, stream));
    CHECK(hipEventSynchronize(t1));
    float msGraph = 0;
    CHECK(hipEventElapsedTime(&msGraph, t0, t1));
    double* xg = (double*)malloc((size_t)n * sizeof(double));
    double rrGraph = 0;
    CHECK(hipMemcpy(xg, dX, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&rrGraph, dS + ((it & 1) ? RR_B : RR_A), sizeof(double), hipMemcpyDeviceToHost));
    const int same = memcmp(x, xg, (size_t)n * sizeof(double)) == 0 && memcmp(&rr, &rrGraph, sizeof(double)) == 0;
    printf("graph replay: %d iterations, %.3f ms total, %.1f us per iteration (eager with host scalars: %.1f us); iterate %s\n",
           it, msGraph, it ? msGraph * 1e3 / it : 0.0, it ? ms * 1e3 / it : 0.0,
           same ? "bit-identical to the eager run" : "DIFFERS from the eager run");
    for (int parity = 0; parity < 2; ++parity) {
        CHECK(hipGraphExecDestroy(step[parity]));
        CHECK(hipGraphDestroy(graph[parity]));
    }

    /* ---- the same iterations from three fused calls: Ap = A p with p.Ap; x, r updates with |r|^2; the direction ---- */
    CHECK(hipMemcpy(dR, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dP, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemset(dX, 0, (size_t)n * sizeof(double)));
    spgpuDdotDevice(h, dS + RR_A, n, dR, dR);
    for (int parity = 0; parity < 2; ++parity) {
        double* rrOld = dS + (parity ? RR_B : RR_A);
        double* rrNew = dS + (parity ? RR_A : RR_B);
        CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
        spgpuDhellspmvDotDevice(h, dS + PAP, NULL, dAp, NULL, 1.0, dV, dI, hackSize, dHo, dRs, n, dP, 0.0, 0);
        spgpuDaxpbyPairDotDevice(h, rrNew, n, dX, dX, dP, dR, dR, dAp, rrOld, dS + PAP);
        spgpuDaxpbyQuotDevice(h, dP, n, rrNew, rrOld, dP, NULL, NULL, 0, dR);
        CHECK(hipStreamEndCapture(stream, &graph[parity]));
        CHECK(hipGraphInstantiate(&step[parity], graph[parity], NULL, NULL, 0));
    }
    CHECK(hipEventRecord(t0, stream));
    for (int i = 0; i < it; ++i)
        CHECK(hipGraphLaunch(step[i & 1], stream));
    CHECK(hipEventRecord(t1, stream));
    CHECK(hipEventSynchronize(t1));
    float msFused = 0;
    CHECK(hipEventElapsedTime(&msFused, t0, t1));
    CHECK(hipMemcpy(xg, dX, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&rrGraph, dS + ((it & 1) ? RR_B : RR_A), sizeof(double), hipMemcpyDeviceToHost));
    const int sameFused = memcmp(x, xg, (size_t)n * sizeof(double)) == 0 && memcmp(&rr, &rrGraph, sizeof(double)) == 0;
    printf("fused replay: %d iterations, %.3f ms total, %.1f us per iteration (5 kernels instead of 8); iterate %s\n", it,
           msFused, it ? msFused * 1e3 / it : 0.0, sameFused ? "bit-identical to the eager run" : "DIFFERS from the eager run");
    for (int parity = 0; parity < 2; ++parity) {
        CHECK(hipGraphExecDestroy(step[parity]));
        CHECK(hipGraphDestroy(graph[parity]));
    }

    spgpuDestroy(h);
    CHECK(hipGetLastError());
    const int converged = rr < rr0 * 1e-4 || sqrt(rr / rr0) <= tol;
    const int ok = (timingOnly ? rr < rr0 : converged) && same && sameFused;
    printf(ok ? "PASSED\n" : "FAILED (residual did not reach the tolerance, or a graph run differs)\n");
    return ok ? 0 : 1;
}
