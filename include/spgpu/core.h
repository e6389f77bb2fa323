#pragma once
/*
 * spgpu-amd: MI355X-native drop-in for the SpMV hot path of spGPU.
 *
 * This header is the handle/runtime part of the C ABI.  Every declaration
 * names the reference interface it replaces (paths relative to the spGPU
 * tree, src/core/).  CUDA types become their HIP twins: cudaStream_t ->
 * hipStream_t, cuFloatComplex/cuDoubleComplex -> hipFloatComplex /
 * hipDoubleComplex (both {x,y} pairs with identical layout).
 *
 * Plain C callers compiled with gcc need -D__HIP_PLATFORM_AMD__ and
 * -I/opt/rocm/include so that the HIP API headers resolve.
 */
#include <stddef.h>
#include <hip/hip_runtime_api.h>
#include <hip/hip_complex.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Pointer-space markers, documentation only (reference: core.h:37-40). */
#ifndef __host
#define __host
#endif
#ifndef __device
#define __device
#endif

/* Status codes of the synchronous entry points (reference: core.h:43-48). */
typedef int spgpuStatus_t;
#define SPGPU_SUCCESS     0
#define SPGPU_UNSUPPORTED 1
#define SPGPU_UNSPECIFIED 2
#define SPGPU_OUTOFMEMORY 3

/* Element type codes (reference: core.h:51-57). */
typedef int spgpuType_t;
#define SPGPU_TYPE_INT            0
#define SPGPU_TYPE_FLOAT          1
#define SPGPU_TYPE_DOUBLE         2
#define SPGPU_TYPE_COMPLEX_FLOAT  3
#define SPGPU_TYPE_COMPLEX_DOUBLE 4

/*
 * Public part of a handle; field names and order follow the reference
 * (core.h:60-82) so that code reading handle->multiProcessorCount etc. keeps
 * compiling.  On MI355X warpSize reads 64 and multiProcessorCount 256.
 * The library allocates a larger private record behind this struct
 * (reduction scratch, cached launch geometry); callers must only create
 * handles through spgpuCreate().
 */
typedef struct spgpuHandleStruct {
    hipStream_t currentStream;  /* stream every call on this handle launches on */
    hipStream_t defaultStream;  /* created by spgpuCreate, owned by the handle  */
    int device;
    int warpSize;
    int maxThreadsPerBlock;
    int maxGridSizeX;
    int maxGridSizeY;
    int maxGridSizeZ;
    int multiProcessorCount;
    int capabilityMajor;
    int capabilityMinor;
} SpgpuHandleStruct;

/* One handle == one GPU (reference: core.h:84-85). */
typedef const SpgpuHandleStruct* spgpuHandle_t;

/* reference: core.h:94 / core.c:11-41.  Re-entrant.  Unlike the reference,
 * *pHandle is NULL (and SPGPU_OUTOFMEMORY / SPGPU_UNSPECIFIED returned) when
 * the device query or an allocation fails. */
spgpuStatus_t spgpuCreate(spgpuHandle_t* pHandle, int device);

/* reference: core.h:101 / core.c:43-48.  Destroys defaultStream and the
 * private scratch; streams made by spgpuStreamCreate stay the caller's. */
void spgpuDestroy(spgpuHandle_t pHandle);

/* reference: core.h:109-116 / core.c:50-62. */
void spgpuStreamCreate(spgpuHandle_t pHandle, hipStream_t* stream);
void spgpuStreamDestroy(hipStream_t stream);

/* reference: core.h:124 / core.c:64-74.  stream == 0 restores defaultStream.
 * Calls of one handle queued on DIFFERENT streams may be in flight together, as with the reference (its kernels share
 * nothing through the handle): the scratch the ELL/HELL SpMV uses for matrices with a row order (rIdx) exists once per
 * stream -- the first time a handle is given a stream, this call allocates ~16 MiB of device memory for it (a blocking
 * allocation: call it outside stream captures; up to 8 streams per handle, later ones run a kernel that needs none).
 * Calls queued on ONE stream run in order.  The reductions (dot, nrm2, asum, amax) synchronise their stream and use
 * one scratch per handle: one reduction of a handle at a time, as the reference's one handle per host thread. */
void spgpuSetStream(spgpuHandle_t pHandle, hipStream_t stream);

/* reference: core.h:131 / core.c:76-80. */
hipStream_t spgpuGetStream(spgpuHandle_t pHandle);

/* reference: core.h:138 / core.c:82-99.  0 for an unknown code. */
size_t spgpuSizeOf(spgpuType_t typeCode);

/* reference: core.h:151-154 (cuFloatComplex_isZero ...). */
#define hipFloatComplex_isZero(a)     ((a).x == 0.0f && (a).y == 0.0f)
#define hipDoubleComplex_isZero(a)    ((a).x == 0.0 && (a).y == 0.0)
#define hipFloatComplex_isNotZero(a)  ((a).x != 0.0f || (a).y != 0.0f)
#define hipDoubleComplex_isNotZero(a) ((a).x != 0.0 || (a).y != 0.0)

#ifdef __cplusplus
}
#endif
