/*
 * Handle / runtime layer of the spgpu-amd C ABI (include/spgpu/core.h).
 * Behavioural model: reference src/core/core.c:11-99.  Written for HIP on
 * MI355X; the handle additionally owns the scratch that the reductions use
 * (the reference keeps that in a process-global __device__ array,
 * kernels/ddot.cu:35).
 */
#include "spgpu_internal.h"
#include "spgpu/tuning.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- deep lists of the ELL/HELL SpMV (spgpu_internal.h): one per stream ---- */
#define DEEP_HEAD_BYTES (SPGPU_DEEP_HEAD_INTS * sizeof(int))
#define DEEP_ENTRY_BYTES ((size_t)SPGPU_DEEP_ENTRIES * sizeof(SpgpuDeepEntry))
#define DEEP_ITEM_ENTRY_BYTES ((size_t)SPGPU_DEEP_ITEMS * sizeof(SpgpuDeepItem))
#define DEEP_PARTIAL_BYTES ((size_t)SPGPU_DEEP_ENTRIES * 32 * 16)
#define DEEP_ITEM_SUM_BYTES ((size_t)SPGPU_DEEP_ITEMS * 32 * 16)

/* Gives `stream` a list if it has none and the table has room (device already current).  Allocates and clears with a
 * blocking call: this runs in spgpuCreate / spgpuSetStream, outside any launch path and outside any stream capture. */
static void deepListFor(SpgpuPrivateHandle* h, hipStream_t stream)
{
    pthread_mutex_lock(&h->formLock);
    int known = 0;
    for (int i = 0; i < h->deepStreams; ++i)
        known |= h->deepStream[i] == stream;
    if (!known && h->deepStreams < SPGPU_DEEP_STREAMS) {
        void* p = NULL;
        hipEvent_t idle = NULL;
        if (hipMalloc(&p, DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES + DEEP_ITEM_ENTRY_BYTES + DEEP_PARTIAL_BYTES + DEEP_ITEM_SUM_BYTES) == hipSuccess) {
            if (spgpuTuning()->poisonScratch)
                (void)hipMemset((char*)p + DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES + DEEP_ITEM_ENTRY_BYTES, 0xFF, DEEP_PARTIAL_BYTES + DEEP_ITEM_SUM_BYTES);
            if (hipMemset(p, 0, DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES + DEEP_ITEM_ENTRY_BYTES) == hipSuccess &&
                hipEventCreateWithFlags(&idle, hipEventDisableTiming) == hipSuccess) {
                h->deepScratch[h->deepStreams] = p;
                h->deepStream[h->deepStreams] = stream;
                h->deepIdle[h->deepStreams] = idle;
                h->deepUsed[h->deepStreams] = 0;
                h->deepPinned[h->deepStreams] = 0;
                h->deepClock[h->deepStreams] = ++h->deepTick;
                h->deepStreams += 1;
            } else {
                hipFree(p);
            }
        }
    } else if (!known) {
        /* Every list has an owner.  A program that creates and destroys streams as it goes would fill the table with
         * the lists of streams that no longer exist, and every ordered SpMV on a later stream would run the kernel that
         * needs no list for good (slower, another order of additions).  So the list that was used longest ago AND whose
         * last user has finished -- the event recorded behind its deep kernels has completed, or it was never used --
         * changes hands: its header is zero (the last finishing workgroup of a call leaves it so), nothing else of it
         * carries over from call to call. */
        int pick = -1;
        for (int i = 0; i < h->deepStreams; ++i) {
            if (h->deepStream[i] == h->pub.defaultStream || h->deepPinned[i])
                continue; /* the default stream always comes back (spgpuSetStream(h, 0)); a captured graph may replay at any time */
            if (h->deepUsed[i] && !spgpuEventDone(h->deepIdle[i]))
                continue;
            if (pick < 0 || h->deepClock[i] < h->deepClock[pick])
                pick = i;
        }
        if (pick >= 0) {
            h->deepStream[pick] = stream;
            h->deepUsed[pick] = 0;
            h->deepClock[pick] = ++h->deepTick;
            h->deepRecycled += 1;
        }
    }
    pthread_mutex_unlock(&h->formLock);
}

spgpuStatus_t spgpuCreate(spgpuHandle_t* pHandle, int device)
{
    if (!pHandle)
        return SPGPU_UNSPECIFIED;
    *pHandle = NULL;

    hipDeviceProp_t prop;
    hipError_t perr = hipGetDeviceProperties(&prop, device);
    if (perr != hipSuccess) {
        fprintf(stderr, "spgpuCreate: hipGetDeviceProperties(%d) failed: %s\n", device, hipGetErrorString(perr));
        return SPGPU_UNSPECIFIED;
    }

    SpgpuPrivateHandle* h = (SpgpuPrivateHandle*)calloc(1, sizeof(SpgpuPrivateHandle));
    if (!h)
        return SPGPU_OUTOFMEMORY;

    int previous = 0;
    hipGetDevice(&previous);
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess)
        err = hipStreamCreate(&h->pub.defaultStream);
    if (err == hipSuccess)
        err = hipMalloc(&h->reduceScratch, SPGPU_REDUCE_SCRATCH_BYTES);
    if (err == hipSuccess && spgpuTuning()->poisonScratch)
        err = hipMemset(h->reduceScratch, 0xFF, SPGPU_REDUCE_SCRATCH_BYTES);
    if (err == hipSuccess)
        err = hipHostMalloc(&h->reduceHost, SPGPU_REDUCE_SCRATCH_BYTES, hipHostMallocDefault);
    if (err == hipSuccess)
        err = hipHostMalloc((void**)&h->formFeedback, (SPGPU_FEEDBACK_ENTRIES + 2) * SPGPU_FEEDBACK_SAMPLES * sizeof(int),
                            hipHostMallocDefault);
    if (err == hipSuccess)
        memset(h->formFeedback, 0, (SPGPU_FEEDBACK_ENTRIES + 2) * SPGPU_FEEDBACK_SAMPLES * sizeof(int));
    hipSetDevice(previous);

    if (err != hipSuccess) {
        fprintf(stderr, "spgpuCreate: device %d setup failed: %s\n", device, hipGetErrorString(err));
        if (h->reduceScratch) hipFree(h->reduceScratch);
        if (h->reduceHost) hipHostFree(h->reduceHost);
        if (h->pub.defaultStream) hipStreamDestroy(h->pub.defaultStream);
        free(h);
        return err == hipErrorOutOfMemory ? SPGPU_OUTOFMEMORY : SPGPU_UNSPECIFIED;
    }

    h->pub.currentStream = h->pub.defaultStream;
    h->pub.device = device;
    h->pub.warpSize = prop.warpSize;
    h->pub.maxThreadsPerBlock = prop.maxThreadsPerBlock;
    h->pub.maxGridSizeX = prop.maxGridSize[0];
    h->pub.maxGridSizeY = prop.maxGridSize[1];
    h->pub.maxGridSizeZ = prop.maxGridSize[2];
    h->pub.multiProcessorCount = prop.multiProcessorCount;
    h->pub.capabilityMajor = prop.major;
    h->pub.capabilityMinor = prop.minor;
    h->magic = SPGPU_HANDLE_MAGIC;
    pthread_mutex_init(&h->formLock, NULL);
    h->spmvForm = SPGPU_SPMV_FORM_AUTO;
    hipSetDevice(device);
    deepListFor(h, h->pub.defaultStream); /* failing that, ordered matrices run the kernel that needs no list */
    h->adopted = (SpgpuAdopted*)calloc(SPGPU_ADOPTED, sizeof(SpgpuAdopted)); /* failing this, nothing can be adopted */
    /* the plan table (spgpu_internal.h): failing this, ordered matrices run without plans */
    h->plans = (SpgpuSpmvPlan*)calloc(SPGPU_PLANS, sizeof(SpgpuSpmvPlan));
    if (h->plans && hipHostMalloc((void**)&h->planPinned, SPGPU_PLANS * SPGPU_PLAN_WORDS * sizeof(int), hipHostMallocDefault) == hipSuccess) {
        memset(h->planPinned, 0, SPGPU_PLANS * SPGPU_PLAN_WORDS * sizeof(int));
        for (int i = 0; i < SPGPU_PLANS; ++i) {
            h->plans[i].pinned = h->planPinned + i * SPGPU_PLAN_WORDS;
            if (hipEventCreateWithFlags(&h->plans[i].built, hipEventDisableTiming) != hipSuccess)
                h->plans[i].state = SPGPU_PLAN_GIVEN_UP;
        }
    } else {
        free(h->plans);
        h->plans = NULL;
        h->planPinned = NULL;
    }
    hipSetDevice(previous);

    *pHandle = &h->pub;
    return SPGPU_SUCCESS;
}

static void freeAdopted(const SpgpuAdopted* e);

void spgpuDestroy(spgpuHandle_t pHandle)
{
    if (!pHandle)
        return;
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    int previous = 0;
    hipGetDevice(&previous);
    hipSetDevice(h->pub.device);
    /* Kernels still queued on ANY stream of this device may write the pinned feedback words and read the reduction
     * scratch (a caller's stream set with spgpuSetStream, a replayed graph): wait for the whole device, not only
     * for defaultStream, before freeing them.  A graph captured from this handle must not be replayed after this. */
    hipDeviceSynchronize();
    hipFree(h->reduceScratch);
    for (int i = 0; i < h->deepStreams; ++i) {
        hipFree(h->deepScratch[i]);
        hipEventDestroy(h->deepIdle[i]);
    }
    if (h->adopted) {
        for (int i = 0; i < SPGPU_ADOPTED; ++i)
            if (h->adopted[i].rows > 0)
                freeAdopted(&h->adopted[i]);
        free(h->adopted);
    }
    if (h->plans) {
        for (int i = 0; i < SPGPU_PLANS; ++i) {
            if (h->plans[i].device)
                hipFree(h->plans[i].device);
            if (h->plans[i].packed)
                hipFree(h->plans[i].packed);
            if (h->plans[i].built)
                hipEventDestroy(h->plans[i].built);
        }
        free(h->plans);
    }
    for (int i = 0; i < h->planGraves; ++i)
        hipFree(h->planGraveyard[i]);
    if (h->planPinned)
        hipHostFree(h->planPinned);
    hipHostFree(h->reduceHost);
    hipHostFree(h->formFeedback);
    hipStreamDestroy(h->pub.defaultStream);
    hipSetDevice(previous);
    pthread_mutex_destroy(&h->formLock);
    h->magic = 0;
    free(h);
}

void spgpuStreamCreate(spgpuHandle_t pHandle, hipStream_t* stream)
{
    int previous = 0;
    hipGetDevice(&previous);
    hipSetDevice(pHandle->device);
    hipStreamCreate(stream);
    hipSetDevice(previous);
}

void spgpuStreamDestroy(hipStream_t stream)
{
    hipStreamDestroy(stream);
}

void spgpuSetStream(spgpuHandle_t pHandle, hipStream_t stream)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    h->pub.currentStream = stream ? stream : h->pub.defaultStream;
    /* a stream the handle has not seen before gets a deep list of its own (see spgpu_internal.h): the reference's SpMV has
     * no state shared between streams (hell_spmv_base_template.cuh:336-345, core.c:64-74), so neither may this one */
    int known = 0;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < h->deepStreams; ++i)
        known |= h->deepStream[i] == h->pub.currentStream;
    pthread_mutex_unlock(&h->formLock);
    if (!known) {
        int previous = 0;
        hipGetDevice(&previous);
        hipSetDevice(h->pub.device);
        deepListFor(h, h->pub.currentStream);
        hipSetDevice(previous);
    }
}

hipStream_t spgpuGetStream(spgpuHandle_t pHandle)
{
    return pHandle->currentStream;
}

size_t spgpuSizeOf(spgpuType_t typeCode)
{
    switch (typeCode) {
    case SPGPU_TYPE_INT:            return sizeof(int);
    case SPGPU_TYPE_FLOAT:          return sizeof(float);
    case SPGPU_TYPE_DOUBLE:         return sizeof(double);
    case SPGPU_TYPE_COMPLEX_FLOAT:  return sizeof(hipFloatComplex);
    case SPGPU_TYPE_COMPLEX_DOUBLE: return sizeof(hipDoubleComplex);
    default:                        return 0;
    }
}

int* spgpuFormFeedback(spgpuHandle_t pHandle, const void* key, int rows, int* calls, int* tag)
{
    /* Two host threads may share a handle (the reference documents one handle per thread, core.h:88-90, but does not
     * enforce it): the table is searched and re-assigned under a lock.  The words themselves are written by the GPU
     * and read without synchronisation by design -- any value selects a correct kernel. */
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    int* slot = NULL;
    for (unsigned e = 0; e < SPGPU_FEEDBACK_ENTRIES && !slot; ++e)
        if (h->formKey[e] == key && h->formRows[e] == rows) {
            slot = h->formFeedback + e * SPGPU_FEEDBACK_SAMPLES;
            *calls = ++h->formCalls[e];
            *tag = h->formGeneration[e] << 8;
        }
    if (!slot) {
        const unsigned e = h->formNext++ % SPGPU_FEEDBACK_ENTRIES; /* oldest entry makes room */
        h->formKey[e] = key;
        h->formRows[e] = rows;
        h->formCalls[e] = 0;
        h->formGeneration[e] = (h->formGeneration[e] + 1) & 0x7FFFFF; /* reports still in flight for the previous owner carry the old one */
        *tag = h->formGeneration[e] << 8;
        *calls = 0;
        slot = h->formFeedback + e * SPGPU_FEEDBACK_SAMPLES;
        for (int i = 0; i < SPGPU_FEEDBACK_SAMPLES; ++i)
            slot[i] = 0;
    }
    pthread_mutex_unlock(&h->formLock);
    return slot;
}

spgpuStatus_t spgpuDeepScratch(spgpuHandle_t pHandle, SpgpuDeepList* list)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    char* base = NULL;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < h->deepStreams; ++i)
        if (h->deepStream[i] == h->pub.currentStream) {
            base = (char*)h->deepScratch[i];
            list->idle = h->deepIdle[i];
            h->deepUsed[i] = 1;
            h->deepClock[i] = ++h->deepTick;
        }
    if (!base)
        h->deepFallbacks += 1;
    pthread_mutex_unlock(&h->formLock);
    if (!base)
        return SPGPU_UNSUPPORTED;
    list->header = (int*)base;
    list->entries = (SpgpuDeepEntry*)(base + DEEP_HEAD_BYTES);
    list->items = (SpgpuDeepItem*)(base + DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES);
    list->partials = base + DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES + DEEP_ITEM_ENTRY_BYTES;
    list->itemSums = base + DEEP_HEAD_BYTES + DEEP_ENTRY_BYTES + DEEP_ITEM_ENTRY_BYTES + DEEP_PARTIAL_BYTES;
    return SPGPU_SUCCESS;
}

void spgpuDeepListPin(spgpuHandle_t pHandle)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < h->deepStreams; ++i)
        if (h->deepStream[i] == h->pub.currentStream)
            h->deepPinned[i] = 1;
    pthread_mutex_unlock(&h->formLock);
}

int* spgpuAnalyseWords(spgpuHandle_t pHandle)
{
    return spgpuPrivate(pHandle)->formFeedback + SPGPU_FEEDBACK_ENTRIES * SPGPU_FEEDBACK_SAMPLES;
}

/* Pinned words the deep kernels report into (ellpack_spmv.hip deepFinishKernel): [0] calls whose deep list overflowed,
 * [1] / [2] the entries / items the last such call asked for. */
int* spgpuDeepOverflowWords(spgpuHandle_t pHandle)
{
    return spgpuPrivate(pHandle)->formFeedback + (SPGPU_FEEDBACK_ENTRIES + 1) * SPGPU_FEEDBACK_SAMPLES;
}

int spgpuDeepListOverflows(spgpuHandle_t pHandle)
{
    return ((volatile int*)spgpuDeepOverflowWords(pHandle))[0];
}

int spgpuDeepListFallbacks(spgpuHandle_t pHandle)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    const int n = h->deepFallbacks;
    pthread_mutex_unlock(&h->formLock);
    return n;
}

int spgpuDeepListsRecycled(spgpuHandle_t pHandle)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    const int n = h->deepRecycled;
    pthread_mutex_unlock(&h->formLock);
    return n;
}

/* ---- per-matrix plans of the ordered ELL/HELL SpMV (spgpu_internal.h, csrc/planned_spmv.hip) ---- */
void spgpuPlanLock(spgpuHandle_t pHandle)
{
    pthread_mutex_lock(&spgpuPrivate(pHandle)->formLock);
}

void spgpuPlanUnlock(spgpuHandle_t pHandle)
{
    pthread_mutex_unlock(&spgpuPrivate(pHandle)->formLock);
}

void spgpuPlanRetire(spgpuHandle_t pHandle, SpgpuSpmvPlan* plan)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    void** const buffers[2] = {&plan->device, &plan->packed};
    for (int b = 0; b < 2; ++b) {
        if (!*buffers[b])
            continue;
        if (h->planGraves == SPGPU_PLAN_GRAVES) {
            /* kernels in flight on any stream may still read a retired plan: wait for the device before the buffers go
             * (once per SPGPU_PLAN_GRAVES retirements; never while a stream of the process captures -- see launchPlanned) */
            hipDeviceSynchronize();
            for (int i = 0; i < h->planGraves; ++i)
                hipFree(h->planGraveyard[i]);
            h->planGraves = 0;
        }
        h->planGraveyard[h->planGraves++] = *buffers[b];
        *buffers[b] = NULL;
    }
    plan->state = SPGPU_PLAN_EMPTY;
    plan->uses = 0;
    plan->deep = 0;
    plan->pinned[0] = 0;
    plan->pinned[1] = 0;
}

static int samePlanKey(const SpgpuSpmvPlan* a, const SpgpuSpmvPlan* b)
{
    return a->rP == b->rP && a->rS == b->rS && a->rIdx == b->rIdx && a->hackOffsets == b->hackOffsets &&
           a->idxStride == b->idxStride && a->rows == b->rows && a->hackSize == b->hackSize && a->baseIndex == b->baseIndex &&
           a->maxNnz == b->maxNnz && a->deepCap == b->deepCap && a->subs == b->subs;
}

SpgpuSpmvPlan* spgpuPlanRecord(spgpuHandle_t pHandle, const SpgpuSpmvPlan* key)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    if (!h->plans)
        return NULL;
    SpgpuSpmvPlan* oldest = NULL;
    for (int i = 0; i < SPGPU_PLANS; ++i) {
        SpgpuSpmvPlan* p = &h->plans[i];
        if (p->rows > 0 && samePlanKey(p, key)) {
            p->clock = ++h->planClock;
            return p;
        }
        /* a record whose analysis is still in flight keeps its buffer and its pinned words until it has landed */
        if (p->state == SPGPU_PLAN_BUILDING && !spgpuEventDone(p->built))
            continue;
        if (!oldest || p->rows == 0 || (oldest->rows != 0 && p->clock < oldest->clock))
            oldest = p;
    }
    if (!oldest)
        return NULL;
    const int givenUp = oldest->built == NULL; /* (its event could not be created: spgpuCreate) */
    spgpuPlanRetire(pHandle, oldest);
    int* pinned = oldest->pinned;
    hipEvent_t built = oldest->built;
    *oldest = *key;
    oldest->pinned = pinned;
    oldest->built = built;
    oldest->device = NULL;
    oldest->packed = NULL;
    oldest->state = givenUp ? SPGPU_PLAN_GIVEN_UP : SPGPU_PLAN_EMPTY;
    oldest->stales = 0;
    oldest->uses = 0;
    oldest->deep = 0;
    oldest->blocks = 0;
    oldest->clock = ++h->planClock;
    return oldest;
}

/* Lock held.  The record with this key, or NULL: looks, never makes room. */
SpgpuSpmvPlan* spgpuPlanFind(spgpuHandle_t pHandle, const SpgpuSpmvPlan* key)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    if (!h->plans)
        return NULL;
    for (int i = 0; i < SPGPU_PLANS; ++i) {
        SpgpuSpmvPlan* p = &h->plans[i];
        if (p->rows > 0 && samePlanKey(p, key)) {
            p->clock = ++h->planClock;
            return p;
        }
    }
    return NULL;
}

void spgpuSpmvPlanCounts(spgpuHandle_t pHandle, int* uses, int* builds, int* stales)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    if (uses) *uses = h->planUses;
    if (builds) *builds = h->planBuilds;
    if (stales) *stales = h->planStales;
    pthread_mutex_unlock(&h->formLock);
}

/* ---- adopted matrices (spgpu_internal.h, csrc/adopted_hell.hip) ---- */
const SpgpuAdopted* spgpuAdoptedFind(spgpuHandle_t pHandle, hipStream_t stream, const void* cM, const int* rP, const int* rS,
                                     const int* hackOffsets, int rows, int hackSize, int baseIndex, long long valPitch, long long idxPitch)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    if (!h->adopted || __atomic_load_n(&h->adoptedCount, __ATOMIC_RELAXED) <= 0)
        return NULL;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return NULL;
    }
    const SpgpuAdopted* found = NULL;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < SPGPU_ADOPTED; ++i) {
        const SpgpuAdopted* e = &h->adopted[i];
        if (e->rows > 0 && e->rP == (const void*)rP && e->cM == cM && e->rS == (const void*)rS && e->hackOffsets == (const void*)hackOffsets &&
            e->rows == rows && e->hackSize == hackSize && e->baseIndex == baseIndex && e->valPitch == valPitch && e->idxPitch == idxPitch) {
            found = e;
            h->adoptedUses += 1;
            break;
        }
    }
    pthread_mutex_unlock(&h->formLock);
    return found; /* (an entry's arrays live until spgpuSpmvThaw, which the caller may not run beside an SpMV on the same matrix) */
}

int spgpuAdoptedAdd(spgpuHandle_t pHandle, const SpgpuAdopted* entry)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    int said = SPGPU_UNSUPPORTED;
    if (!h->adopted)
        return said;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < SPGPU_ADOPTED; ++i) {
        if (h->adopted[i].rows == 0) {
            h->adopted[i] = *entry;
            h->adoptedCount += 1;
            said = SPGPU_SUCCESS;
            break;
        }
    }
    pthread_mutex_unlock(&h->formLock);
    return said;
}

int spgpuAdoptedRemove(spgpuHandle_t pHandle, const int* rP, SpgpuAdopted* out)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    int n = 0;
    if (!h->adopted)
        return 0;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < SPGPU_ADOPTED; ++i) {
        if (h->adopted[i].rows > 0 && (rP == NULL || h->adopted[i].rP == (const void*)rP)) {
            out[n++] = h->adopted[i];
            memset(&h->adopted[i], 0, sizeof(SpgpuAdopted));
            h->adoptedCount -= 1;
        }
    }
    pthread_mutex_unlock(&h->formLock);
    return n;
}

int spgpuSpmvAdoptedUses(spgpuHandle_t pHandle)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    pthread_mutex_lock(&h->formLock);
    const int n = h->adoptedUses;
    pthread_mutex_unlock(&h->formLock);
    return n;
}

/* spgpu?SpmvFreeze's counterpart (include/spgpu/tuning.h): the plans of the matrix with this index array lose their 16-bit copies
 * -- retired whole, the next SpMV analyses the matrix again. */
static void freeAdopted(const SpgpuAdopted* e)
{
    hipFree(e->values);
    hipFree(e->indices);
    hipFree(e->hackOffsetsOrdered);
    hipFree(e->lengths);
    hipFree(e->order);
}

int spgpuSpmvThaw(spgpuHandle_t pHandle, const int* rP)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    int thawed = 0;
    if (!h || !h->plans || !rP)
        return SPGPU_UNSPECIFIED;
    {
        /* an adopted matrix: the plan of its ordered copy goes first (keyed by the copy's arrays), then the copy -- behind a
         * device-wide wait: SpMVs in flight read it */
        SpgpuAdopted gone[SPGPU_ADOPTED];
        const int n = spgpuAdoptedRemove(pHandle, rP, gone);
        if (n > 0)
            hipDeviceSynchronize();
        for (int i = 0; i < n; ++i) {
            (void)spgpuSpmvThaw(pHandle, gone[i].indices);
            freeAdopted(&gone[i]);
            thawed += 1;
        }
    }
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; i < SPGPU_PLANS; ++i) {
        SpgpuSpmvPlan* p = &h->plans[i];
        if (p->rows > 0 && p->rP == (const void*)rP && p->packed) {
            spgpuPlanRetire(pHandle, p);
            thawed += 1;
        }
    }
    h->planFrozenSlabs = 0;
    for (int i = 0; i < SPGPU_PLANS; ++i)
        h->planFrozenSlabs += (h->plans[i].rows > 0 && h->plans[i].subs < 0 && h->plans[i].packed) ? 1 : 0;
    pthread_mutex_unlock(&h->formLock);
    return thawed ? SPGPU_SUCCESS : SPGPU_UNSUPPORTED;
}

long long spgpuSpmvFrozenBytes(spgpuHandle_t pHandle)
{
    SpgpuPrivateHandle* h = spgpuPrivate(pHandle);
    long long bytes = 0;
    if (!h || !h->plans)
        return 0;
    pthread_mutex_lock(&h->formLock);
    for (int i = 0; h->adopted && i < SPGPU_ADOPTED; ++i)
        bytes += h->adopted[i].rows > 0 ? h->adopted[i].bytes : 0;
    for (int i = 0; i < SPGPU_PLANS; ++i)
        bytes += h->plans[i].packed ? h->plans[i].packedBytes : 0;
    pthread_mutex_unlock(&h->formLock);
    return bytes;
}

/* ---- per-handle kernel-form hint (include/spgpu/tuning.h) ---- */
void spgpuSetSpmvForm(spgpuHandle_t pHandle, int form)
{
    if (form < SPGPU_SPMV_FORM_AUTO || form > SPGPU_SPMV_FORM_SWEEP)
        form = SPGPU_SPMV_FORM_AUTO;
    __atomic_store_n(&spgpuPrivate(pHandle)->spmvForm, form, __ATOMIC_RELAXED);
}

int spgpuGetSpmvForm(spgpuHandle_t pHandle)
{
    return __atomic_load_n(&spgpuPrivate(pHandle)->spmvForm, __ATOMIC_RELAXED);
}

int spgpuGetLastSpmvForm(spgpuHandle_t pHandle)
{
    return __atomic_load_n(&spgpuPrivate(pHandle)->lastSpmvForm, __ATOMIC_RELAXED);
}

void spgpuNoteSpmvForm(spgpuHandle_t pHandle, int form)
{
    __atomic_store_n(&spgpuPrivate(pHandle)->lastSpmvForm, form, __ATOMIC_RELAXED);
}

/* ---- tuning knobs (include/spgpu/tuning.h) ---- */
static SpgpuTuning tuning;
static int tuningLoaded;

static int envInt(const char* name, int fallback)
{
    const char* s = getenv(name);
    return s && *s ? atoi(s) : fallback;
}

void spgpuTuningReload(void)
{
    SpgpuTuning t;
    t.spmvVariant = envInt("SPGPU_SPMV_VARIANT", 0);
    t.ntLoads = envInt("SPGPU_NT_LOADS", 1);
    t.tailLanes = envInt("SPGPU_TAIL_LANES", -1);
    t.hdiaVariant = envInt("SPGPU_HDIA_VARIANT", 0);
    t.hdiaBlock = envInt("SPGPU_HDIA_BLOCK", 512);
    t.hdiaNarrow = envInt("SPGPU_HDIA_NARROW", 0);
    t.xcdOrder = envInt("SPGPU_XCD_ORDER", 0);
    t.spmmVariant = envInt("SPGPU_SPMM_VARIANT", 0);
    t.l1Blocks = envInt("SPGPU_L1_BLOCKS", 0);
    t.xStrips = envInt("SPGPU_X_STRIPS", -1);
    t.xTile = envInt("SPGPU_X_TILE", -1);
    t.autoSweep = envInt("SPGPU_AUTO_SWEEP", 1);
    t.poisonScratch = envInt("SPGPU_POISON_SCRATCH", 0);
    t.slide = envInt("SPGPU_SLIDE", 0);
    t.xTileShape = envInt("SPGPU_X_TILE_SHAPE", 0);
    t.deepSplit = envInt("SPGPU_DEEP_SPLIT", -1);
    t.deepCap = envInt("SPGPU_DEEP_CAP", 256);
    t.deepKeep = envInt("SPGPU_DEEP_KEEP", 64);
    t.ragged = envInt("SPGPU_RAGGED", 1);
    t.raggedShape = envInt("SPGPU_RAGGED_SHAPE", 0);
    t.pipeGroups = envInt("SPGPU_PIPE_GROUPS", 0);
    t.raggedSplit = envInt("SPGPU_RAGGED_SPLIT", -1);
    t.l1Nt = envInt("SPGPU_L1_NT", -1);
    t.plan = envInt("SPGPU_PLAN", 1);
    t.planDeepSpread = envInt("SPGPU_PLAN_DEEP_SPREAD", 60);
    t.planDeepPerBlock = envInt("SPGPU_PLAN_DEEP_PER_BLOCK", 8);
    t.planDeepRuns = envInt("SPGPU_PLAN_DEEP_RUNS", 1);
    t.freezeEscapesPct = envInt("SPGPU_FREEZE_MAX_ESCAPES_PCT", 1);
    t.stageLate = envInt("SPGPU_STAGE_LATE", 1);
    tuning = t;
    __atomic_store_n(&tuningLoaded, 1, __ATOMIC_RELEASE);
}

const SpgpuTuning* spgpuTuning(void)
{
    if (!__atomic_load_n(&tuningLoaded, __ATOMIC_ACQUIRE))
        spgpuTuningReload(); /* two threads racing here store the same values */
    return &tuning;
}
