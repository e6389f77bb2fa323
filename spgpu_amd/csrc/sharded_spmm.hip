/*
 * Row-sharded HELL SpMM, the per-rank driver in C (include/spgpu/sharded.h).  New functionality: the reference has no
 * multi-GPU code (core.h:84-85).  Everything on the data path of a step is issued here: the packing kernel, the RCCL
 * calls (through dlopen, so that libspgpu.so carries no link-time dependency on RCCL), the two products
 * (spgpuDhellspmm) and the events that order the communication stream against the handle's stream.
 *
 * Needed-rows set-up (device): the column indices of `rest` are sorted and reduced to their distinct values (rocPRIM;
 * this is set-up, not the path); position in that list = new column number; the list is cut at the row-block
 * boundaries, which gives the rows wanted from every owner.
 */
#include "numeric.hip.h"
#include "spgpu_internal.h"

#include "spgpu/sharded.h"
#include "spgpu/spmm.h"

#include <dlfcn.h>
#include <rccl/rccl.h> /* types and enums only: the functions are resolved with dlsym */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace spgpu {

/* ---- RCCL, opened on first use ------------------------------------------------------------------------------------ */
struct Rccl {
    void* lib;
    ncclResult_t (*getUniqueId)(ncclUniqueId*);
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*commInitAll)(ncclComm_t*, int, const int*);
    ncclResult_t (*commDestroy)(ncclComm_t);
    ncclResult_t (*allGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*groupStart)(void);
    ncclResult_t (*groupEnd)(void);
    ncclResult_t (*commAbort)(ncclComm_t); /* optional */
    const char* (*errorString)(ncclResult_t);
};
static Rccl rccl;
static int rcclState; /* 0 untried, 1 loaded, -1 unavailable */
static pthread_mutex_t rcclLock = PTHREAD_MUTEX_INITIALIZER;

static bool loadRccl()
{
    pthread_mutex_lock(&rcclLock);
    if (rcclState == 0) {
        /* the copy the process already uses first (torch ships its own librccl.so), then the system's */
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* lib = nullptr;
        /* SPGPU_RCCL_LIBRARY: a particular build (or, in the tests, an in-process stand-in that lets several ranks run as
         * threads on one GPU: tests/mock_rccl.c) */
        const char* named = getenv("SPGPU_RCCL_LIBRARY");
        if (named && named[0])
            lib = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
        for (int pass = 0; pass < 2 && !lib; ++pass)
            for (const char* name : names) {
                lib = dlopen(name, (pass == 0 ? RTLD_NOLOAD : 0) | RTLD_NOW | RTLD_GLOBAL);
                if (lib)
                    break;
            }
        rcclState = -1;
        if (lib) {
            rccl.lib = lib;
#define SPGPU_RCCL_SYM(field, symbol) *(void**)(&rccl.field) = dlsym(lib, symbol)
            SPGPU_RCCL_SYM(getUniqueId, "ncclGetUniqueId");
            SPGPU_RCCL_SYM(commInitRank, "ncclCommInitRank");
            SPGPU_RCCL_SYM(commInitAll, "ncclCommInitAll");
            SPGPU_RCCL_SYM(commDestroy, "ncclCommDestroy");
            SPGPU_RCCL_SYM(allGather, "ncclAllGather");
            SPGPU_RCCL_SYM(send, "ncclSend");
            SPGPU_RCCL_SYM(recv, "ncclRecv");
            SPGPU_RCCL_SYM(groupStart, "ncclGroupStart");
            SPGPU_RCCL_SYM(groupEnd, "ncclGroupEnd");
            SPGPU_RCCL_SYM(commAbort, "ncclCommAbort");
            SPGPU_RCCL_SYM(errorString, "ncclGetErrorString");
#undef SPGPU_RCCL_SYM
            if (rccl.getUniqueId && rccl.commInitRank && rccl.commInitAll && rccl.commDestroy && rccl.allGather && rccl.send &&
                rccl.recv && rccl.groupStart && rccl.groupEnd)
                rcclState = 1;
        }
    }
    const bool ok = rcclState == 1;
    pthread_mutex_unlock(&rcclLock);
    return ok;
}

static bool rcclOk(ncclResult_t r, const char* what)
{
    if (r == ncclSuccess)
        return true;
    fprintf(stderr, "spgpu sharded: %s failed: %s\n", what, rccl.errorString ? rccl.errorString(r) : "RCCL error");
    return false;
}

static bool hipOk(hipError_t e, const char* what)
{
    if (e == hipSuccess)
        return true;
    fprintf(stderr, "spgpu sharded: %s failed: %s\n", what, hipGetErrorString(e));
    return false;
}

/* ---- kernels ------------------------------------------------------------------------------------------------------ */
constexpr int kShThreads = 256;

static unsigned gridOverSh(long long n)
{
    const long long blocks = (n + kShThreads - 1) / kShThreads;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1048576 ? 1048576 : blocks));
}

/* zero-based columns of the REAL entries as unsigned keys (slot k of a row is real iff k < rS[row]); the array was
 * filled with 0xFFFFFFFF before, so padding slots sort behind every column and never become a "needed row" */
__global__ __launch_bounds__(kShThreads) void columnKeysKernel(unsigned* keys, const int* rP, const int* rS, const int* hackOffsets,
                                                               int hackSize, int rows, int base)
{
    const long long stride = (long long)gridDim.x * kShThreads;
    for (long long r = (long long)blockIdx.x * kShThreads + threadIdx.x; r < rows; r += stride) {
        const long long slot0 = (long long)hackOffsets[r / hackSize] + r % hackSize;
        const int len = rS[r];
        for (int k = 0; k < len; ++k)
            keys[slot0 + (long long)k * hackSize] = (unsigned)(rP[slot0 + (long long)k * hackSize] - base);
    }
}

__device__ inline long long lowerBound(const unsigned* sorted, long long n, unsigned value)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sorted[mid] < value)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

/* bounds[r] = number of needed rows below blockFirstRow[r], r = 0 .. world */
__global__ void ownerBoundsKernel(long long* bounds, const unsigned* needed, long long count, const long long* blockFirstRow,
                                  int world)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= world) {
        const long long first = blockFirstRow[r];
        bounds[r] = first > 0xFFFFFFFFll ? count : lowerBound(needed, count, (unsigned)first);
    }
}

__global__ __launch_bounds__(kShThreads) void renumberKernel(int* newRP, const int* rP, long long slots, int base,
                                                             const unsigned* needed, long long count)
{
    const long long stride = (long long)gridDim.x * kShThreads;
    for (long long i = (long long)blockIdx.x * kShThreads + threadIdx.x; i < slots; i += stride) {
        const unsigned col = (unsigned)(rP[i] - base);
        const long long at = lowerBound(needed, count, col);
        newRP[i] = (at < count && needed[at] == col) ? (int)at + base : base; /* a padding slot: any valid column */
    }
}

/* ask[i] = needed[i] - first row of its owner's block */
__global__ __launch_bounds__(kShThreads) void askKernel(int* ask, const unsigned* needed, long long count, const long long* bounds,
                                                        const long long* blockFirstRow, int world)
{
    const long long stride = (long long)gridDim.x * kShThreads;
    for (long long i = (long long)blockIdx.x * kShThreads + threadIdx.x; i < count; i += stride) {
        int owner = 0;
        while (owner + 1 < world && bounds[owner + 1] <= i)
            ++owner;
        ask[i] = (int)((long long)needed[i] - blockFirstRow[owner]);
    }
}

/* dst[i][0..count) = X[index[i]][0..count): one lane per element, rows of `count` doubles are contiguous */
__global__ __launch_bounds__(kShThreads) void packRowsKernel(double* dst, const double* X, const int* index, long long rows,
                                                             int count)
{
    const long long total = rows * count;
    const long long stride = (long long)gridDim.x * kShThreads;
    for (long long e = (long long)blockIdx.x * kShThreads + threadIdx.x; e < total; e += stride) {
        const long long i = e / count;
        const int j = (int)(e - i * count);
        dst[e] = X[(long long)index[i] * count + j];
    }
}

} // namespace spgpu

using namespace spgpu;

struct spgpuShardedSpmmPlan {
    spgpuHandle_t handle;
    ncclComm_t comm;
    int rank, world, count, exchange;
    long long* blockFirst; /* host, world + 1 */
    spgpuHellBlockD own, rest;
    bool hasRest, equalBlocks;
    hipStream_t commStream;
    hipEvent_t ready, landed;
    /* all-gather */
    double* xFull;
    /* needed rows */
    int* restRP;          /* rest's indices renumbered into the needed list */
    long long neededRows; /* length of that list */
    long long* want;      /* host, world + 1: bounds of the needed list per owner */
    long long* give;      /* host, world + 1: prefix of the rows every rank wants from me */
    int* sendIndex;       /* device: my rows, in the order they are sent */
    double* xNeeded;
    double* sendBuffer;
};

static void freePlan(spgpuShardedSpmm_t p)
{
    if (!p)
        return;
    int previous = 0;
    (void)hipGetDevice(&previous);
    (void)hipSetDevice(p->handle->device);
    if (p->commStream) {
        (void)hipStreamSynchronize(p->commStream);
        (void)hipStreamDestroy(p->commStream);
    }
    if (p->ready) (void)hipEventDestroy(p->ready);
    if (p->landed) (void)hipEventDestroy(p->landed);
    (void)hipFree(p->xFull);
    (void)hipFree(p->restRP);
    (void)hipFree(p->sendIndex);
    (void)hipFree(p->xNeeded);
    (void)hipFree(p->sendBuffer);
    (void)hipSetDevice(previous);
    free(p->blockFirst);
    free(p->want);
    free(p->give);
    free(p);
}

/* Do ALL ranks say yes?  One ncclAllGather of a status word through the handle's reduction scratch (allocated in
 * spgpuCreate: no allocation here that could fail on one rank only).  Every rank of the communicator must call this at the same
 * point of the set-up, whatever happened to it locally before: a rank that left the set-up on a local failure (hipMalloc,
 * rocPRIM, host malloc) without saying so would leave its peers inside the next collective for ever. */
static bool everyRankOk(spgpuShardedSpmm_t p, bool mine)
{
    if (p->world == 1)
        return mine;
    hipStream_t s = p->handle->currentStream;
    int* word = (int*)spgpuPrivate(p->handle)->reduceScratch; /* [0]: mine, [64 ...]: everyone's (world <= 1024) */
    int* all = word + 64;
    int host[1024];
    const int flag = mine ? 1 : 0;
    if (!hipOk(hipMemcpyAsync(word, &flag, sizeof(int), hipMemcpyHostToDevice, s), "copy") ||
        !rcclOk(rccl.allGather(word, all, 1, ncclInt32, p->comm, s), "ncclAllGather(status)") ||
        !hipOk(hipMemcpyAsync(host, all, (size_t)p->world * sizeof(int), hipMemcpyDeviceToHost, s), "copy") ||
        !hipOk(hipStreamSynchronize(s), "sync"))
        return false;
    bool ok = true;
    for (int r = 0; r < p->world; ++r)
        ok = ok && host[r] == 1;
    return ok;
}

/* the needed-rows set-up; collective over the communicator */
static spgpuStatus_t setUpNeeded(spgpuShardedSpmm_t p)
{
    hipStream_t s = p->handle->currentStream;
    const long long slots = p->hasRest ? p->rest.slots : 0;
    const int world = p->world;
    unsigned *keys = nullptr, *sorted = nullptr, *needed = nullptr;
    long long *dBounds = nullptr, *dFirst = nullptr;
    size_t* dCount = nullptr;
    void* temp = nullptr;
    int* ask = nullptr;
    int *dWant = nullptr, *dAllWant = nullptr;
    spgpuStatus_t status = SPGPU_UNSPECIFIED;
    long long count = 0;

    int* hostWant = nullptr;
    int* hostAll = nullptr;
    /* The set-up is a sequence of LOCAL phases (allocations, rocPRIM, kernels, host memory), each closed by an agreement of
     * all ranks (everyRankOk) before the collective that follows: either every rank enters that collective or none does. */
    bool local = false;
    do { /* phase 1 (local): the distinct off-block columns, cut at the row-block boundaries, rest renumbered */
        if (!hipOk(hipMalloc(&dBounds, (world + 1) * sizeof(long long)), "hipMalloc") ||
            !hipOk(hipMalloc(&dFirst, (world + 1) * sizeof(long long)), "hipMalloc") ||
            !hipOk(hipMemcpyAsync(dFirst, p->blockFirst, (world + 1) * sizeof(long long), hipMemcpyHostToDevice, s), "copy"))
            break;
        if (slots > 0) {
            size_t sortBytes = 0, uniqueBytes = 0;
            if (rocprim::radix_sort_keys(nullptr, sortBytes, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)slots) != hipSuccess ||
                rocprim::unique(nullptr, uniqueBytes, (unsigned*)nullptr, (unsigned*)nullptr, (size_t*)nullptr, (size_t)slots,
                                rocprim::equal_to<unsigned>()) != hipSuccess)
                break;
            const size_t tempBytes = sortBytes > uniqueBytes ? sortBytes : uniqueBytes;
            if (!hipOk(hipMalloc(&keys, slots * sizeof(unsigned)), "hipMalloc") ||
                !hipOk(hipMalloc(&sorted, slots * sizeof(unsigned)), "hipMalloc") ||
                !hipOk(hipMalloc(&dCount, sizeof(size_t)), "hipMalloc") || !hipOk(hipMalloc(&temp, tempBytes ? tempBytes : 16), "hipMalloc"))
                break;
            if (!hipOk(hipMemsetAsync(keys, 0xFF, slots * sizeof(unsigned), s), "hipMemsetAsync"))
                break;
            hipLaunchKernelGGL(columnKeysKernel, dim3(gridOverSh(p->rest.rows)), dim3(kShThreads), 0, s, keys, p->rest.rP, p->rest.rS,
                               p->rest.hackOffsets, p->rest.hackSize, p->rest.rows, p->rest.baseIndex);
            size_t bytes = tempBytes;
            if (rocprim::radix_sort_keys(temp, bytes, keys, sorted, (size_t)slots, 0, 32, s) != hipSuccess)
                break;
            bytes = tempBytes;
            needed = keys; /* the distinct values overwrite the unsorted copy */
            if (rocprim::unique(temp, bytes, sorted, needed, dCount, (size_t)slots, rocprim::equal_to<unsigned>(), s) != hipSuccess)
                break;
            size_t distinct = 0;
            if (!hipOk(hipMemcpyAsync(&distinct, dCount, sizeof(size_t), hipMemcpyDeviceToHost, s), "copy") ||
                !hipOk(hipStreamSynchronize(s), "sync"))
                break;
            count = (long long)distinct;
        }
        /* cut the list at the row-block boundaries; what lies behind the last boundary is padding, not a row
         * (world + 1 <= 1024 lanes: spgpuDhellspmmShardedCreate refuses larger communicators) */
        hipLaunchKernelGGL(ownerBoundsKernel, dim3(1), dim3(world + 1), 0, s, dBounds, needed, count, dFirst, world);
        if (!hipOk(hipMemcpyAsync(p->want, dBounds, (world + 1) * sizeof(long long), hipMemcpyDeviceToHost, s), "copy") ||
            !hipOk(hipStreamSynchronize(s), "sync"))
            break;
        p->neededRows = p->want[world];
        if (slots > 0) {
            if (!hipOk(hipMalloc(&p->restRP, slots * sizeof(int)), "hipMalloc"))
                break;
            hipLaunchKernelGGL(renumberKernel, dim3(gridOverSh(slots)), dim3(kShThreads), 0, s, p->restRP, p->rest.rP, slots,
                               p->rest.baseIndex, needed, p->neededRows);
        }
        hostWant = (int*)malloc((size_t)world * sizeof(int));
        hostAll = (int*)malloc((size_t)world * world * sizeof(int));
        if (!hostWant || !hostAll) {
            status = SPGPU_OUTOFMEMORY;
            break;
        }
        for (int r = 0; r < world; ++r)
            hostWant[r] = (int)(p->want[r + 1] - p->want[r]);
        if (world > 1 && (!hipOk(hipMalloc(&dWant, world * sizeof(int)), "hipMalloc") ||
                          !hipOk(hipMalloc(&dAllWant, (size_t)world * world * sizeof(int)), "hipMalloc") ||
                          !hipOk(hipMemcpyAsync(dWant, hostWant, world * sizeof(int), hipMemcpyHostToDevice, s), "copy")))
            break;
        /* tests: SPGPU_TEST_FAIL_SETUP_RANK = r makes rank r's local phase fail here (tests/test_gpu_sharded_c.py) */
        const char* failing = getenv("SPGPU_TEST_FAIL_SETUP_RANK");
        local = !(failing && failing[0] && atoi(failing) == p->rank);
    } while (0);

    do {
        if (!everyRankOk(p, local))
            break;
        /* who wants how many of my rows: all-gather of every rank's per-owner counts */
        if (world > 1) {
            if (!rcclOk(rccl.allGather(dWant, dAllWant, (size_t)world, ncclInt32, p->comm, s), "ncclAllGather(counts)") ||
                !hipOk(hipMemcpyAsync(hostAll, dAllWant, (size_t)world * world * sizeof(int), hipMemcpyDeviceToHost, s), "copy") ||
                !hipOk(hipStreamSynchronize(s), "sync"))
                break;
        } else {
            hostAll[0] = hostWant[0];
        }
        p->give[0] = 0;
        for (int r = 0; r < world; ++r)
            p->give[r + 1] = p->give[r] + hostAll[(size_t)r * world + p->rank];

        /* phase 2 (local): the row numbers this rank asks for, room for the ones it is asked for */
        const long long sendRows = p->give[world];
        local = true;
        if (p->neededRows > 0) {
            local = hipOk(hipMalloc(&ask, p->neededRows * sizeof(int)), "hipMalloc");
            if (local)
                hipLaunchKernelGGL(askKernel, dim3(gridOverSh(p->neededRows)), dim3(kShThreads), 0, s, ask, needed, p->neededRows, dBounds,
                                   dFirst, world);
        }
        if (local && sendRows > 0)
            local = hipOk(hipMalloc(&p->sendIndex, sendRows * sizeof(int)), "hipMalloc");
        if (!everyRankOk(p, local))
            break;
        /* every rank tells every owner which of its rows it wants */
        if (world > 1 && !rcclOk(rccl.groupStart(), "ncclGroupStart"))
            break;
        bool ok = true;
        for (int r = 0; r < world && ok; ++r) {
            const long long wantRows = p->want[r + 1] - p->want[r], giveRows = p->give[r + 1] - p->give[r];
            if (r == p->rank) {
                if (wantRows > 0) /* == giveRows */
                    ok = hipOk(hipMemcpyAsync(p->sendIndex + p->give[r], ask + p->want[r], wantRows * sizeof(int), hipMemcpyDeviceToDevice, s),
                               "copy");
                continue;
            }
            if (wantRows > 0)
                ok = ok && rcclOk(rccl.send(ask + p->want[r], (size_t)wantRows, ncclInt32, r, p->comm, s), "ncclSend(rows)");
            if (giveRows > 0)
                ok = ok && rcclOk(rccl.recv(p->sendIndex + p->give[r], (size_t)giveRows, ncclInt32, r, p->comm, s), "ncclRecv(rows)");
        }
        if (world > 1 && !ok && rccl.commAbort) {
            /* a half-posted group must not be launched: the communicator is given up instead (its peers' calls then fail
             * rather than wait); the caller must not use it again */
            (void)rccl.commAbort(p->comm);
            p->comm = nullptr;
            break;
        }
        if (world > 1 && !rcclOk(rccl.groupEnd(), "ncclGroupEnd"))
            break;
        if (!ok || !hipOk(hipStreamSynchronize(s), "sync"))
            break;

        /* phase 3 (local): the exchange buffers; the ranks agree once more so that all return the same status */
        const size_t rowBytes = (size_t)p->count * sizeof(double);
        local = true;
        if (p->neededRows > 0)
            local = hipOk(hipMalloc(&p->xNeeded, p->neededRows * rowBytes), "hipMalloc");
        if (local && sendRows > 0)
            local = hipOk(hipMalloc(&p->sendBuffer, sendRows * rowBytes), "hipMalloc");
        if (!everyRankOk(p, local))
            break;
        status = SPGPU_SUCCESS;
    } while (0);
    free(hostWant);
    free(hostAll);

    (void)hipStreamSynchronize(s);
    (void)hipFree(keys);
    (void)hipFree(sorted);
    (void)hipFree(dBounds);
    (void)hipFree(dFirst);
    (void)hipFree(dCount);
    (void)hipFree(temp);
    (void)hipFree(ask);
    (void)hipFree(dWant);
    (void)hipFree(dAllWant);
    return status;
}

extern "C" {

int spgpuCommAvailable(void)
{
    return loadRccl() ? 1 : 0;
}

spgpuStatus_t spgpuCommGetUniqueId(void* id128)
{
    if (!id128 || !loadRccl())
        return SPGPU_UNSUPPORTED;
    ncclUniqueId id;
    if (!rcclOk(rccl.getUniqueId(&id), "ncclGetUniqueId"))
        return SPGPU_UNSPECIFIED;
    memcpy(id128, &id, sizeof(id));
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCommInitRank(void** comm, int world, const void* id128, int rank)
{
    if (!comm || !id128 || !loadRccl())
        return SPGPU_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (!rcclOk(rccl.commInitRank(&c, world, id, rank), "ncclCommInitRank"))
        return SPGPU_UNSPECIFIED;
    *comm = c;
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuCommInitAll(void** comms, int ndev, const int* devices)
{
    if (!comms || ndev <= 0 || !loadRccl())
        return SPGPU_UNSUPPORTED;
    return rcclOk(rccl.commInitAll((ncclComm_t*)comms, ndev, devices), "ncclCommInitAll") ? SPGPU_SUCCESS : SPGPU_UNSPECIFIED;
}

void spgpuCommDestroy(void* comm)
{
    if (comm && loadRccl())
        rccl.commDestroy((ncclComm_t)comm);
}

spgpuStatus_t spgpuDhellspmmShardedCreate(spgpuShardedSpmm_t* plan, spgpuHandle_t handle, void* comm, int rank, int world,
                                          const long long* blockFirstRow, const spgpuHellBlockD* own, const spgpuHellBlockD* rest,
                                          int count, int exchange)
{
    if (!plan)
        return SPGPU_UNSPECIFIED;
    *plan = nullptr;
    if (!handle || !blockFirstRow || !own || world < 1 || rank < 0 || rank >= world || count <= 0 ||
        (exchange != SPGPU_EXCHANGE_ALLGATHER && exchange != SPGPU_EXCHANGE_NEEDED))
        return SPGPU_UNSPECIFIED;
    if (world > 960) /* the set-up cuts the needed rows with one lane per rank and agrees through 4 KiB of status words */
        return SPGPU_UNSUPPORTED;
    if (world > 1 && (!comm || !loadRccl()))
        return SPGPU_UNSUPPORTED;
    /* From here on a rank that fails says so to the others before anyone enters a collective of the needed-rows set-up
     * (everyRankOk): what can go wrong locally -- a row partition that does not fit this rank's block, host memory, the stream and
     * the events -- is folded into one status, agreed on ONCE before setUpNeeded, and setUpNeeded agrees again before each of its
     * own collectives.  (Only the checks above return without a word: a rank without a handle, a communicator or a sane rank
     * number has nothing to speak through.) */
    spgpuStatus_t status = SPGPU_SUCCESS;
    for (int r = 0; r < world; ++r)
        if (blockFirstRow[r + 1] < blockFirstRow[r])
            status = SPGPU_UNSPECIFIED;
    if (blockFirstRow[rank + 1] - blockFirstRow[rank] != own->rows || (rest && rest->rows > 0 && rest->rows != own->rows))
        status = SPGPU_UNSPECIFIED;

    spgpuShardedSpmm_t p = status == SPGPU_SUCCESS ? (spgpuShardedSpmm_t)calloc(1, sizeof(spgpuShardedSpmmPlan)) : nullptr;
    if (!p && status == SPGPU_SUCCESS)
        status = SPGPU_OUTOFMEMORY;
    int previous = 0;
    (void)hipGetDevice(&previous);
    (void)hipSetDevice(handle->device);
    if (p) {
        p->handle = handle;
        p->comm = (ncclComm_t)comm;
        p->rank = rank;
        p->world = world;
        p->count = count;
        p->exchange = exchange;
        p->own = *own;
        p->hasRest = rest && rest->rows > 0 && rest->slots > 0;
        if (p->hasRest)
            p->rest = *rest;
        p->blockFirst = (long long*)malloc((world + 1) * sizeof(long long));
        p->want = (long long*)calloc(world + 1, sizeof(long long));
        p->give = (long long*)calloc(world + 1, sizeof(long long));
        if (!p->blockFirst || !p->want || !p->give) {
            status = SPGPU_OUTOFMEMORY;
        } else {
            memcpy(p->blockFirst, blockFirstRow, (world + 1) * sizeof(long long));
            p->equalBlocks = true;
            for (int r = 1; r < world; ++r)
                p->equalBlocks = p->equalBlocks && (blockFirstRow[r + 1] - blockFirstRow[r] == blockFirstRow[1] - blockFirstRow[0]);
            if (!hipOk(hipStreamCreateWithFlags(&p->commStream, hipStreamNonBlocking), "hipStreamCreate") ||
                !hipOk(hipEventCreateWithFlags(&p->ready, hipEventDisableTiming), "hipEventCreate") ||
                !hipOk(hipEventCreateWithFlags(&p->landed, hipEventDisableTiming), "hipEventCreate"))
                status = SPGPU_UNSPECIFIED;
        }
    }
    if (exchange == SPGPU_EXCHANGE_NEEDED && world > 1) {
        /* the first agreement: through a stand-in plan, so that a rank whose own plan could not even be allocated takes part */
        spgpuShardedSpmmPlan voice{};
        voice.handle = handle;
        voice.comm = (ncclComm_t)comm;
        voice.rank = rank;
        voice.world = world;
        if (!everyRankOk(&voice, status == SPGPU_SUCCESS) && status == SPGPU_SUCCESS)
            status = SPGPU_UNSPECIFIED; /* another rank failed: nobody goes on */
    }
    if (status == SPGPU_SUCCESS) {
        if (exchange == SPGPU_EXCHANGE_NEEDED) {
            status = setUpNeeded(p);
        } else if (p->hasRest || world > 1) {
            const size_t bytes = (size_t)blockFirstRow[world] * count * sizeof(double);
            if (!hipOk(hipMalloc(&p->xFull, bytes ? bytes : 16), "hipMalloc"))
                status = SPGPU_OUTOFMEMORY;
        }
    }
    (void)hipSetDevice(previous);
    if (status != SPGPU_SUCCESS) {
        if (p)
            freePlan(p);
        return status;
    }
    *plan = p;
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuDhellspmmShardedExchange(spgpuShardedSpmm_t p, const double* X)
{
    if (!p)
        return SPGPU_UNSPECIFIED;
    hipStream_t compute = p->handle->currentStream, cs = p->commStream;
    const size_t rowBytes = (size_t)p->count * sizeof(double);
    const long long myRows = p->blockFirst[p->rank + 1] - p->blockFirst[p->rank];
    bool ok = true;
    /* after everything queued so far: X_local is there, and the previous step no longer reads the exchange buffer */
    ok = hipOk(hipEventRecord(p->ready, compute), "hipEventRecord") && hipOk(hipStreamWaitEvent(cs, p->ready, 0), "hipStreamWaitEvent");
    if (ok && p->exchange == SPGPU_EXCHANGE_NEEDED) {
        const long long sendRows = p->give[p->world];
        if (sendRows > 0)
            hipLaunchKernelGGL(packRowsKernel, dim3(gridOverSh(sendRows * p->count)), dim3(kShThreads), 0, cs, p->sendBuffer, X,
                               (const int*)p->sendIndex, sendRows, p->count);
        if (p->world > 1)
            ok = rcclOk(rccl.groupStart(), "ncclGroupStart");
        for (int r = 0; r < p->world && ok; ++r) {
            const long long wantRows = p->want[r + 1] - p->want[r], giveRows = p->give[r + 1] - p->give[r];
            if (r == p->rank) {
                if (wantRows > 0)
                    ok = hipOk(hipMemcpyAsync(p->xNeeded + p->want[r] * p->count, p->sendBuffer + p->give[r] * p->count, wantRows * rowBytes,
                                              hipMemcpyDeviceToDevice, cs), "copy");
                continue;
            }
            if (giveRows > 0)
                ok = ok && rcclOk(rccl.send(p->sendBuffer + p->give[r] * p->count, (size_t)(giveRows * p->count), ncclDouble, r, p->comm, cs),
                                  "ncclSend");
            if (wantRows > 0)
                ok = ok && rcclOk(rccl.recv(p->xNeeded + p->want[r] * p->count, (size_t)(wantRows * p->count), ncclDouble, r, p->comm, cs),
                                  "ncclRecv");
        }
        if (p->world > 1)
            ok = rcclOk(rccl.groupEnd(), "ncclGroupEnd") && ok;
    } else if (ok && p->xFull) {
        if (p->world > 1 && p->equalBlocks) {
            ok = rcclOk(rccl.allGather(X, p->xFull, (size_t)(myRows * p->count), ncclDouble, p->comm, cs), "ncclAllGather");
        } else {
            if (myRows > 0)
                ok = hipOk(hipMemcpyAsync(p->xFull + p->blockFirst[p->rank] * p->count, X, myRows * rowBytes, hipMemcpyDeviceToDevice, cs), "copy");
            if (p->world > 1) {
                ok = ok && rcclOk(rccl.groupStart(), "ncclGroupStart");
                for (int r = 0; r < p->world && ok; ++r) {
                    if (r == p->rank)
                        continue;
                    const long long theirs = p->blockFirst[r + 1] - p->blockFirst[r];
                    if (myRows > 0)
                        ok = ok && rcclOk(rccl.send(X, (size_t)(myRows * p->count), ncclDouble, r, p->comm, cs), "ncclSend");
                    if (theirs > 0)
                        ok = ok && rcclOk(rccl.recv(p->xFull + p->blockFirst[r] * p->count, (size_t)(theirs * p->count), ncclDouble, r, p->comm, cs),
                                          "ncclRecv");
                }
                ok = rcclOk(rccl.groupEnd(), "ncclGroupEnd") && ok;
            }
        }
    }
    ok = ok && hipOk(hipEventRecord(p->landed, cs), "hipEventRecord");
    return ok ? SPGPU_SUCCESS : SPGPU_UNSPECIFIED;
}

spgpuStatus_t spgpuDhellspmmShardedExchangeWait(spgpuShardedSpmm_t p)
{
    if (!p)
        return SPGPU_UNSPECIFIED;
    return hipOk(hipStreamWaitEvent(p->handle->currentStream, p->landed, 0), "hipStreamWaitEvent") ? SPGPU_SUCCESS : SPGPU_UNSPECIFIED;
}

static void productsAfterExchange(spgpuShardedSpmm_t p, double* Z, const double* Y, double alpha, const double* X, double beta, bool wait)
{
    const spgpuHellBlockD& o = p->own;
    spgpuDhellspmm(p->handle, Z, Y, alpha, o.cM, o.rP, o.hackSize, o.hackOffsets, o.rS, nullptr, o.avgNnzPerRow, o.rows, X, beta,
                   o.baseIndex, p->count, p->count, p->count);
    if (wait)
        (void)hipStreamWaitEvent(p->handle->currentStream, p->landed, 0);
    if (p->hasRest) {
        const spgpuHellBlockD& r = p->rest;
        const bool needed = p->exchange == SPGPU_EXCHANGE_NEEDED;
        if (!needed || p->neededRows > 0)
            spgpuDhellspmm(p->handle, Z, Z, alpha, r.cM, needed ? p->restRP : r.rP, r.hackSize, r.hackOffsets, r.rS, nullptr, r.avgNnzPerRow,
                           r.rows, needed ? p->xNeeded : p->xFull, 1.0, r.baseIndex, p->count, p->count, p->count);
    }
}

spgpuStatus_t spgpuDhellspmmShardedProducts(spgpuShardedSpmm_t p, double* Z, const double* Y, double alpha, const double* X, double beta)
{
    if (!p)
        return SPGPU_UNSPECIFIED;
    productsAfterExchange(p, Z, Y, alpha, X, beta, false);
    return SPGPU_SUCCESS;
}

spgpuStatus_t spgpuDhellspmmShardedStep(spgpuShardedSpmm_t p, double* Z, const double* Y, double alpha, const double* X, double beta)
{
    if (!p)
        return SPGPU_UNSPECIFIED;
    const spgpuStatus_t status = spgpuDhellspmmShardedExchange(p, X);
    if (status != SPGPU_SUCCESS)
        return status;
    productsAfterExchange(p, Z, Y, alpha, X, beta, true);
    spgpuDebugCheck(p->handle, "hellspmmSharded");
    return SPGPU_SUCCESS;
}

long long spgpuDhellspmmShardedRowsReceived(spgpuShardedSpmm_t p)
{
    if (!p)
        return 0;
    if (p->exchange == SPGPU_EXCHANGE_NEEDED)
        return p->neededRows - (p->want[p->rank + 1] - p->want[p->rank]);
    return p->xFull ? p->blockFirst[p->world] - (p->blockFirst[p->rank + 1] - p->blockFirst[p->rank]) : 0;
}

const double* spgpuDhellspmmShardedExchanged(spgpuShardedSpmm_t p, long long* rows)
{
    if (!p)
        return nullptr;
    if (rows)
        *rows = p->exchange == SPGPU_EXCHANGE_NEEDED ? p->neededRows : (p->xFull ? p->blockFirst[p->world] : 0);
    return p->exchange == SPGPU_EXCHANGE_NEEDED ? p->xNeeded : p->xFull;
}

void spgpuDhellspmmShardedDestroy(spgpuShardedSpmm_t p)
{
    freePlan(p);
}

} // extern "C"
