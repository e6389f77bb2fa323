/*
 * TEST INFRASTRUCTURE: an in-process stand-in for the ten RCCL entry points the sharded SpMM driver resolves with dlsym
 * (spgpu_amd/csrc/sharded_spmm.hip), so that several ranks can run as THREADS of one process on ONE GPU and the
 * driver's world > 1 logic -- who needs which rows, the request exchange, the per-step sends and receives -- can be
 * checked against the oracle where no second GPU exists.  Loaded through SPGPU_RCCL_LIBRARY (tests/test_gpu_sharded_c.py).
 * Not a communication library: every collective is a rendezvous of the ranks' threads plus device-to-device copies,
 * ordered with events exactly as the real calls order them with respect to the streams they are given:
 *   - a receive sees what the sender's stream had produced when the send was issued;
 *   - a sender's stream does not run past the collective before its buffer has been read.
 * It also CHECKS what real RCCL would turn into a hang or silent corruption: every receive must meet a send of the same
 * size from that peer, and no send may stay unmatched.
 *
 * Build: gcc -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/mock_rccl.c -L/opt/rocm/lib -lamdhip64 -lpthread
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_RANKS 16
#define MAX_OPS 256
#define MAX_WORLDS 64

typedef struct {
    const void* sendPtr;
    void* recvPtr;
    size_t bytes;
    int peer;
    int isSend;
    hipStream_t stream;
    hipEvent_t ready; /* sends: the data is there */
    int matched;
} MockOp;

typedef struct {
    unsigned long long id;
    int nranks, joined;
    pthread_barrier_t barrier;
    pthread_cond_t everyoneHere;
    MockOp ops[MAX_RANKS][MAX_OPS]; /* the operations of the collective in progress, per rank */
    int count[MAX_RANKS];
    hipEvent_t done[MAX_RANKS];     /* rank r has issued all its copies */
    int failed;
} MockWorld;

struct ncclComm {
    MockWorld* world;
    int rank;
};

static pthread_mutex_t tableLock = PTHREAD_MUTEX_INITIALIZER;
static MockWorld* worlds[MAX_WORLDS];
static unsigned long long nextId = 1;

static __thread int grouping;
static __thread MockOp pending[MAX_OPS];
static __thread int pendingCount;
static __thread struct ncclComm* pendingComm;
static __thread struct ncclComm* threadComm; /* a rank is a thread: the communicator it initialised last */

static size_t sizeOf(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

#define HIP_OK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "mock rccl: %s -> %s\n", #call, hipGetErrorString(e_));          \
            return ncclUnhandledCudaError;                                                   \
        }                                                                                    \
    } while (0)

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error (see stderr)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    pthread_mutex_lock(&tableLock);
    const unsigned long long mine = nextId++;
    pthread_mutex_unlock(&tableLock);
    memcpy(id->internal, &mine, sizeof(mine));
    memcpy(id->internal + 8, "mock", 4);
    return ncclSuccess;
}

static MockWorld* newWorld(unsigned long long id, int nranks)
{
    MockWorld* w = (MockWorld*)calloc(1, sizeof(MockWorld));
    w->id = id;
    w->nranks = nranks;
    pthread_barrier_init(&w->barrier, NULL, (unsigned)nranks);
    pthread_cond_init(&w->everyoneHere, NULL);
    return w;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks)
        return ncclInvalidArgument;
    unsigned long long id;
    memcpy(&id, commId.internal, sizeof(id));
    pthread_mutex_lock(&tableLock);
    MockWorld* w = NULL;
    int freeSlot = -1;
    for (int i = 0; i < MAX_WORLDS; ++i) {
        if (worlds[i] && worlds[i]->id == id)
            w = worlds[i];
        if (!worlds[i] && freeSlot < 0)
            freeSlot = i;
    }
    if (!w) {
        if (freeSlot < 0) {
            pthread_mutex_unlock(&tableLock);
            return ncclInternalError;
        }
        w = worlds[freeSlot] = newWorld(id, nranks);
    }
    if (w->nranks != nranks) {
        pthread_mutex_unlock(&tableLock);
        fprintf(stderr, "mock rccl: ranks disagree on the size of the communicator\n");
        return ncclInvalidArgument;
    }
    w->joined += 1;
    if (w->joined == nranks)
        pthread_cond_broadcast(&w->everyoneHere);
    while (w->joined < nranks) /* as the real call: returns when every rank has arrived */
        pthread_cond_wait(&w->everyoneHere, &tableLock);
    pthread_mutex_unlock(&tableLock);
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof(*c));
    c->world = w;
    c->rank = rank;
    HIP_OK(hipEventCreateWithFlags(&w->done[rank], hipEventDisableTiming));
    threadComm = c;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    (void)devlist;
    if (ndev != 1) {
        fprintf(stderr, "mock rccl: ncclCommInitAll with one device only (ranks are threads here)\n");
        return ncclInvalidArgument;
    }
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    return ncclCommInitRank(&comms[0], 1, id, 0);
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    free(comm); /* the world stays in the table: tests create a handful */
    return ncclSuccess;
}

/* The collective itself: every rank of the world arrives with its list of sends and receives. */
static ncclResult_t rendezvous(struct ncclComm* c, MockOp* ops, int count)
{
    MockWorld* w = c->world;
    const int me = c->rank;
    ncclResult_t result = ncclSuccess;
    for (int i = 0; i < count; ++i) {
        ops[i].matched = 0;
        if (ops[i].isSend) {
            if (hipEventCreateWithFlags(&ops[i].ready, hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(ops[i].ready, ops[i].stream) != hipSuccess)
                result = ncclUnhandledCudaError;
        }
    }
    memcpy(w->ops[me], ops, (size_t)count * sizeof(MockOp));
    w->count[me] = count;
    pthread_barrier_wait(&w->barrier); /* every list is posted */

    hipStream_t myStream = count ? ops[0].stream : NULL;
    for (int i = 0; i < count && result == ncclSuccess; ++i) {
        MockOp* r = &w->ops[me][i];
        if (r->isSend)
            continue;
        MockOp* s = NULL;
        if (r->peer >= 0 && r->peer < w->nranks)
            for (int j = 0; j < w->count[r->peer]; ++j) {
                MockOp* cand = &w->ops[r->peer][j];
                if (cand->isSend && cand->peer == me && !cand->matched) {
                    s = cand;
                    break;
                }
            }
        if (!s) {
            fprintf(stderr, "mock rccl: rank %d receives %zu bytes from rank %d, which sends nothing to it (real RCCL: a hang)\n", me,
                    r->bytes, r->peer);
            result = ncclInvalidUsage;
            break;
        }
        if (s->bytes != r->bytes) {
            fprintf(stderr, "mock rccl: rank %d receives %zu bytes from rank %d, which sends %zu\n", me, r->bytes, r->peer, s->bytes);
            result = ncclInvalidUsage;
            break;
        }
        s->matched = 1; /* only the receiving rank touches this send (one receiver per send) */
        if (hipStreamWaitEvent(r->stream, s->ready, 0) != hipSuccess ||
            (r->bytes && hipMemcpyAsync(r->recvPtr, s->sendPtr, r->bytes, hipMemcpyDeviceToDevice, r->stream) != hipSuccess))
            result = ncclUnhandledCudaError;
    }
    if (myStream && hipEventRecord(w->done[me], myStream) != hipSuccess)
        result = ncclUnhandledCudaError;
    if (result != ncclSuccess)
        w->failed = 1;
    pthread_barrier_wait(&w->barrier); /* every copy is issued */

    for (int i = 0; i < count; ++i) {
        MockOp* s = &w->ops[me][i];
        if (!s->isSend)
            continue;
        if (!s->matched && !w->failed) {
            fprintf(stderr, "mock rccl: rank %d sends %zu bytes to rank %d, which does not receive them (real RCCL: a hang)\n", me,
                    s->bytes, s->peer);
            result = ncclInvalidUsage;
            w->failed = 1;
        }
        /* the sender's stream may reuse the buffer only after the receiver's copy */
        if (s->matched && w->count[s->peer] && hipStreamWaitEvent(s->stream, w->done[s->peer], 0) != hipSuccess)
            result = ncclUnhandledCudaError;
        (void)hipEventDestroy(s->ready);
    }
    pthread_barrier_wait(&w->barrier); /* nobody reads the lists any more */
    return w->failed ? (result != ncclSuccess ? result : ncclInvalidUsage) : result;
}

static ncclResult_t enqueue(struct ncclComm* c, MockOp op)
{
    if (!grouping) {
        MockOp one = op;
        return rendezvous(c, &one, 1);
    }
    if (pendingComm && pendingComm != c) {
        fprintf(stderr, "mock rccl: one communicator per group\n");
        return ncclInvalidUsage;
    }
    if (pendingCount >= MAX_OPS)
        return ncclInternalError;
    pendingComm = c;
    pending[pendingCount++] = op;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    grouping += 1;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (grouping <= 0)
        return ncclInvalidUsage;
    grouping -= 1;
    if (grouping > 0)
        return ncclSuccess;
    struct ncclComm* c = pendingComm;
    const int count = pendingCount;
    pendingComm = NULL;
    pendingCount = 0;
    if (!c)
        c = threadComm; /* a rank with nothing to send or receive still meets the others (here every group is collective) */
    if (!c)
        return ncclSuccess;
    return rendezvous(c, pending, count);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    MockOp op;
    memset(&op, 0, sizeof(op));
    op.sendPtr = sendbuff;
    op.bytes = count * sizeOf(datatype);
    op.peer = peer;
    op.isSend = 1;
    op.stream = stream;
    return enqueue(comm, op);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    MockOp op;
    memset(&op, 0, sizeof(op));
    op.recvPtr = recvbuff;
    op.bytes = count * sizeOf(datatype);
    op.peer = peer;
    op.isSend = 0;
    op.stream = stream;
    return enqueue(comm, op);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    /* every rank sends its block to every rank (itself included) and receives everybody's at rank * bytes */
    MockWorld* w = comm->world;
    const size_t bytes = sendcount * sizeOf(datatype);
    MockOp ops[2 * MAX_RANKS];
    int n = 0;
    for (int r = 0; r < w->nranks; ++r) {
        MockOp s, v;
        memset(&s, 0, sizeof(s));
        memset(&v, 0, sizeof(v));
        s.sendPtr = sendbuff; s.bytes = bytes; s.peer = r; s.isSend = 1; s.stream = stream;
        v.recvPtr = (char*)recvbuff + (size_t)r * bytes; v.bytes = bytes; v.peer = r; v.isSend = 0; v.stream = stream;
        ops[n++] = s;
        ops[n++] = v;
    }
    return rendezvous(comm, ops, n);
}
