#pragma once
/*
 * Row-sharded HELL SpMM across the GPUs of one node:  Z = alpha*A*X + beta*Y, `count` right-hand sides.
 * NEW: the reference is a one-GPU library (core.h:84-85: one handle == one device); BASELINE.json's north_star
 * partitions the multi-vector product by rows over the 8 GPUs of a node with an RCCL exchange of the dense X
 * (SURVEY.md section 8(e)).  This header is that driver in C: one call per rank and step, every data-path
 * operation (packing, RCCL, the products) issued from here -- no Python, no torch on the path.
 *
 * Model: one rank per GPU (one process per GPU with ncclCommInitRank, or one process driving several devices
 * with ncclCommInitAll -- both work, a rank is a (handle, communicator) pair; in one process every rank needs a
 * host thread of its own: Create with SPGPU_EXCHANGE_NEEDED and every Step are collectives that a single thread
 * calling them rank after rank would never complete -- tests/run_sharded_ranks.py drives 8 ranks that way).  Rows of A, Y, Z and X are cut
 * into contiguous blocks, boundaries multiples of the hack size; rank r owns rows
 * [blockFirstRow[r], blockFirstRow[r+1]).  The caller has cut its row block of A by column ownership:
 *     own   entries whose column lies in the rank's own block, columns REBASED to that block (they index X_local)
 *     rest  the other entries, global columns
 * (spgpu_amd/synth.py split_uniform_hell_by_columns / sharded.py do this for the tests and bench.py).
 * A step is then:  exchange of X starts on the plan's communication stream;  Z = alpha*own*X_local + beta*Y runs on
 * the handle's stream meanwhile;  Z += alpha*rest*X_exchanged follows when the exchange has landed.
 *
 * Exchanges:
 *   SPGPU_EXCHANGE_ALLGATHER  ncclAllGather of the X row blocks (blocks of equal size; grouped ncclSend/ncclRecv
 *                             otherwise).  What north_star names.  Per step a rank receives all of X.
 *   SPGPU_EXCHANGE_NEEDED     only the X rows `rest` names travel.  Create lists them (sorted, unique, on the
 *                             device), asks every owner for its share (one ncclAllGather of counts, one grouped
 *                             exchange of row numbers) and renumbers rest's columns into that list; a step packs
 *                             the rows the others asked for and moves them with ONE group of ncclSend/ncclRecv.
 *                             Banded matrix: a halo per neighbour instead of all of X.
 *
 * RCCL is opened with dlopen on first use (librccl.so as loaded by the process, else /opt/rocm/lib): libspgpu.so
 * has no link-time dependency on it and single-GPU users never load it.  comm == NULL with world == 1 runs
 * without RCCL at all.
 *
 * Stream contract: a step is asynchronous.  It is ordered after everything already queued on
 * spgpuGetStream(handle) (X_local and Y must be produced there or be complete) and Z is complete when that
 * stream reaches the end of the step.  The plan owns its communication stream and events.  One step of a plan
 * at a time.  All ranks of the communicator must call Create / Step in the same order.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPGPU_EXCHANGE_ALLGATHER 0
#define SPGPU_EXCHANGE_NEEDED    1

/* One HELL row block in device memory, arguments as spgpuDhellspmm takes them (spmm.h). */
typedef struct spgpuHellBlockD {
    const __device double* cM;
    const __device int* rP;
    int hackSize;
    const __device int* hackOffsets;
    const __device int* rS;
    int rows;
    int avgNnzPerRow;
    int baseIndex;
    long long slots; /* entries of cM / rP (hackSize * allocation height): what the needed-rows set-up renumbers */
} spgpuHellBlockD;

typedef struct spgpuShardedSpmmPlan* spgpuShardedSpmm_t;

/* ---- RCCL through dlopen: enough for a caller that does not want <rccl/rccl.h> itself -------------------------- */
/* 1 if the RCCL library could be opened: the copy already in the process (torch ships one), else the system's;
 * the environment variable SPGPU_RCCL_LIBRARY names a particular one. */
int spgpuCommAvailable(void);
/* 128 bytes (ncclUniqueId) from ncclGetUniqueId; hand them to the other ranks by any means. */
spgpuStatus_t spgpuCommGetUniqueId(__host void* id128);
/* ncclCommInitRank on the calling thread's current device; *comm is an ncclComm_t. */
spgpuStatus_t spgpuCommInitRank(__host void** comm, int world, __host const void* id128, int rank);
/* ncclCommInitAll: one process, ndev devices (devices == NULL: 0 .. ndev-1). */
spgpuStatus_t spgpuCommInitAll(__host void** comms, int ndev, __host const int* devices);
void spgpuCommDestroy(void* comm);

/* ---- the plan ------------------------------------------------------------------------------------------------- */
/* blockFirstRow: world + 1 host entries.  rest may be NULL (or rows == 0) when the block has no foreign columns.
 * comm: the rank's ncclComm_t (any RCCL the process uses), NULL allowed for world == 1.
 * Collective over the communicator for SPGPU_EXCHANGE_NEEDED: every rank must call it.  Synchronises the handle's stream.
 * world <= 960 (SPGPU_UNSUPPORTED beyond).  Failure: every local step -- the check of the row partition against this rank's
 * block, the plan's host memory, its stream and events, then the set-up's allocations and sorting -- is followed by an
 * agreement of all ranks before the next collective, so a failure on ONE rank makes EVERY rank return an error from
 * this call instead of leaving the others inside a collective (the one exception: a rank called without a handle, without
 * a communicator or with a rank number outside 0 .. world-1 returns at once -- it has nothing to speak through); if a send or receive of the row-number exchange cannot be
 * posted, the communicator is aborted (ncclCommAbort, where the RCCL in use has it) rather than a half-posted group launched
 * -- it must not be used again.  An error inside RCCL itself (a peer that died) is RCCL's to report. */
spgpuStatus_t spgpuDhellspmmShardedCreate(spgpuShardedSpmm_t* plan, spgpuHandle_t handle, void* comm, int rank, int world,
                                          __host const long long* blockFirstRow, const spgpuHellBlockD* own,
                                          const spgpuHellBlockD* rest, int count, int exchange);

/* One product.  X_local, Y, Z: this rank's row blocks, interleaved, leading dimension `count`.  Z may alias Y. */
spgpuStatus_t spgpuDhellspmmShardedStep(spgpuShardedSpmm_t plan, __device double* Z, const __device double* Y, double alpha,
                                        const __device double* X_local, double beta);

/* The two halves of a step on their own (bench.py times them): the exchange alone (returns when queued; the result is
 * in the plan's buffer once the handle's stream has passed the step), and the products alone on whatever the plan's
 * buffer holds. */
spgpuStatus_t spgpuDhellspmmShardedExchange(spgpuShardedSpmm_t plan, const __device double* X_local);
/* makes the handle's stream wait for the exchange queued last (a step does this itself before the second product) */
spgpuStatus_t spgpuDhellspmmShardedExchangeWait(spgpuShardedSpmm_t plan);
spgpuStatus_t spgpuDhellspmmShardedProducts(spgpuShardedSpmm_t plan, __device double* Z, const __device double* Y,
                                            double alpha, const __device double* X_local, double beta);

/* X rows this rank receives per step from OTHER ranks, and the buffer they land in (device; all-gather: all of X in
 * rank order, needed: the needed rows in ascending global row number). */
long long spgpuDhellspmmShardedRowsReceived(spgpuShardedSpmm_t plan);
const __device double* spgpuDhellspmmShardedExchanged(spgpuShardedSpmm_t plan, long long* rows);

void spgpuDhellspmmShardedDestroy(spgpuShardedSpmm_t plan);

#ifdef __cplusplus
}
#endif
