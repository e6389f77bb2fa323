#pragma once
/* Library-private declarations shared by the host C files and the HIP kernels. */
#include "spgpu/core.h"
#include "spgpu/tuning.h"

#include <pthread.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPGPU_HANDLE_MAGIC 0x53504750u /* 'SPGP' */

/* Reduction scratch: up to SPGPU_REDUCE_MAX_BLOCKS partials of the widest
 * type (double complex, 16 B) per reduction call. */
#define SPGPU_REDUCE_MAX_BLOCKS 1024
#define SPGPU_REDUCE_SCRATCH_BYTES (SPGPU_REDUCE_MAX_BLOCKS * 16)

/* The public struct is the FIRST member, so a spgpuHandle_t is also a pointer
 * to this record. */
#define SPGPU_DEEP_STREAMS 8
#define SPGPU_PLANS 8
#define SPGPU_PLAN_WORDS 4
#define SPGPU_PLAN_GRAVES 32
typedef struct SpgpuPrivateHandle {
    SpgpuHandleStruct pub;
    unsigned magic;
    void* reduceScratch; /* device, SPGPU_REDUCE_SCRATCH_BYTES */
    void* reduceHost;    /* pinned host mirror of reduceScratch */
    /* ELL/HELL SpMV, kernel-form feedback (ellpack_spmv.hip): per recently seen matrix, SPGPU_FEEDBACK_SAMPLES ints
     * in pinned host memory that sample wavefronts of the strip-capable kernel write (0 unknown, 1 ran as gathers,
     * 2 ran as strips) and the host reads -- without synchronising -- at a later call on the same matrix. */
    int* formFeedback;                              /* pinned, SPGPU_FEEDBACK_ENTRIES * SPGPU_FEEDBACK_SAMPLES ints */
    const void* formKey[8];
    int formRows[8];
    int formCalls[8];                               /* SpMV calls seen for the entry */
    int formGeneration[8];                          /* bumped when the entry is given to another matrix (tags the reports) */
    unsigned formNext;
    pthread_mutex_t formLock; /* guards formKey / formRows / formNext */
    int spmvForm;             /* SPGPU_SPMV_FORM_* set by spgpuSetSpmvForm (atomic) */
    /* deep lists of the ELL/HELL SpMV, one per stream the handle has been given (spgpuCreate: the default stream;
     * spgpuSetStream: every new one): two SpMVs of one handle in flight on two streams never share a list */
    void* deepScratch[SPGPU_DEEP_STREAMS];
    hipStream_t deepStream[SPGPU_DEEP_STREAMS];
    hipEvent_t deepIdle[SPGPU_DEEP_STREAMS];  /* recorded behind the deep kernels of the list's latest call: complete = nobody uses the list */
    int deepUsed[SPGPU_DEEP_STREAMS];         /* the event has been recorded at least once */
    int deepPinned[SPGPU_DEEP_STREAMS];       /* a captured launch carries this list's addresses: it never changes hands */
    unsigned deepClock[SPGPU_DEEP_STREAMS];   /* last handed out (the least recently used idle list changes hands) */
    unsigned deepTick;
    int deepStreams;
    int deepFallbacks;                        /* ordered SpMV calls on a stream without a list (spgpuDeepListFallbacks) */
    int deepRecycled;                         /* lists that changed hands (spgpuDeepListsRecycled) */
    int lastSpmvForm;         /* form of the most recent ELL/HELL SpMV launch (diagnostic, atomic) */
    /* per-matrix plans of the ELL/HELL SpMV with a row order (below; guarded by formLock) */
    struct SpgpuSpmvPlan* plans;                    /* [SPGPU_PLANS] */
    int* planPinned;                                /* pinned, SPGPU_PLANS * SPGPU_PLAN_WORDS ints */
    void* planGraveyard[SPGPU_PLAN_GRAVES];         /* device buffers of retired plans: kernels in flight may still read them */
    int planGraves;
    unsigned planClock;
    int planUses, planBuilds, planStales;           /* diagnostics (spgpuSpmvPlanCounts) */
    int planFreezes;                                /* spgpu?SpmvFreeze calls that left a matrix frozen */
    /* ADOPTED matrices (spgpuHellSpmvAdopt, include/spgpu/tuning.h; csrc/adopted_hell.hip): a HELL matrix without a row order whose rows
     * are ragged, of which the library keeps its own copy with the rows ordered by length; guarded by formLock */
    struct SpgpuAdopted* adopted;                   /* [SPGPU_ADOPTED] */
    int adoptedCount;                               /* entries in use: an SpMV without rIdx looks one up only when > 0 */
    int adoptedUses;
    int planFrozenSlabs;                            /* frozen records of matrices WITHOUT a row order (subs < 0): an SpMV of the default kernels looks one up only when > 0 */
} SpgpuPrivateHandle;
#define SPGPU_FEEDBACK_ENTRIES 8 /* + one more group of words behind them for spgpu?SpmvForm, and one for the deep list's overflow report */
#define SPGPU_FEEDBACK_SAMPLES 4

static inline SpgpuPrivateHandle* spgpuPrivate(spgpuHandle_t h)
{
    return (SpgpuPrivateHandle*)(void*)h;
}

/* Deep list of the ELL/HELL SpMV (csrc/ellpack_spmv.hip, DEEP form): device memory owned by the handle, allocated on
 * the first call that needs it.  The main kernel registers every 32-row sub-group deeper than deepCap as one ENTRY and
 * its columns beyond the cap as ITEMS of deepChunk columns; deepItemsKernel gives every item to a wavefront,
 * deepFinishKernel adds an entry's item sums in item order, writes z and -- the workgroup that finishes last -- zeroes the
 * header for the next call.  A list belongs to ONE stream of the handle (calls on one stream run in order); the handle
 * keeps a list for each of the first SPGPU_DEEP_STREAMS streams it is given -- allocated in spgpuCreate / spgpuSetStream,
 * never inside an SpMV call -- and an SpMV on a stream without a list runs the kernel that needs none
 * (share_spmv.hip.h). */
typedef struct SpgpuDeepEntry {
    int row0;      /* first row of the 32-row sub-group (a multiple of 32) */
    int depth;     /* its longest row */
    int firstItem; /* its items are firstItem .. firstItem + items - 1 */
    int items;     /* 0: the item list was full, the main kernel kept the whole sub-group */
} SpgpuDeepEntry;
/* What a wavefront of deepItemsKernel needs to know about its item, in one 16-byte load: no look-up of the entry (a dependent
 * round trip) stands between the header and the item's first loads. */
typedef struct SpgpuDeepItem {
    int row0;      /* first row of the sub-group */
    unsigned base; /* slot of row0's first entry in the slab arrays (hackOffsets[hack] + row0 % hackSize; row0 for ELL) */
    int depth;     /* the sub-group's longest row */
    int chunk;     /* this item: columns deepKeep + chunk * deepChunk ... */
} SpgpuDeepItem;
#define SPGPU_DEEP_ENTRIES 8192
#define SPGPU_DEEP_ITEMS 20480
/* HEAD_CUT: SPGPU_DEEP_ITEMS - (first item of the first registration whose items did not fit), or 0: the items below
 * SPGPU_DEEP_ITEMS - HEAD_CUT were all written by THIS call (a registration that does not fit leaves its slots as an earlier call
 * left them, and every later one starts behind it) */
enum { SPGPU_DEEP_HEAD_ENTRIES = 0, SPGPU_DEEP_HEAD_ITEMS = 1, SPGPU_DEEP_HEAD_TICKET = 2, SPGPU_DEEP_HEAD_CUT = 3, SPGPU_DEEP_HEAD_INTS = 64 };
typedef struct SpgpuDeepList {
    int* header;             /* [SPGPU_DEEP_HEAD_INTS]: entries registered, items handed out (both may exceed the capacity), finish ticket */
    SpgpuDeepEntry* entries; /* [SPGPU_DEEP_ENTRIES] */
    SpgpuDeepItem* items;    /* [SPGPU_DEEP_ITEMS] */
    void* partials;          /* [SPGPU_DEEP_ENTRIES][32] x 16 bytes: row sums over the columns < deepCap */
    void* itemSums;          /* [SPGPU_DEEP_ITEMS][32] x 16 bytes */
    hipEvent_t idle;         /* to be recorded behind the kernels that use the list */
} SpgpuDeepList;
/* Device pointers of the current stream's list, or SPGPU_UNSUPPORTED when that stream has none. */
spgpuStatus_t spgpuDeepScratch(spgpuHandle_t h, SpgpuDeepList* list);
/* The current stream's list has gone into a captured graph: it stays with that stream for the handle's lifetime. */
void spgpuDeepListPin(spgpuHandle_t h);

/*
 * The plan of one matrix with a row order (csrc/planned_spmv.hip; the north_star target).  The queue kernel for ordered rows
 * spends a third of its prologue finding out two things that do not change from call to call: where the slice of x lies that
 * a block of rows touches (a dependent round trip: the first and last column of every row), and which 32-row sub-groups
 * are too deep for a block (registered through atomics in a list, worked off by two launches BEHIND the main kernel).  An
 * analysis pass -- started by the first SpMV that sees the matrix, on its stream, not waited for -- writes both down once:
 * a 32-byte record per block of SUBS sub-groups, and the list of deep sub-groups, which then get workgroups of their own in the
 * SAME launch.  Later calls on the same arrays find the plan by its key.
 *
 * A plan decides only WHO computes a sub-group and WHERE the LDS tile lies -- never what is read or in which order it is
 * added: every kernel reads the row lengths, slab bases and destinations of the matrix as it is at the call and derives the
 * chunks of a sub-group from its true depth.  A plan that has gone stale (another matrix at the same addresses) therefore
 * still gives the same bits as no plan; the kernels notice (a sub-group's true depth contradicts the plan), say so in a
 * pinned word, and the next call builds a new one.
 */
typedef struct SpgpuPlanBlock {
    int lowest, highest; /* columns (0-based) the block's rows reach: first and last entry of every row outside deep sub-groups */
    int middle;          /* mean of (first + last) / 2 over those rows */
    int probed;          /* how many rows took part (0: no tile) */
    unsigned long long deepMask; /* bit s: sub-group s of the block is in the deep list */
    int firstDeep;       /* index of its first entry there */
    int packBase;        /* frozen matrices (spgpu?SpmvFreeze): the column the block's 16-bit packed indices count from */
} SpgpuPlanBlock;
typedef struct SpgpuSpmvPlan {
    /* key: the arrays the analysis read, and what it assumed */
    const void *rP, *rS, *rIdx, *hackOffsets;
    long long idxStride;
    int rows, hackSize, baseIndex, maxNnz, deepCap, subs; /* subs: 32-row sub-groups per block; < 0: the frozen record of a matrix WITHOUT a
                                                           * row order (ellpack_spmv.hip freezeSlab): -rows per group, `device` = the groups' bases */
    /* state */
    int state;      /* SPGPU_PLAN_EMPTY ... */
    int stales;     /* times it was found stale */
    int uses;       /* launches since it was built */
    unsigned clock; /* last looked up (least recently used goes first) */
    int blocks;     /* blocks of `subs` sub-groups */
    int deep;       /* deep sub-groups (read from pinned[0] once the analysis has completed) */
    void* device;   /* one allocation: SpgpuPlanBlock[blocks] | int counts[blocks] | int deepSubs[sub-groups] */
    long long packedBytes;
    void* packed;   /* frozen matrices only (spgpu?SpmvFreeze, include/spgpu/tuning.h), else NULL: the column indices of the rows the blocks walk
                     * themselves as 16-bit offsets from the block's packBase, slot for slot as in rP (0xFFFF: ask rP) */
    int* pinned;    /* [0] deep sub-groups, written by the analysis; [1] != 0: a kernel found the plan stale; [2] slots of the index array (freeze) */
    hipEvent_t built;
} SpgpuSpmvPlan;
enum { SPGPU_PLAN_EMPTY = 0, SPGPU_PLAN_BUILDING = 1, SPGPU_PLAN_READY = 2, SPGPU_PLAN_GIVEN_UP = 3 };
/* The plan table is used with the handle's lock held from the look-up to the launch that reads the plan's arrays (a second
 * host thread on the handle may retire a plan and free retired buffers only under the same lock, after a device-wide wait). */
void spgpuPlanLock(spgpuHandle_t h);
void spgpuPlanUnlock(spgpuHandle_t h);
/* Lock held.  The record with this key; or, if there is none, the least recently used record, retired and re-keyed
 * (state EMPTY).  Never NULL once the handle exists (NULL: the handle has no plan table -- its allocation failed). */
SpgpuSpmvPlan* spgpuPlanRecord(spgpuHandle_t h, const SpgpuSpmvPlan* key);
/* Lock held.  The record with this key, or NULL (nothing is retired, nothing re-keyed). */
SpgpuSpmvPlan* spgpuPlanFind(spgpuHandle_t h, const SpgpuSpmvPlan* key);
/* An adopted matrix: the caller's arrays (the key) and the library's ordered copy of them. */
#define SPGPU_ADOPTED 4
typedef struct SpgpuAdopted {
    const void *cM, *rP, *rS, *hackOffsets; /* the caller's (key; rows == 0: free entry); ELL: hackOffsets NULL, hackSize 0 */
    int rows, hackSize, baseIndex, type;
    long long valPitch, idxPitch;           /* ELL: the caller's pitches (key); HELL: 0 */
    void* values;       /* ordered copy: coefficients */
    int* indices;       /* column indices */
    int* hackOffsetsOrdered;
    int* lengths;       /* row lengths in the new order */
    int* order;         /* rIdx: original row of every position */
    long long bytes;    /* device memory of the copy */
} SpgpuAdopted;
/* No lock held.  The ordered copy of these arrays, or NULL (none, or the stream is capturing: a graph would outlive the copy). */
const SpgpuAdopted* spgpuAdoptedFind(spgpuHandle_t h, hipStream_t stream, const void* cM, const int* rP, const int* rS, const int* hackOffsets,
                                     int rows, int hackSize, int baseIndex, long long valPitch, long long idxPitch);
/* No lock held.  Takes a free entry for `entry` (copied); SPGPU_UNSUPPORTED when the table is full. */
int spgpuAdoptedAdd(spgpuHandle_t h, const SpgpuAdopted* entry);
/* No lock held.  Removes the entries keyed by rP (NULL: all); their arrays are returned in `out` (at most SPGPU_ADOPTED) for the caller to free. */
int spgpuAdoptedRemove(spgpuHandle_t h, const int* rP, SpgpuAdopted* out);

/* Lock held.  The plan's device buffer goes to the graveyard (kernels in flight may read it); a full graveyard is emptied
 * after a device-wide wait.  State EMPTY afterwards. */
void spgpuPlanRetire(spgpuHandle_t h, SpgpuSpmvPlan* plan);

/* Has everything recorded into `event` finished?  hipEventQuery says hipErrorNotReady for "not yet" -- an answer, not a
 * failure -- but the runtime also leaves it behind as the thread's last error, where a caller in the reference's style
 * (hellPerf.cpp:385-390: cudaGetLastError after the launches) would find it and report a failed SpMV.  It is taken back here;
 * any other error the caller has not looked at yet stays where it is. */
static inline int spgpuEventDone(hipEvent_t event)
{
    const hipError_t said = hipEventQuery(event);
    if (said != hipSuccess && hipPeekAtLastError() == said) /* "not ready" (or: recorded inside a capture) is this poll's business only */
        (void)hipGetLastError();
    return said == hipSuccess;
}

/* With -DSPGPU_DEBUG every launch is followed by a synchronising error check
 * that prints and exits, as the reference does under -DDEBUG
 * (kernels/cudadebug.h:12-25).  Otherwise launch errors surface through the
 * caller's own hipGetLastError(), again as in the reference. */
void spgpuDebugCheck(spgpuHandle_t h, const char* what);

void spgpuNoteSpmvForm(spgpuHandle_t h, int form);

/* The feedback ints of the matrix identified by (key, rows): found or newly assigned (and zeroed); *calls = how many
 * SpMV calls have asked for this entry before. */
int* spgpuFormFeedback(spgpuHandle_t h, const void* key, int rows, int* calls, int* tag);
/* A report is (generation of the table entry << 8) | answer.  A probe queued for a matrix whose entry has meanwhile been
 * given to another matrix lands in that matrix' words with the OLD generation: read as "nothing said" (0). */
static inline int spgpuFeedbackSaid(int word, int tag)
{
    return (word & ~0xFF) == tag ? (word & 0xFF) : 0;
}

/* Environment knobs (include/spgpu/tuning.h), read once and cached: no getenv in a launch path. */
typedef struct SpgpuTuning {
    int spmvVariant; /* 0 */
    int ntLoads;     /* 1 */
    int tailLanes;   /* -1: kernel default */
    int hdiaVariant; /* 0 */
    int hdiaBlock;   /* 512 */
    int hdiaNarrow;  /* 0 */
    int xcdOrder;    /* 0 */
    int spmmVariant; /* 0 */
    int l1Blocks;    /* 0: kernel default */
    int xStrips;     /* -1: by feedback */
    int xTile;       /* -1: by the handle's hint */
    int autoSweep;   /* 1: AUTO may pick the SWEEP form (SPGPU_AUTO_SWEEP=0: never) */
    int poisonScratch; /* 0; SPGPU_POISON_SCRATCH=1 (testing): device scratch the library allocates and does not have to initialise -- the deep lists'
                        * sums, a plan's tables, the reduction scratch -- is filled with 0xFF bytes (NaN / -1) when it is allocated: a kernel that
                        * read such a word before writing it would show */
    int slide;       /* 0; lab builds: 1 = the moving x tile of slide_spmv.hip.h for 8-byte elements (experiment) */
    int xTileShape;  /* 0 */
    int deepSplit;   /* -1: when rIdx is given */
    int deepCap;     /* 256 */
    int deepKeep;    /* 64: columns of a sub-group deeper than deepCap that stay in the main kernel (-1 or >= deepCap: deepCap) */
    int ragged;      /* 1: the queue-driven kernel where the deep split is on */
    int raggedShape; /* 0 */
    int pipeGroups;  /* 0: one workgroup per CU (tests: fewer, so that small matrices run several blocks per workgroup) */
    int raggedSplit; /* -1: about 96 columns per chunk; 0: sub-groups are never cut; > 0: columns per chunk (rounded up to what LDS can park) */
    int l1Nt;        /* -1: by size */
    int plan;        /* 1: ordered matrices get a per-matrix plan (planned_spmv.hip); 0: never */
    int planDeepSpread; /* 60: the deep sub-groups' workgroups are spread over the first 60 % of the grid; 0: all in front; < 0: all behind */
    int planDeepPerBlock; /* 8: deep sub-groups per such workgroup (1 .. 8) */
    int freezeEscapesPct; /* 1: spgpu?SpmvFreeze keeps a 16-bit copy only if at most this many entries in a hundred are escapes (0xFFFF: the column is
                           * in rP after all) -- a matrix with scattered columns gains nothing from the copy and pays for every escape */
    int planDeepRuns;     /* 1: such a workgroup takes a run of consecutive sub-groups of the list; 0: every deepBlocks-th */
    int stageLate;   /* 1: the queue kernel stages its destinations under the tile's round trip (0: before the first requests, as round 3) */
} SpgpuTuning;
const SpgpuTuning* spgpuTuning(void);

/* Pinned words for the synchronous analysis calls (spgpuHellSpmvForm / spgpuEllSpmvForm): one analysis of a handle at a time. */
int* spgpuAnalyseWords(spgpuHandle_t h);
int* spgpuDeepOverflowWords(spgpuHandle_t h);

#ifdef __cplusplus
}
#endif
