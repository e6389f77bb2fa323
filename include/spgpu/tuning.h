#pragma once
/*
 * Tuning knobs of the gfx950 kernels (no counterpart in the reference, whose
 * only compile-time knobs are THREAD_BLOCK / MAX_NNZ_PER_WG in the *_base.cuh
 * files).  The library reads these environment variables ONCE, on the first
 * call that needs them -- never in a launch path -- and again whenever
 * spgpuTuningReload() is called (the A/B tools and the kernel-shape parity
 * tests change them between launches):
 *
 *   SPGPU_SPMV_VARIANT   ELL/HELL SpMV kernel shape (0 = default, see csrc/ellpack_spmv.hip)
 *   SPGPU_NT_LOADS       0: no non-temporal hint on the coefficient/index streams (default 1)
 *   SPGPU_X_STRIPS       ELL/HELL SpMV: x values of a strip of rows with one 16-byte load where the rows name consecutive
 *                        columns.  Unset: learnt per matrix from the kernel's own feedback (csrc/ellpack_spmv.hip);
 *                        1: the strip-capable kernel always; 0: the gather-only kernel always
 *   SPGPU_TAIL_LANES     busy lanes below which a wavefront switches to whole-wave rows (default 16)
 *   SPGPU_HDIA_VARIANT   2: 8 diagonals per stage instead of 4
 *   SPGPU_HDIA_BLOCK     HDIA workgroup size 256 / 512 (default) / 1024
 *   SPGPU_HDIA_NARROW    1: one row per lane even when 16-byte accesses are possible
 *   SPGPU_XCD_ORDER      HDIA: 0 hardware workgroup order (default), n: XCD-contiguous runs of n
 *   SPGPU_DEEP_CAP       ELL/HELL SpMV with a row order: a 32-row sub-group deeper than this many columns (default 256) hands
 *                        columns to the deep kernels (the deep list below); SPGPU_DEEP_SPLIT=1 / 0 forces that on without a
 *                        row order / off with one
 *   SPGPU_DEEP_KEEP      of a 32-row sub-group deeper than SPGPU_DEEP_CAP, the columns the main kernel walks itself (default 64;
 *                        -1 or >= the cap: all of the first SPGPU_DEEP_CAP).  The rest are items of 64 columns for the deep
 *                        kernels.  With 256 the workgroups that hold the set-aside long rows -- 64 sub-groups of 256 columns
 *                        each -- lived as long as the whole launch (profiles/r03_ragged_workgroup_trace.txt)
 *   SPGPU_RAGGED_SPLIT   ELL/HELL SpMV with a row order: columns per chunk of a 32-row sub-group that several wavefronts share
 *                        (unset: about 96; 0: never cut; rounded up to what LDS can park; csrc/ragged_spmv.hip.h, SPLIT)
 *   SPGPU_SPMM_VARIANT   SpMM kernel shape (0 = default, see csrc/hell_spmm.hip)
 *   SPGPU_L1_NT          Level-1 streams non-temporal: 1 always, 0 never, unset: vectors beyond the Infinity Cache
 *   SPGPU_L1_BLOCKS      grid cap of the Level-1 kernels (default 16384)
 *   SPGPU_POISON_SCRATCH 1 (testing): device scratch the library allocates without having to initialise it -- the deep lists'
 *                        sums, a plan's tables, the reduction scratch -- is filled with 0xFF bytes (NaN / -1) at allocation, so
 *                        that a kernel reading such a word before writing it shows in the results
 *
 * Every setting computes the same values up to the summation order documented
 * per kernel; the defaults are the measured best on MI355X.
 */
#include "core.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * The deep list.  Until a matrix has its plan (below), an ELL/HELL SpMV that is given a row order (rIdx) hands the columns beyond
 * SPGPU_DEEP_KEEP of its 32-row sub-groups deeper than SPGPU_DEEP_CAP to two small kernels through a list the handle owns (one per
 * stream, 8 192 sub-groups / 20 480 items of 64 columns).  A matrix with more such sub-groups than the list holds is multiplied
 * all the same and, since round 4, TO THE SAME BITS: a sub-group that finds the list full is worked off by its own workgroup,
 * behind that workgroup's stream, in the chunks and the order of the deep kernels (csrc/deep_rows.hip.h) -- which sub-groups
 * are surplus depends on scheduling, their sums do not.  Only speed suffers (those workgroups live longer).
 * spgpuDeepListOverflows says how many completed calls of the handle overflowed (0 for every matrix the tests and benches
 * use, whose long rows were set aside by spgpuOellOrderDevice; a matrix ordered with longRows = 0 can overflow on its first
 * calls; planned calls have no list).
 */
int spgpuDeepListOverflows(spgpuHandle_t handle);
/* The handle keeps a deep list for each of 8 streams.  A ninth stream takes over the list of the least recently used stream
 * whose calls have all finished (spgpuDeepListsRecycled counts those hand-overs); while none is idle, an ordered SpMV on a stream
 * without a list runs the same kernel family without state (csrc/planned_spmv.hip: the matrix' plan if it is ready, else no plan
 * at all) -- slower, the SAME bits -- and spgpuDeepListFallbacks counts those calls. */
int spgpuDeepListFallbacks(spgpuHandle_t handle);
int spgpuDeepListsRecycled(spgpuHandle_t handle);

/*
 * Plans.  An ELL/HELL SpMV with a row order analyses a matrix the first time it sees it (behind the call, on its stream) and
 * keeps what it learnt -- where the slice of x lies that each block of rows touches, which 32-row sub-groups are deeper
 * than SPGPU_DEEP_CAP -- under the addresses of the matrix' arrays (8 matrices per handle).  Later calls on the same arrays run
 * ONE launch with a shorter prologue: no probing of columns, no deep list, nothing behind the main kernel; the deep
 * sub-groups get workgroups of their own in the same grid.  The bits of z are the same with and without a plan, and with a
 * plan that has gone stale (another matrix at the same addresses): a plan decides who computes, never what or in which order.
 * A stale plan is noticed by the kernels and rebuilt by the next call.  Launches captured into a HIP graph never use a plan.
 *   SPGPU_PLAN=0                 no plans
 *   SPGPU_PLAN_DEEP_PER_BLOCK    deep sub-groups per workgroup of theirs (default 8, 1 .. 8)
 *   SPGPU_PLAN_DEEP_RUNS         1 (default): such a workgroup takes a RUN of consecutive deep sub-groups of the list -- after an
 *                                ordering these are neighbours in the matrix (one window of set-aside long rows), so their gathers
 *                                of x meet in one L2; 0: every N-th sub-group (the dealing of the first planned kernels)
 *   SPGPU_PLAN_DEEP_SPREAD       those workgroups are spread over the first N per cent of the grid (default 60; 0: all in front;
 *                                -1: all behind the blocks of rows)
 * spgpuSpmvPlanCounts: launches that ran with a plan, analyses started, plans found stale (any pointer may be NULL).
 */
void spgpuSpmvPlanCounts(spgpuHandle_t handle, int* uses, int* builds, int* stales);
/*
 * For a caller who does not want its first calls to differ from its later ones (a benchmark without warm-up calls, a graph
 * captured right after the matrix was built): everything a first SpMV on these arrays would leave to later calls -- the
 * ordered matrix' workgroup shape, its plan -- is worked out now, on the handle's current stream, and WAITED for.  Arguments as
 * in the SpMV call that will follow (cM is only looked at for its alignment).  SPGPU_SUCCESS: the next spgpu?hellspmv /
 * spgpu?ellspmv on these arrays (same rIdx, baseIndex, form) runs from the plan; SPGPU_UNSUPPORTED: calls of this kind have no
 * plan (no rIdx, SPGPU_PLAN=0, a layout that takes the narrow kernels) -- nothing to prepare, nothing wrong.
 */
int spgpuHellSpmvPrepare(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                         const int* rIdx, int rows, int baseIndex);
int spgpuEllSpmvPrepare(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, const int* rIdx,
                        int maxNnzPerRow, int rows, int baseIndex);

/*
 * FROZEN matrices (no counterpart in the reference).  An iterative solver multiplies by one matrix thousands of times
 * (hellPerf.cpp:301-313 is that loop); its index arrays never change in between.  A caller who can PROMISE that -- rP, rS,
 * hackOffsets and rIdx of this matrix stay byte for byte as they are until spgpuSpmvThaw (cM may change at any time: the
 * coefficients are always read from the caller's array) -- lets the library keep what it derives from them: spgpu?SpmvFreeze
 * stores the column indices of the matrix as 16-bit offsets (2 bytes per slot of rP; 0xFFFF where a column lies out of reach:
 * such entries are still read from rP):
 *   with a row order (rIdx != NULL): what spgpu?SpmvPrepare does, then the copy with the matrix' plan, counted from the first
 *     column each block of 1 024 / 2 048 ordered rows reaches;
 *   without one (rIdx == NULL; the default kernels, BASELINE configs[1]): counted from the lowest column of every group of rows
 *     one wavefront owns (128 rows for the 8-byte types, 32 for fp32); a matrix in which more than one entry in a hundred would
 *     be out of reach -- scattered columns -- is NOT frozen (its SpMV is bound by the gathers, not by the index stream).
 * The same rule holds with a row order: more than one escape in a hundred entries and the matrix keeps its plan but gets no copy
 * (SPGPU_UNSUPPORTED; SPGPU_FREEZE_MAX_ESCAPES_PCT, default 1, is the knob -- 100 freezes anything).
 * Later spgpu?hellspmv / spgpu?ellspmv calls on these arrays -- the same ABI calls, nothing else changes for the caller -- stream
 * 2 bytes of index per stored entry instead of 4: 10 instead of 12 bytes per nonzero in fp64, 6 instead of 8 in fp32.  Same
 * columns, same x, same order of additions: the bits of z are those of the unfrozen call.
 *   SPGPU_SUCCESS      frozen (or already so);
 *   SPGPU_UNSUPPORTED  calls of this kind have no packed form (complex fp64; arrays not aligned for 16-byte slab loads; the
 *                      ordered gather form; SPGPU_PLAN=0; scattered columns without a row order) or there was no memory for
 *                      the copy: nothing is frozen, nothing wrong.  (The LDS-tile form of the default kernels has no packed
 *                      variant either: a frozen matrix AUTO runs in that form runs as before.)
 * Breaking the promise is undefined behaviour in the usual sense: wrong results for the entries that changed, and reads of x at
 * columns the copy never named if rows grew (the library cannot see it: checking would mean reading rP, which is what the copy
 * saves; an ordered matrix whose row LENGTHS changed is noticed like any stale plan -- by the call that has already used the copy).
 * Launches captured into a HIP graph never use a frozen copy (a graph outlives a Thaw): they run as unfrozen calls.
 * A frozen plan ends with spgpuSpmvThaw(handle, rP), when it is the least recently used of 8 matrices, or with the handle.
 * spgpuSpmvFrozenBytes: device memory the handle's frozen plans hold.
 */
int spgpuHellSpmvFreeze(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                        const int* rIdx, int rows, int baseIndex);
int spgpuEllSpmvFreeze(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS, const int* rIdx,
                       int maxNnzPerRow, int rows, int baseIndex);
int spgpuSpmvThaw(spgpuHandle_t handle, const int* rP);
long long spgpuSpmvFrozenBytes(spgpuHandle_t handle);

/*
 * ADOPTED matrices (no counterpart in the reference; csrc/adopted_hell.hip).  A HELL matrix with very unequal row lengths stores
 * several slots per nonzero when its rows come as they are (4.99 on the north_star target), and any kernel that takes it as it
 * is pays for them; the reference's remedy is the caller's -- order the rows by length and pass the permutation as rIdx
 * (ellToOell, hellPerf.cpp:324-378).  spgpuHellSpmvAdopt does that FOR a caller who promises more than Freeze asks: NONE of the
 * matrix' arrays -- cM, rP, rS, hackOffsets -- changes until spgpuSpmvThaw(handle, rP).  The library then keeps its own copy of
 * the matrix with the rows in spgpuOellOrderAlignedDevice's order (windows of 2 048, rows longer than 256 set aside), frozen;
 * spgpu?hellspmv calls on the caller's arrays with rIdx == NULL run on that copy and write z through its row order: z[i] for
 * the caller's row i, as ever, the value the ordered kernel computes (equal to the plain kernel's within rounding; bit for bit
 * what the caller would get by ordering the matrix himself with the same device calls).  Cost: device memory for the ordered
 * matrix (spgpuSpmvFrozenBytes counts it) and ~17 ms once for the 10 M-row target.  SPGPU_UNSUPPORTED: hackSize not a multiple of 32, no memory, four
 * matrices adopted already -- or a matrix that stores less than 1.25 x the slots its ordered copy would (rows about equally long:
 * nothing to gain; spgpuHellSpmvFreeze is the call for such a matrix).  Launches captured into a HIP graph run on the caller's arrays.
 * spgpuSpmvAdoptedUses: calls that ran on a copy.  (Adopt, Freeze and Thaw synchronise the handle's stream: not inside a capture.)
 */
int spgpuHellSpmvAdopt(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                       int rows, int baseIndex);
/* The ELL flavour (same promise, same Thaw): spgpu?ellspmv calls with rIdx == NULL and these arrays run on an ordered HELL copy (hack
 * size 32) -- rows x maxNnzPerRow slots in the caller's layout, ~1.1 per nonzero in the copy.  Needs rS. */
int spgpuEllSpmvAdopt(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int cMPitch, int rPPitch, const int* rS,
                      int maxNnzPerRow, int rows, int baseIndex);
int spgpuSpmvAdoptedUses(spgpuHandle_t handle);
/* One call for a solver (Adopt's promise: no array of the matrix changes until spgpuSpmvThaw): Adopt if the matrix comes without a row
 * order and is ragged, else Freeze if its columns allow a 16-bit copy, else nothing.  Returns what the later calls will run on. */
#define SPGPU_SPMV_AS_IS 0
#define SPGPU_SPMV_FROZEN 1
#define SPGPU_SPMV_ADOPTED 2
int spgpuHellSpmvOptimize(spgpuHandle_t handle, spgpuType_t type, const void* cM, const int* rP, int hackSize, const int* hackOffsets, const int* rS,
                          const int* rIdx, int rows, int baseIndex);

void spgpuTuningReload(void);
/* 1 if the library was built with -DSPGPU_TUNING_VARIANTS: the non-default kernel shapes (SPGPU_SPMV_VARIANT, SPGPU_X_TILE_SHAPE,
 * SPGPU_RAGGED_SHAPE, SPGPU_RAGGED=0) exist only in such a build; the product build carries the defaults and ignores those knobs. */
int spgpuTuningVariantsBuilt(void);

/*
 * Per-handle hint: how the ELL/HELL SpMV kernels fetch x (no counterpart in the reference, whose only per-call hint
 * is avgNnzPerRow, hell.h:45-59).  Every form computes the same values; AUTO, GATHER, STRIPS and XTILE also the same
 * bits (the order in which a row's products are added does not depend on the form).
 *
 *   AUTO    the library decides per matrix: sample wavefronts report what they saw (consecutive columns in neighbouring
 *           rows / columns inside a window that fits an LDS tile / scattered) and the next launch on the same arrays
 *           uses it.  A first call runs the strip-capable kernel, whose own wavefronts report; the other forms are
 *           looked at again by a three-wavefront probe with their first and then every fourth call (another matrix may
 *           have come to live at the address).  spgpuHellSpmvForm / spgpuEllSpmvForm below give the same answer at once, to keep.
 *   GATHER  one global load per nonzero.
 *   STRIPS  the x values of a lane's consecutive rows with ONE 16-byte load where those rows name consecutive
 *           columns (stencil and band matrices in natural order); falls back to gathers stage by stage.
 *   XTILE   a workgroup copies the slice of x its rows touch into LDS once, coalesced, and gathers from LDS
 *           (columns near the diagonal but not consecutive; length-sorted rows used through rIdx); entries outside
 *           the tile are gathered from global memory.
 *
 *   SWEEP   for matrices whose columns are scattered over all of x but ascend inside a row: a lane carries 32 rows (16 for
 *           complex fp64) through the slab columns in step, so that at any moment the rows in flight gather from the same
 *           quantile of x and meet in L2 (10 M x 32 scattered, fp64: 4.66 ms against 6.0).  Without rIdx only (with rIdx: as
 *           AUTO).  On matrices with locality between neighbouring rows, or with rows of very unequal length (a lane walks
 *           its 32 rows to the longest of them), this form is several times SLOWER than the others.
 *           Order of additions: for the 8-byte types (fp64, complex fp32) the bits of AUTO / GATHER / STRIPS / XTILE -- ascending
 *           k, and the last rows of a 128-row group finished by the whole wavefront exactly where their default kernel does it;
 *           for fp32 and complex fp64 a row's products in ascending k, the reference's one-thread-per-row order
 *           (hell_spmv_base_template.cuh:104-215), which is NOT the bit pattern of their other forms.
 *           AUTO takes this form by itself for the 8-byte types (since round 4) when the three-wavefront probe finds, in two of
 *           its three groups of rows: columns reaching over half the matrix' rows and more (the matrix is taken to be about
 *           square: no call says how long x is), ascending in every sampled row, rows about equally long (slots of the group
 *           <= 1.5 x its nonzeros) -- and the matrix has 2 Mi rows or more (the form wants a grid that fills the chip).
 *           SPGPU_AUTO_SWEEP=0 keeps AUTO out of it.
 *
 * The hint applies to every later SpMV call on the handle, from any thread; SPGPU_X_STRIPS / SPGPU_X_TILE in the
 * environment override it.
 */
#define SPGPU_SPMV_FORM_AUTO   0
#define SPGPU_SPMV_FORM_GATHER 1
#define SPGPU_SPMV_FORM_STRIPS 2
#define SPGPU_SPMV_FORM_XTILE  3
#define SPGPU_SPMV_FORM_SWEEP  4
void spgpuSetSpmvForm(spgpuHandle_t handle, int form);
int spgpuGetSpmvForm(spgpuHandle_t handle);
/* Diagnostic: the form (GATHER / STRIPS / XTILE / SWEEP) the most recent ELL/HELL SpMV call on this handle was launched in. */
int spgpuGetLastSpmvForm(spgpuHandle_t handle);

/*
 * The answer AUTO would settle on for ONE matrix, for a caller who wants to hold it instead of leaving it to AUTO's table
 * of the eight most recent (rP, rows) pairs -- which learns a call late, and a few calls late when another matrix comes
 * to live at the address of an old one:
 *
 *     form = spgpuHellSpmvForm(handle, SPGPU_TYPE_DOUBLE, rP, hackSize, hackOffsets, rS, rows, baseIndex);   // once per matrix
 *     spgpuSetSpmvForm(handle, form);  spgpuDhellspmv(handle, ...);                                           // every call
 *
 * Three wavefronts look at the column indices of three groups of rows (near the quarter points of the matrix) on the
 * handle's current stream; the call WAITS for them (it synchronises that stream) and returns SPGPU_SPMV_FORM_STRIPS,
 * _XTILE, _GATHER or -- where AUTO would take it -- _SWEEP (AUTO if the device call failed).  One analysis of a handle at a time.  rP / hackOffsets / rS: the
 * device arrays of the SpMV call; rS may be NULL for ELL (every row maxNnzPerRow long).
 */
int spgpuHellSpmvForm(spgpuHandle_t handle, spgpuType_t type, const int* rP, int hackSize, const int* hackOffsets, const int* rS, int rows,
                      int baseIndex);
int spgpuEllSpmvForm(spgpuHandle_t handle, spgpuType_t type, const int* rP, int rPPitch, const int* rS, int maxNnzPerRow, int rows,
                     int baseIndex);

#ifdef __cplusplus
}
#endif
