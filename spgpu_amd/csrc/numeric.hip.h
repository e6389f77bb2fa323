#pragma once
#include <type_traits>
/*
 * Device-side arithmetic shared by every kernel: the four value types of the
 * ABI (float, double, float complex, double complex), their multiply-add in
 * the operation order the reference uses, vector-width global loads and
 * wavefront (64-lane) shuffles.
 *
 * Arithmetic model (reference: kernels/hell_spmv_base.cuh:29-51):
 *   real    fma(a,b,c) is written (a*b)+c there and contracts to one fused
 *           multiply-add under nvcc's default -fmad; here it IS one fma.
 *   complex fma is cuCfma / cuCfmaf; the same expression tree as
 *           hipCfma (amd_hip_complex.h), contracted.
 * oracle/spgpu_oracle.c uses the identical expression trees with C fma(), so
 * the kernels and the oracle agree to the last bit whenever they also agree
 * on the order in which a row's products are added.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spgpu {

constexpr int kWave = 64;

template <typename R> struct Cx { R x, y; };
using cfloat = Cx<float>;
using cdouble = Cx<double>;

/* ---- zero / tests ------------------------------------------------------- */
template <typename T> __device__ __host__ inline T zeroOf();
template <> __device__ __host__ inline float zeroOf<float>() { return 0.0f; }
template <> __device__ __host__ inline double zeroOf<double>() { return 0.0; }
template <> __device__ __host__ inline cfloat zeroOf<cfloat>() { return cfloat{0.0f, 0.0f}; }
template <> __device__ __host__ inline cdouble zeroOf<cdouble>() { return cdouble{0.0, 0.0}; }

__device__ __host__ inline bool isNotZero(float a) { return a != 0.0f; }
__device__ __host__ inline bool isNotZero(double a) { return a != 0.0; }
template <typename R> __device__ __host__ inline bool isNotZero(Cx<R> a) { return a.x != R(0) || a.y != R(0); }

/* ---- real --------------------------------------------------------------- */
__device__ inline float mulAdd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double mulAdd(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float mul(float a, float b) { return a * b; }
__device__ inline double mul(double a, double b) { return a * b; }
__device__ __host__ inline float add(float a, float b) { return a + b; }
__device__ __host__ inline double add(double a, double b) { return a + b; }

/* ---- complex: p*q + r with the cuCfma expression tree --------------------
 *   re = (p.x*q.x + r.x) - p.y*q.y ;  im = (q.x*p.y + r.y) + p.x*q.y      */
template <typename R> __device__ inline Cx<R> mulAdd(Cx<R> p, Cx<R> q, Cx<R> r)
{
    R re = mulAdd(p.x, q.x, r.x);
    R im = mulAdd(q.x, p.y, r.y);
    re = mulAdd(-p.y, q.y, re);
    im = mulAdd(p.x, q.y, im);
    return Cx<R>{re, im};
}
/* cuCmul: re = a.x*b.x - a.y*b.y ; im = a.x*b.y + a.y*b.x (second product fused in). */
template <typename R> __device__ inline Cx<R> mul(Cx<R> a, Cx<R> b)
{
    return Cx<R>{mulAdd(a.x, b.x, -(a.y * b.y)), mulAdd(a.x, b.y, a.y * b.x)};
}
template <typename R> __device__ __host__ inline Cx<R> add(Cx<R> a, Cx<R> b) { return Cx<R>{a.x + b.x, a.y + b.y}; }

/* ---- branch-free select.  Written per component for complex values: a ?: on the
 * whole struct makes hipcc keep the running sums in scratch memory. */
__device__ inline float pick(bool c, float a, float b) { return c ? a : b; }
__device__ inline double pick(bool c, double a, double b) { return c ? a : b; }
template <typename R> __device__ inline Cx<R> pick(bool c, Cx<R> a, Cx<R> b)
{
    return Cx<R>{c ? a.x : b.x, c ? a.y : b.y};
}

/* ---- SpMV epilogue (reference: hell_spmv_base_template.cuh:219-222) ------ */
template <bool HAS_BETA, typename T> __device__ inline T epilogue(T alpha, T rowSum, T beta, T yVal)
{
    if constexpr (HAS_BETA)
        return mulAdd(beta, yVal, mul(alpha, rowSum));
    else
        return mul(alpha, rowSum);
}

/* ---- N consecutive elements moved by ONE global load/store -------------- */
template <typename E, int N> struct alignas(sizeof(E) * N) Pack { E v[N]; };

template <int BYTES> struct RawBits;
template <> struct RawBits<4> { using type = uint32_t; };
template <> struct RawBits<8> { using type = uint32_t __attribute__((ext_vector_type(2))); };
template <> struct RawBits<16> { using type = uint32_t __attribute__((ext_vector_type(4))); };

/* NT = streamed once: non-temporal hint keeps the coefficient/index streams
 * from displacing x in the L2 / Infinity Cache. */
template <bool NT, typename E, int N> __device__ inline Pack<E, N> loadPack(const E* p)
{
    using Raw = typename RawBits<sizeof(E) * N>::type;
    Raw raw;
    if constexpr (NT)
        raw = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(p));
    else
        raw = *reinterpret_cast<const Raw*>(p);
    Pack<E, N> out;
    __builtin_memcpy(&out, &raw, sizeof(out));
    return out;
}

/* N consecutive elements at an address that is only element-aligned: still ONE load instruction (global memory
 * takes 16-byte accesses at any dword address), the compiler just may not assume more than alignof(E). */
template <typename E, int N> __device__ inline Pack<E, N> loadPackElementAligned(const E* p)
{
    using Raw = typename RawBits<sizeof(E) * N>::type;
    typedef Raw LooseRaw __attribute__((aligned(alignof(E)))); /* the vector type, minus its alignment promise */
    const Raw raw = *reinterpret_cast<const LooseRaw*>(p);
    Pack<E, N> out;
    __builtin_memcpy(&out, &raw, sizeof(out));
    return out;
}

template <typename E, int N> __device__ inline void storePack(E* p, const Pack<E, N>& value)
{
    using Raw = typename RawBits<sizeof(E) * N>::type;
    Raw raw;
    __builtin_memcpy(&raw, &value, sizeof(raw));
    *reinterpret_cast<Raw*>(p) = raw;
}

/* The store counterpart of loadPackElementAligned. */
template <typename E, int N> __device__ inline void storePackElementAligned(E* p, const Pack<E, N>& value)
{
    using Raw = typename RawBits<sizeof(E) * N>::type;
    typedef Raw LooseRaw __attribute__((aligned(alignof(E))));
    Raw raw;
    __builtin_memcpy(&raw, &value, sizeof(raw));
    *reinterpret_cast<LooseRaw*>(p) = raw;
}

template <bool NT, typename E, int N> __device__ inline void storePackMaybeNT(E* p, const Pack<E, N>& value)
{
    using Raw = typename RawBits<sizeof(E) * N>::type;
    Raw raw;
    __builtin_memcpy(&raw, &value, sizeof(raw));
    if constexpr (NT)
        __builtin_nontemporal_store(raw, reinterpret_cast<Raw*>(p));
    else
        *reinterpret_cast<Raw*>(p) = raw;
}

/* ---- XCD-aware workgroup order ------------------------------------------------
 * MI355X deals consecutive workgroup ids round-robin over its 8 XCDs (observed dispatch
 * behaviour, MI355X_MICROARCH.md; used for speed only, never for correctness), and every
 * XCD has a private 4 MiB L2.  This bijection of [0, groups) hands XCD j the j-th
 * CONTIGUOUS eighth of the work, so that neighbouring row blocks -- which read
 * neighbouring pieces of x -- share one L2 instead of each pulling the same lines
 * over the fabric. */
constexpr unsigned kXcds = 8;
__device__ inline unsigned xcdContiguous(unsigned id, unsigned groups)
{
    const unsigned q = groups / kXcds, r = groups % kXcds;
    const unsigned xcd = id % kXcds, slot = id / kXcds;
    return xcd * q + (xcd < r ? xcd : r) + slot;
}
/* Finer grain: runs of `run` consecutive work items per XCD (run = 1 is the hardware's own
 * order).  Bijective on the first floor(groups / (8*run)) * 8*run ids, identity on the tail. */
__device__ inline unsigned xcdRuns(unsigned id, unsigned groups, unsigned run)
{
    const unsigned span = kXcds * run;
    if (id >= groups / span * span)
        return id;
    const unsigned xcd = id % kXcds, slot = id / kXcds;
    return (slot / run * kXcds + xcd) * run + slot % run;
}

/* ---- wavefront shuffles (lower to DPP / ds_bpermute, no LDS allocation) -- */
__device__ inline float laneXor(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ inline double laneXor(double v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ inline int laneXor(int v, int mask) { return __shfl_xor(v, mask, kWave); }
template <typename R> __device__ inline Cx<R> laneXor(Cx<R> v, int mask)
{
    return Cx<R>{laneXor(v.x, mask), laneXor(v.y, mask)};
}

__device__ inline int waveMax(int v)
{
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        const int other = laneXor(v, m);
        v = other > v ? other : v;
    }
    return v;
}

__device__ inline int waveMin(int v)
{
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        const int other = laneXor(v, m);
        v = other < v ? other : v;
    }
    return v;
}

/* Cross-lane reductions of the prologue as DPP moves and readlanes (VALU / SALU only).  __shfl lowers to ds_bpermute, an LDS-queue
 * instruction: harmless in an idle CU, but the workgroup beside this one is streaming through its LDS tile at that moment and
 * every one of the ~70 dependent shuffles of the prologue waited its turn behind that traffic -- the prologue's compute-only
 * stretches measured 2-5 us each (profiles/r03_ragged_workgroup_trace.txt). */
template <int CTRL, int ROW_MASK> __device__ inline int dppMove(int keep, int v)
{
    return __builtin_amdgcn_update_dpp(keep, v, CTRL, ROW_MASK, 0xF, false);
}
struct MaxOf { static constexpr int identity = -0x7fffffff - 1; __device__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct MinOf { static constexpr int identity = 0x7fffffff; __device__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct SumOf { static constexpr int identity = 0; __device__ int operator()(int a, int b) const { return a + b; } };
/* over the 16 lanes of each DPP row, result in every lane: neighbours, pairs, then the two mirrors */
template <typename Op> __device__ inline int rowReduce(int v, Op op)
{
    v = op(v, dppMove<0xB1, 0xF>(v, v));  /* quad_perm [1,0,3,2] */
    v = op(v, dppMove<0x4E, 0xF>(v, v));  /* quad_perm [2,3,0,1] */
    v = op(v, dppMove<0x141, 0xF>(v, v)); /* row_half_mirror */
    v = op(v, dppMove<0x140, 0xF>(v, v)); /* row_mirror */
    return v;
}
/* over each 32-lane half of the wavefront, result in every lane of the half */
template <typename Op> __device__ inline int halfReduce(int v, Op op)
{
    v = rowReduce(v, op);
    v = op(v, dppMove<0x142, 0xA>(Op::identity, v)); /* row_bcast:15 into rows 1 and 3 */
    const int low = __builtin_amdgcn_readlane(v, 31), high = __builtin_amdgcn_readlane(v, 63);
    return (threadIdx.x & 32) ? high : low;
}
/* over the wavefront, as a scalar */
template <typename Op> __device__ inline int waveReduce(int v, Op op)
{
    v = rowReduce(v, op);
    v = op(v, dppMove<0x142, 0xA>(Op::identity, v)); /* row_bcast:15 into rows 1 and 3 */
    v = op(v, dppMove<0x143, 0xC>(Op::identity, v)); /* row_bcast:31 into rows 2 and 3 */
    return __builtin_amdgcn_readlane(v, 63);
}
/* sum of exact integers kept as doubles (row sums of column numbers: below 2^53), over the wavefront, as lane 63's value */
__device__ inline double waveSumExact(double v)
{
    auto moved = [](double x, auto ctrl, auto rowMask) {
        constexpr int CTRL = decltype(ctrl)::value, MASK = decltype(rowMask)::value;
        const int lo = dppMove<CTRL, MASK>(0, __double2loint(x)), hi = dppMove<CTRL, MASK>(0, __double2hiint(x));
        return __hiloint2double(hi, lo); /* lanes outside the row mask receive +0.0 */
    };
    v += moved(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});
    v += moved(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});
    v += moved(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});
    v += moved(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});
    v += moved(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
    v += moved(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
/* set bits of a wave-wide mask below this lane */
__device__ inline int bitsBelowLane(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

} // namespace spgpu
