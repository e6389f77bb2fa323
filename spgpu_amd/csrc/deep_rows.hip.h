/*
 * Whole 32-row sub-groups worked off by ONE workgroup, chunk by chunk -- the deep sub-groups of a planned ELL / HELL SpMV
 * (raggedSpmvKernel<..., PLAN>, planned_spmv.hip).  Included by ragged_spmv.hip.h (namespace spgpu).
 *
 * After an ordering by length (ellToOell, reference ell.c:85-202; spgpuOellOrder*Device) the long rows sit together: whole
 * hacks are hundreds or thousands of columns deep, far more than a wavefront of the queue kernel should own.  Without a
 * plan such a sub-group registers in the stream's deep list and two launches behind the main kernel finish it
 * (deepItemsKernel / deepFinishKernel, ellpack_spmv.hip).  With a plan the sub-groups are known before the launch, a few
 * of them make up a workgroup of the same grid, and that workgroup does everything: its wavefronts take the sub-groups'
 * chunks round-robin, a chunk's sum waits in LDS, a half-wave per sub-group adds the chunk sums in chunk order, applies
 * the epilogue and stores z through rIdx.  No list, no scratch in global memory, no launch behind.
 *
 * The chunks of a sub-group and the order in which they are added are the ones the list path uses -- chunksOf below is the
 * one definition, a function of the sub-group's TRUE depth at the call (orc_?spmv_split restates it):
 *   depth <= deepCap:  the walked part [0, depth) in pieces of `split` columns if it is longer than that, else whole
 *   depth >  deepCap:  [0, deepKeep) likewise, then pieces of deepChunk (64) columns from deepKeep on
 * A chunk's PH phase sums (entries k = first + p, + PH, ... ascending) are combined pairwise (own + partner: the lane-xor
 * tree), the chunk sums added first to last.  So a stale plan, or a sub-group the plan does not know, changes who computes
 * and never a bit of the result.
 */

constexpr int kPlanDeepMost = 8; /* sub-groups one batch holds (their tables live in LDS) */

struct ChunkPlan {
    int mainChunks; /* pieces of the walked part */
    int items;      /* pieces of deepChunk columns behind it */
    int walked;     /* columns of the walked part */
};
__device__ inline ChunkPlan chunksOf(int depth, int deepCap, int deepKeep, int split, int deepChunk)
{
    const bool deep = depth > deepCap;
    const int walked = deep ? deepKeep : depth;
    const int mainChunks = (split > 0 && walked > split) ? (walked + split - 1) / split : 1;
    const int items = deep ? (depth - deepKeep + deepChunk - 1) / deepChunk : 0;
    return ChunkPlan{mainChunks, items, walked};
}
__device__ inline void chunkColumns(const ChunkPlan& cp, int chunk, int depth, int deepKeep, int split, int deepChunk, int& k0, int& k1)
{
    if (chunk < cp.mainChunks) {
        k0 = cp.mainChunks > 1 ? chunk * split : 0;
        k1 = (cp.mainChunks > 1 && k0 + split < cp.walked) ? k0 + split : cp.walked;
    } else {
        k0 = deepKeep + (chunk - cp.mainChunks) * deepChunk;
        k1 = k0 + deepChunk < depth ? k0 + deepChunk : depth;
    }
}

/*
 * count (<= kPlanDeepMost, workgroup-uniform) sub-groups, sub-group j being number subOf(j); every wavefront of the
 * workgroup calls this together.  `lds` is LDS_BYTES of shared memory nobody else uses during the call (the x tile of the
 * kernel: a workgroup of deep sub-groups has no use for one).  expectDeep: the plan sent these sub-groups here because they
 * were deeper than deepCap -- one that is not says so in the plan's pinned word.
 */
template <typename T, int RPL, bool IS_HELL, int WAVES, int LDS_BYTES, typename SubOf>
__device__ inline void wholeSubgroups(const SlabArgs<T>& a, void* lds, int count, SubOf subOf, bool expectDeep)
{
    constexpr int LPC = 32 / RPL;
    constexpr int PH = kWave / LPC;
    constexpr int UD = 16 / PH >= 1 ? 16 / PH : 1; /* 16 slab columns per stage */
    constexpr int STEP = PH * UD;
    constexpr int MOST = kPlanDeepMost;
    static_assert(WAVES * kWave >= MOST * 32, "a lane per row of the batch");
    /* the batch's tables, then the chunk sums */
    constexpr int TOTALS_BYTES = MOST * 32 * (int)sizeof(T);
    constexpr int TABLE_BYTES = TOTALS_BYTES + MOST * 32 * 4 /* lengths */ + MOST * 32 * 4 /* slab bases */ + MOST * 4 * 2 /* first row, depth */;
    constexpr int PARKS = (LDS_BYTES - TABLE_BYTES) / (32 * (int)sizeof(T));
    static_assert(PARKS >= 2 * WAVES, "room for a round of chunk sums");
    T* const totals = reinterpret_cast<T*>(lds);
    int* const lens = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + TOTALS_BYTES);
    unsigned* const bases = reinterpret_cast<unsigned*>(lens + MOST * 32);
    int* const firstRow = reinterpret_cast<int*>(bases + MOST * 32);
    int* const depths = firstRow + MOST;
    T* const parks = reinterpret_cast<T*>(reinterpret_cast<char*>(lds) + (TABLE_BYTES + 15) / 16 * 16);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane % LPC, phase = lane / LPC;
    const T* __restrict__ x = a.x;

    __syncthreads(); /* whoever used this memory before is done */
    { /* one round trip: the rows' lengths and slab bases, as the matrix is now */
        const int i = threadIdx.x;
        int len = 0, row0 = 0;
        unsigned base = 0;
        if (i < count * 32) {
            row0 = subOf(i >> 5) * 32;
            const long long r = (long long)row0 + (i & 31);
            if (r < a.rows) {
                len = a.rS ? a.rS[r] : a.maxNnz;
                if constexpr (IS_HELL) {
                    const unsigned u = (unsigned)r, hs = (unsigned)a.hackSize;
                    base = (unsigned)a.hackOffsets[u / hs] + u % hs;
                } else {
                    base = (unsigned)r;
                }
            }
        }
        const int depth = halfReduce(len, MaxOf{});
        if (i < count * 32) {
            lens[i] = len;
            bases[i] = base;
            if ((i & 31) == 0) {
                firstRow[i >> 5] = row0;
                depths[i >> 5] = depth;
                if (expectDeep && depth <= a.deepCap && a.planFlags)
                    a.planFlags[1] = 1; /* the plan is of another matrix: correct all the same, the host builds a new one */
            }
        }
    }
    __syncthreads();

    /* where each sub-group's chunks start in the batch's list (every lane works the few numbers out for itself) */
    int firstChunk[MOST + 1];
    firstChunk[0] = 0;
#pragma unroll
    for (int j = 0; j < MOST; ++j) {
        int n = 0;
        if (j < count) {
            const ChunkPlan cp = chunksOf(depths[j], a.deepCap, a.deepKeep, a.split, a.deepChunk);
            n = cp.mainChunks + cp.items;
        }
        firstChunk[j + 1] = firstChunk[j] + n;
    }
    const int chunks = firstChunk[MOST];

    struct Stage {
        Pack<T, RPL> v[UD];
        Pack<int, RPL> c[UD];
    };
    struct Slot {
        Stage st;
        int len[RPL];
        int kBase;
        int park;   /* where the chunk's sum goes (this round) */
        bool last;  /* last stage of its chunk */
        bool valid;
    };
    const bool hasBeta = isNotZero(a.beta);

    for (int roundStart = 0; roundStart < chunks; roundStart += PARKS) {
        const int roundEnd = roundStart + PARKS < chunks ? roundStart + PARKS : chunks;
        /* fetch cursor: the chunk being requested */
        int f = roundStart + wave;
        long long slab = 0;
        int fLen[RPL], fLongest = 0, fk = 0, fEnd = 0;
        auto openChunk = [&]() {
            if (f >= roundEnd)
                return;
            int j = 0;
#pragma unroll
            for (int q = 1; q < MOST; ++q)
                j += f >= firstChunk[q] ? 1 : 0;
            int first = 0;
#pragma unroll
            for (int q = 0; q < MOST; ++q)
                first = q == j ? firstChunk[q] : first;
            const int depth = depths[j];
            const ChunkPlan cp = chunksOf(depth, a.deepCap, a.deepKeep, a.split, a.deepChunk);
            chunkColumns(cp, f - first, depth, a.deepKeep, a.split, a.deepChunk, fk, fEnd);
            int strip = sub;
            asm volatile("" : "+v"(strip)); /* (as in raggedSpmvKernel: keeps the LDS addresses out of the loop's registers) */
            slab = (long long)bases[j * 32 + strip * RPL];
            fLongest = 0;
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                const int l = lens[j * 32 + strip * RPL + t];
                fLen[t] = l < fEnd ? l : fEnd;
                fLongest = fLen[t] > fLongest ? fLen[t] : fLongest;
            }
        };
        auto fetchNext = [&](Slot& slot) {
            slot.valid = f < roundEnd;
            slot.last = false;
            if (!slot.valid)
                return;
#pragma unroll
            for (int t = 0; t < RPL; ++t)
                slot.len[t] = fLen[t];
            slot.kBase = fk;
            slot.park = f - roundStart;
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const int k = fk + u * PH + phase;
                if (k < fLongest) {
                    slot.st.v[u] = loadPack<true, T, RPL>(a.cM + slab + (long long)k * a.valStride);
                    slot.st.c[u] = loadPack<true, int, RPL>(a.rP + slab + (long long)k * a.idxStride);
                } else {
#pragma unroll
                    for (int t = 0; t < RPL; ++t) {
                        slot.st.v[u].v[t] = zeroOf<T>();
                        slot.st.c[u].v[t] = a.baseIndex;
                    }
                }
            }
            fk += STEP;
            slot.last = fk >= fEnd;
            if (slot.last) { /* wavefront-uniform */
                f += WAVES;
                openChunk();
            }
        };
        T sum[RPL];
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            sum[t] = zeroOf<T>();
        auto consume = [&](Slot& cur, Slot& refill) {
            T xv[UD][RPL];
            bool use[UD][RPL];
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const int k = cur.kBase + u * PH + phase;
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int col = cur.st.c[u].v[t] - a.baseIndex;
                    use[u][t] = k < cur.len[t] && col >= 0;
                    xv[u][t] = x[use[u][t] ? col : 0];
                }
            }
            const int park = cur.park;
            const bool last = cur.last;
            fetchNext(refill); /* behind the gathers in issue order: waiting for them leaves it in flight */
#pragma unroll
            for (int u = 0; u < UD; ++u) {
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = pick(use[u][t], mulAdd(cur.st.v[u].v[t], xv[u][t], sum[t]), sum[t]);
            }
            if (last) {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    if constexpr (LPC <= 8)
                        sum[t] = add(sum[t], partnerOf<8>(sum[t]));
                    if constexpr (LPC <= 16)
                        sum[t] = add(sum[t], partnerOf<16>(sum[t]));
                    sum[t] = add(sum[t], partnerOf<32>(sum[t]));
                }
                if (phase == 0) {
#pragma unroll
                    for (int t = 0; t < RPL; ++t)
                        parks[park * 32 + sub * RPL + t] = sum[t];
                }
#pragma unroll
                for (int t = 0; t < RPL; ++t)
                    sum[t] = zeroOf<T>();
            }
        };
        Slot ring[2];
        openChunk();
        fetchNext(ring[0]);
        for (;;) {
            if (!ring[0].valid) break;
            consume(ring[0], ring[1]);
            if (!ring[1].valid) break;
            consume(ring[1], ring[0]);
        }
        __syncthreads(); /* the round's chunk sums are in LDS */
        for (int j = wave * 2 + (lane >> 5); j < count; j += WAVES * 2) { /* a half-wave per sub-group: its chunks of this round, in order */
            int first = 0, next = 0;
#pragma unroll
            for (int q = 0; q < MOST; ++q) {
                first = q == j ? firstChunk[q] : first;
                next = q == j ? firstChunk[q + 1] : next;
            }
            const int from = first > roundStart ? first : roundStart, to = next < roundEnd ? next : roundEnd;
            if (from < to) {
                const int row = lane & 31;
                T total = from == first ? parks[(from - roundStart) * 32 + row] : add(totals[j * 32 + row], parks[(from - roundStart) * 32 + row]);
                for (int c = from + 1; c < to; ++c)
                    total = add(total, parks[(c - roundStart) * 32 + row]);
                totals[j * 32 + row] = total;
            }
        }
        __syncthreads();
    }
    { /* epilogue and store, a lane per row */
        const int i = threadIdx.x;
        if (i < count * 32) {
            const long long r = (long long)firstRow[i >> 5] + (i & 31);
            if (r < a.rows) {
                const T total = totals[i];
                const int outRow = a.rIdx ? a.rIdx[r] : (int)r;
                a.z[outRow] = hasBeta ? epilogue<true>(a.alpha, total, a.beta, a.y[outRow]) : epilogue<false>(a.alpha, total, a.beta, zeroOf<T>());
            }
        }
    }
}
