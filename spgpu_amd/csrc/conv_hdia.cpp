/*
 * COO -> HDIA on the host.  Own implementation of the behaviour specified by
 * the reference's src/core/hdia.cpp:8-11,161-349 (see
 * include/spgpu/hdia_conv.h); output arrays are byte-identical to the
 * reference's for the same input.
 *
 * The reference keeps one std::vector per hack and a std::map per hack; here
 * the entries are bucketed by hack with a counting sort (stable, so COO order
 * survives inside a hack) and the diagonals of a hack are a sorted, deduplicated
 * key array that is binary-searched.
 */
#include "spgpu/hdia_conv.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct HackBuckets {
    std::vector<size_t> start; /* hacks+1 */
    std::vector<int> entry;    /* COO entry ids grouped by hack, COO order kept */
};

HackBuckets bucketByHack(int hacks, int hackSize, int nnz, const int* cooRows, int base)
{
    HackBuckets b;
    b.start.assign((size_t)hacks + 1, 0);
    for (int e = 0; e < nnz; ++e)
        b.start[(size_t)((cooRows[e] - base) / hackSize) + 1] += 1;
    for (int h = 0; h < hacks; ++h)
        b.start[(size_t)h + 1] += b.start[(size_t)h];
    b.entry.resize((size_t)nnz);
    std::vector<size_t> cursor(b.start.begin(), b.start.end() - 1);
    for (int e = 0; e < nnz; ++e)
        b.entry[cursor[(size_t)((cooRows[e] - base) / hackSize)]++] = e;
    return b;
}

/* Key that identifies a diagonal inside one hack: column minus the row's
 * position in the hack.  Ascending key == ascending (column - row). */
inline int diagKey(int row0, int col0, int hackSize) { return col0 - row0 % hackSize; }

void hackKeys(std::vector<int>& keys, const HackBuckets& b, int h, int hackSize, const int* cooRows,
              const int* cooCols, int base)
{
    keys.clear();
    for (size_t p = b.start[(size_t)h]; p < b.start[(size_t)h + 1]; ++p) {
        const int e = b.entry[p];
        keys.push_back(diagKey(cooRows[e] - base, cooCols[e] - base, hackSize));
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
}

} // namespace

extern "C" int getHdiaHacksCount(int hackSize, int rowsCount)
{
    return (rowsCount + hackSize - 1) / hackSize;
}

extern "C" void computeHdiaHackOffsetsFromCoo(int* allocationHeight, int* hackOffsets, int hackSize,
                                              int rowsCount, int columnsCount, int nonZerosCount,
                                              const int* cooRowIndices, const int* cooColsIndices,
                                              int cooBaseIndex)
{
    (void)columnsCount;
    const int hacks = getHdiaHacksCount(hackSize, rowsCount);
    const HackBuckets b = bucketByHack(hacks, hackSize, nonZerosCount, cooRowIndices, cooBaseIndex);
    std::vector<int> keys;
    hackOffsets[0] = 0;
    for (int h = 0; h < hacks; ++h) {
        hackKeys(keys, b, h, hackSize, cooRowIndices, cooColsIndices, cooBaseIndex);
        hackOffsets[h + 1] = hackOffsets[h] + (int)keys.size();
    }
    *allocationHeight = hackOffsets[hacks];
}

extern "C" void cooToHdia(void* hdiaValues, int* hdiaOffsets, const int* hackOffsets, int hackSize,
                          int rowsCount, int columnsCount, int nonZerosCount, const int* cooRowIndices,
                          const int* cooColsIndices, const void* cooValues, int cooBaseIndex,
                          spgpuType_t valuesType)
{
    (void)columnsCount;
    const size_t elem = spgpuSizeOf(valuesType);
    const int hacks = getHdiaHacksCount(hackSize, rowsCount);
    const HackBuckets b = bucketByHack(hacks, hackSize, nonZerosCount, cooRowIndices, cooBaseIndex);
    std::vector<int> keys;
    char* out = static_cast<char*>(hdiaValues);
    const char* in = static_cast<const char*>(cooValues);

    for (int h = 0; h < hacks; ++h) {
        hackKeys(keys, b, h, hackSize, cooRowIndices, cooColsIndices, cooBaseIndex);
        const size_t firstDiag = (size_t)hackOffsets[h];
        for (size_t p = 0; p < keys.size(); ++p)
            hdiaOffsets[firstDiag + p] = keys[p] - h * hackSize; /* == column - row */

        for (size_t q = b.start[(size_t)h]; q < b.start[(size_t)h + 1]; ++q) {
            const int e = b.entry[q];
            const int row0 = cooRowIndices[e] - cooBaseIndex;
            const int key = diagKey(row0, cooColsIndices[e] - cooBaseIndex, hackSize);
            const size_t p = (size_t)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
            const size_t slot = (firstDiag + p) * (size_t)hackSize + (size_t)(row0 % hackSize);
            std::memcpy(out + slot * elem, in + (size_t)e * elem, elem);
        }
    }
}

/* ---- DIA -> HDIA (reference: hdia.cpp:13-153) ----------------------------------------------- */
namespace {

/* A DIA diagonal belongs to a hack iff any byte of its values in the hack's rows is non-zero. */
bool diagonalTouchesHack(const char* diaValues, size_t elem, int diagonal, int pitch, int firstRow, int endRow)
{
    const char* p = diaValues + ((size_t)diagonal * (size_t)pitch + (size_t)firstRow) * elem;
    const char* end = p + (size_t)(endRow - firstRow) * elem;
    for (; p != end; ++p)
        if (*p != 0)
            return true;
    return false;
}

} // namespace

extern "C" void computeHdiaHackOffsets(int* allocationHeight, int* hackOffsets, int hackSize, const void* diaValues,
                                       int diaValuesPitch, int diagonals, int rowsCount, spgpuType_t valuesType)
{
    const size_t elem = spgpuSizeOf(valuesType);
    const int hacks = getHdiaHacksCount(hackSize, rowsCount);
    const char* vals = static_cast<const char*>(diaValues);
    hackOffsets[0] = 0;
    for (int h = 0; h < hacks; ++h) {
        const int first = h * hackSize;
        const int end = first + hackSize < rowsCount ? first + hackSize : rowsCount;
        int kept = 0;
        for (int d = 0; d < diagonals; ++d)
            kept += diagonalTouchesHack(vals, elem, d, diaValuesPitch, first, end) ? 1 : 0;
        hackOffsets[h + 1] = hackOffsets[h] + kept;
    }
    *allocationHeight = hackOffsets[hacks];
}

extern "C" void diaToHdia(void* hdiaValues, int* hdiaOffsets, const int* hackOffsets, int hackSize,
                          const void* diaValues, const int* diaOffsets, int diaValuesPitch, int diagonals,
                          int rowsCount, spgpuType_t valuesType)
{
    const size_t elem = spgpuSizeOf(valuesType);
    const int hacks = getHdiaHacksCount(hackSize, rowsCount);
    const char* vals = static_cast<const char*>(diaValues);
    char* out = static_cast<char*>(hdiaValues);
    for (int h = 0; h < hacks; ++h) {
        const int first = h * hackSize;
        const int end = first + hackSize < rowsCount ? first + hackSize : rowsCount;
        size_t slot = (size_t)hackOffsets[h];
        for (int d = 0; d < diagonals; ++d) {
            if (!diagonalTouchesHack(vals, elem, d, diaValuesPitch, first, end))
                continue;
            hdiaOffsets[slot] = diaOffsets[d];
            std::memcpy(out + slot * (size_t)hackSize * elem,
                        vals + ((size_t)d * (size_t)diaValuesPitch + (size_t)first) * elem, (size_t)(end - first) * elem);
            ++slot;
        }
    }
}
