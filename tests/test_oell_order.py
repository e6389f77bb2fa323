"""CPU: the host row order (ell_conv.h oellOrder, csrc/conv_ell.c).  One window and no long-row group must be the
reference's ellToOell order (checked against the reference's own object, oracle/_ref); the windowed and long-row forms
have no counterpart in the reference and are checked against their definition."""
import numpy as np
import pytest

import oracle_api as O
from spgpu_amd import formats, synth


def brute_order(lengths, window, long_rows):
    """The definition, spelt out: the long rows first, in windows 32 times as large, then the others in their windows;
    (length, row) descending in the first window of a class and in every other one from there, ascending in between."""
    lengths = np.asarray(lengths)
    n = lengths.size
    out = []
    is_long = [long_rows > 0 and lengths[r] > long_rows for r in range(n)]
    for rows, w in (([r for r in range(n) if is_long[r]], 32 * window if window > 0 else max(n, 1)),
                    ([r for r in range(n) if not is_long[r]], window if window > 0 else max(n, 1))):
        for g in range((n + w - 1) // w):
            members = [r for r in rows if r // w == g]
            key = (lambda r: (-lengths[r], -r)) if g % 2 == 0 else (lambda r: (lengths[r], r))
            out += sorted(members, key=key)
    return np.array(out, np.int32)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 257, 1000])
@pytest.mark.parametrize("window,long_rows", [(0, 0), (8, 0), (64, 0), (0, 5), (16, 5), (2, 3), (100, 12), (5000, 0)])
def test_order_matches_its_definition(n, window, long_rows):
    rng = np.random.default_rng(n * 131 + window * 7 + long_rows)
    lengths = np.minimum(rng.zipf(1.7, size=n), 40).astype(np.int32)
    r_idx, dst = formats.oell_order(lengths, window, long_rows)
    if n == 2 and long_rows <= 0 and (window <= 0 or window >= n):
        assert r_idx.tolist() == [0, 1]             # the reference never sorts exactly two rows (ell.c:131-157)
    else:
        assert r_idx.tolist() == brute_order(lengths, window, long_rows).tolist()
    assert dst.tolist() == lengths[r_idx].tolist()
    assert sorted(r_idx.tolist()) == list(range(n))


def brute_aligned_order(lengths, window, long_rows):
    """oellOrderAligned's definition: the long rows as in oellOrder; the others, in their original order, fill the positions
    behind them, and the windows are the runs of positions between multiples of `window`."""
    lengths = np.asarray(lengths)
    n = lengths.size
    if window <= 0 or long_rows <= 0:
        return brute_order(lengths, window, long_rows)
    longs = [r for r in range(n) if lengths[r] > long_rows]
    shorts = [r for r in range(n) if lengths[r] <= long_rows]
    out = []
    w = 32 * window
    for g in range((n + w - 1) // w):
        members = [r for r in longs if r // w == g]
        out += sorted(members, key=(lambda r: (-lengths[r], -r)) if g % 2 == 0 else (lambda r: (lengths[r], r)))
    P = len(longs)
    first_window = P // window
    position = P
    while position < n:
        end = min(n, (position // window + 1) * window)
        members = shorts[position - P:end - P]
        g = position // window - first_window
        out += sorted(members, key=(lambda r: (-lengths[r], -r)) if g % 2 == 0 else (lambda r: (lengths[r], r)))
        position = end
    return np.array(out, np.int32)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 257, 1000, 5003])
@pytest.mark.parametrize("window,long_rows", [(0, 0), (8, 0), (0, 5), (16, 5), (2, 3), (100, 12), (64, 1), (64, 39), (64, 40)])
def test_aligned_order_matches_its_definition(n, window, long_rows):
    """oellOrderAligned: equal to oellOrder without a window or without rows set aside; otherwise every window of the shorter
    rows but the first starts at a multiple of the window in the new order, holds `window` consecutive shorter rows, and is
    sorted by (length, row) in alternating directions."""
    rng = np.random.default_rng(n * 31 + window * 7 + long_rows)
    lengths = np.minimum(rng.zipf(1.7, size=n), 40).astype(np.int32)
    r_idx, dst = formats.oell_order(lengths, window, long_rows, aligned=True)
    if window <= 0 or long_rows <= 0:
        plain, _ = formats.oell_order(lengths, window, long_rows)
        assert r_idx.tolist() == plain.tolist()
    else:
        assert r_idx.tolist() == brute_aligned_order(lengths, window, long_rows).tolist()
        P = int((lengths > long_rows).sum())
        assert (lengths[r_idx[:P]] > long_rows).all() and (lengths[r_idx[P:]] <= long_rows).all()
        for start in range((P // window + 1) * window, n, window):      # the aligned windows: consecutive shorter rows
            members = np.sort(r_idx[start:start + window])
            shorter_between = np.flatnonzero(lengths[members[0]:members[-1] + 1] <= long_rows) + members[0]
            assert members.tolist() == shorter_between.tolist()
    assert dst.tolist() == lengths[r_idx].tolist()
    assert sorted(r_idx.tolist()) == list(range(n))


@pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref is built in the build container only")
@pytest.mark.parametrize("n", list(range(1, 40)) + [100, 257, 1024, 4099])
def test_one_window_is_the_reference_order(n):
    rng = np.random.default_rng(n)
    lengths = rng.integers(0, 9, size=n).astype(np.int32)
    ell = dict(letter="S", rows=n, values=np.zeros(((n + 31) // 32 * 32) * 9, np.float32),
               indices=np.zeros(((n + 31) // 32 * 32) * 9, np.int32), pitch=(n + 31) // 32 * 32, max_row=9,
               row_lengths=lengths, base=0)
    _, want = O.reference_converters().ell_to_oell(ell)
    got, dst = formats.oell_order(lengths, 0, 0)
    assert got.tolist() == want.tolist()
    got2, _ = formats.oell_order(lengths, n + 5, 0)       # a window that holds every row is one window
    assert got2.tolist() == want.tolist()


def test_padding_of_the_north_star_lengths():
    """What the order is for: power-law lengths (mean 32, max 2048) at hack 32 store 5 slots per nonzero as they come,
    ~1.0 after a global sort, and <= 1.1 with windows of 4096 rows once rows longer than 256 are set aside."""
    lengths = synth.power_law_lengths(1_000_000, 32.0, 2048, 5)

    def slots_per_nnz(order):
        L = lengths[order].astype(np.int64)
        pad = (-L.size) % 32
        return np.concatenate([L, np.zeros(pad, np.int64)]).reshape(-1, 32).max(1).sum() * 32 / L.sum()

    assert slots_per_nnz(np.arange(lengths.size)) > 4.5
    assert slots_per_nnz(formats.oell_order(lengths, 0, 0)[0]) < 1.002
    assert slots_per_nnz(formats.oell_order(lengths, 4096, 256)[0]) < 1.05
    windowed, _ = formats.oell_order(lengths, 4096, 0)
    assert 1.1 < slots_per_nnz(windowed) < 1.4
    assert np.max(np.abs(windowed.astype(np.int64) - np.arange(lengths.size))) < 4096   # every row stays inside its window
