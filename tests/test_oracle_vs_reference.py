"""CPU: oracle and product host converters against the reference's OWN converters.

oracle/_ref/libspgpu_ref.so is ell.c / hell.c / hdia.cpp / core.c of the reference,
compiled unmodified (oracle/Makefile).  It exists in the build container and
travels to the GPU box prebuilt; where it is absent these tests skip and the
committed fixtures (test_oracle_golden.py) carry the pin alone.
"""
import numpy as np
import pytest

import oracle_api as O
from spgpu_amd import formats, synth
from test_oracle_golden import _ProductConverters

pytestmark = pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref not built (no /root/reference here)")


def _same_dict(a, b, keys):
    for k in keys:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), k
        else:
            assert x == y, k


def _random_coo(rng, letter, base):
    n_rows = int(rng.integers(1, 400))
    n_cols = int(rng.integers(1, 400))
    style = rng.integers(0, 3)
    if style == 0:      # ragged, random columns
        lengths = rng.integers(0, 12, n_rows)
    elif style == 1:    # a few very long rows
        lengths = np.where(rng.random(n_rows) < 0.05, rng.integers(20, min(n_cols, 90) + 21, n_rows), rng.integers(0, 4, n_rows))
    else:               # banded
        lengths = np.full(n_rows, int(rng.integers(1, 8)))
    rows = np.repeat(np.arange(n_rows), lengths)
    if style == 2:
        k = np.concatenate([np.arange(l) for l in lengths]) if rows.size else np.zeros(0, int)
        cols = np.clip(rows + k - 3, 0, n_cols - 1)
    else:
        cols = rng.integers(0, n_cols, rows.size)
    perm = rng.permutation(rows.size)       # arbitrary COO order, duplicates allowed
    rows, cols = rows[perm], cols[perm]
    vals = rng.standard_normal(rows.size)
    if letter in "CZ":
        vals = vals + 1j * rng.standard_normal(rows.size)
    vals = vals.astype(O.NP_DTYPE[letter])
    return n_rows, n_cols, (rows + base).astype(np.int32), (cols + base).astype(np.int32), vals


@pytest.mark.parametrize("impl", [O.oracle_converters, _ProductConverters], ids=lambda i: i.label)
@pytest.mark.parametrize("letter", "SDCZ")
def test_converters_bit_exact_on_random_inputs(letter, impl):
    ref = O.reference_converters()
    rng = np.random.default_rng({"S": 1, "D": 2, "C": 3, "Z": 4}[letter])
    for trial in range(40):
        base = int(rng.integers(0, 2))
        hs = int(rng.choice([32, 64, 96]))
        n_rows, n_cols, r, c, v = _random_coo(rng, letter, base)
        e_ref = ref.coo_to_ell(n_rows, r, c, v, coo_base=base, ell_base=base)
        e = impl.coo_to_ell(n_rows, r, c, v, coo_base=base, ell_base=base)
        _same_dict(e, e_ref, ("max_row", "pitch", "row_lengths", "indices", "values"))
        h_ref, h = ref.ell_to_hell(e_ref, hs), impl.ell_to_hell(e, hs)
        _same_dict(h, h_ref, ("height", "hack_offsets", "indices", "values"))
        d_ref = ref.coo_to_hdia(n_rows, n_cols, r, c, v, hs, coo_base=base)
        d = impl.coo_to_hdia(n_rows, n_cols, r, c, v, hs, coo_base=base)
        _same_dict(d, d_ref, ("height", "hack_offsets", "offsets", "values"))


def test_index_base_translation():
    """cooToEll stores col - cooBase + ellBase (ell.c:72)."""
    ref = O.reference_converters()
    n, m, r, c, v = synth.random_rows_coo(50, 60, np.full(50, 3), seed=4, letter="S", base=1)
    for impl in (O.oracle_converters, _ProductConverters):
        for ell_base in (0, 1):
            _same_dict(impl.coo_to_ell(n, r, c, v, coo_base=1, ell_base=ell_base),
                       ref.coo_to_ell(n, r, c, v, coo_base=1, ell_base=ell_base), ("indices", "values", "row_lengths"))


def test_size_of_matches_reference():
    import ctypes as C
    from spgpu_amd import capi
    addr = O.orc.orc_ref_symbol(b"spgpuSizeOf")
    ref_size_of = C.CFUNCTYPE(C.c_size_t, C.c_int)(addr)
    for code in range(-1, 7):
        assert ref_size_of(code) == capi.spgpuSizeOf(code) == O.orc.orc_sizeOf(code)
