"""CPU: the f3 converters (COO->DIA, DIA->HDIA, ELL->OELL) of the oracle and of the product against the
reference's own build (oracle/_ref), byte for byte; plus properties that hold without the reference."""
import numpy as np
import pytest

import oracle_api as O
from spgpu_amd import formats, synth


class _Product:
    label = "product"
    coo_to_ell = staticmethod(lambda n, r, c, v, coo_base=0, ell_base=0: formats.coo_to_ell(n, r, c, v, coo_base, ell_base))
    coo_to_dia = staticmethod(lambda n, m, r, c, v, coo_base=0: formats.coo_to_dia(n, m, r, c, v, coo_base))
    dia_to_hdia = staticmethod(lambda d, hs=32: formats.dia_to_hdia(d, hs))
    ell_to_oell = staticmethod(lambda e: formats.ell_to_oell(e))
    coo_to_hdia = staticmethod(lambda n, m, r, c, v, hs=32, coo_base=0: formats.coo_to_hdia(n, m, r, c, v, hs, coo_base))


IMPLS = [O.oracle_converters, _Product]
needs_ref = pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref not built")


def _same(a, b, keys):
    for k in keys:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), k
        else:
            assert x == y, k


def _banded_coo(rng, letter, base):
    n_rows, n_cols = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    offs = rng.choice(np.arange(-n_rows + 1, n_cols), size=min(int(rng.integers(1, 12)), n_rows + n_cols - 1), replace=False)
    rows, cols = [], []
    for o in offs:
        r = np.arange(max(0, -o), min(n_rows, n_cols - o))
        keep = rng.random(r.size) < 0.7
        rows.append(r[keep]); cols.append(r[keep] + o)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(rows.size)
    rows, cols = rows[perm], cols[perm]
    vals = rng.standard_normal(rows.size)
    vals[rng.random(rows.size) < 0.1] = 0.0        # explicit zeros: DIA->HDIA tests bytes, not entries
    if letter in "CZ":
        vals = vals + 1j * rng.standard_normal(rows.size)
    return n_rows, n_cols, (rows + base).astype(np.int32), (cols + base).astype(np.int32), vals.astype(O.NP_DTYPE[letter])


@needs_ref
@pytest.mark.parametrize("impl", IMPLS, ids=lambda i: i.label)
@pytest.mark.parametrize("letter", "SDCZ")
def test_dia_and_dia_to_hdia_bit_exact(letter, impl):
    ref = O.reference_converters()
    rng = np.random.default_rng(ord(letter))
    for trial in range(30):
        base = int(rng.integers(0, 2))
        n, m, r, c, v = _banded_coo(rng, letter, base)
        d_ref, d = ref.coo_to_dia(n, m, r, c, v, coo_base=base), impl.coo_to_dia(n, m, r, c, v, coo_base=base)
        _same(d, d_ref, ("diags", "pitch", "offsets", "values"))
        for hs in (32, 64):
            _same(impl.dia_to_hdia(d, hs), ref.dia_to_hdia(d_ref, hs), ("height", "hack_offsets", "offsets", "values"))


@needs_ref
@pytest.mark.parametrize("impl", IMPLS, ids=lambda i: i.label)
def test_ell_to_oell_bit_exact_for_every_small_size(impl):
    """The reference's merge sort has size-dependent merge schedules (ell.c:131-157); every size up to 130
    and a few larger ones, with many ties, must give its exact permutation."""
    ref = O.reference_converters()
    rng = np.random.default_rng(7)
    for n in list(range(1, 131)) + [255, 256, 257, 1000, 4097]:
        lengths = rng.integers(0, 5, n)                      # few distinct lengths: ties everywhere
        n_, m_, r, c, v = synth.random_rows_coo(n, 50, lengths, seed=n, letter="D")
        e_ref, e = ref.coo_to_ell(n, r, c, v), impl.coo_to_ell(n, r, c, v)
        (o_ref, idx_ref), (o, idx) = ref.ell_to_oell(e_ref), impl.ell_to_oell(e)
        assert np.array_equal(idx, idx_ref), n
        _same(o, o_ref, ("row_lengths", "indices", "values"))


@pytest.mark.parametrize("impl", IMPLS, ids=lambda i: i.label)
def test_oell_order_and_survey_probe(impl):
    # lengths {2,3,2,3,1,2} -> rIdx = 3,1,5,2,0,4
    n, m, r, c, v = synth.random_rows_coo(6, 9, np.array([2, 3, 2, 3, 1, 2]), seed=1, letter="S")
    oell, idx = impl.ell_to_oell(impl.coo_to_ell(6, r, c, v))
    assert idx.tolist() == [3, 1, 5, 2, 0, 4]
    assert oell["row_lengths"].tolist() == [3, 3, 2, 2, 2, 1]
    # general: length descending, ties by descending original row
    lengths = np.random.default_rng(3).integers(0, 7, 500)
    n, m, r, c, v = synth.random_rows_coo(500, 40, lengths, seed=2, letter="D")
    _, idx = impl.ell_to_oell(impl.coo_to_ell(500, r, c, v))
    assert idx.tolist() == sorted(range(500), key=lambda i: (-lengths[i], -i))


@pytest.mark.parametrize("impl", IMPLS, ids=lambda i: i.label)
def test_dia_route_equals_coo_route_for_hdia(impl):
    """For a matrix without explicit zeros COO->DIA->HDIA and COO->HDIA store the same thing."""
    n, m, r, c, v = synth.laplacian_3d_7pt(8)
    direct = impl.coo_to_hdia(n, m, r, c, v, 32)
    via = impl.dia_to_hdia(impl.coo_to_dia(n, m, r, c, v), 32)
    _same(via, direct, ("height", "hack_offsets", "offsets", "values"))


def test_oracle_dia_spmv_and_csput():
    n, m, r, c, v = synth.laplacian_2d_5pt(12)
    dia = O.oracle_converters.coo_to_dia(n, m, r, c, v)
    hdia = O.oracle_converters.coo_to_hdia(n, m, r, c, v, 32)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(m), rng.standard_normal(n)
    assert O.dia_spmv(dia, x, y, 1.5, -0.5).tobytes() == O.hdia_spmv(hdia, x, y, 1.5, -0.5).tobytes()
    # csput: overwrite the diagonal coefficients of an ELL matrix whose rows have ascending columns
    ell = O.oracle_converters.coo_to_ell(n, r, c, v)
    rows = np.arange(n, dtype=np.int32)
    new = O.ell_csput(ell, rows, rows, np.full(n, 9.0), 0)
    changed = dict(ell, values=new)
    z0, z1 = O.ell_spmv(ell, x, None, 1.0, 0.0), O.ell_spmv(changed, x, None, 1.0, 0.0)
    assert np.allclose(z1 - z0, 5.0 * x)      # diagonal 4 -> 9
    # a column that is not stored, and a negative row, are ignored
    assert O.ell_csput(ell, [0, -1], [n - 1, 0], [7.0, 7.0], 0).tobytes() == ell["values"].tobytes()
