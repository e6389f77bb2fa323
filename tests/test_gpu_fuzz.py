"""GPU: seeded random cases through the C ABI against EXTENDED PRECISION (not the kernel-shaped oracle): every type, hack
sizes that do and do not divide 32, both index bases, empty / short / power-law / a few huge rows, random / local / band
columns, rIdx absent / a random permutation / the ordering by length, every x-fetch form incl. SWEEP, alpha and beta incl. 0,
z == y.
The bound is north_star's: |z - z*| <= tol * (|alpha| * sum |a_ij x_j| + |beta y_i|), tol 1e-6 (fp64) / 1e-4 (fp32)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu

TOL = {"S": 1e-4, "C": 1e-4, "D": 1e-6, "Z": 1e-6}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _case(seed):
    rng = np.random.default_rng(seed)
    letter = "SDCZ"[seed % 4]
    n = int(rng.choice([1, 2, 31, 32, 33, 100, 777, 2048, 5003, 20011]))
    cols_n = int(rng.choice([n, max(1, n // 3), n + 17]))
    hack = int(rng.choice([1, 2, 3, 8, 16, 30, 32, 33, 64, 96]))
    base = int(rng.integers(0, 2))
    kind = rng.choice(["uniform", "powerlaw", "huge", "empty_mix"])
    if kind == "uniform":
        lengths = np.full(n, int(rng.integers(1, 40)))
    elif kind == "powerlaw":
        lengths = np.minimum((3.0 * rng.random(n) ** -0.5).astype(np.int64), 700)
    elif kind == "huge":
        lengths = rng.integers(0, 6, n)
        lengths[rng.integers(0, n, max(1, n // 200))] = int(rng.integers(300, 1500))
    else:
        lengths = rng.integers(0, 3, n) * rng.integers(0, 9, n)
    lengths = np.minimum(lengths, cols_n).astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
    nnz = int(rows.size)
    pattern = rng.choice(["random", "near", "band"])
    k = np.arange(nnz, dtype=np.int64) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    if pattern == "random":
        cols = rng.integers(0, cols_n, nnz)
    elif pattern == "near":
        cols = (rows * cols_n // max(n, 1) + rng.integers(-300, 300, nnz)) % cols_n
    else:
        cols = (rows * cols_n // max(n, 1) - np.repeat(lengths, lengths) // 2 + k) % cols_n
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    vals = rng.standard_normal(nnz).astype(real)
    if letter in "CZ":
        vals = (vals + 1j * rng.standard_normal(nnz).astype(real)).astype(O.NP_DTYPE[letter])
    perm = rng.permutation(nnz)                      # COO in any order
    return dict(letter=letter, n=n, cols_n=cols_n, hack=hack, base=base, rows=rows[perm], cols=cols[perm], vals=vals[perm],
                rng=rng, kind=str(kind), pattern=str(pattern))


def _vector(rng, letter, n):
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    v = rng.standard_normal(n).astype(real)
    if letter in "CZ":
        v = (v + 1j * rng.standard_normal(n).astype(real)).astype(O.NP_DTYPE[letter])
    return v


def _exact(case, x, y, alpha, beta):
    cplx = case["letter"] in "CZ"
    wide = np.clongdouble if cplx else np.longdouble
    prod = case["vals"].astype(wide) * x.astype(wide)[case["cols"]]
    acc = np.zeros(case["n"], wide)
    np.add.at(acc, case["rows"], prod)
    mag = np.zeros(case["n"], np.longdouble)
    np.add.at(mag, case["rows"], np.abs(prod))
    z = wide(alpha) * acc + (wide(beta) * y.astype(wide) if beta != 0 else 0)
    scale = abs(alpha) * mag + (np.abs(wide(beta) * y.astype(wide)) if beta != 0 else 0)
    return z, scale.astype(np.float64)


@pytest.mark.parametrize("seed", range(72))
def test_random_case_within_the_bound(gpu, seed):
    import torch
    from spgpu_amd import capi, formats
    case = _case(seed)
    letter, n, base, hack, rng = case["letter"], case["n"], case["base"], case["hack"], case["rng"]
    ell = formats.coo_to_ell(n, case["rows"] + base, case["cols"] + base, case["vals"], coo_base=base, ell_base=base)
    order = rng.choice(["none", "permutation", "by_length", "windowed"])
    r_idx = None
    if order == "by_length":
        ell, r_idx = formats.ell_to_oell(ell)
    elif order == "windowed":
        r_idx, _ = formats.oell_order(ell["row_lengths"], window=int(rng.choice([32, 64, 512])), long_rows=int(rng.choice([0, 20])))
        # apply the order on the host: row i of the ordered matrix is row r_idx[i] of the original
        inverse = np.empty(n, np.int64)
        inverse[r_idx] = np.arange(n)
        ell = formats.coo_to_ell(n, inverse[case["rows"]] + base, case["cols"] + base, case["vals"], coo_base=base, ell_base=base)
    elif order == "permutation":
        r_idx = rng.permutation(n).astype(np.int32)   # any bijection: z[r_idx[i]] = row i
    hell = formats.ell_to_hell(ell, hack)
    x = _vector(rng, letter, case["cols_n"])
    y = _vector(rng, letter, n)
    alpha = [1.0, -0.75, 2.5][int(rng.integers(0, 3))]
    beta = [0.0, 1.0, -0.5][int(rng.integers(0, 3))]
    in_place = beta != 0 and bool(rng.integers(0, 2))
    # what the product must be, in terms of the ORIGINAL rows
    if order == "permutation":
        z_rows, scale_rows = _exact(case, x, np.zeros(n, y.dtype), alpha, 0.0)        # row sums first, then through rIdx
        want = np.zeros(n, z_rows.dtype)
        want[r_idx] = z_rows
        scale = np.zeros(n)
        scale[r_idx] = scale_rows
        if beta != 0:
            wide = want.dtype.type
            want = want + wide(beta) * y.astype(want.dtype)
            scale = scale + np.abs(wide(beta) * y.astype(want.dtype)).astype(np.float64)
    else:
        want, scale = _exact(case, x, y, alpha, beta)
    dx, dy = formats.to_device(x), formats.to_device(y)
    rI = formats.to_device(r_idx)
    for form in (capi.FORM_AUTO, capi.FORM_GATHER, capi.FORM_STRIPS, capi.FORM_XTILE, capi.FORM_SWEEP):
        capi.spgpuSetSpmvForm(gpu, form)
        try:
            # an ordered matrix a second time under AUTO: the first call starts the per-matrix plan, the second,
            # after the synchronisation, runs on it -- same arrays, so the plan's key matches
            repeats = 2 if (r_idx is not None and form == capi.FORM_AUTO) else 1
            kept = {}
            for fmt in ("hell", "ell"):
                for repeat in range(repeats):
                    dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
                    yy = dz if in_place else (dy if beta != 0 else None)
                    if fmt == "hell":
                        mat = kept.setdefault("hell", formats.DeviceHell(hell, r_idx=None))
                        capi.hellspmv[letter](gpu, _p(dz), _p(yy), capi.scalar(letter, alpha), _p(mat.cM), _p(mat.rP), hack,
                                              _p(mat.hack_offsets), _p(mat.rS), _p(rI), 8, n, _p(dx), capi.scalar(letter, beta), base)
                    else:
                        cM, rP, rS = kept.setdefault("ell", (formats.to_device(ell["values"]), formats.to_device(ell["indices"]),
                                                             formats.to_device(ell["row_lengths"])))
                        capi.ellspmv[letter](gpu, _p(dz), _p(yy), capi.scalar(letter, alpha), _p(cM), _p(rP), ell["pitch"], ell["pitch"],
                                             _p(rS), _p(rI), 8, ell["max_row"], n, _p(dx), capi.scalar(letter, beta), base)
                    torch.cuda.synchronize()
                    got = dz.cpu().numpy()
                    err = np.abs(got.astype(want.dtype) - want).astype(np.float64)
                    bound = TOL[letter] * scale + 1e-300
                    worst = int(np.argmax(err - bound))
                    assert np.all(err <= bound), (seed, letter, case["kind"], case["pattern"], str(order), fmt, form, repeat, n, hack,
                                                  base, worst, got[worst], want[worst])
        finally:
            capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("seed", range(40))
def test_random_diagonal_case_within_the_bound(gpu, seed):
    """HDIA built from COO, DIA, and HDIA built from DIA: a random set of diagonals (some only partly filled, some outside
    short rectangular matrices), every type, hack sizes 1 ... 96, against extended precision."""
    import torch
    from spgpu_amd import capi, formats
    rng = np.random.default_rng(1000 + seed)
    letter = "SDCZ"[seed % 4]
    n = int(rng.choice([1, 5, 32, 33, 257, 1000, 4099, 30011]))
    cols_n = int(rng.choice([n, n + 40, max(1, n - 13)]))
    hack = int(rng.choice([1, 2, 8, 30, 32, 33, 64, 96]))
    count = int(rng.integers(1, 12))
    offsets = np.unique(np.concatenate(([0], rng.integers(-min(n, 3000) + 1, min(cols_n, 3000), count))))
    rows_l, cols_l = [], []
    for off in offsets:
        r = np.arange(n, dtype=np.int64)
        c = r + off
        keep = (c >= 0) & (c < cols_n) & (rng.random(n) < rng.choice([1.0, 1.0, 0.6, 0.05]))   # full, or with holes
        rows_l.append(r[keep])
        cols_l.append(c[keep])
    rows, cols = np.concatenate(rows_l), np.concatenate(cols_l)
    if rows.size == 0:
        rows, cols = np.array([0], np.int64), np.array([0], np.int64)
    nnz = int(rows.size)
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    vals = rng.standard_normal(nnz).astype(real)
    if letter in "CZ":
        vals = (vals + 1j * rng.standard_normal(nnz).astype(real)).astype(O.NP_DTYPE[letter])
    perm = rng.permutation(nnz)
    rows, cols, vals = rows[perm], cols[perm], vals[perm]
    case = dict(letter=letter, n=n, rows=rows, cols=cols, vals=vals)
    x, y = _vector(rng, letter, cols_n), _vector(rng, letter, n)
    alpha = [1.0, -0.75, 2.5][int(rng.integers(0, 3))]
    beta = [0.0, 1.0, -0.5][int(rng.integers(0, 3))]
    want, scale = _exact(case, x, y, alpha, beta)
    bound = TOL[letter] * scale + 1e-300
    dx, dy = formats.to_device(x), formats.to_device(y)
    dia = formats.coo_to_dia(n, cols_n, rows, cols, vals)
    mats = [("hdia from coo", formats.DeviceHdia(formats.coo_to_hdia(n, cols_n, rows, cols, vals, hack))),
            ("dia", formats.DeviceDia(dia)),
            ("hdia from dia", formats.DeviceHdia(formats.dia_to_hdia(dia, hack)))]
    for name, mat in mats:
        for in_place in ((False, True) if beta != 0 else (False,)):
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            mat.spmv(gpu, dz, dz if in_place else (dy if beta != 0 else None), alpha, dx, beta)
            torch.cuda.synchronize()
            got = dz.cpu().numpy()
            err = np.abs(got.astype(want.dtype) - want).astype(np.float64)
            worst = int(np.argmax(err - bound))
            assert np.all(err <= bound), (seed, letter, name, n, cols_n, hack, offsets.tolist(), worst, got[worst], want[worst])
