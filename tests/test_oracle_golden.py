"""CPU: the oracle (oracle/spgpu_oracle.c) against the committed golden fixtures.

Conversion arrays in tests/golden/*.npz were produced by the REFERENCE's own
converters (oracle/make_golden.py); the oracle and the product's host converters
must reproduce them byte for byte.  SpMV expectations in the same files are
extended-precision products; tolerance 1e-6 (fp64) / 1e-4 (fp32) of the row
magnitude |alpha| sum|a x| + |beta y|, as north_star states it.
"""
import glob
import json
import os

import numpy as np
import pytest

import oracle_api as O
from spgpu_amd import formats, synth

TOL = {"S": 1e-4, "C": 1e-4, "D": 1e-6, "Z": 1e-6}


def _cases(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_dir, "*.npz")))


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, what
    assert a.tobytes() == b.tobytes(), f"{what}: bytes differ"


class _ProductConverters:
    """spgpu_amd.formats (the product's host C converters) behind the oracle_api driving interface."""
    label = "product"

    @staticmethod
    def coo_to_ell(n_rows, rows, cols, vals, coo_base=0, ell_base=0):
        return formats.coo_to_ell(n_rows, rows, cols, vals, coo_base, ell_base)

    @staticmethod
    def ell_to_hell(ell, hack_size=32):
        return formats.ell_to_hell(ell, hack_size)

    @staticmethod
    def coo_to_hdia(n_rows, n_cols, rows, cols, vals, hack_size=32, coo_base=0):
        return formats.coo_to_hdia(n_rows, n_cols, rows, cols, vals, hack_size, coo_base)


IMPLEMENTATIONS = [O.oracle_converters, _ProductConverters]


@pytest.mark.parametrize("impl", IMPLEMENTATIONS, ids=lambda i: i.label)
@pytest.mark.parametrize("name", _cases())
def test_conversions_match_reference_fixture(golden_dir, name, impl):
    g = _load(golden_dir, name)
    base, hs = int(g["base"]), int(g["hack_size"])
    n_rows, n_cols = int(g["n_rows"]), int(g["n_cols"])
    ell = impl.coo_to_ell(n_rows, g["coo_rows"], g["coo_cols"], g["coo_vals"], coo_base=base, ell_base=base)
    assert ell["max_row"] == int(g["ell_max_row"]) and ell["pitch"] == int(g["ell_pitch"])
    _same(ell["row_lengths"], g["row_lengths"], "row lengths")
    _same(ell["indices"], g["ell_indices"], "ELL indices")
    _same(ell["values"], g["ell_values"], "ELL values")
    hell = impl.ell_to_hell(ell, hs)
    assert hell["height"] == int(g["hell_height"])
    _same(hell["hack_offsets"], g["hell_hack_offsets"], "HELL hackOffsets")
    _same(hell["indices"], g["hell_indices"], "HELL indices")
    _same(hell["values"], g["hell_values"], "HELL values")
    hdia = impl.coo_to_hdia(n_rows, n_cols, g["coo_rows"], g["coo_cols"], g["coo_vals"], hs, coo_base=base)
    assert hdia["height"] == int(g["hdia_height"])
    _same(hdia["hack_offsets"], g["hdia_hack_offsets"], "HDIA hackOffsets")
    _same(hdia["offsets"], g["hdia_offsets"], "HDIA offsets")
    _same(hdia["values"], g["hdia_values"], "HDIA values")


@pytest.mark.parametrize("impl", IMPLEMENTATIONS, ids=lambda i: i.label)
def test_baseline_config1_checksums(golden_dir, impl):
    """BASELINE config 1: 5-point Laplacian 1024x1024 through the CPU converters (plumbing, no GPU)."""
    with open(os.path.join(golden_dir, "checksums.json")) as f:
        want = json.load(f)["lap2d_1024_d"]
    n, m, r, c, v = synth.laplacian_2d_5pt(1024)
    assert (n, r.size) == (want["n"], want["nnz"])
    ell = impl.coo_to_ell(n, r, c, v)
    hell = impl.ell_to_hell(ell, 32)
    hdia = impl.coo_to_hdia(n, m, r, c, v, 32)
    assert (ell["max_row"], ell["pitch"]) == (want["ell_max_row"], want["ell_pitch"])
    assert (hell["height"], hell["values"].size, int(hell["hack_offsets"][-1])) == (
        want["hell_height"], want["hell_slots"], want["hell_last_hack_offset"])
    assert hdia["height"] == want["hdia_height"]
    got = dict(ell_indices=O.fnv(ell["indices"]), ell_values=O.fnv(ell["values"]),
               hell_indices=O.fnv(hell["indices"]), hell_values=O.fnv(hell["values"]),
               hell_hack_offsets=O.fnv(hell["hack_offsets"]), row_lengths=O.fnv(ell["row_lengths"]),
               hdia_offsets=O.fnv(hdia["offsets"]), hdia_hack_offsets=O.fnv(hdia["hack_offsets"]),
               hdia_values=O.fnv(hdia["values"]))
    assert got == want["fnv"]


def test_survey_known_answers(golden_dir):
    """Structural known answers SURVEY.md 8(a) captured from the compiled reference."""
    with open(os.path.join(golden_dir, "checksums.json")) as f:
        want = json.load(f)["survey_8a_structure"]
    C = O.oracle_converters
    n, m, r, c, v = synth.laplacian_2d_5pt(32)
    ell = C.coo_to_ell(n, r, c, v)
    hell = C.ell_to_hell(ell, 32)
    w = want["lap2d_32"]
    assert (n, r.size, ell["max_row"], ell["pitch"], hell["height"], hell["values"].size,
            int(hell["hack_offsets"][-1])) == (w["n"], w["nnz"], w["ell_max_row"], w["ell_pitch"], w["hell_height"],
                                               w["hell_slots"], w["hell_last_hack_offset"])
    n, m, r, c, v = synth.laplacian_3d_7pt(16)
    hd = C.coo_to_hdia(n, m, r, c, v, 32)
    w = want["lap3d_16"]
    ho = hd["hack_offsets"]
    assert (n, r.size, ho.size - 1, hd["height"]) == (w["n"], w["nnz"], w["hacks"], w["hdia_height"])
    assert hd["offsets"][ho[0]:ho[1]].tolist() == w["hack0_offsets"]
    assert hd["offsets"][ho[8]:ho[9]].tolist() == w["hack8_offsets"]
    n, m, r, c, v = synth.ctest_matrix()
    ell = C.coo_to_ell(n, r, c, v)
    hell = C.ell_to_hell(ell, 32)
    hd = C.coo_to_hdia(n, m, r, c, v, 32)
    w = want["ctest"]
    assert (ell["max_row"], ell["pitch"], hell["height"]) == (w["ell_max_row"], w["ell_pitch"], w["hell_height"])
    assert hell["hack_offsets"].tolist() == w["hell_hack_offsets"]
    assert (hd["hack_offsets"].size - 1, hd["height"]) == (w["hdia_hacks"], w["hdia_height"])
    assert hd["hack_offsets"].tolist() == w["hdia_hack_offsets"]


def _check(z, g, letter):
    err = np.abs(z.astype(np.complex128 if letter in "CZ" else np.float64) - g["z_expected"])
    bound = TOL[letter] * g["z_scale"] + np.finfo(np.float64).tiny
    worst = float(np.max(err / bound)) if err.size else 0.0
    assert worst <= 1.0, f"error {worst:.3g} x tolerance"


@pytest.mark.parametrize("phases", [1, 2, 4, 8])
@pytest.mark.parametrize("name", [n for n in _cases() if n not in ("empty_d", "onerow_z")])
def test_oracle_spmv_matches_extended_precision(golden_dir, name, phases):
    """Every format's oracle SpMV, in every summation order the kernels use, against the
    independent extended-precision product."""
    g = _load(golden_dir, name)
    letter = O.LETTER_OF[g["coo_vals"].dtype]
    base, hs = int(g["base"]), int(g["hack_size"])
    alpha, beta = g["alpha"][()], g["beta"][()]
    y = g["y"] if beta != 0 else None
    ell = dict(letter=letter, rows=int(g["n_rows"]), values=g["ell_values"], indices=g["ell_indices"],
               pitch=int(g["ell_pitch"]), max_row=int(g["ell_max_row"]), row_lengths=g["row_lengths"], base=base)
    hell = dict(letter=letter, rows=int(g["n_rows"]), values=g["hell_values"], indices=g["hell_indices"],
                hack_offsets=g["hell_hack_offsets"], hack_size=hs, row_lengths=g["row_lengths"], base=base)
    _check(O.ell_spmv(ell, g["x"], y, alpha, beta, phases=phases), g, letter)
    _check(O.hell_spmv(hell, g["x"], y, alpha, beta, phases=phases), g, letter)
    if phases == 1:
        # rS == NULL walks the zero padding too (ell_spmv_base_nors.cuh)
        _check(O.ell_spmv(ell, g["x"], y, alpha, beta, with_row_sizes=False), g, letter)
        # HDIA merges duplicate (row, col) entries (last wins), so only duplicate-free inputs compare
        if not name.startswith("ctest"):
            key = g["coo_rows"].astype(np.int64) * (int(g["n_cols"]) + 2) + g["coo_cols"]
            if np.unique(key).size == key.size:
                hdia = dict(letter=letter, rows=int(g["n_rows"]), cols=int(g["n_cols"]), values=g["hdia_values"],
                            offsets=g["hdia_offsets"], hack_offsets=g["hdia_hack_offsets"], hack_size=hs)
                _check(O.hdia_spmv(hdia, g["x"], y, alpha, beta), g, letter)


def test_ctest_identity(golden_dir):
    """ctest.c:25-39,105: A = 2I (two unit entries per diagonal slot), alpha 2, beta -3  =>  z = 4x - 3y."""
    g = _load(golden_dir, "ctest_s")
    ell = dict(letter="S", rows=100, values=g["ell_values"], indices=g["ell_indices"], pitch=128, max_row=2,
               row_lengths=g["row_lengths"], base=0)
    hell = dict(letter="S", rows=100, values=g["hell_values"], indices=g["hell_indices"],
                hack_offsets=g["hell_hack_offsets"], hack_size=32, row_lengths=g["row_lengths"], base=0)
    want = 4.0 * g["x"].astype(np.float64) - 3.0 * g["y"].astype(np.float64)
    for z in (O.ell_spmv(ell, g["x"], g["y"], 2.0, -3.0), O.hell_spmv(hell, g["x"], g["y"], 2.0, -3.0)):
        assert np.max(np.abs(z - want)) <= 1e-5
    # the reference's only signal: dot(z, z) equal across formats
    ze, zh = O.ell_spmv(ell, g["x"], g["y"], 2.0, -3.0), O.hell_spmv(hell, g["x"], g["y"], 2.0, -3.0)
    assert O.dot("S", ze, ze) == O.dot("S", zh, zh)


def test_cross_format_equality_bitwise(golden_dir):
    """hellPerf.cpp:234,297 / diaPerf.cpp:227,321 compare dot(z,z) across formats; with one
    summation order the oracle's ELL and HELL results are the same bits."""
    for name in ("lap2d_32_d", "powerlaw_d_b0_h32", "powerlaw_z_b1_h64"):
        g = _load(golden_dir, name)
        letter = O.LETTER_OF[g["coo_vals"].dtype]
        base = int(g["base"])
        ell = dict(letter=letter, rows=int(g["n_rows"]), values=g["ell_values"], indices=g["ell_indices"],
                   pitch=int(g["ell_pitch"]), max_row=int(g["ell_max_row"]), row_lengths=g["row_lengths"], base=base)
        hell = dict(letter=letter, rows=int(g["n_rows"]), values=g["hell_values"], indices=g["hell_indices"],
                    hack_offsets=g["hell_hack_offsets"], hack_size=int(g["hack_size"]),
                    row_lengths=g["row_lengths"], base=base)
        for ph in (1, 2, 4):
            a = O.ell_spmv(ell, g["x"], g["y"], 1.0, 0.5, phases=ph)
            b = O.hell_spmv(hell, g["x"], g["y"], 1.0, 0.5, phases=ph)
            assert a.tobytes() == b.tobytes()


def test_rows_permutation_and_level1():
    rng = np.random.default_rng(5)
    n, m, r, c, v = synth.random_rows_coo(257, 300, synth.power_law_lengths(257, 6.0, 40, seed=9), seed=3, letter="D")
    ell = O.oracle_converters.coo_to_ell(n, r, c, v)
    hell = O.oracle_converters.ell_to_hell(ell, 32)
    x, y = rng.standard_normal(m), rng.standard_normal(n)
    perm = rng.permutation(n).astype(np.int32)
    plain = O.hell_spmv(hell, x, y, 1.25, 0.0)
    scattered = O.hell_spmv(hell, x, None, 1.25, 0.0, r_idx=perm)
    assert np.array_equal(scattered[perm], plain)
    # level 1 against numpy
    a, b = rng.standard_normal(1000), rng.standard_normal(1000)
    assert abs(O.dot("D", a, b) - float(np.dot(a, b))) <= 1e-12 * float(np.sum(np.abs(a * b)))
    assert abs(O.nrm2("D", a) - float(np.linalg.norm(a))) <= 1e-13 * float(np.linalg.norm(a))
    assert np.allclose(O.axpby("D", 1000, 0.5, b, -2.0, a), 0.5 * b - 2.0 * a, rtol=1e-15, atol=1e-15)
    assert np.array_equal(O.axpby("D", 1000, 0.0, None, -2.0, a), -2.0 * a)
    za = (a + 1j * b).astype(np.complex128)
    zb = (b - 1j * a).astype(np.complex128)
    assert abs(O.dot("Z", za, zb) - np.sum(za * zb)) <= 1e-11 * float(np.sum(np.abs(za * zb)))  # un-conjugated


@pytest.mark.parametrize("name", [n for n in _cases() if n.startswith("powerlaw")])
def test_kernel_order_restatements_match_extended_precision(golden_dir, name):
    """The oracle in the summation orders of the kernels a row order selects (oracle_api.slab_shape: queue kernel with the
    deep split and with sub-groups cut into chunks, share kernel) against the extended-precision product, on the ragged
    fixtures; a chunk length beyond every row changes no bit, and a row permutation only moves the results."""
    g = _load(golden_dir, name)
    letter = O.LETTER_OF[g["coo_vals"].dtype]
    base, hs = int(g["base"]), int(g["hack_size"])
    alpha, beta = g["alpha"][()], g["beta"][()]
    y = g["y"] if beta != 0 else None
    n = int(g["n_rows"])
    hell = dict(letter=letter, rows=n, values=g["hell_values"], indices=g["hell_indices"],
                hack_offsets=g["hell_hack_offsets"], hack_size=hs, row_lengths=g["row_lengths"], base=base)
    ell = dict(letter=letter, rows=n, values=g["ell_values"], indices=g["ell_indices"],
               pitch=int(g["ell_pitch"]), max_row=int(g["ell_max_row"]), row_lengths=g["row_lengths"], base=base)
    for mat in (hell, ell):
        for cap, split in ((256, -1), (16, 0), (64, 24), (64, 1 << 20)):
            shape = O.slab_shape(letter, "ragged", deep_cap=cap, split=split)
            _check(O.spmv_tail(mat, g["x"], y, alpha, beta, **shape), g, letter)
        _check(O.spmv_tail(mat, g["x"], y, alpha, beta, **O.slab_shape(letter, "share")), g, letter)
        whole = O.spmv_tail(mat, g["x"], y, alpha, beta, **O.slab_shape(letter, "ragged", deep_cap=64, split=0))
        never = O.spmv_tail(mat, g["x"], y, alpha, beta, **O.slab_shape(letter, "ragged", deep_cap=64, split=1 << 20))
        assert whole.tobytes() == never.tobytes()
    # through a row order: z[rIdx[r]] = row r (y gathered the same way)
    r_idx = np.random.default_rng(1).permutation(n).astype(np.int32)
    shape = O.slab_shape(letter, "ragged", deep_cap=64, split=24)
    straight = O.spmv_tail(hell, g["x"], y, alpha, beta, **shape)
    y_moved = None
    if y is not None:
        y_moved = np.empty_like(y)
        y_moved[r_idx] = y
    moved = O.spmv_tail(hell, g["x"], y_moved, alpha, beta, r_idx=r_idx, **shape)
    assert moved[r_idx].tobytes() == straight.tobytes()
