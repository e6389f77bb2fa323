"""GPU: parity of the HIP kernels, called through the C ABI, against the oracle and the
golden fixtures.  Tolerance (north_star): 1e-6 (fp64) / 1e-4 (fp32) of the row magnitude;
in addition the kernels are compared BIT FOR BIT with the oracle run in the kernel's own
summation order (default dispatch: D/C ascending k like the reference's one-thread-per-row
kernel, with the entries of a long row's tail split 64 ways; S 8 phases; Z 2 phases like the
reference's two-threads-per-row kernel), which is
stricter than the contract and catches indexing slips that a tolerance would hide."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu

TOL = {"S": 1e-4, "C": 1e-4, "D": 1e-6, "Z": 1e-6}
# default dispatch of ellpack_spmv.hip (launchSlabFamily): see oracle_api.default_spmv
_ONE, _TWO = dict.fromkeys("SDCZ", 1), dict.fromkeys("SDCZ", 2)
_PHX2 = {"S": 8, "D": 4, "C": 4, "Z": 2}       # wide kernel with 2*RPL phases (Z has RPL 1: narrow, 2 phases)
VARIANT_PHASES = {1: _PHX2, 2: {**_ONE, "Z": 2}, 3: _TWO, 4: _ONE, 6: _PHX2, 12: {**_ONE, "Z": 2}, 13: _TWO}
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def _load(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def _mats(g):
    letter = O.LETTER_OF[g["coo_vals"].dtype]
    base, hs = int(g["base"]), int(g["hack_size"])
    ell = dict(letter=letter, rows=int(g["n_rows"]), values=g["ell_values"], indices=g["ell_indices"],
               pitch=int(g["ell_pitch"]), max_row=int(g["ell_max_row"]), row_lengths=g["row_lengths"], base=base)
    hell = dict(letter=letter, rows=int(g["n_rows"]), values=g["hell_values"], indices=g["hell_indices"],
                hack_offsets=g["hell_hack_offsets"], hack_size=hs, row_lengths=g["row_lengths"], base=base,
                height=int(g["hell_height"]))
    hdia = dict(letter=letter, rows=int(g["n_rows"]), cols=int(g["n_cols"]), values=g["hdia_values"],
                offsets=g["hdia_offsets"], hack_offsets=g["hdia_hack_offsets"], hack_size=hs,
                height=int(g["hdia_height"]))
    return letter, ell, hell, hdia


def _within(z, g, letter):
    err = np.abs(z.astype(np.complex128 if letter in "CZ" else np.float64) - g["z_expected"])
    bound = TOL[letter] * g["z_scale"] + np.finfo(np.float64).tiny
    return float(np.max(err / bound)) if err.size else 0.0


def _run(handle, mat, x, y, alpha, beta, in_place=False):
    """One C-ABI SpMV call; returns z as numpy."""
    import torch
    from spgpu_amd import formats
    dx = formats.to_device(x)
    dy = formats.to_device(y) if y is not None else None
    if in_place and dy is not None:
        dz = dy
    else:
        dz = torch.full((mat.rows,), float("nan"), dtype=dx.dtype, device=dx.device)
    mat.spmv(handle, dz, dy, alpha, dx, beta)
    torch.cuda.synchronize()
    return dz.cpu().numpy()


@pytest.mark.parametrize("name", [n for n in CASES if n not in ("empty_d", "onerow_z")])
def test_fixture_parity_all_formats(gpu, name):
    from spgpu_amd import formats
    g = _load(name)
    letter, ell, hell, hdia = _mats(g)
    alpha, beta = g["alpha"][()], g["beta"][()]
    y = g["y"] if beta != 0 else None

    z = _run(gpu, formats.DeviceHell(hell), g["x"], y, alpha, beta)
    assert _within(z, g, letter) <= 1.0
    assert z.tobytes() == O.default_spmv(hell, g["x"], y, alpha, beta).tobytes()

    z = _run(gpu, formats.DeviceEll(ell), g["x"], y, alpha, beta)
    assert _within(z, g, letter) <= 1.0
    assert z.tobytes() == O.default_spmv(ell, g["x"], y, alpha, beta).tobytes()

    # rS == NULL: iterate maxNnzPerRow over the zero padding (index 0 - baseIndex may be -1: skipped)
    z = _run(gpu, formats.DeviceEll(ell, with_row_sizes=False), g["x"], y, alpha, beta)
    assert _within(z, g, letter) <= 1.0

    key = g["coo_rows"].astype(np.int64) * (int(g["n_cols"]) + 2) + g["coo_cols"]
    if np.unique(key).size == key.size:  # HDIA merges duplicates
        z = _run(gpu, formats.DeviceHdia(hdia), g["x"], y, alpha, beta)
        assert _within(z, g, letter) <= 1.0
        assert z.tobytes() == O.hdia_spmv(hdia, g["x"], y, alpha, beta).tobytes()


@pytest.mark.parametrize("variant", [0] + [pytest.param(v, marks=pytest.mark.skipif("not config._lab_build", reason="a forced kernel shape: -DSPGPU_TUNING_VARIANTS build"))
                                           for v in (1, 2, 3, 4, 6, 12, 13, 17, 18, 21, 22)])
@pytest.mark.parametrize("name", ["powerlaw_s_b1_h64", "powerlaw_d_b0_h32", "powerlaw_c_b0_h32", "powerlaw_z_b1_h64"])
def test_every_kernel_variant_bit_exact(gpu, name, variant, tuning):
    from spgpu_amd import formats
    tuning(SPGPU_SPMV_VARIANT=variant)
    g = _load(name)
    letter, ell, hell, _ = _mats(g)
    # Z (16-byte elements) has no wide form: wide requests run narrow 2x4 pipe
    rpl = 16 // np.dtype(O.NP_DTYPE[letter]).itemsize
    if variant == 0:          # what the library picks for the type (the only shapes of the product build)
        want_of = lambda m, yy, b: O.default_spmv(m, g["x"], yy, g["alpha"][()], b)
    elif variant in (17, 18, 21, 22) and letter != "Z":
        one_phase = variant in (17, 21)
        ph = 1 if one_phase else 2 * rpl                         # 18/22: wide kernel with 2*RPL phases x 2 columns
        shape = dict(group_rows=(64 // ph) * rpl, rows_per_lane=rpl, step=8 if one_phase else 2 * ph, tail_lanes=16, phases=ph)
        want_of = lambda m, yy, b: O.spmv_tail(m, g["x"], yy, g["alpha"][()], b, **shape)
    else:
        ph = 2 if (variant in (17, 18, 21, 22) and letter == "Z") else VARIANT_PHASES[variant][letter]
        want_of = lambda m, yy, b: (O.hell_spmv if "hack_offsets" in m else O.ell_spmv)(m, g["x"], yy, g["alpha"][()], b, phases=ph)
    for beta in (0.0, g["beta"][()] if g["beta"][()] != 0 else 0.5):
        y = g["y"] if beta != 0 else None
        z = _run(gpu, formats.DeviceHell(hell), g["x"], y, g["alpha"][()], beta)
        assert z.tobytes() == want_of(hell, y, beta).tobytes()
        z = _run(gpu, formats.DeviceEll(ell), g["x"], y, g["alpha"][()], beta)
        assert z.tobytes() == want_of(ell, y, beta).tobytes()


@pytest.mark.parametrize("name", ["powerlaw_d_b1_h64", "powerlaw_s_b0_h32", "powerlaw_z_b0_h32"])
def test_row_reorder_in_place_and_beta_zero_ignores_y(gpu, name):
    import torch
    from spgpu_amd import formats
    g = _load(name)
    letter, ell, hell, _ = _mats(g)
    rng = np.random.default_rng(3)
    perm = rng.permutation(hell["rows"]).astype(np.int32)
    alpha, beta = g["alpha"][()], (0.5 if letter in "SD" else 0.5 - 0.25j)

    # rIdx: row r of the storage is row rIdx[r] of y and z (hell_spmv_base_template.cuh:227-252)
    z = _run(gpu, formats.DeviceHell(hell, r_idx=perm), g["x"], g["y"], alpha, beta)
    assert z.tobytes() == O.default_spmv(hell, g["x"], g["y"], alpha, beta, r_idx=perm).tobytes()
    z = _run(gpu, formats.DeviceEll(ell, r_idx=perm), g["x"], g["y"], alpha, beta)
    assert z.tobytes() == O.default_spmv(ell, g["x"], g["y"], alpha, beta, r_idx=perm).tobytes()

    # z may alias y exactly (hell.h:30)
    z = _run(gpu, formats.DeviceHell(hell), g["x"], g["y"], alpha, beta, in_place=True)
    assert z.tobytes() == O.default_spmv(hell, g["x"], g["y"], alpha, beta).tobytes()

    # beta == 0 must not read y: NaN-filled y may not leak into z
    mat = formats.DeviceHell(hell)
    dx = formats.to_device(g["x"])
    dy = torch.full((hell["rows"],), float("nan"), dtype=dx.dtype, device="cuda:0")
    dz = torch.empty_like(dy)
    mat.spmv(gpu, dz, dy, alpha, dx, 0.0)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == O.default_spmv(hell, g["x"], None, alpha, 0.0).tobytes()


def test_unaligned_streams_take_the_narrow_kernel(gpu):
    """cM / rP / z offset by one element from a 16-byte boundary: results must not change."""
    import torch
    from spgpu_amd import capi, formats
    g = _load("powerlaw_d_b0_h32")
    letter, ell, hell, _ = _mats(g)
    want = O.hell_spmv(hell, g["x"], g["y"], 1.0, 0.5, phases=2)  # narrow slab kernel: 2 phases

    def shifted(a):
        t = torch.empty(a.size + 1, dtype=torch.from_numpy(a[:1].copy()).dtype, device="cuda:0")
        t[1:] = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        return t[1:]

    cM, rP = shifted(hell["values"]), shifted(hell["indices"])
    ho, rS = formats.to_device(hell["hack_offsets"]), formats.to_device(hell["row_lengths"])
    x, y = formats.to_device(g["x"]), shifted(g["y"])
    z = shifted(np.zeros(hell["rows"]))
    p = lambda t: C.c_void_p(t.data_ptr())
    capi.hellspmv["D"](gpu, p(z), p(y), 1.0, p(cM), p(rP), 32, p(ho), p(rS), None, 0, hell["rows"], p(x), 0.5, 0)
    torch.cuda.synchronize()
    assert z.cpu().numpy().tobytes() == want.tobytes()


def test_degenerate_shapes(gpu):
    import torch
    from spgpu_amd import formats
    # no entries at all: z = beta*y (alpha*0), and beta == 0 gives zeros
    g = _load("empty_d")
    _, ell, hell, hdia = _mats(g)
    y = np.linspace(-1, 1, 40)
    x = np.ones(40)
    for mat, orc in ((formats.DeviceHell(hell), lambda yy, b: O.hell_spmv(hell, x, yy, 2.0, b)),
                     (formats.DeviceEll(ell), lambda yy, b: O.ell_spmv(ell, x, yy, 2.0, b)),
                     (formats.DeviceHdia(hdia), lambda yy, b: O.hdia_spmv(hdia, x, yy, 2.0, b))):
        assert np.array_equal(_run(gpu, mat, x, y, 2.0, -0.5), orc(y, -0.5))
        assert np.array_equal(_run(gpu, mat, x, None, 2.0, 0.0), np.zeros(40))
    # a single row, double complex
    g = _load("onerow_z")
    _, ell, hell, hdia = _mats(g)
    x = (np.arange(5) + 1j * np.arange(5)[::-1]).astype(np.complex128)
    want = O.hell_spmv(hell, x, None, 1.0, 0.0, phases=2)
    assert _run(gpu, formats.DeviceHell(hell), x, None, 1.0, 0.0).tobytes() == want.tobytes()
    assert _run(gpu, formats.DeviceEll(ell), x, None, 1.0, 0.0).tobytes() == want.tobytes()
    assert _run(gpu, formats.DeviceHdia(hdia), x, None, 1.0, 0.0).tobytes() == O.hdia_spmv(hdia, x, None, 1.0, 0.0).tobytes()
    # rows == 0 is a no-op
    from spgpu_amd import capi
    capi.hellspmv["D"](gpu, None, None, 1.0, None, None, 32, None, None, None, 0, 0, None, 0.0, 0)
    torch.cuda.synchronize()


def test_ctest_program(gpu):
    """The reference's ctest.c:105-149 sequence: ELL then HELL, alpha 2, beta -3, dot(z,z) printed."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, m, r, c, v = synth.ctest_matrix(np.float32)
    ell = formats.coo_to_ell(n, r, c, v)
    hell = formats.ell_to_hell(ell, 32)
    g = _load("ctest_s")
    dx, dy = formats.to_device(g["x"]), formats.to_device(g["y"])
    dz = torch.empty_like(dy)
    dots = []
    for mat in (formats.DeviceEll(ell), formats.DeviceHell(hell)):
        mat.spmv(gpu, dz, dy, 2.0, dx, -3.0, avg_nnz=ell["max_row"])
        dots.append(capi.dot["S"](gpu, n, C.c_void_p(dz.data_ptr()), C.c_void_p(dz.data_ptr())))
        want = 4.0 * g["x"].astype(np.float64) - 3.0 * g["y"].astype(np.float64)
        assert np.max(np.abs(dz.cpu().numpy() - want)) <= 1e-5
    assert dots[0] == dots[1]
    assert abs(dots[0] - float(np.sum(want * want))) <= 1e-4 * float(np.sum(want * want))


def test_handle_and_streams(gpu):
    """core.c:11-80 behaviour: device properties cached, SetStream(0) restores the default stream."""
    import torch
    from spgpu_amd import capi
    h = gpu.contents
    assert h.warpSize == 64 and h.multiProcessorCount >= 1 and h.maxThreadsPerBlock == 1024
    assert h.currentStream == h.defaultStream and h.defaultStream
    s = C.c_void_p()
    capi.spgpuStreamCreate(gpu, C.byref(s))
    assert s.value
    capi.spgpuSetStream(gpu, s)
    assert capi.spgpuGetStream(gpu) == s.value
    capi.spgpuSetStream(gpu, None)
    assert capi.spgpuGetStream(gpu) == gpu.contents.defaultStream
    capi.spgpuStreamDestroy(s)
    # a torch stream can carry the library's work
    ts = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(ts.cuda_stream))
    a = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    z = torch.empty_like(a)
    torch.cuda.synchronize()   # `a` was filled on torch's stream; `ts` does not wait for it by itself
    with torch.cuda.stream(ts):
        capi.axpby["D"](gpu, C.c_void_p(z.data_ptr()), 1000, 0.0, None, 2.0, C.c_void_p(a.data_ptr()))
    ts.synchronize()
    assert torch.equal(z, 2.0 * a)
    capi.spgpuSetStream(gpu, None)


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("fmt", ["hell", "ell"])
@pytest.mark.parametrize("n,hack,base", [(1, 32, 0), (77, 32, 1), (5000, 64, 0), (40_003, 32, 0), (700_001, 32, 1)])
def test_sweep_form_is_the_one_phase_order(gpu, letter, fmt, n, hack, base):
    """SPGPU_SPMV_FORM_SWEEP (tuning.h): 32 rows per lane carried through the columns in step.  fp32 and complex fp64: a row's
    products added in ascending k = the oracle with one phase, bit for bit.  The 8-byte types: the bits of their DEFAULT kernel
    (one phase too, and the last rows of a 128-row group finished by the whole wavefront exactly where that kernel does it) --
    which is what lets AUTO pick the form.  Ragged rows incl. empty ones and rows long enough for the tail, rows not a multiple
    of the pack, beta != 0, in place."""
    import torch
    from spgpu_amd import capi, formats, synth
    lengths = np.minimum(synth.power_law_lengths(n, mean=7.0, max_len=90, seed=n), n)
    lengths[2::9] = 0
    _, _, r, c, v = synth.random_rows_coo(n, n, lengths, seed=n + 1, letter=letter, base=base)
    ell = formats.coo_to_ell(n, r, c, v, coo_base=base, ell_base=base)
    hell = formats.ell_to_hell(ell, hack)
    mat = formats.DeviceHell(hell) if fmt == "hell" else formats.DeviceEll(ell)
    x, y = synth.values_for(letter, 3, n), synth.values_for(letter, 4, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_SWEEP)
    try:
        for alpha, beta, in_place in ((1.0, 0.0, False), (-0.75, 0.5, False), (2.0, 1.0, True)):
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            mat.spmv(gpu, dz, dz if in_place else (dy if beta != 0 else None), alpha, dx, beta)
            torch.cuda.synchronize()
            assert capi.spgpuGetLastSpmvForm(gpu) == capi.FORM_SWEEP
            oracle = O.hell_spmv if fmt == "hell" else O.ell_spmv
            host = hell if fmt == "hell" else ell
            if letter in "DC":
                want = O.default_spmv(host, x, y if beta != 0 else None, alpha, beta)
            else:
                want = oracle(host, x, y if beta != 0 else None, alpha, beta, phases=1)
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (alpha, beta, in_place)
            if letter in "DC":   # and the default kernel itself says the same
                capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER)
                dg = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
                mat.spmv(gpu, dg, dg if in_place else (dy if beta != 0 else None), alpha, dx, beta)
                torch.cuda.synchronize()
                capi.spgpuSetSpmvForm(gpu, capi.FORM_SWEEP)
                assert torch.equal(dg.view(torch.uint8), dz.view(torch.uint8)), (alpha, beta, in_place)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("letter", ["D", "C"])
@pytest.mark.parametrize("hint", [1, 5, 8])
def test_short_row_hint_same_bits(gpu, letter, hint):
    """avgNnzPerRow in 1 .. 8 selects the kernel without a prefetch ring for the 8-byte types (ellpack_spmv.hip launchLean; the
    hint is the reference's own, hell_spmv_base_template.cuh:306-325, and only an average): on rows of 0 .. 300 entries, empty
    rows, rows longer than a stage and long enough for the whole-wave tail, HELL and ELL (ELL with maxNnzPerRow <= 16 takes the
    kernel, the long one keeps the prefetching kernel), beta != 0 and in place -- the bytes of the default kernel (hint 0) and
    of the oracle in its order."""
    from spgpu_amd import formats, synth
    n = 5000 + 3
    rng = np.random.default_rng(hint + (7 if letter == "C" else 0))
    for longest in (300, 14):
        lengths = np.minimum(rng.zipf(1.4, size=n), longest).astype(np.int64)
        lengths[rng.integers(0, n, 200)] = 0
        lengths[[1, n // 2, n - 1]] = [longest, longest - 1, longest]
        _, _, r, c, v = synth.random_rows_coo(n, n, lengths, seed=11 + hint, letter=letter)
        ell = formats.coo_to_ell(n, r, c, v)
        hell = formats.ell_to_hell(ell, 32)
        x, y = synth.values_for(letter, 3, n), synth.values_for(letter, 4, n)
        for mat, dev in ((hell, formats.DeviceHell(hell)), (ell, formats.DeviceEll(ell))):
            for alpha, beta, in_place in ((1.0, 0.0, False), (-0.5, 1.5, False), (2.0, -1.0, True)):
                import torch
                dx, dy = formats.to_device(x), formats.to_device(y)
                outs = []
                for avg in (0, hint):
                    dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
                    dev.spmv(gpu, dz, dz if in_place else (dy if beta != 0 else None), alpha, dx, beta, avg_nnz=avg)
                    torch.cuda.synchronize()
                    outs.append(dz.cpu().numpy())
                want = O.default_spmv(mat, x, y if beta != 0 else None, alpha, beta)
                assert outs[0].tobytes() == want.tobytes(), (longest, "hell" if mat is hell else "ell", alpha, beta)
                assert outs[1].tobytes() == want.tobytes(), (longest, "hell" if mat is hell else "ell", alpha, beta, "hint")
