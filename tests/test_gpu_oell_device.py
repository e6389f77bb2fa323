"""GPU: rows ordered by length in HBM (include/spgpu/oell_device.h) and the x-tile form of the ELL/HELL SpMV that the
ordered matrices are run with (spgpuSetSpmvForm, include/spgpu/tuning.h).

Order: byte-identical to the host oellOrder, which for one window is the reference's ellToOell order
(tests/test_oell_order.py pins that against the reference's own object).  SpMV: every form against the oracle in
the kernel's summation order, bit for bit, and against the extended-precision fixtures within the north_star bound."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _device_order(gpu, lengths, window, long_rows, aligned=False):
    import torch
    from spgpu_amd import capi
    n = int(lengths.size)
    rs = torch.from_numpy(np.ascontiguousarray(lengths, np.int32)).cuda() if n else torch.zeros(1, dtype=torch.int32, device="cuda")
    work = torch.empty(max(capi.spgpuOellOrderWorkBytes(n), 256), dtype=torch.uint8, device="cuda")
    r_idx = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    dst = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    call = capi.spgpuOellOrderAlignedDevice if aligned else capi.spgpuOellOrderDevice
    assert call(gpu, _dp(r_idx), _dp(dst), _dp(rs), n, window, long_rows, _dp(work)) == capi.SPGPU_SUCCESS
    torch.cuda.synchronize()
    return r_idx[:n].cpu().numpy(), dst[:n].cpu().numpy()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 1000, 70001])
@pytest.mark.parametrize("window,long_rows", [(0, 0), (32, 0), (0, 6), (512, 6), (64, 1), (2048, 20), (100000, 3)])
def test_aligned_order_equals_host_order(gpu, n, window, long_rows):
    """spgpuOellOrderAlignedDevice against the host oellOrderAligned (tests/test_oell_order.py pins that to its definition)."""
    from spgpu_amd import formats
    rng = np.random.default_rng(n + 13 * window + long_rows)
    lengths = np.minimum(rng.zipf(1.6, size=n), 60).astype(np.int32)
    want_idx, want_len = formats.oell_order(lengths, window, long_rows, aligned=True)
    got_idx, got_len = _device_order(gpu, lengths, window, long_rows, aligned=True)
    assert got_idx.tobytes() == want_idx.tobytes()
    assert got_len.tobytes() == want_len.tobytes()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 1000, 70001])
@pytest.mark.parametrize("window,long_rows", [(0, 0), (32, 0), (4096, 0), (0, 6), (512, 6), (100000, 0)])
def test_order_equals_host_order(gpu, n, window, long_rows):
    from spgpu_amd import formats
    rng = np.random.default_rng(n + 17 * window + long_rows)
    lengths = np.minimum(rng.zipf(1.6, size=n), 60).astype(np.int32)
    want_idx, want_len = formats.oell_order(lengths, window, long_rows)
    got_idx, got_len = _device_order(gpu, lengths, window, long_rows)
    assert got_idx.tobytes() == want_idx.tobytes()
    assert got_len.tobytes() == want_len.tobytes()


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("window,long_rows", [(0, 0), (64, 8)])
def test_ell_to_oell_device_equals_host(gpu, letter, window, long_rows):
    """spgpuEllToOellDevice against the host ellToOell (window 0: the reference's call) / the host order + copy."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 777
    lengths = np.minimum(np.random.default_rng(3).zipf(1.5, size=n), 30)
    _, _, r, c, v = synth.random_rows_coo(n, 900, lengths, seed=4, letter=letter)
    ell = formats.coo_to_ell(n, r, c, v)
    if (window, long_rows) == (0, 0):
        want, want_idx = formats.ell_to_oell(ell)
    else:
        want_idx, want_len = formats.oell_order(ell["row_lengths"], window, long_rows)
        vals, idx = np.zeros_like(ell["values"]), np.zeros_like(ell["indices"])
        for i, src in enumerate(want_idx):
            for k in range(ell["row_lengths"][src]):
                vals[i + k * ell["pitch"]] = ell["values"][src + k * ell["pitch"]]
                idx[i + k * ell["pitch"]] = ell["indices"][src + k * ell["pitch"]]
        want = dict(ell, values=vals, indices=idx, row_lengths=want_len)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sv, si, rs = d(ell["values"]), d(ell["indices"]), d(ell["row_lengths"])
    dv, di = torch.zeros_like(sv), torch.zeros_like(si)
    r_idx, dst_rs = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    work = torch.empty(capi.spgpuOellOrderWorkBytes(n), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert capi.spgpuEllToOellDevice(gpu, _dp(r_idx), _dp(dv), _dp(di), _dp(dst_rs), _dp(sv), _dp(si), _dp(rs), ell["pitch"],
                                     ell["pitch"], n, capi.TYPE_CODE[letter], window, long_rows, _dp(work)) == capi.SPGPU_SUCCESS
    torch.cuda.synchronize()
    assert r_idx.cpu().numpy().tobytes() == np.ascontiguousarray(want_idx, np.int32).tobytes()
    assert dst_rs.cpu().numpy().tobytes() == np.ascontiguousarray(want["row_lengths"], np.int32).tobytes()
    assert dv.cpu().numpy().tobytes() == want["values"].tobytes()
    assert di.cpu().numpy().tobytes() == want["indices"].tobytes()


def _ordered_case(gpu, n, letter, window, long_rows, pattern="near", near=300, hack=32, base=0):
    """A ragged matrix through the whole device route; returns (device HELL dict, host COO, host HELL built on the host
    from the permuted COO)."""
    import torch
    from spgpu_amd import formats, synth
    lengths = np.minimum(synth.power_law_lengths(n, 12.0, 400, 9), 2 * near)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, near, letter, seed=7)
    if base:
        rows_t, cols_t = rows_t + base, cols_t + base
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, hack, window, long_rows, coo_base=base,
                                           hell_base=base)
    return h, (rows_t.cpu().numpy(), cols_t.cpu().numpy(), vals_t.cpu().numpy()), lengths


@pytest.mark.parametrize("window,long_rows", [(0, 0), (256, 0), (512, 40)])
@pytest.mark.parametrize("base", [0, 1])
def test_coo_route_builds_the_ordered_hell(gpu, window, long_rows, base):
    """Order + permuted COO rows + spgpuCooToHellDevice == the host converters on the host-permuted COO, byte for byte;
    row i of the result is row rIdx[i] of the original."""
    from spgpu_amd import formats
    n = 5000
    h, (r, c, v), lengths = _ordered_case(gpu, n, "D", window, long_rows, base=base)
    want_idx, want_len = formats.oell_order(lengths, window, long_rows)
    assert h["rIdx"].cpu().numpy().tobytes() == want_idx.tobytes()
    inverse = np.empty(n, np.int64)
    inverse[want_idx] = np.arange(n)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, inverse[r - base] + base, c, v, coo_base=base, ell_base=base), 32)
    assert h["rS"][:n].cpu().numpy().tobytes() == want_len.tobytes() == hell["row_lengths"].tobytes()
    assert h["hack_offsets"].cpu().numpy().tobytes() == hell["hack_offsets"].tobytes()
    assert h["cM"][:h["slots"]].cpu().numpy().tobytes() == hell["values"].tobytes()
    assert h["rP"][:h["slots"]].cpu().numpy().tobytes() == hell["indices"].tobytes()


def tile_shape(letter, shape, deep=True):
    """spmv_tail parameters of the x-tile kernels; with a row order the deep split is on (cap 128)."""
    return O.slab_shape(letter, "xtile", shape, deep_cap=O.DEEP_CAP if deep else 0)


@pytest.mark.skipif("not config._lab_build", reason="non-default kernel shape: -DSPGPU_TUNING_VARIANTS build")     # SPGPU_RAGGED=0: the deep split with fixed rows per wavefront
@pytest.mark.parametrize("shape", [0, 1, 2, 3])
@pytest.mark.parametrize("letter", ["S", "D"])
@pytest.mark.parametrize("window,long_rows,pattern", [(512, 40, "near"), (0, 0, "near"), (1024, 0, "random")])
def test_tile_form_through_ridx_bit_exact(gpu, tuning, letter, shape, window, long_rows, pattern):
    """Ordered ragged matrix + rIdx, x-tile form: equals the oracle in the kernel's order bit for bit (columns inside the
    tile come from LDS, the others from global memory -- "random" and the global sort exercise the mix), beta != 0 and
    in-place included; and equals the SAME product computed from the unordered matrix within the north_star bound."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_X_TILE_SHAPE=shape, SPGPU_RAGGED=0)
    n = 6000 + 13
    h, (r, c, v), lengths = _ordered_case(gpu, n, letter, window, long_rows, pattern=pattern)
    sub = dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
               hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=32, row_lengths=h["rS"][:n].cpu().numpy(), base=0)
    r_idx = h["rIdx"].cpu().numpy()
    x = synth.values_for(letter, 11, n)
    y = synth.values_for(letter, 12, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_XTILE)
    try:
        for alpha, beta, in_place in ((1.0, 0.0, False), (-0.75, 0.5, False), (2.0, 1.0, True)):
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            capi.hellspmv[letter](gpu, _dp(dz), _dp(dz if in_place else (dy if beta != 0 else None)), capi.scalar(letter, alpha),
                                  _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]), 12, n,
                                  _dp(dx), capi.scalar(letter, beta), 0)
            torch.cuda.synchronize()
            got = dz.cpu().numpy()
            want = O.spmv_tail(sub, x, y if beta != 0 else None, alpha, beta, r_idx=r_idx, **tile_shape(letter, shape))
            assert got.tobytes() == want.tobytes(), (alpha, beta, in_place)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    # independent of any summation order: alpha*A*x from the ORIGINAL triplets in extended precision
    exact = np.zeros(n, np.longdouble)
    scale = np.zeros(n, np.longdouble)
    np.add.at(exact, r, v.astype(np.longdouble) * x[c].astype(np.longdouble))
    np.add.at(scale, r, np.abs(v.astype(np.longdouble) * x[c].astype(np.longdouble)))
    dz = torch.empty(n, dtype=dx.dtype, device="cuda")
    capi.spgpuSetSpmvForm(gpu, capi.FORM_XTILE)
    try:
        capi.hellspmv[letter](gpu, _dp(dz), None, capi.scalar(letter, 1.0), _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]),
                              _dp(h["rS"]), _dp(h["rIdx"]), 12, n, _dp(dx), capi.scalar(letter, 0.0), 0)
        torch.cuda.synchronize()
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    tol = {"S": 1e-4, "D": 1e-6}[letter]
    assert np.all(np.abs(dz.cpu().numpy().astype(np.longdouble) - exact) <= tol * scale + np.finfo(np.float64).tiny)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))
                                        if os.path.basename(p)[:-4] not in ("empty_d", "onerow_z")))
@pytest.mark.parametrize("form", ["gather", "strips", "xtile"])
def test_every_form_on_the_fixtures(gpu, name, form):
    """The hint changes how x is fetched, never the sums: all four types, both formats, against the fixtures'
    extended-precision vectors, and gather == strips bit for bit (same kernel shape)."""
    import torch
    from spgpu_amd import capi, formats
    from test_gpu_spmv import _load, _mats, _run, _within
    g = _load(name)
    letter, ell, hell, _ = _mats(g)
    alpha, beta = g["alpha"][()], g["beta"][()]
    y = g["y"] if beta != 0 else None
    capi.spgpuSetSpmvForm(gpu, {"gather": capi.FORM_GATHER, "strips": capi.FORM_STRIPS, "xtile": capi.FORM_XTILE}[form])
    try:
        assert capi.spgpuGetSpmvForm(gpu) != capi.FORM_AUTO
        for mat in (formats.DeviceHell(hell), formats.DeviceEll(ell), formats.DeviceEll(ell, with_row_sizes=False)):
            z = _run(gpu, mat, g["x"], y, alpha, beta)
            assert _within(z, g, letter) <= 1.0
            if form != "xtile" and not (isinstance(mat, formats.DeviceEll) and mat.rS is None):
                assert z.tobytes() == O.default_spmv(hell if isinstance(mat, formats.DeviceHell) else ell, g["x"], y, alpha, beta).tobytes()
            elif form == "xtile" and letter != "Z" and not (isinstance(mat, formats.DeviceEll) and mat.rS is None):
                want = O.spmv_tail(hell if isinstance(mat, formats.DeviceHell) else ell, g["x"], y, alpha, beta,
                                   **tile_shape(letter, 0, deep=False))
                assert z.tobytes() == want.tobytes()
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


def test_one_column_matrix_strips_form(gpu):
    """x shorter than a strip (1 column): the strip-capable kernel must not read past it (an absent strip loads from the
    coefficient array instead).  Result checked; the out-of-bounds read itself would need a sanitizer to see."""
    import torch
    from spgpu_amd import capi, formats
    n = 300
    r = np.arange(n, dtype=np.int32)
    c = np.zeros(n, np.int32)
    v = np.linspace(1.0, 2.0, n)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
    x = np.array([3.0])
    capi.spgpuSetSpmvForm(gpu, capi.FORM_STRIPS)
    try:
        from test_gpu_spmv import _run
        z = _run(gpu, formats.DeviceHell(hell), x, None, 1.0, 0.0)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    assert z.tobytes() == (v * 3.0).tobytes()


@pytest.mark.skipif("not config._lab_build", reason="non-default kernel shape: -DSPGPU_TUNING_VARIANTS build")     # SPGPU_RAGGED=0
@pytest.mark.parametrize("cap", [16, 128])
@pytest.mark.parametrize("form", ["gather", "xtile"])
@pytest.mark.parametrize("name", ["powerlaw_s_b1_h64", "powerlaw_d_b0_h32", "powerlaw_c_b0_h32", "powerlaw_z_b1_h64"])
def test_deep_split_every_type_bit_exact(gpu, tuning, name, form, cap):
    """The deep split on its own (SPGPU_DEEP_SPLIT=1, no row order): sub-groups deeper than the cap are finished by
    deepSpmvKernel; HELL and ELL, all four types, hack 32 and 64, base 0 and 1, beta == 0 and != 0, against the oracle's
    restatement of that order bit for bit and against the extended-precision fixtures."""
    from spgpu_amd import capi, formats
    from test_gpu_spmv import _load, _mats, _run, _within
    tuning(SPGPU_DEEP_SPLIT=1, SPGPU_DEEP_CAP=cap, SPGPU_RAGGED=0)
    g = _load(name)
    letter, ell, hell, _ = _mats(g)
    shape = O.slab_shape(letter, form, 0, deep_cap=cap)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_XTILE if form == "xtile" else capi.FORM_GATHER)
    try:
        for beta in (0.0, g["beta"][()] if g["beta"][()] != 0 else 0.5):
            y = g["y"] if beta != 0 else None
            for mat, host in ((formats.DeviceHell(hell), hell), (formats.DeviceEll(ell), ell)):
                z = _run(gpu, mat, g["x"], y, g["alpha"][()], beta)
                assert z.tobytes() == O.spmv_tail(host, g["x"], y, g["alpha"][()], beta, **shape).tobytes()
                if beta == g["beta"][()]:
                    assert _within(z, g, letter) <= 1.0
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


def test_deep_queue_overflow_stays_correct(gpu, tuning):
    """More deep sub-groups than the deep list holds (cap 1 on 1.2 M rows of length 3: 37 500 sub-groups, capacity 8 192).
    Round 4: a sub-group that finds the list full is worked off by its own block behind its stream, in the chunks and the
    order of the deep path (deep_rows.hip.h) -- so WHICH sub-groups were surplus (that depends on scheduling) no longer shows in
    the result: every call gives the oracle's bits.  The handle still counts such calls (include/spgpu/tuning.h)."""
    import torch
    from spgpu_amd import capi, synth
    tuning(SPGPU_DEEP_SPLIT=1, SPGPU_DEEP_CAP=1)
    n = 1_200_000
    h = synth.hell_uniform_on_device(n, 3, "banded", "D", 32, seed=3)
    x = synth.device_vector(n, "D", 5)
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    want = O.spmv_tail(synth.hell_rows_to_host(h, 0, n), x.cpu().numpy(), None, 1.0, 0.0, **O.slab_shape("D", "ragged", deep_cap=1))
    before = capi.spgpuDeepListOverflows(gpu)
    for call in range(2):
        z.fill_(float("nan"))
        capi.hellspmv["D"](gpu, _dp(z), None, capi.scalar("D", 1.0), _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]),
                           _dp(h["rS"]), None, 3, n, _dp(x), capi.scalar("D", 0.0), 0)
        torch.cuda.synchronize()
        assert z.cpu().numpy().tobytes() == want.tobytes(), call
        assert capi.spgpuDeepListOverflows(gpu) == before + call + 1      # the handle says so (include/spgpu/tuning.h)
    # a call whose list does not overflow leaves the count alone
    tuning(SPGPU_DEEP_SPLIT=1, SPGPU_DEEP_CAP=2)
    capi.hellspmv["D"](gpu, _dp(z[:6400]), None, capi.scalar("D", 1.0), _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]),
                       _dp(h["rS"]), None, 3, 6400, _dp(x), capi.scalar("D", 0.0), 0)
    torch.cuda.synchronize()
    assert capi.spgpuDeepListOverflows(gpu) == before + 2


@pytest.mark.parametrize("shape,form", [(0, "auto"), (0, "gather"), (4, "auto"), (5, "auto")] + [pytest.param(k, "auto", marks=pytest.mark.skipif("not config._lab_build", reason="non-default kernel shape: -DSPGPU_TUNING_VARIANTS build")) for k in (1, 2, 3)])
@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("window,long_rows,pattern,hack", [(512, 40, "near", 32), (0, 0, "near", 64), (1024, 0, "random", 32), (256, 100, "near", 96)])
def test_ragged_kernel_through_ridx_bit_exact(gpu, tuning, letter, shape, form, window, long_rows, pattern, hack):
    """The queue-driven kernel a row order selects (ragged_spmv.hip.h) with the deep split behind it: all four types,
    hack sizes 32 / 64 / 96, every workgroup shape (4 and 5: results staged in LDS by destination and written in whole lines;
    4: 2 048 rows per workgroup), tile and gathers, beta != 0 and in place; against the oracle in the kernel's order (2 * rows-per-lane phases, items of 64 columns beyond the cap) bit for bit."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_RAGGED_SHAPE=shape)
    n = 7000 + 5
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, 12.0, 400, 9), 600)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, 300, real, seed=7)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, hack, window, long_rows)
    sub = dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
               hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=hack, row_lengths=h["rS"][:n].cpu().numpy(), base=0)
    r_idx = h["rIdx"].cpu().numpy()
    x, y = synth.values_for(letter, 11, n), synth.values_for(letter, 12, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    shape_args = O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER if form == "gather" else capi.FORM_AUTO)
    try:
        for alpha, beta, in_place in ((1.0, 0.0, False), (-0.75, 0.5, False), (2.0, 1.0, True)):
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            capi.hellspmv[letter](gpu, _dp(dz), _dp(dz if in_place else (dy if beta != 0 else None)), capi.scalar(letter, alpha),
                                  _dp(h["cM"]), _dp(h["rP"]), hack, _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]), 12, n,
                                  _dp(dx), capi.scalar(letter, beta), 0)
            torch.cuda.synchronize()
            want = O.spmv_tail(sub, x, y if beta != 0 else None, alpha, beta, r_idx=r_idx, **shape_args)
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (alpha, beta, in_place)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("pattern", ["near", "random"])
def test_aligned_order_every_type_auto_bit_exact(gpu, letter, pattern):
    """Rows ordered by spgpuOellOrderAlignedDevice (windows of 2 048, rows > 60 set aside), AUTO: the probe finds that the
    kernel's 2 048-row blocks are the windows (answer 6) and the later calls run the 2 048-row shape (4- and 8-byte types;
    complex fp64 keeps the default shape) -- every call, before and after the answer, equals the oracle bit for bit, with the
    set-aside rows' columns beyond SPGPU_DEEP_KEEP in the deep kernels."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 9 * 2048 + 77
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, 14.0, 900, 8), 900)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, 500, real, seed=5)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, 32, 2048, 60, aligned=True)
    want_idx, _ = formats.oell_order(lengths, 2048, 60, aligned=True)
    assert h["rIdx"].cpu().numpy().tobytes() == want_idx.tobytes()
    sub = dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
               hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=32, row_lengths=h["rS"][:n].cpu().numpy(), base=0)
    x, y = synth.values_for(letter, 21, n), synth.values_for(letter, 22, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    want = O.spmv_tail(sub, x, y, -0.5, 2.0, r_idx=want_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    for call in range(4):
        dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
        capi.hellspmv[letter](gpu, _dp(dz), _dp(dy), capi.scalar(letter, -0.5), _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]),
                              _dp(h["rS"]), _dp(h["rIdx"]), 14, n, _dp(dx), capi.scalar(letter, 2.0), 0)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == want.tobytes(), call


@pytest.mark.parametrize("split", [0, 48, 96, 150])
@pytest.mark.parametrize("letter,shape,form", [("D", 0, "auto"), ("D", 4, "auto"), ("S", 4, "auto"), ("C", 0, "gather"), ("Z", 4, "auto")])
def test_split_sub_groups_bit_exact(gpu, tuning, letter, shape, form, split):
    """SPLIT of the queue kernel (ragged_spmv.hip.h): sub-groups deeper than SPGPU_RAGGED_SPLIT columns (default: about 96) and
    not beyond the deep cap are walked as chunks by several wavefronts, the chunk sums added in chunk order -- the oracle's
    mainChunk.  0 switches it off, 48 asks for more chunks than LDS parks for 8-byte types (rounded up by the library and by
    oracle_api.ragged_split alike); window 256 puts a deep head into every workgroup, the long-row hacks are split AND deep;
    complex fp64 never splits."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_RAGGED_SHAPE=shape, SPGPU_RAGGED_SPLIT=split)
    n = 9000 + 7
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, 14.0, 700, 4), 700)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, "near", 400, real, seed=3)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, 32, 256, 300)
    sub = dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
               hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=32, row_lengths=h["rS"][:n].cpu().numpy(), base=0)
    r_idx = h["rIdx"].cpu().numpy()
    x, y = synth.values_for(letter, 11, n), synth.values_for(letter, 12, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    shape_args = O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP, split=split)
    if split == 0 or letter == "Z":
        assert shape_args["main_chunk"] == 0
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER if form == "gather" else capi.FORM_AUTO)
    try:
        for alpha, beta, in_place in ((1.0, 0.0, False), (2.0, 1.0, True)):
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            capi.hellspmv[letter](gpu, _dp(dz), _dp(dz if in_place else None), capi.scalar(letter, alpha),
                                  _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]), 14, n,
                                  _dp(dx), capi.scalar(letter, beta), 0)
            torch.cuda.synchronize()
            want = O.spmv_tail(sub, x, y if beta != 0 else None, alpha, beta, r_idx=r_idx, **shape_args)
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (alpha, beta, in_place)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("pattern,expect", [("banded", "strips"), ("near512", "xtile"), ("random", "gather")])
def test_auto_form_settles_on_the_matrix(gpu, pattern, expect):
    """AUTO: sample wavefronts of every launch report what they saw and the next launch on the same arrays uses it --
    consecutive columns -> strips, columns inside a window an LDS tile holds -> x-tile, scattered -> gathers.  The sums
    do not depend on the form (uniform rows: every shape adds a row's products in ascending k)."""
    import torch
    from spgpu_amd import capi, synth
    n = 400_000
    h = synth.hell_uniform_on_device(n, 32, pattern, "D", 32, seed=3)
    x = synth.device_vector(n, "D", 5)
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    results = []
    for _ in range(10):     # the arrays may sit where the previous case's did: the forms that do not report are probed every 4th call
        capi.hellspmv["D"](gpu, _dp(z), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), None, 32, n,
                           _dp(x), 0.0, 0)
        torch.cuda.synchronize()
        results.append(z.cpu().numpy().tobytes())
    assert capi.spgpuGetLastSpmvForm(gpu) == {"strips": capi.FORM_STRIPS, "xtile": capi.FORM_XTILE, "gather": capi.FORM_GATHER}[expect]
    assert len(set(results)) == 1
    sub = synth.hell_rows_to_host(h, 0, 2048)
    assert z[:2048].cpu().numpy().tobytes() == O.default_spmv(sub, x.cpu().numpy(), None, 1.0, 0.0).tobytes()


def test_auto_takes_the_sweep_form_for_a_large_scattered_matrix(gpu, tuning):
    """AUTO and the SWEEP form (include/spgpu/tuning.h): the probe that looks at a matrix in the gather form again also finds out
    whether its columns reach over all of x, ascend inside the rows and the rows are about equally long; two of three samples
    saying so select the SWEEP form for the 8-byte types on matrices of 2 Mi rows and more -- the same bits as the gather kernel
    and as the oracle in the default order.  The analysis call says the same; a 65 536-column window stays with the gathers; and
    SPGPU_AUTO_SWEEP=0 keeps AUTO out of it."""
    import torch
    from spgpu_amd import capi, synth
    n, nnz = 4 * 1024 * 1024 + 4096, 16
    x = synth.device_vector(n, "D", 5)
    z = torch.empty(n, dtype=torch.float64, device="cuda")

    def run(h, calls):
        forms, results = [], []
        for _ in range(calls):
            capi.hellspmv["D"](gpu, _dp(z), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), None, nnz, n,
                               _dp(x), 0.0, 0)
            torch.cuda.synchronize()
            forms.append(capi.spgpuGetLastSpmvForm(gpu))
            results.append(z[:4096].cpu().numpy().tobytes() + z[-4096:].cpu().numpy().tobytes())
        return forms, results

    capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    h = synth.hell_uniform_on_device(n, nnz, "random", "D", 32, seed=3)
    torch.cuda.synchronize()
    forms, results = run(h, 8)
    assert forms[-1] == capi.FORM_SWEEP and capi.FORM_SWEEP in forms[:6], forms
    assert len(set(results)) == 1
    whole = z.clone()
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER)
    try:
        run(h, 1)
        assert torch.equal(whole, z)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    sub = synth.hell_rows_to_host(h, 0, 2048)
    assert whole[:2048].cpu().numpy().tobytes() == O.default_spmv(sub, x.cpu().numpy(), None, 1.0, 0.0).tobytes()
    assert capi.spgpuHellSpmvForm(gpu, capi.TYPE_CODE["D"], _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), n, 0) == capi.FORM_SWEEP
    assert capi.spgpuHellSpmvForm(gpu, capi.TYPE_CODE["S"], _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), n, 0) == capi.FORM_GATHER
    # the knob
    tuning(SPGPU_AUTO_SWEEP=0)
    forms, _ = run(h, 6)
    assert capi.FORM_SWEEP not in forms, forms
    tuning(SPGPU_AUTO_SWEEP=1)
    del h
    # columns inside a window: scattered for a tile, but nowhere near all of x.  (The arrays may sit where the scattered matrix'
    # did: AUTO then starts from what it knew of that one and looks again within four calls.)
    h = synth.hell_uniform_on_device(n, nnz, "window", "D", 32, seed=4)
    torch.cuda.synchronize()
    forms, results = run(h, 12)
    assert capi.FORM_SWEEP not in forms[6:] and forms[-1] == capi.FORM_GATHER, forms
    assert len(set(results)) == 1


def test_auto_sweep_ell_complex(gpu):
    """The same choice for ELL and the other 8-byte type (complex fp32): AUTO settles on the SWEEP form for the scattered matrix, and
    its z is the gather kernel's, byte for byte."""
    import torch
    from spgpu_amd import capi, synth
    n, nnz = 4 * 1024 * 1024 + 4096, 16
    h = synth.hell_uniform_on_device(n, nnz, "random", "C", 32, seed=9)
    cM = h["cM"].view(n // 32, nnz, 32).permute(1, 0, 2).reshape(-1).contiguous()     # the same slots as ELL: pitch = n
    rP = h["rP"].view(n // 32, nnz, 32).permute(1, 0, 2).reshape(-1).contiguous()
    del h
    x = synth.device_vector(n, "C", 5)
    y = synth.device_vector(n, "C", 6)
    z = torch.empty_like(x)
    alpha, beta = capi.scalar("C", 0.75), capi.scalar("C", -0.5)
    torch.cuda.synchronize()
    call = lambda: capi.ellspmv["C"](gpu, _dp(z), _dp(y), alpha, _dp(cM), _dp(rP), n, n, None, None, nnz, nnz, n, _dp(x), beta, 0)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    forms = []
    for _ in range(8):
        call()
        torch.cuda.synchronize()
        forms.append(capi.spgpuGetLastSpmvForm(gpu))
    assert forms[-1] == capi.FORM_SWEEP, forms
    swept = z.clone()
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER)
    try:
        call()
        torch.cuda.synchronize()
        assert capi.spgpuGetLastSpmvForm(gpu) == capi.FORM_GATHER
        assert torch.equal(torch.view_as_real(swept).view(torch.int32), torch.view_as_real(z).view(torch.int32))
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    assert capi.spgpuEllSpmvForm(gpu, capi.TYPE_CODE["C"], _dp(rP), n, None, nnz, n, 0) == capi.FORM_SWEEP


def _deep_case(gpu, n, seed):
    """A matrix whose ordered form has deep sub-groups (rows up to 900 entries, cap 256), as a device HELL + host copy."""
    import torch
    from spgpu_amd import formats, synth
    lengths = np.minimum(synth.power_law_lengths(n, 14.0, 900, seed), 900)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, "near", 450, "D", seed=seed)
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, "D", 32, 512, 200)
    sub = dict(letter="D", rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
               hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=32, row_lengths=h["rS"][:n].cpu().numpy(), base=0)
    assert int(h["rS"].max()) > O.DEEP_CAP
    return h, sub


def test_deep_list_in_a_replayed_graph(gpu):
    """The deep list's header is zeroed by the finish kernel's last workgroup: an SpMV with deep rows captured ONCE and
    replayed five times (new x every time) gives the oracle's bits every time."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 9000
    h, sub = _deep_case(gpu, n, 21)
    r_idx = h["rIdx"].cpu().numpy()
    shape_args = O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP)
    dx = formats.to_device(synth.values_for("D", 1, n))
    dz = torch.zeros(n, dtype=torch.float64, device="cuda")
    call = lambda: capi.hellspmv["D"](gpu, _dp(dz), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                      _dp(h["rIdx"]), 14, n, _dp(dx), 0.0, 0)
    side = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()   # dz was zeroed on torch's stream; `side` does not wait for it by itself
    try:
        with torch.cuda.stream(side):
            call()                               # warm-up outside the capture (the stream's list exists since spgpuSetStream)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            call()
        for rep in range(5):
            x = synth.values_for("D", 100 + rep, n)
            dx.copy_(formats.to_device(x))
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            want = O.spmv_tail(sub, x, None, 1.0, 0.0, r_idx=r_idx, **shape_args)
            assert dz.cpu().numpy().tobytes() == want.tobytes(), rep
    finally:
        capi.spgpuSetStream(gpu, None)


def test_two_handles_on_two_threads_with_deep_rows(gpu):
    """Every handle owns its deep list: two host threads, a handle and a stream each, 20 SpMVs with deep rows each at the
    same time -- both get the oracle's bits on every call."""
    import threading
    import torch
    from spgpu_amd import capi, formats, synth
    cases = []
    for seed, n in ((31, 7000), (32, 11000)):
        h, sub = _deep_case(gpu, n, seed)
        x = synth.values_for("D", seed, n)
        want = O.spmv_tail(sub, x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
        cases.append((h, n, formats.to_device(x), want))
    torch.cuda.synchronize()
    failures = []

    def work(case):
        h, n, dx, want = case
        handle = capi.create_handle(0)
        stream = torch.cuda.Stream()
        capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
        dz = torch.zeros(n, dtype=torch.float64, device="cuda")
        try:
            for rep in range(20):
                capi.hellspmv["D"](handle, _dp(dz), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                   _dp(h["rIdx"]), 14, n, _dp(dx), 0.0, 0)
                stream.synchronize()
                if dz.cpu().numpy().tobytes() != want.tobytes():
                    failures.append((n, rep))
                dz.zero_()
                torch.cuda.synchronize()
        finally:
            capi.spgpuSetStream(handle, None)
            capi.spgpuDestroy(handle)

    threads = [threading.Thread(target=work, args=(case,)) for case in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures


def test_two_streams_of_one_handle_with_deep_rows(gpu):
    """The reference's SpMV shares nothing between the streams of a handle (hell_spmv_base_template.cuh:336-345; the caller
    switches streams with spgpuSetStream, core.c:64-74): two ordered power-law SpMVs queued on two streams of ONE handle,
    20 rounds without a synchronisation in between, both bit for bit the single-stream result.  Every stream the handle is
    given owns a deep list."""
    import torch
    from spgpu_amd import capi, formats, synth
    cases = []
    for seed, n in ((41, 60000), (42, 90000)):
        h, sub = _deep_case(gpu, n, seed)
        x = synth.values_for("D", seed, n)
        want = O.spmv_tail(sub, x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
        cases.append((h, n, formats.to_device(x), want, torch.zeros(n, dtype=torch.float64, device="cuda"), torch.cuda.Stream()))
    torch.cuda.synchronize()
    try:
        for rep in range(20):
            for h, n, dx, want, dz, stream in cases:        # queued back to back: the two calls overlap on the device
                capi.spgpuSetStream(gpu, C.c_void_p(stream.cuda_stream))
                capi.hellspmv["D"](gpu, _dp(dz), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                   _dp(h["rIdx"]), 14, n, _dp(dx), 0.0, 0)
            torch.cuda.synchronize()
            for h, n, dx, want, dz, stream in cases:
                assert dz.cpu().numpy().tobytes() == want.tobytes(), (n, rep)
                dz.zero_()
            torch.cuda.synchronize()
    finally:
        capi.spgpuSetStream(gpu, None)


def test_more_streams_than_deep_lists(gpu):
    """A handle keeps a deep list for 8 streams.  Eleven ordered SpMVs queued on eleven streams with nothing waited for: a
    stream beyond the eighth takes over a list whose owner has finished, or -- all of them busy -- runs the same kernel family
    without any state (planned_spmv.hip: no plan, every deep sub-group worked off by its own block).  Whichever it was, the
    bits are the ones of the list path and of the oracle: z does not depend on the handle's stream history."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 30000
    handle = capi.create_handle(0)
    h, sub = _deep_case(handle, n, 51)
    x = synth.values_for("D", 51, n)
    r_idx = h["rIdx"].cpu().numpy()
    want = O.spmv_tail(sub, x, None, 1.0, 0.0, r_idx=r_idx, **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
    dx = formats.to_device(x)
    streams = [torch.cuda.Stream() for _ in range(11)]
    outs = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    try:
        for rep in range(3):
            for stream, dz in zip(streams, outs):
                capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
                capi.hellspmv["D"](handle, _dp(dz), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                   _dp(h["rIdx"]), 14, n, _dp(dx), 0.0, 0)
            torch.cuda.synchronize()
            for k, dz in enumerate(outs):
                assert dz.cpu().numpy().tobytes() == want.tobytes(), (rep, k)
                dz.zero_()
            torch.cuda.synchronize()
        assert capi.spgpuDeepListFallbacks(handle) + capi.spgpuDeepListsRecycled(handle) + capi.plan_counts(handle)[0] > 0
    finally:
        capi.spgpuSetStream(handle, None)
        capi.spgpuDestroy(handle)


@pytest.mark.parametrize("pattern,expect", [("banded", "strips"), ("near512", "xtile"), ("random", "gather")])
@pytest.mark.parametrize("letter", ["D", "S"])
def test_analysis_call_gives_the_form_at_once(gpu, pattern, expect, letter):
    """spgpuHellSpmvForm / spgpuEllSpmvForm: the answer AUTO settles on, synchronously, for a caller who holds it --
    consecutive columns -> strips, columns inside a window an LDS tile holds -> x-tile, scattered -> gathers; the SpMV run
    in that form gives the bits of the default run."""
    import torch
    from spgpu_amd import capi, synth
    n = 200_000
    h = synth.hell_uniform_on_device(n, 32, pattern, letter, 32, seed=3)
    x = synth.device_vector(n, letter, 5)
    z, z2 = torch.empty_like(x), torch.empty_like(x)
    torch.cuda.synchronize()
    want = {"strips": capi.FORM_STRIPS, "xtile": capi.FORM_XTILE, "gather": capi.FORM_GATHER}[expect]
    form = capi.spgpuHellSpmvForm(gpu, capi.TYPE_CODE[letter], _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), n, 0)
    assert form == want
    # the same slots as ELL (uniform rows: pitch = n)
    rP_ell = h["rP"].view(n // 32, 32, 32).permute(1, 0, 2).reshape(-1).contiguous()
    torch.cuda.synchronize()
    assert capi.spgpuEllSpmvForm(gpu, capi.TYPE_CODE[letter], _dp(rP_ell), n, None, 32, n, 0) == want
    one, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.0)
    call = lambda out: capi.hellspmv[letter](gpu, _dp(out), None, one, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                             None, 32, n, _dp(x), zero, 0)
    capi.spgpuSetSpmvForm(gpu, form)
    try:
        call(z)
        torch.cuda.synchronize()
        assert capi.spgpuGetLastSpmvForm(gpu) == want
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    for _ in range(3):
        call(z2)
        torch.cuda.synchronize()
    assert torch.equal(z, z2)


def test_auto_notices_another_matrix_at_the_same_address(gpu):
    """AUTO's table is keyed by (rP, rows): when a scattered matrix is overwritten in place by a banded one, the gather
    form -- which does not report by itself -- is looked at again by the probe with every fourth call, and the
    strip form takes over; results stay the oracle's throughout."""
    import torch
    from spgpu_amd import capi, synth
    n = 100_000
    scattered = synth.hell_uniform_on_device(n, 32, "random", "D", 32, seed=3)
    banded = synth.hell_uniform_on_device(n, 32, "banded", "D", 32, seed=4)
    x = synth.device_vector(n, "D", 5)
    z = torch.empty_like(x)
    torch.cuda.synchronize()
    h = scattered
    call = lambda: capi.hellspmv["D"](gpu, _dp(z), None, 1.0, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), None, 32,
                                      n, _dp(x), 0.0, 0)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
    for _ in range(3):
        call()
        torch.cuda.synchronize()
    assert capi.spgpuGetLastSpmvForm(gpu) == capi.FORM_GATHER
    h["rP"].copy_(banded["rP"])          # another matrix, same arrays
    h["cM"].copy_(banded["cM"])
    torch.cuda.synchronize()
    forms = []
    for _ in range(40):
        call()
        torch.cuda.synchronize()
        forms.append(capi.spgpuGetLastSpmvForm(gpu))
    assert forms[-1] == capi.FORM_STRIPS and capi.FORM_STRIPS in forms[:10]
    sub = synth.hell_rows_to_host(h, 0, 2048)
    assert z[:2048].cpu().numpy().tobytes() == O.default_spmv(sub, x.cpu().numpy(), None, 1.0, 0.0).tobytes()
