"""GPU: what lies in the padding slots of the ELL / HELL arrays never reaches z.

The reference's converters leave the slots beyond a row's length as they find them (ell.c:65-78, hell.c:93-96: only the row's
entries are written) and its kernels never read them with rS given (`for j < rowSize`, hell_spmv_base_template.cuh:112-118).  In
a long-lived process those slots hold whatever lived there before, so every kernel here that loads a slab column wholesale must
select, not multiply.  Each slot (r, k) with k >= rowLength[r] gets a NaN coefficient and a random valid column; every form of
the unordered SpMV and the ordered paths (first call, planned, SPGPU_PLAN=0) still give the oracle's bytes."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from test_gpu_plan import _call, _host, _matrix

pytestmark = pytest.mark.gpu


def _poison_hell(cM, rP, lengths, offsets, hack, n):
    """NaN / random columns into the padding slots of device HELL arrays (lengths, offsets: host)."""
    import torch
    rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
    ks = np.arange(int(lengths.sum()), dtype=np.int64) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    used = np.zeros(cM.numel(), bool)
    used[offsets[rows // hack] + rows % hack + ks * hack] = True
    padding = torch.from_numpy(~used).cuda()
    count = int(padding.sum().item())
    cM[padding] = float("nan")
    rP[padding] = torch.randint(0, n, (count,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    return count


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("window,long_rows,aligned,hack", [(2048, 60, True, 32), (512, 40, False, 32), (0, 0, False, 64), (256, 100, False, 96)])
def test_ordered_paths_never_use_padding(gpu, tuning, letter, window, long_rows, aligned, hack):
    import torch
    from spgpu_amd import capi, formats, synth
    n = 6 * 2048 + 300
    h = _matrix(gpu, n, letter, window, long_rows, aligned, hack=hack, longest=900, near=500, seed=77)
    lengths = h["rS"][:n].cpu().numpy().astype(np.int64)
    offsets = h["hack_offsets"].cpu().numpy().astype(np.int64)
    assert _poison_hell(h["cM"][:h["slots"]], h["rP"][:h["slots"]], lengths, offsets, hack, n) > 0
    x, y = synth.values_for(letter, 31, n), synth.values_for(letter, 32, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    want = O.spmv_tail(_host(h, letter, n, hack), x, y, -0.5, 2.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    assert not np.isnan(want.view(np.float32 if letter in "SC" else np.float64)).any()
    for plan in (1, 0):
        tuning(SPGPU_PLAN=plan)
        for call in range(5):
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            torch.cuda.synchronize()
            _call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)
            torch.cuda.synchronize()
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (plan, call, capi.plan_counts(gpu))


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("pattern", ["near", "band"])
def test_unordered_forms_never_use_padding(gpu, letter, pattern):
    """Rows as they come (ragged, so that every hack has padding): AUTO, gathers, strips, the LDS tile and the sweep."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, hack = 5 * 2048 + 77, 32
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, 10.0, 400, 5), 400)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, 300, real, seed=9)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, hack, 0, 0, order=False)
    lens = h["rS"][:n].cpu().numpy().astype(np.int64)
    offsets = h["hack_offsets"].cpu().numpy().astype(np.int64)
    assert _poison_hell(h["cM"][:h["slots"]], h["rP"][:h["slots"]], lens, offsets, hack, n) > 0
    x = synth.values_for(letter, 41, n)
    dx = formats.to_device(x)
    host = _host(h, letter, n, hack)
    one, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.0)
    p = lambda t: C.c_void_p(t.data_ptr())
    try:
        for form in (capi.FORM_AUTO, capi.FORM_GATHER, capi.FORM_STRIPS, capi.FORM_XTILE, capi.FORM_SWEEP):
            capi.spgpuSetSpmvForm(gpu, form)
            for call in range(3):
                dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
                torch.cuda.synchronize()
                capi.hellspmv[letter](gpu, p(dz), None, one, p(h["cM"]), p(h["rP"]), hack, p(h["hack_offsets"]), p(h["rS"]), None, 10, n, p(dx), zero, 0)
                torch.cuda.synchronize()
                if form == capi.FORM_SWEEP and letter in "SZ":
                    want = O.hell_spmv(host, x, None, 1.0, 0.0, phases=1)
                else:
                    want = O.default_spmv(host, x, None, 1.0, 0.0)
                assert dz.cpu().numpy().tobytes() == want.tobytes(), (form, call)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("n,cols_n,hack", [(4099, 4099, 32), (30011, 30051, 64), (1000, 987, 30), (257, 257, 96)])
def test_diagonal_formats_never_use_slots_outside_the_matrix(gpu, letter, n, cols_n, hack):
    """HDIA and DIA store whole diagonals: where a diagonal leaves the matrix (row + offset outside [0, cols), or a row beyond the last
    in the final hack) the slot exists and is never used (hdia_spmv_base_template.cuh:111-118, dia_spmv_base_template.cuh:20-216).
    Those slots get NaN; z is the oracle's, byte for byte, beta = 0 and beta != 0."""
    import torch
    from spgpu_amd import formats, synth
    rng = np.random.default_rng(n + hack)
    offsets = np.unique(np.concatenate(([0, 1, -1], rng.integers(-min(n, 2500) + 1, min(cols_n, 2500), 9))))
    rows_l, cols_l = [], []
    for off in offsets:
        r = np.arange(n, dtype=np.int64)
        c = r + off
        keep = (c >= 0) & (c < cols_n)
        rows_l.append(r[keep])
        cols_l.append(c[keep])
    rows, cols = np.concatenate(rows_l), np.concatenate(cols_l)
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    vals = rng.standard_normal(rows.size).astype(real)
    if letter in "CZ":
        vals = (vals + 1j * rng.standard_normal(rows.size).astype(real)).astype(O.NP_DTYPE[letter])
    x, y = synth.values_for(letter, 51, cols_n), synth.values_for(letter, 52, n)
    dx, dy = formats.to_device(x), formats.to_device(y)

    hdia = formats.coo_to_hdia(n, cols_n, rows, cols, vals, hack)
    values, offs, hack_offsets = hdia["values"], hdia["offsets"], hdia["hack_offsets"]
    poisoned = 0
    for h in range(len(hack_offsets) - 1):
        for pos in range(hack_offsets[h], hack_offsets[h + 1]):
            r = h * hack + np.arange(hack, dtype=np.int64)
            c = r + offs[pos]
            outside = (r >= n) | (c < 0) | (c >= cols_n)
            values[pos * hack + np.flatnonzero(outside)] = np.nan
            poisoned += int(outside.sum())
    assert poisoned > 0
    dia = formats.coo_to_dia(n, cols_n, rows, cols, vals)
    for d in range(dia["diags"]):
        r = np.arange(dia["pitch"], dtype=np.int64)
        c = r + dia["offsets"][d]
        dia["values"][d * dia["pitch"] + np.flatnonzero((r >= n) | (c < 0) | (c >= cols_n))] = np.nan
    for mat, oracle, host in ((formats.DeviceHdia(hdia), O.hdia_spmv, hdia), (formats.DeviceDia(dia), O.dia_spmv, dia)):
        for alpha, beta in ((1.0, 0.0), (-0.75, 0.5)):
            want = oracle(host, x, y if beta != 0 else None, alpha, beta)
            assert not np.isnan(want.view(np.float32 if letter in "SC" else np.float64)).any()
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            torch.cuda.synchronize()
            mat.spmv(gpu, dz, dy if beta != 0 else None, alpha, dx, beta)
            torch.cuda.synchronize()
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (type(mat).__name__, alpha, beta)


@pytest.mark.parametrize("letter", ["S", "D"])
@pytest.mark.parametrize("pattern,count", [("band", 16), ("near", 16), ("band", 8), ("near", 5)])
def test_spmm_never_uses_padding(gpu, letter, pattern, count):
    """spgpu?hellspmm on a ragged HELL whose padding slots hold NaN and random columns: band wavefronts (the sliding window of X rows),
    the strip kernel's other wavefronts and the narrow right-hand-side counts all select, none multiplies."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, hack = 5 * 2048 + 77, 32
    lengths = np.minimum(synth.power_law_lengths(n, 10.0, 200, 6), 200)
    lengths[: 2048] = 24                       # a stretch of even rows: band wavefronts (pattern "band") next to ragged ones
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, 150, letter, seed=10)
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, hack, 0, 0, order=False)
    lens = h["rS"][:n].cpu().numpy().astype(np.int64)
    offsets = h["hack_offsets"].cpu().numpy().astype(np.int64)
    assert _poison_hell(h["cM"][:h["slots"]], h["rP"][:h["slots"]], lens, offsets, hack, n) > 0
    X = synth.values_for(letter, 61, n * count).reshape(n, count)
    Y = synth.values_for(letter, 62, n * count).reshape(n, count)
    dX, dY = formats.to_device(X), formats.to_device(Y)
    host = _host(h, letter, n, hack)
    p = lambda t: C.c_void_p(t.data_ptr())
    for beta in (0.0, -0.5):
        dZ = torch.full_like(dY, float("nan"))
        torch.cuda.synchronize()
        capi.hellspmm[letter](gpu, p(dZ), p(dY), capi.scalar(letter, 1.25), p(h["cM"]), p(h["rP"]), hack, p(h["hack_offsets"]), p(h["rS"]),
                              None, 0, n, p(dX), capi.scalar(letter, beta), 0, count, count, count)
        torch.cuda.synchronize()
        want = O.hell_spmm(host, X, Y if beta != 0 else None, 1.25, beta)
        assert not np.isnan(want).any()
        assert dZ.cpu().numpy().tobytes() == want.tobytes(), beta
