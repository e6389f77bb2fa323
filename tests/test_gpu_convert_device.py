"""GPU: device-side COO -> ELL / HELL construction (include/spgpu/convert_device.h) against the host converters
(which are byte-identical to the reference's): same bytes for any COO entry order, duplicates, both index bases,
all four value types, several hack sizes."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _device_convert(gpu, n_rows, r, c, v, base, hack_size):
    """Runs the four device calls; returns (ell dict, hell dict) with numpy arrays."""
    import torch
    from spgpu_amd import capi, formats
    letter = formats.LETTER_OF[np.dtype(v.dtype)]
    code = capi.TYPE_CODE[letter]
    nnz = int(r.size)
    one = lambda a, dt: formats.to_device(a if a.size else np.zeros(1, dt))
    dr, dc, dv = one(r.astype(np.int32), np.int32), one(c.astype(np.int32), np.int32), one(v, v.dtype)
    work = torch.empty(capi.spgpuCooConvertWorkBytes(n_rows, nnz), dtype=torch.uint8, device="cuda:0")
    rs = torch.empty(max(n_rows, 1), dtype=torch.int32, device="cuda:0")
    max_row = C.c_int(-1)
    st = capi.spgpuCooRowLengthsDevice(gpu, _p(rs), C.byref(max_row), n_rows, nnz, _p(dr), base, _p(work))
    assert st == capi.SPGPU_SUCCESS
    pitch = capi.computeEllAllocPitch(n_rows)
    ell_v = torch.zeros(max(max_row.value * pitch, 1), dtype=dv.dtype, device="cuda:0")
    ell_i = torch.zeros(max(max_row.value * pitch, 1), dtype=torch.int32, device="cuda:0")
    assert capi.spgpuCooToEllDevice(gpu, _p(ell_v), _p(ell_i), pitch, pitch, base, n_rows, nnz, _p(dr), _p(dc), _p(dv), base,
                                    code, _p(rs), _p(work)) == capi.SPGPU_SUCCESS
    hacks = (n_rows + hack_size - 1) // hack_size
    ho = torch.zeros(max(hacks, 1), dtype=torch.int32, device="cuda:0")
    height = C.c_int(-1)
    assert capi.spgpuHellPlanDevice(gpu, C.byref(height), _p(ho), hack_size, n_rows, _p(rs), _p(work)) == capi.SPGPU_SUCCESS
    slots = hack_size * height.value
    hell_v = torch.zeros(max(slots, 1), dtype=dv.dtype, device="cuda:0")
    hell_i = torch.zeros(max(slots, 1), dtype=torch.int32, device="cuda:0")
    assert capi.spgpuCooToHellDevice(gpu, _p(hell_v), _p(hell_i), _p(ho), hack_size, base, n_rows, nnz, _p(dr), _p(dc), _p(dv),
                                     base, code, _p(rs), _p(work)) == capi.SPGPU_SUCCESS
    torch.cuda.synchronize()
    ell = dict(max_row=max_row.value, pitch=pitch, row_lengths=rs.cpu().numpy()[:n_rows],
               values=ell_v.cpu().numpy()[:max_row.value * pitch], indices=ell_i.cpu().numpy()[:max_row.value * pitch])
    hell = dict(height=height.value, hack_offsets=ho.cpu().numpy()[:hacks], values=hell_v.cpu().numpy()[:slots],
                indices=hell_i.cpu().numpy()[:slots])
    return ell, hell


def _hosts():
    """Whom the device converters are compared with: the product's host converters (pinned to the reference's build in the CPU
    suite, tests/test_oracle_vs_reference.py), the ORACLE's restatement of the reference's converters (ell.c:5-80, hell.c:4-104,
    hdia.cpp:161-349, dia.c:5-104), and -- where oracle/_ref travelled with the tree -- the reference's own objects."""
    from spgpu_amd import formats
    sets = [("product host converters", formats), ("oracle", O.oracle_converters)]
    if O.reference_available():
        sets.append(("reference build", O.reference_converters()))
    return sets


def _same(dev, host, keys):
    for k in keys:
        a, b = dev[k], host[k]
        if isinstance(b, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k
        else:
            assert a == b, k


@pytest.mark.parametrize("letter", "SDCZ")
def test_random_coo_matches_host_converters(gpu, letter):
    from spgpu_amd import formats
    from test_oracle_vs_reference import _random_coo
    rng = np.random.default_rng(100 + ord(letter))
    for trial in range(12):
        base, hs = int(rng.integers(0, 2)), int(rng.choice([32, 64, 96]))
        n_rows, n_cols, r, c, v = _random_coo(rng, letter, base)
        ell_d, hell_d = _device_convert(gpu, n_rows, r, c, v, base, hs)
        for _, host in _hosts():
            ell_h = host.coo_to_ell(n_rows, r, c, v, coo_base=base, ell_base=base)
            hell_h = host.ell_to_hell(ell_h, hs)
            _same(ell_d, ell_h, ("max_row", "pitch", "row_lengths", "indices", "values"))
            _same(hell_d, hell_h, ("height", "hack_offsets", "indices", "values"))


def test_large_shuffled_power_law(gpu):
    """2 M rows, power-law lengths up to 2048, COO order shuffled, duplicates present: the rank recount must
    restore encounter order exactly."""
    from spgpu_amd import formats, synth
    n = 2_000_000 // 4
    lengths = synth.power_law_lengths(n, 12.0, 2048, seed=3)
    n_, m_, r, c, v = synth.random_rows_coo(n, 5000, lengths, seed=4, letter="D", shuffle=True)   # few columns: duplicates
    ell_h = formats.coo_to_ell(n, r, c, v)
    hell_h = formats.ell_to_hell(ell_h, 32)
    ell_d, hell_d = _device_convert(gpu, n, r, c, v, 0, 32)
    _same(ell_d, ell_h, ("max_row", "row_lengths"))
    _same(hell_d, hell_h, ("height", "hack_offsets", "indices", "values"))


def test_degenerate_and_bad_input(gpu):
    import torch
    from spgpu_amd import capi, formats
    # no entries
    ell_d, hell_d = _device_convert(gpu, 70, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 0, 32)
    assert ell_d["max_row"] == 0 and hell_d["height"] == 0 and not ell_d["row_lengths"].any()
    # a row index outside the matrix is reported, not dereferenced
    r = formats.to_device(np.array([0, 5, 99], np.int32))
    work = torch.empty(capi.spgpuCooConvertWorkBytes(10, 3), dtype=torch.uint8, device="cuda:0")
    rs = torch.empty(10, dtype=torch.int32, device="cuda:0")
    mx = C.c_int(0)
    assert capi.spgpuCooRowLengthsDevice(gpu, _p(rs), C.byref(mx), 10, 3, _p(r), 0, _p(work)) == capi.SPGPU_UNSUPPORTED


# ---- COO -> HDIA ------------------------------------------------------------------------------------------------
def _device_hdia(gpu, n_rows, n_cols, r, c, v, base, hack_size):
    import torch
    from spgpu_amd import capi, formats
    letter = formats.LETTER_OF[np.dtype(v.dtype)]
    nnz = int(r.size)
    one = lambda a, dt: formats.to_device(a if a.size else np.zeros(1, dt))
    dr, dc, dv = one(r.astype(np.int32), np.int32), one(c.astype(np.int32), np.int32), one(v, v.dtype)
    work_bytes = capi.spgpuCooHdiaPlanWorkBytes(n_rows, nnz)
    assert work_bytes > 0
    work = torch.empty(work_bytes, dtype=torch.uint8, device="cuda:0")
    hacks = capi.getHdiaHacksCount(hack_size, n_rows)
    ho = torch.full((hacks + 1,), -1, dtype=torch.int32, device="cuda:0")
    height = C.c_int(-1)
    st = capi.spgpuCooHdiaPlanDevice(gpu, C.byref(height), _p(ho), hack_size, n_rows, n_cols, nnz, _p(dr), _p(dc), base, _p(work))
    if st != capi.SPGPU_SUCCESS:
        return st, None
    slots = hack_size * height.value
    values = torch.zeros(max(slots, 1), dtype=dv.dtype, device="cuda:0")
    offsets = torch.zeros(max(height.value, 1), dtype=torch.int32, device="cuda:0")
    scratch = torch.empty(capi.spgpuCooToHdiaScratchBytes(hack_size, height.value), dtype=torch.uint8, device="cuda:0")
    st = capi.spgpuCooToHdiaDevice(gpu, _p(values), _p(offsets), _p(ho), hack_size, n_rows, n_cols, nnz, _p(dr), _p(dc), _p(dv),
                                   base, capi.TYPE_CODE[letter], height.value, _p(work), _p(scratch))
    torch.cuda.synchronize()
    return st, dict(height=height.value, hack_offsets=ho.cpu().numpy(), values=values.cpu().numpy()[:slots],
                    offsets=offsets.cpu().numpy()[:height.value])


def _diagonal_coo(rng, letter, base, n_rows, n_cols, diagonals, fill, duplicates):
    """Entries on a few diagonals (HDIA's home ground), each present with probability `fill`, some repeated with
    other values, then shuffled."""
    from spgpu_amd import synth
    rows = np.arange(n_rows)
    parts_r, parts_c = [], []
    for d in diagonals:
        keep = (rows + d >= 0) & (rows + d < n_cols) & (rng.random(n_rows) < fill)
        parts_r.append(rows[keep])
        parts_c.append(rows[keep] + d)
    r, c = np.concatenate(parts_r), np.concatenate(parts_c)
    if duplicates and r.size:
        again = rng.integers(0, r.size, size=duplicates)
        r, c = np.concatenate([r, r[again]]), np.concatenate([c, c[again]])
    order = rng.permutation(r.size)
    r, c = r[order], c[order]
    v = synth.values_for(letter, int(rng.integers(1, 1 << 30)), r.size)
    return (r + base).astype(np.int32), (c + base).astype(np.int32), v


@pytest.mark.parametrize("letter", "SDCZ")
def test_device_coo_to_hdia_matches_host(gpu, letter):
    """Same hackOffsets, offsets and values as computeHdiaHackOffsetsFromCoo + cooToHdia, byte for byte: shuffled COO,
    duplicates (the last one in COO order wins), both index bases, partly filled diagonals, rectangular matrices,
    hack sizes 32/64/96, a last hack that is partly filled."""
    from spgpu_amd import capi, formats
    rng = np.random.default_rng(300 + ord(letter))
    for trial in range(10):
        base, hs = int(rng.integers(0, 2)), int(rng.choice([32, 64, 96]))
        n_rows = int(rng.integers(1, 3000))
        n_cols = n_rows if trial % 2 else int(rng.integers(1, 3000))
        diagonals = np.unique(rng.integers(-n_rows + 1, n_cols, size=int(rng.integers(1, 12))))
        r, c, v = _diagonal_coo(rng, letter, base, n_rows, n_cols, diagonals, float(rng.choice([1.0, 0.6, 0.05])),
                                int(rng.integers(0, 50)))
        st, dev = _device_hdia(gpu, n_rows, n_cols, r, c, v, base, hs)
        assert st == capi.SPGPU_SUCCESS
        for _, converters in _hosts():
            host = converters.coo_to_hdia(n_rows, n_cols, r, c, v, hs, coo_base=base)
            _same(dev, host, ("height", "hack_offsets", "offsets", "values"))


def test_device_coo_to_hdia_scattered_and_empty(gpu):
    """Not a diagonal matrix at all (every entry its own diagonal in its hack), an empty matrix, and an entry outside
    the matrix (reported, as the ELL/HELL calls do)."""
    from spgpu_amd import capi, formats
    from test_oracle_vs_reference import _random_coo
    rng = np.random.default_rng(41)
    for _ in range(4):
        n_rows, n_cols, r, c, v = _random_coo(rng, "D", 1)
        host = formats.coo_to_hdia(n_rows, n_cols, r, c, v, 32, coo_base=1)
        st, dev = _device_hdia(gpu, n_rows, n_cols, r, c, v, 1, 32)
        assert st == capi.SPGPU_SUCCESS
        _same(dev, host, ("height", "hack_offsets", "offsets", "values"))
    none = np.zeros(0, dtype=np.int32)
    st, dev = _device_hdia(gpu, 100, 100, none, none, np.zeros(0), 0, 32)
    assert st == capi.SPGPU_SUCCESS and dev["height"] == 0 and not dev["hack_offsets"].any()
    st, _ = _device_hdia(gpu, 10, 10, np.array([3, 10], dtype=np.int32), np.array([1, 1], dtype=np.int32), np.ones(2), 0, 32)
    assert st == capi.SPGPU_UNSUPPORTED


def test_device_hdia_laplacian(gpu):
    """7-point Laplacian 48^3 (BASELINE configs[3]'s matrix family) from COO on the device: equals the host-built HDIA."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, _, r, c, v = synth.laplacian_3d_7pt(48)
    st, dev = _device_hdia(gpu, n, n, r, c, v, 0, 32)
    assert st == capi.SPGPU_SUCCESS
    for _, converters in _hosts():
        host = converters.coo_to_hdia(n, n, r, c, v, 32)
        _same(dev, host, ("height", "hack_offsets", "offsets", "values"))


# ---- COO -> DIA -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("letter", "SDCZ")
def test_device_coo_to_dia_matches_host(gpu, letter):
    """Same diagonal count, offsets and values as computeDiaDiagonalsCount + coo2dia, byte for byte: shuffled COO,
    duplicates (the last one wins), both index bases, rectangular matrices, partly filled diagonals."""
    import torch
    from spgpu_amd import capi, formats
    rng = np.random.default_rng(500 + ord(letter))
    for trial in range(8):
        base = int(rng.integers(0, 2))
        n_rows = int(rng.integers(1, 2500))
        n_cols = n_rows if trial % 2 else int(rng.integers(1, 2500))
        diagonals = np.unique(rng.integers(-n_rows + 1, n_cols, size=int(rng.integers(1, 10))))
        r, c, v = _diagonal_coo(rng, letter, base, n_rows, n_cols, diagonals, float(rng.choice([1.0, 0.5])), int(rng.integers(0, 40)))
        host = formats.coo_to_dia(n_rows, n_cols, r, c, v, coo_base=base)
        nnz = int(r.size)
        one = lambda a, dt: formats.to_device(a if a.size else np.zeros(1, dt))
        dr, dc, dv = one(r, np.int32), one(c, np.int32), one(v, v.dtype)
        work = torch.empty(capi.spgpuCooDiaWorkBytes(n_rows, n_cols), dtype=torch.uint8, device="cuda:0")
        count = C.c_int(-1)
        assert capi.spgpuCooDiaPlanDevice(gpu, C.byref(count), n_rows, n_cols, nnz, _p(dr), _p(dc), base, _p(work)) == capi.SPGPU_SUCCESS
        assert count.value == host["diags"]
        pitch = capi.computeDiaAllocPitch(n_rows)
        values = torch.zeros(max(pitch * count.value, 1), dtype=dv.dtype, device="cuda:0")
        offsets = torch.zeros(max(count.value, 1), dtype=torch.int32, device="cuda:0")
        scratch = torch.empty(capi.spgpuCooToDiaScratchBytes(pitch, count.value), dtype=torch.uint8, device="cuda:0")
        assert capi.spgpuCooToDiaDevice(gpu, _p(values), _p(offsets), pitch, count.value, n_rows, n_cols, nnz, _p(dr), _p(dc), _p(dv),
                                        base, capi.TYPE_CODE[letter], _p(work), _p(scratch)) == capi.SPGPU_SUCCESS
        torch.cuda.synchronize()
        for _, converters in _hosts():
            host = converters.coo_to_dia(n_rows, n_cols, r, c, v, coo_base=base)
            assert count.value == host["diags"]
            assert offsets.cpu().numpy()[:count.value].tobytes() == host["offsets"].tobytes()
            assert values.cpu().numpy()[:pitch * count.value].tobytes() == host["values"].tobytes()
    # an entry outside the matrix is reported
    bad = formats.to_device(np.array([0, 7], dtype=np.int32))
    work = torch.empty(capi.spgpuCooDiaWorkBytes(5, 5), dtype=torch.uint8, device="cuda:0")
    count = C.c_int(-1)
    assert capi.spgpuCooDiaPlanDevice(gpu, C.byref(count), 5, 5, 2, _p(bad), _p(bad), 0, _p(work)) == capi.SPGPU_UNSUPPORTED


def test_one_row_of_300k_entries(gpu):
    """A row far longer than anything a hack of 32 was made for (a hub of a power-law graph): 300 000 entries in one row
    among 5 000 ordinary rows, COO order shuffled.  The position of an entry inside its row is its sorted position minus
    the row's start -- cost linear in the entries (round 1 recounted it by scanning the row's bucket: 9e10 steps here)."""
    from spgpu_amd import formats
    rng = np.random.default_rng(4)
    n_rows, hub = 5000, 1234
    lengths = rng.integers(0, 9, size=n_rows)
    lengths[hub] = 300_000
    r = np.repeat(np.arange(n_rows), lengths)
    c = rng.integers(0, 400_000, size=r.size)
    v = rng.standard_normal(r.size)
    order = rng.permutation(r.size)
    r, c, v = r[order], c[order], v[order]
    ell_h = formats.coo_to_ell(n_rows, r, c, v)
    hell_h = formats.ell_to_hell(ell_h, 32)
    ell_d, hell_d = _device_convert(gpu, n_rows, r, c, v, 0, 32)
    _same(ell_d, ell_h, ("max_row", "pitch", "row_lengths", "indices", "values"))
    _same(hell_d, hell_h, ("height", "hack_offsets", "indices", "values"))
