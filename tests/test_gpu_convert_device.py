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
        ell_h = formats.coo_to_ell(n_rows, r, c, v, coo_base=base, ell_base=base)
        hell_h = formats.ell_to_hell(ell_h, hs)
        ell_d, hell_d = _device_convert(gpu, n_rows, r, c, v, base, hs)
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
