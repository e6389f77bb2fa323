"""GPU: DIA SpMV, ELL csput and the OELL (rIdx) route through the C ABI, bit for bit against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("letter", "SDCZ")
def test_dia_spmv(gpu, letter):
    import torch
    from spgpu_amd import formats, synth
    rng = np.random.default_rng(5)
    for n, m, offs in ((200, 200, [-37, -1, 0, 1, 2, 64]), (97, 150, [-96, -3, 0, 5, 149]), (1, 1, [0])):
        rows, cols = [], []
        for o in offs:
            r = np.arange(max(0, -o), min(n, m - o))
            rows.append(r); cols.append(r + o)
        r, c = np.concatenate(rows), np.concatenate(cols)
        v = synth.values_for(letter, 3, r.size)
        dia = formats.coo_to_dia(n, m, r, c, v)
        x, y = synth.values_for(letter, 4, m), synth.values_for(letter, 5, n)
        mat = formats.DeviceDia(dia)
        dx, dy = formats.to_device(x), formats.to_device(y)   # keep device buffers alive across the async calls
        for alpha, beta in ((1.0, 0.0), (0.5, -1.5)):
            dz = torch.full((n,), float("nan"), dtype=dy.dtype, device="cuda:0")
            mat.spmv(gpu, dz, dy, alpha, dx, beta)
            torch.cuda.synchronize()
            want = O.dia_spmv(dia, x, y if beta else None, alpha, beta)
            assert dz.cpu().numpy().tobytes() == want.tobytes()
        # the same matrix through DIA -> HDIA gives the same bits
        hd = formats.dia_to_hdia(dia, 32)
        dz2 = torch.empty_like(dz)
        formats.DeviceHdia(hd).spmv(gpu, dz2, None, 1.0, dx, 0.0)
        torch.cuda.synchronize()
        assert dz2.cpu().numpy().tobytes() == O.dia_spmv(dia, x, None, 1.0, 0.0).tobytes()


@pytest.mark.parametrize("letter", "SDCZ")
def test_ell_csput(gpu, letter):
    import torch
    from spgpu_amd import capi, formats, synth
    n, m, r, c, v = synth.laplacian_2d_5pt(20, dtype=O.NP_DTYPE[letter], base=1)
    ell = formats.coo_to_ell(n, r, c, v, coo_base=1, ell_base=1)     # rows ascend in column: binary search valid
    rng = np.random.default_rng(9)
    pick = rng.permutation(r.size)[:300]
    a_i, a_j = r[pick].copy(), c[pick].copy()
    a_i[::40] = 0                      # row -1 with base 1: ignored
    a_j[1::40] = n + 5                 # column not stored: ignored
    a_val = synth.values_for(letter, 6, 300)
    mat = formats.DeviceEll(ell)
    d_i, d_j, d_val = formats.to_device(a_i), formats.to_device(a_j), formats.to_device(a_val)   # alive until the sync
    capi.ellcsput[letter](gpu, capi.scalar(letter, 3.0), _p(mat.cM), _p(mat.rP), mat.pitch, mat.pitch, _p(mat.rS), 300,
                          _p(d_i), _p(d_j), _p(d_val), 1)
    torch.cuda.synchronize()
    assert mat.cM.cpu().numpy().tobytes() == O.ell_csput(ell, a_i, a_j, a_val, 1).tobytes()


def test_oell_route_matches_plain_ell(gpu):
    """hellPerf.cpp:319-378: ELL reordered by ellToOell and run with rIdx gives the plain ELL result."""
    import torch
    from spgpu_amd import formats, synth
    lengths = synth.power_law_lengths(1500, 8.0, 60, seed=4)
    n, m, r, c, v = synth.random_rows_coo(1500, 1500, lengths, seed=5, letter="D")
    ell = formats.coo_to_ell(n, r, c, v)
    oell, r_idx = formats.ell_to_oell(ell)
    x, y = synth.values_for("D", 7, m), synth.values_for("D", 8, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    z0, z1 = torch.empty_like(dy), torch.empty_like(dy)
    formats.DeviceEll(ell).spmv(gpu, z0, dy, 1.0, dx, 0.5)
    formats.DeviceEll(oell, r_idx=r_idx).spmv(gpu, z1, dy, 1.0, dx, 0.5)
    torch.cuda.synchronize()
    # a long row's tail is summed 64 ways from the column where its wavefront's other rows have ended, and the
    # reordering changes which rows share a wavefront: equal within rounding, not bitwise
    assert (z0 - z1).abs().max().item() <= 1e-13 * float(np.abs(v).max() * np.abs(x).max() * lengths.max())
    assert z1.cpu().numpy().tobytes() == O.default_spmv(oell, x, y, 1.0, 0.5, r_idx=r_idx).tobytes()
