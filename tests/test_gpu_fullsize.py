"""GPU: size-independent properties at (or near) BASELINE.json's full sizes, and the device-side
generators the full-size runs depend on, checked against the host converters on small instances."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("m", [32, 64])
def test_device_hdia_laplacian_equals_the_converter(gpu, m):
    """synth.hdia_laplacian7_on_device == cooToHdia(laplacian_3d_7pt) byte for byte."""
    from spgpu_amd import formats, synth
    n, nc, r, c, v = synth.laplacian_3d_7pt(m)
    host = formats.coo_to_hdia(n, nc, r, c, v, 32)
    dev = synth.hdia_laplacian7_on_device(m, "D", 32)
    assert dev["height"] == host["height"] and dev["nnz"] == r.size
    assert np.array_equal(dev["hack_offsets"].cpu().numpy(), host["hack_offsets"])
    assert np.array_equal(dev["offsets"].cpu().numpy(), host["offsets"])
    assert dev["dM"].cpu().numpy().tobytes() == host["values"].tobytes()


def test_device_uniform_hell_equals_the_converters(gpu):
    """synth.hell_uniform_on_device lays HELL out as cooToEll + ellToHell do."""
    from spgpu_amd import formats, synth
    h = synth.hell_uniform_on_device(640, 8, "random", "D", 32, seed=3)
    cols = h["rP"].view(20, 8, 32).permute(0, 2, 1).reshape(640, 8).cpu().numpy()     # [row][k]
    vals = h["cM"].view(20, 8, 32).permute(0, 2, 1).reshape(640, 8).cpu().numpy()
    rows = np.repeat(np.arange(640), 8)
    hell = formats.ell_to_hell(formats.coo_to_ell(640, rows, cols.reshape(-1), vals.reshape(-1)), 32)
    assert hell["indices"].tobytes() == h["rP"].cpu().numpy().tobytes()
    assert hell["values"].tobytes() == h["cM"].cpu().numpy().tobytes()
    assert np.array_equal(hell["hack_offsets"], h["hack_offsets"].cpu().numpy())


@pytest.mark.parametrize("pattern", ["banded", "random"])
def test_full_size_hell_fp64_properties(gpu, pattern):
    """BASELINE configs[1] size (10 M rows x 32): linearity in x, ELL == HELL bit for bit (one summation
    order), alpha/beta epilogue, and oracle parity on row windows."""
    import torch
    from spgpu_amd import capi, synth
    n, L = 10_000_000, 32
    h = synth.hell_uniform_on_device(n, L, pattern, "D", 32, seed=1)
    x1, x2, y = (synth.device_vector(n, "D", s) for s in (3, 4, 5))
    z1, z2, z12, zb = (torch.empty_like(y) for _ in range(4))
    torch.cuda.synchronize()

    def hell(z, yy, alpha, x, beta):
        capi.hellspmv["D"](gpu, _p(z), _p(yy), alpha, _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), None,
                           L, n, _p(x), beta, 0)

    hell(z1, None, 1.0, x1, 0.0)
    hell(z2, None, 1.0, x2, 0.0)
    x12 = x1 + 2.0 * x2
    torch.cuda.synchronize()
    hell(z12, None, 1.0, x12, 0.0)
    hell(zb, y, -0.5, x1, 2.0)
    torch.cuda.synchronize()
    # linearity A(x1 + 2 x2) = A x1 + 2 A x2 within rounding of 32-term sums of values in [0,1)
    err = (z12 - (z1 + 2.0 * z2)).abs().max().item()
    assert err <= 1e-12 * 3 * L
    # epilogue: zb = -0.5 * (A x1) + 2 y, with one rounding for the fma
    ref = torch.addcmul(2.0 * y, z1, torch.tensor(-0.5, dtype=torch.float64, device=y.device))
    assert (zb - ref).abs().max().item() <= 1e-14 * (L + 4)
    # the same slots read as ELL (uniform rows: pitch = n, slot (r,k) = r + k*n) give the same bits
    cM_ell = h["cM"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    rP_ell = h["rP"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    ze = torch.empty_like(y)
    torch.cuda.synchronize()
    capi.ellspmv["D"](gpu, _p(ze), None, 1.0, _p(cM_ell), _p(rP_ell), n, n, _p(h["rS"]), None, L, L, n, _p(x1), 0.0, 0)
    torch.cuda.synchronize()
    assert torch.equal(ze, z1)
    # dot(z,z) as the reference's harness prints it, against torch
    d = capi.dot["D"](gpu, n, _p(z1), _p(z1))
    assert abs(d - float(torch.dot(z1, z1))) <= 1e-10 * d
    # oracle parity on three windows
    xs = x1.cpu().numpy()
    for first in (0, 4_999_936, n - 2048):
        sub = synth.hell_rows_to_host(h, first, 2048)
        assert z1[first:first + 2048].cpu().numpy().tobytes() == O.default_spmv(sub, xs, None, 1.0, 0.0).tobytes()
