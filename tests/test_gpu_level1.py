"""GPU: axpby / maxpby (bit-exact vs the oracle) and dot / nrm2 (tolerance: the order of
addition of a parallel reduction is not the oracle's) through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _vec(letter, seed, n):
    from spgpu_amd import synth
    return synth.values_for(letter, seed, n)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("n", [1, 3, 255, 4097, 1_000_003])
@pytest.mark.parametrize("letter", "SDCZ")
def test_axpby(gpu, letter, n):
    import torch
    from spgpu_amd import capi, formats
    x, y = _vec(letter, 1, n), _vec(letter, 2, n)
    alpha = 1.5 if letter in "SD" else 1.5 - 0.5j
    for beta in (0.0, -0.75 if letter in "SD" else -0.75 + 2j):
        dx, dy = formats.to_device(x), formats.to_device(y)
        dz = torch.empty_like(dx)
        capi.axpby[letter](gpu, _p(dz), n, capi.scalar(letter, beta), _p(dy), capi.scalar(letter, alpha), _p(dx))
        torch.cuda.synchronize()
        want = O.axpby(letter, n, beta, y if beta != 0 else None, alpha, x)
        assert dz.cpu().numpy().tobytes() == want.tobytes()
        # in place on y (z == y) and on x (z == x)
        capi.axpby[letter](gpu, _p(dy), n, capi.scalar(letter, beta), _p(dy), capi.scalar(letter, alpha), _p(dx))
        torch.cuda.synchronize()
        assert dy.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("letter", "SDCZ")
def test_maxpby_multivector(gpu, letter):
    import torch
    from spgpu_amd import capi, formats
    n, count, pitch = 1001, 5, 1024
    x, y = _vec(letter, 3, count * pitch), _vec(letter, 4, count * pitch)
    dx, dy = formats.to_device(x), formats.to_device(y)
    dz = torch.zeros_like(dx)
    alpha, beta = (2.0, 0.25) if letter in "SD" else (2.0 + 1j, 0.25j)
    capi.maxpby[letter](gpu, _p(dz), n, capi.scalar(letter, beta), _p(dy), capi.scalar(letter, alpha), _p(dx), count, pitch)
    torch.cuda.synchronize()
    got = dz.cpu().numpy()
    for j in range(count):
        s = slice(j * pitch, j * pitch + n)
        assert got[s].tobytes() == O.axpby(letter, n, beta, y[s], alpha, x[s]).tobytes()
        assert not np.any(got[j * pitch + n:(j + 1) * pitch])  # gap between vectors untouched


@pytest.mark.parametrize("n", [1, 1234, 300_001])
@pytest.mark.parametrize("letter", "SDCZ")
def test_dot_and_nrm2(gpu, letter, n):
    from spgpu_amd import capi, formats
    a, b = _vec(letter, 5, n), _vec(letter, 6, n)
    da, db = formats.to_device(a), formats.to_device(b)
    eps = 1.2e-7 if letter in "SC" else 2.3e-16
    got = capi.dot[letter](gpu, n, _p(da), _p(db))
    got = got if letter in "SD" else complex(got.x, got.y)
    wide = np.clongdouble if letter in "CZ" else np.longdouble
    exact = np.sum(a.astype(wide) * b.astype(wide))          # un-conjugated (zdot.cu:54)
    mag = float(np.sum(np.abs(a.astype(wide) * b.astype(wide))))
    assert abs(got - complex(exact) if letter in "CZ" else got - float(exact)) <= 64 * eps * mag + 1e-300
    assert abs(complex(O.dot(letter, a, b)) - complex(exact)) <= n * eps * mag + 1e-300
    nr = capi.nrm2[letter](gpu, n, _p(da))
    exact_n = float(np.sqrt(np.sum(np.abs(a.astype(wide)) ** 2)))
    assert abs(nr - exact_n) <= 64 * eps * exact_n


def test_testdensevector_program(gpu):
    """The reference's testDenseVector.c:31-32,51-76: x[i] = i, n = 1234, dot and nrm2."""
    import torch
    from spgpu_amd import capi
    n = 1234
    x = torch.arange(n, dtype=torch.float32, device="cuda:0")
    d = capi.dot["S"](gpu, n, _p(x), _p(x))
    exact = float(sum(i * i for i in range(n)))
    assert abs(d - exact) <= 1e-6 * exact
    assert abs(capi.nrm2["S"](gpu, n, _p(x)) - exact ** 0.5) <= 1e-6 * exact ** 0.5


@pytest.mark.parametrize("letter", "DZ")
def test_mdot_mnrm2(gpu, letter):
    from spgpu_amd import capi, formats
    n, count, pitch = 777, 4, 800
    a, b = _vec(letter, 7, count * pitch), _vec(letter, 8, count * pitch)
    da, db = formats.to_device(a), formats.to_device(b)
    out = np.zeros(count, O.NP_DTYPE[letter])
    capi.mdot[letter](gpu, C.c_void_p(out.ctypes.data), n, _p(da), _p(db), count, pitch)
    nout = np.zeros(count, np.float64)
    capi.mnrm2[letter](gpu, C.c_void_p(nout.ctypes.data), n, _p(da), count, pitch)
    for j in range(count):
        s = slice(j * pitch, j * pitch + n)
        assert abs(out[j] - np.sum(a[s] * b[s])) <= 1e-12 * float(np.sum(np.abs(a[s] * b[s])))
        assert abs(nout[j] - np.linalg.norm(a[s])) <= 1e-13 * np.linalg.norm(a[s])
