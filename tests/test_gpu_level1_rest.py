"""GPU: the rest of the reference's Level-1 (scal, abs, axy, axypbz, gath, scat, setscal, asum, amax)
through the C ABI.  Element-wise results bit for bit against the oracle; sums within rounding.
The sparse-vector test reproduces the reference's own testSparseVector.c (1234-vector, 123 indices
(17*i) % 1234, scatter with beta = 2, exact comparison)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _vec(letter, seed, n):
    from spgpu_amd import synth
    return synth.values_for(letter, seed, n)


def _dev(a):
    from spgpu_amd import formats
    return formats.to_device(a)


@pytest.mark.parametrize("n", [1, 5, 1023, 200_003])
@pytest.mark.parametrize("letter", "SDCZ")
def test_scal_abs_axy_axypbz(gpu, letter, n):
    import torch
    from spgpu_amd import capi
    x, y, z = _vec(letter, 1, n), _vec(letter, 2, n), _vec(letter, 3, n)
    alpha = -1.75 if letter in "SD" else -1.75 + 0.5j
    beta = 0.25 if letter in "SD" else 0.25 - 2j
    sc = lambda v: capi.scalar(letter, v)
    dx, dy, dz = _dev(x), _dev(y), _dev(z)
    out = torch.empty_like(dx)

    capi.scal[letter](gpu, _p(out), n, sc(alpha), _p(dx))
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == O.level1_map(letter, "scal", n, alpha, x).tobytes()

    for a in (alpha, 1.0):   # alpha == 1 takes the reference's no-multiply path
        capi.vabs[letter](gpu, _p(out), n, sc(a), _p(dx))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == O.level1_map(letter, "abs", n, a, x).tobytes()

    capi.axy[letter](gpu, _p(out), n, sc(alpha), _p(dx), _p(dy))
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == O.level1_map(letter, "axy", n, alpha, x, y).tobytes()

    for a, b in ((alpha, beta), (0.0, beta), (alpha, 0.0)):
        capi.axypbz[letter](gpu, _p(out), n, sc(b), _p(dz), sc(a), _p(dx), _p(dy))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == O.level1_map(letter, "axypbz", n, a, x, y, b, z).tobytes()

    # in place: y <- alpha * x * y
    capi.axy[letter](gpu, _p(dy), n, sc(alpha), _p(dx), _p(dy))
    torch.cuda.synchronize()
    assert dy.cpu().numpy().tobytes() == O.level1_map(letter, "axy", n, alpha, x, y).tobytes()


@pytest.mark.parametrize("letter", "SZ")
def test_multivector_forms(gpu, letter):
    import torch
    from spgpu_amd import capi
    n, count, pitch = 501, 3, 512
    x, y, z = (_vec(letter, s, count * pitch) for s in (4, 5, 6))
    alpha, beta = (0.5, -2.0) if letter == "S" else (0.5 + 1j, -2.0j)
    dx, dy, dz = _dev(x), _dev(y), _dev(z)
    out = torch.zeros_like(dx)
    capi.maxy[letter](gpu, _p(out), n, capi.scalar(letter, alpha), _p(dx), _p(dy), count, pitch)
    out2 = torch.zeros_like(dx)
    capi.maxypbz[letter](gpu, _p(out2), n, capi.scalar(letter, beta), _p(dz), capi.scalar(letter, alpha), _p(dx), _p(dy), count, pitch)
    torch.cuda.synchronize()
    a, b = out.cpu().numpy(), out2.cpu().numpy()
    for j in range(count):
        s = slice(j * pitch, j * pitch + n)
        assert a[s].tobytes() == O.level1_map(letter, "axy", n, alpha, x[s], y[s]).tobytes()
        assert b[s].tobytes() == O.level1_map(letter, "axypbz", n, alpha, x[s], y[s], beta, z[s]).tobytes()
        assert not np.any(a[j * pitch + n:(j + 1) * pitch])


@pytest.mark.parametrize("letter", "SD")
def test_sparse_vector_program(gpu, letter):
    """testSparseVector.c:51-122: y of 1234 elements, 123 indices (17*i) % 1234, scat with beta = 2 then gath,
    each compared exactly with a host loop."""
    import torch
    from spgpu_amd import capi
    n, m = 1234, 123
    dt = O.NP_DTYPE[letter]
    y = np.arange(n, dtype=dt)
    idx = ((17 * np.arange(m)) % n).astype(np.int32)
    vals = (np.arange(m) * 0.5 + 1.0).astype(dt)
    dy, didx, dvals = _dev(y), _dev(idx), _dev(vals)
    capi.scat[letter](gpu, _p(dy), m, _p(dvals), _p(didx), 0, capi.scalar(letter, 2.0))
    torch.cuda.synchronize()
    want = y.copy()
    for i in range(m):
        want[idx[i]] = dt(2.0) * want[idx[i]] + vals[i]
    assert np.array_equal(dy.cpu().numpy(), want)
    assert np.array_equal(O.scat(letter, y, vals, idx, 0, 2.0), want)
    got = torch.zeros(m, dtype=dy.dtype, device="cuda:0")
    capi.gath[letter](gpu, _p(got), m, _p(didx), 0, _p(dy))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want[idx])


@pytest.mark.parametrize("letter", "ISDCZ")
def test_gath_scat_setscal_all_types(gpu, letter):
    import torch
    from spgpu_amd import capi
    rng = np.random.default_rng(11)
    n, m, base = 5000, 777, 1
    dt = np.int32 if letter == "I" else O.NP_DTYPE[letter]
    y = (rng.integers(-50, 50, n).astype(dt) if letter == "I" else _vec(letter, 7, n))
    vals = (rng.integers(-50, 50, m).astype(dt) if letter == "I" else _vec(letter, 8, m))
    idx = (rng.permutation(n)[:m] + base).astype(np.int32)   # distinct: scatter is race-free
    idx[::50] = 0                                            # position -1 with base 1: skipped
    beta = 3 if letter == "I" else (1.5 if letter in "SD" else 1.5 - 1j)
    sc = (lambda v: C.c_int(int(v))) if letter == "I" else (lambda v: capi.scalar(letter, v))
    dy, dvals, didx = _dev(y), _dev(vals), _dev(idx)
    for b in (beta, 0):
        dyy = dy.clone()
        capi.scat[letter](gpu, _p(dyy), m, _p(dvals), _p(didx), base, sc(b))
        torch.cuda.synchronize()
        assert dyy.cpu().numpy().tobytes() == O.scat(letter, y, vals, idx, base, b).tobytes()
    g_in = np.full(m, 9, dt) if letter == "I" else _vec(letter, 9, m)
    dg = _dev(g_in)
    capi.gath[letter](gpu, _p(dg), m, _p(didx), base, _p(dy))
    torch.cuda.synchronize()
    assert dg.cpu().numpy().tobytes() == O.gath(letter, g_in, idx, base, y).tobytes()
    val = 42 if letter == "I" else (2.5 if letter in "SD" else 2.5 + 4j)
    capi.setscal[letter](gpu, 11, 4000, 1, sc(val), _p(dy))
    torch.cuda.synchronize()
    assert dy.cpu().numpy().tobytes() == O.setscal(letter, y, 11, 4000, 1, val).tobytes()


@pytest.mark.parametrize("n", [1, 1234, 300_001])
@pytest.mark.parametrize("letter", "SDCZ")
def test_asum_amax(gpu, letter, n):
    from spgpu_amd import capi
    x = _vec(letter, 12, n)
    if n > 10:
        x[n // 3] *= 7   # a unique maximum
    dx = _dev(x)
    eps = 1.2e-7 if letter in "SC" else 2.3e-16
    wide = np.abs(x.astype(np.complex128 if letter in "CZ" else np.float64)).astype(np.longdouble)
    s = capi.asum[letter](gpu, n, _p(dx))
    assert abs(s - float(wide.sum())) <= 64 * eps * float(wide.sum())
    assert abs(float(O.asum(letter, x)) - float(wide.sum())) <= n * eps * float(wide.sum())
    m = capi.amax[letter](gpu, n, _p(dx))
    assert m == O.amax(letter, x)            # max is order-independent: exact
    assert abs(m - float(wide.max())) <= 4 * eps * float(wide.max())


def test_masum_mamax(gpu):
    from spgpu_amd import capi
    n, count, pitch = 700, 3, 704
    x = _vec("D", 13, count * pitch)
    dx = _dev(x)
    s, m = np.zeros(count), np.zeros(count)
    capi.masum["D"](gpu, C.c_void_p(s.ctypes.data), n, _p(dx), count, pitch)
    capi.mamax["D"](gpu, C.c_void_p(m.ctypes.data), n, _p(dx), count, pitch)
    for j in range(count):
        seg = x[j * pitch:j * pitch + n]
        assert abs(s[j] - np.abs(seg).sum()) <= 1e-12 * np.abs(seg).sum()
        assert m[j] == np.abs(seg).max()
