"""GPU: ADOPTED matrices (include/spgpu/tuning.h spgpuHellSpmvAdopt; spgpu_amd/csrc/adopted_hell.hip).

A ragged HELL matrix handed over as its rows come (no rIdx) -- what the reference's plain kernel takes
(hell_spmv_base_template.cuh:112-225) -- of which the library, under the caller's promise not to touch its arrays, keeps an ordered,
frozen copy of its own: what the reference's harness does by hand with ellToOell + rIdx (hellPerf.cpp:324-378).  Pinned here: a
call on the CALLER's arrays after Adopt writes z in the caller's row order, with the bits of (a) the oracle run in the ordered
kernel's order on the matrix ordered the documented way (spgpuOellOrderAlignedDevice, 2 048 / 256) and (b) the library's own call
on a matrix the test ordered itself with the same device calls; within the north_star tolerance of the plain call; Thaw gives
the plain call's bits back; the copy's memory is counted and returned."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
TOL = {"S": 1e-4, "C": 1e-4, "D": 1e-6, "Z": 1e-6}


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _coo(n, letter, pattern, near, longest, mean, seed):
    import torch
    from spgpu_amd import synth
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, mean, longest, seed + 2), longest)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, near, real, seed=seed)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    return rows_t, cols_t, vals_t


def _host(h, letter, n, hack=32):
    return dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
                hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=hack, row_lengths=h["rS"][:n].cpu().numpy(), base=0)


def _call(gpu, letter, h, n, dz, dy, dx, alpha, beta, hack=32, r_idx="own"):
    from spgpu_amd import capi
    capi.hellspmv[letter](gpu, _dp(dz), _dp(dy) if beta != 0 else None, capi.scalar(letter, alpha), _dp(h["cM"]), _dp(h["rP"]), hack,
                          _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]) if r_idx == "own" else None, 12, n, _dp(dx), capi.scalar(letter, beta), 0)


@pytest.mark.parametrize("letter", ["D", "S", "C", "Z"])
@pytest.mark.parametrize("pattern,near,hack", [("near", 500, 32), ("band", 0, 32), ("near", 500, 64)])
def test_adopted_call_equals_the_ordered_call_and_oracle(gpu, letter, pattern, near, hack):
    import torch
    from spgpu_amd import capi, formats, synth
    n = 9 * 2048 + 77
    coo = _coo(n, letter, pattern, near, 900, 12.0, 7)
    plain = formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, hack, 0, 0, order=False)
    ordered = formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, hack, 2048, 256, aligned=True)
    x, y = synth.values_for(letter, 31, n), synth.values_for(letter, 32, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    alpha, beta = -0.5, 2.0
    want_plain = O.default_spmv(_host(plain, letter, n, hack), x, y, alpha, beta)
    want = O.spmv_tail(_host(ordered, letter, n, hack), x, y, alpha, beta, r_idx=ordered["rIdx"].cpu().numpy(),
                       **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, letter, plain, n, dz, dy, dx, alpha, beta, hack, r_idx=None)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want_plain.tobytes()
    code = capi.TYPE_CODE[letter]
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    assert capi.spgpuHellSpmvAdopt(gpu, code, _dp(plain["cM"]), _dp(plain["rP"]), hack, _dp(plain["hack_offsets"]), _dp(plain["rS"]), n, 0) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) >= ordered["slots"] * (plain["cM"].element_size() + 4)
    assert capi.spgpuHellSpmvAdopt(gpu, code, _dp(plain["cM"]), _dp(plain["rP"]), hack, _dp(plain["hack_offsets"]), _dp(plain["rS"]), n, 0) == capi.SPGPU_SUCCESS
    uses0 = capi.spgpuSpmvAdoptedUses(gpu)
    for _ in range(3):
        dz.fill_(float("nan"))
        _call(gpu, letter, plain, n, dz, dy, dx, alpha, beta, hack, r_idx=None)     # the CALLER's arrays, no rIdx
        torch.cuda.synchronize()
        got = dz.cpu().numpy()
        assert got.tobytes() == want.tobytes()
    assert capi.spgpuSpmvAdoptedUses(gpu) - uses0 == 3
    # the same bits as the library's own call on the matrix the test ordered itself
    dz2 = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, letter, ordered, n, dz2, dy, dx, alpha, beta, hack)
    torch.cuda.synchronize()
    assert dz2.cpu().numpy().tobytes() == got.tobytes()
    # and within the north_star tolerance of the plain call (another order of additions)
    scale = np.abs(alpha) * np.abs(want_plain - beta * y) + np.abs(beta * y) + np.finfo(np.float64).tiny
    assert np.max(np.abs(got - want_plain) / (TOL[letter] * (scale + np.max(np.abs(want_plain))))) <= 1.0
    # in place, beta = 0
    want0 = O.spmv_tail(_host(ordered, letter, n, hack), x, None, 1.25, 0.0, r_idx=ordered["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz.fill_(float("nan"))
    _call(gpu, letter, plain, n, dz, None, dx, 1.25, 0.0, hack, r_idx=None)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want0.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(plain["rP"])) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    dz.fill_(float("nan"))
    _call(gpu, letter, plain, n, dz, dy, dx, alpha, beta, hack, r_idx=None)         # thawed: the plain kernel again
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want_plain.tobytes()


def test_a_call_with_a_row_order_of_its_own_is_not_redirected(gpu):
    """Adopt keys the caller's arrays WITHOUT rIdx: the same arrays passed with a row order run as given."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, letter = 4 * 2048, "D"
    coo = _coo(n, letter, "near", 400, 600, 10.0, 3)
    plain = formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, 32, 0, 0, order=False)
    assert capi.spgpuHellSpmvAdopt(gpu, capi.TYPE_CODE[letter], _dp(plain["cM"]), _dp(plain["rP"]), 32, _dp(plain["hack_offsets"]), _dp(plain["rS"]), n, 0) == capi.SPGPU_SUCCESS
    perm = torch.randperm(n, device="cuda", dtype=torch.int64).to(torch.int32)
    x = synth.values_for(letter, 5, n)
    dx = formats.to_device(x)
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    uses0 = capi.spgpuSpmvAdoptedUses(gpu)
    capi.hellspmv[letter](gpu, _dp(dz), None, 1.0, _dp(plain["cM"]), _dp(plain["rP"]), 32, _dp(plain["hack_offsets"]), _dp(plain["rS"]), _dp(perm), 12, n,
                          _dp(dx), 0.0, 0)
    torch.cuda.synchronize()
    assert capi.spgpuSpmvAdoptedUses(gpu) == uses0
    want = O.spmv_tail(_host(plain, letter, n), x, None, 1.0, 0.0, r_idx=perm.cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(plain["rP"])) == capi.SPGPU_SUCCESS


def test_fifth_matrix_is_not_adopted_and_odd_hack_sizes_are_refused(gpu):
    from spgpu_amd import capi, formats
    n, letter = 2 * 2048, "D"
    mats = []
    for i in range(5):
        coo = _coo(n, letter, "near", 300, 400, 8.0, 20 + i)
        mats.append(formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, 32, 0, 0, order=False))
    code = capi.TYPE_CODE[letter]
    said = [capi.spgpuHellSpmvAdopt(gpu, code, _dp(m["cM"]), _dp(m["rP"]), 32, _dp(m["hack_offsets"]), _dp(m["rS"]), n, 0) for m in mats]
    assert said == [capi.SPGPU_SUCCESS] * 4 + [capi.SPGPU_UNSUPPORTED]
    for m in mats[:4]:
        assert capi.spgpuSpmvThaw(gpu, _dp(m["rP"])) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    coo = _coo(n, letter, "near", 300, 400, 8.0, 40)
    odd = formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, 48, 0, 0, order=False)
    assert capi.spgpuHellSpmvAdopt(gpu, code, _dp(odd["cM"]), _dp(odd["rP"]), 48, _dp(odd["hack_offsets"]), _dp(odd["rS"]), n, 0) == capi.SPGPU_UNSUPPORTED


def test_a_matrix_with_rows_of_equal_length_is_not_adopted(gpu):
    """Nothing to gain from another order: SPGPU_UNSUPPORTED, no memory held (Freeze is the call for such a matrix)."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 20 * 1024
    h = synth.hell_uniform_on_device(n, 16, "banded", "D", 32, seed=1)
    assert capi.spgpuHellSpmvAdopt(gpu, capi.TYPE_CODE["D"], _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), n, 0) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0


@pytest.mark.parametrize("letter", ["D", "S"])
def test_adopted_ell_runs_on_an_ordered_hell_copy(gpu, letter):
    """spgpuEllSpmvAdopt: a ragged ELL matrix (rows x maxNnzPerRow slots) of which the library keeps an ordered HELL copy; the
    spgpu?ellspmv call without rIdx then gives the bits of the explicitly ordered HELL call (the same COO through
    spgpuOellOrderAlignedDevice + spgpuCooToHellDevice) and of the oracle; Thaw gives the plain ELL kernel's bits back."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 5 * 2048 + 100
    coo = _coo(n, letter, "near", 400, 600, 10.0, 9)
    rows_h, cols_h, vals_h = (t.cpu().numpy() for t in coo)
    ell = formats.coo_to_ell(n, rows_h, cols_h, vals_h)
    dev = formats.DeviceEll(ell)
    ordered = formats.coo_to_ordered_hell_device(gpu, n, *coo, letter, 32, 2048, 256, aligned=True)
    x, y = synth.values_for(letter, 3, n), synth.values_for(letter, 4, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    want_plain = O.default_spmv(ell, x, y, 1.5, -1.0)
    want = O.spmv_tail(_host(ordered, letter, n), x, y, 1.5, -1.0, r_idx=ordered["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))

    def product():
        dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
        dev.spmv(gpu, dz, dy, 1.5, dx, -1.0)
        torch.cuda.synchronize()
        return dz.cpu().numpy()

    assert product().tobytes() == want_plain.tobytes()
    assert capi.spgpuEllSpmvAdopt(gpu, capi.TYPE_CODE[letter], _dp(dev.cM), _dp(dev.rP), dev.pitch, dev.pitch, _dp(dev.rS), dev.max_row, n, 0) == capi.SPGPU_SUCCESS
    assert 0 < capi.spgpuSpmvFrozenBytes(gpu) < dev.pitch * dev.max_row * (dev.cM.element_size() + 4)     # the copy is smaller than the caller's ELL
    uses0 = capi.spgpuSpmvAdoptedUses(gpu)
    assert product().tobytes() == want.tobytes()
    assert capi.spgpuSpmvAdoptedUses(gpu) == uses0 + 1
    assert capi.spgpuEllSpmvAdopt(gpu, capi.TYPE_CODE[letter], _dp(dev.cM), _dp(dev.rP), dev.pitch, dev.pitch, None, dev.max_row, n, 0) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvThaw(gpu, _dp(dev.rP)) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    assert product().tobytes() == want_plain.tobytes()


def test_optimize_picks_adopt_freeze_or_nothing(gpu):
    """spgpuHellSpmvOptimize: a ragged matrix without an order is adopted, a band matrix frozen, a matrix with scattered columns and
    rows of equal length left as it is, an ordered matrix frozen -- and every later call gives the bits the chosen path is pinned to."""
    import torch
    from spgpu_amd import capi, formats, synth
    code = capi.TYPE_CODE["D"]
    n = 4 * 2048
    ragged = formats.coo_to_ordered_hell_device(gpu, n, *_coo(n, "D", "near", 400, 600, 10.0, 5), "D", 32, 0, 0, order=False)
    band = synth.hell_uniform_on_device(n, 16, "banded", "D", 32, seed=1)
    big = 300 * 1024
    scattered = synth.hell_uniform_on_device(big, 8, "random", "D", 32, seed=2)
    ordered = formats.coo_to_ordered_hell_device(gpu, n, *_coo(n, "D", "near", 400, 600, 10.0, 6), "D", 32, 2048, 256, aligned=True)
    opt = lambda h, rows, r_idx=None: capi.spgpuHellSpmvOptimize(gpu, code, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(r_idx), rows, 0)
    assert opt(ragged, n) == capi.SPMV_ADOPTED
    assert opt(band, n) == capi.SPMV_FROZEN
    assert opt(scattered, big) == capi.SPMV_AS_IS
    assert opt(ordered, n, ordered["rIdx"]) == capi.SPMV_FROZEN
    x = synth.values_for("D", 8, n)
    dx = formats.to_device(x)
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, "D", ordered, n, dz, None, dx, 1.0, 0.0)
    torch.cuda.synchronize()
    want = O.spmv_tail(_host(ordered, "D", n), x, None, 1.0, 0.0, r_idx=ordered["rIdx"].cpu().numpy(), **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    for h in (ragged, band, ordered):
        assert capi.spgpuSpmvThaw(gpu, _dp(h["rP"])) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
