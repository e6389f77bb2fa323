"""GPU: FROZEN matrices (include/spgpu/tuning.h spgpu?SpmvFreeze; spgpu_amd/csrc/planned_spmv.hip planPackKernel,
ragged_spmv.hip.h PACKED).

The reference multiplies by one ordered matrix thousands of times (hellPerf.cpp:333-378); a caller who promises not to touch
the index arrays in between lets the library keep a 16-bit copy of the column indices with the matrix' plan, and the same
spgpu?hellspmv / spgpu?ellspmv calls then read 2 bytes of index per stored entry instead of 4.  What these tests pin: a frozen
call gives the bits of the unfrozen call and of the oracle (tests/oracle_api.py: the queue kernel's order) -- for columns
inside the block's LDS tile, beside it (16-bit offsets that are gathered from global memory), beyond the 16 bits (escapes that
still ask rP) and for negative column numbers; coefficients may change under a frozen matrix; Thaw ends it; the calls that
have no packed form say so and stay as they were."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _matrix(gpu, n, letter, window, long_rows, aligned, hack=32, mean=12.0, longest=600, pattern="near", seed=7, near=300):
    import torch
    from spgpu_amd import formats, synth
    real = {"S": "S", "D": "D", "C": "S", "Z": "D"}[letter]
    lengths = np.minimum(synth.power_law_lengths(n, mean, longest, seed + 2), longest)
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, near, real, seed=seed)
    if letter in "CZ":
        vals_t = torch.complex(vals_t, torch.flip(vals_t, [0]))
    return formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, letter, hack, window, long_rows, aligned=aligned)


def _host(h, letter, n, hack=32):
    return dict(letter=letter, rows=n, values=h["cM"][:h["slots"]].cpu().numpy(), indices=h["rP"][:h["slots"]].cpu().numpy(),
                hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=hack, row_lengths=h["rS"][:n].cpu().numpy(), base=h.get("base", 0))


def _call(gpu, letter, h, n, dz, dy, dx, alpha, beta, hack=32, base=0):
    from spgpu_amd import capi
    capi.hellspmv[letter](gpu, _dp(dz), _dp(dy) if beta != 0 else None, capi.scalar(letter, alpha), _dp(h["cM"]), _dp(h["rP"]), hack,
                          _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]), 12, n, _dp(dx), capi.scalar(letter, beta), base)


def _freeze(gpu, letter, h, n, hack=32, base=0):
    from spgpu_amd import capi
    return capi.spgpuHellSpmvFreeze(gpu, capi.TYPE_CODE[letter], _dp(h["cM"]), _dp(h["rP"]), hack, _dp(h["hack_offsets"]), _dp(h["rS"]),
                                    _dp(h["rIdx"]), n, base)


@pytest.mark.parametrize("letter", ["S", "D", "C"])
@pytest.mark.parametrize("window,long_rows,aligned,hack,pattern,near", [
    (2048, 60, True, 32, "near", 500),      # the aligned order: 2 048-row staged shape, every column inside the tile
    (2048, 60, True, 32, "band", 0),        # consecutive columns
    (512, 40, False, 32, "near", 500),      # drifting windows: the 1 024-row shape
    (2048, 60, True, 64, "near", 6000),     # columns +-6 000: beside the tile -- 16-bit offsets gathered from global memory
    (0, 0, False, 32, "near", 500),         # one global sort: a block's rows come from everywhere -- escapes (0xFFFF -> rP) beside offsets
    (2048, 60, True, 32, "random", 0),      # scattered over all of x: mostly escapes
])
def test_frozen_call_equals_unfrozen_call_and_oracle(gpu, letter, window, long_rows, aligned, hack, pattern, near):
    import torch
    from spgpu_amd import capi, formats, synth
    n = 80 * 1024 + 77 if pattern == "random" or window == 0 else 9 * 2048 + 77
    h = _matrix(gpu, n, letter, window, long_rows, aligned, hack=hack, longest=900, pattern=pattern, near=near)
    sub, r_idx = _host(h, letter, n, hack), h["rIdx"].cpu().numpy()
    x, y = synth.values_for(letter, 31, n), synth.values_for(letter, 32, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    want = O.spmv_tail(sub, x, y, -0.5, 2.0, r_idx=r_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)           # unfrozen (first call: no plan yet)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    assert _freeze(gpu, letter, h, n, hack) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) >= 2 * h["slots"]
    assert _freeze(gpu, letter, h, n, hack) == capi.SPGPU_SUCCESS   # already frozen
    uses0 = capi.plan_counts(gpu)[0]
    for _ in range(3):
        dz.fill_(float("nan"))
        _call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.plan_counts(gpu)[0] - uses0 == 3                    # every frozen call ran from the plan
    # in place (z = y), beta = 0
    want0 = O.spmv_tail(sub, x, None, 1.25, 0.0, r_idx=r_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz.fill_(float("nan"))
    _call(gpu, letter, h, n, dz, None, dx, 1.25, 0.0, hack)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want0.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(h["rP"])) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    assert capi.spgpuSpmvThaw(gpu, _dp(h["rP"])) == capi.SPGPU_UNSUPPORTED
    dz.fill_(float("nan"))
    _call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)           # thawed: analysed again behind this call
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()


def test_coefficients_may_change_under_a_frozen_matrix(gpu):
    """The promise covers the index arrays only: the coefficients are read from the caller's array at every call."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, letter = 7 * 2048, "D"
    h = _matrix(gpu, n, letter, 2048, 60, True, longest=700, near=400)
    x = synth.values_for(letter, 5, n)
    dx = formats.to_device(x)
    assert _freeze(gpu, letter, h, n) == capi.SPGPU_SUCCESS
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")
    for scale in (1.0, -3.0):
        h["cM"].mul_(scale)
        torch.cuda.synchronize()
        want = O.spmv_tail(_host(h, letter, n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
        dz.fill_(float("nan"))
        _call(gpu, letter, h, n, dz, None, dx, 1.0, 0.0)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(h["rP"])) == capi.SPGPU_SUCCESS


def test_negative_columns_and_base_index_one(gpu):
    """Slots whose stored column is below baseIndex are skipped by every kernel of the family (col >= 0); frozen they are escapes.
    baseIndex 1 with the packed words counted from the 0-based column."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, letter = 5 * 2048 + 64, "D"
    h = _matrix(gpu, n, letter, 2048, 60, True, longest=500)
    h["rP"].add_(1)                                     # 1-based
    slots = h["slots"]
    holes = torch.arange(7, slots, 97, device="cuda")
    h["rP"][holes] = 0                                  # column -1 in a 1-based matrix: never used
    torch.cuda.synchronize()
    h["base"] = 1
    sub, r_idx = _host(h, letter, n), h["rIdx"].cpu().numpy()
    x = synth.values_for(letter, 9, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(sub, x, None, 2.0, 0.0, r_idx=r_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, letter, h, n, dz, None, dx, 2.0, 0.0, base=1)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert _freeze(gpu, letter, h, n, base=1) == capi.SPGPU_SUCCESS
    dz.fill_(float("nan"))
    _call(gpu, letter, h, n, dz, None, dx, 2.0, 0.0, base=1)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(h["rP"])) == capi.SPGPU_SUCCESS


def test_calls_without_a_packed_form_say_so(gpu):
    """No row order: no plan, nothing to freeze.  Complex fp64 (16-byte elements: one row per lane, the 2 048-row shape does not
    exist): a plan, no packed form.  Both stay what they were."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 4 * 2048
    h = _matrix(gpu, n, "D", 2048, 60, True)
    code = capi.TYPE_CODE["D"]
    assert capi.spgpuHellSpmvFreeze(gpu, code, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), None, n, 0) == capi.SPGPU_UNSUPPORTED
    hz = _matrix(gpu, n, "Z", 2048, 60, True)
    assert _freeze(gpu, "Z", hz, n) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    x = synth.values_for("Z", 3, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(hz, "Z", n), x, None, 1.0, 0.0, r_idx=hz["rIdx"].cpu().numpy(), **O.slab_shape("Z", "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, "Z", hz, n, dz, None, dx, 1.0, 0.0)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()


def test_frozen_ell(gpu):
    """spgpuEllSpmvFreeze: ELL with a row order (the arrays ellToOell leaves, reference ell.c:161-202) through the same kernels;
    a random order scatters a block's rows over the matrix -- offsets beside the tile and escapes."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 6000
    lengths = np.minimum(np.random.default_rng(5).zipf(1.5, size=n), 300)
    _, _, r, c, v = synth.random_rows_coo(n, n, lengths, seed=6, letter="D")
    ell = formats.coo_to_ell(n, r, c, v)
    perm = np.random.default_rng(7).permutation(n).astype(np.int32)
    dev = formats.DeviceEll(ell, r_idx=perm)
    x = synth.values_for("D", 99, n)
    dx = formats.to_device(x)
    want = O.default_spmv(ell, x, None, 1.0, 0.0, r_idx=perm)
    torch.cuda.synchronize()
    assert capi.spgpuEllSpmvFreeze(gpu, capi.TYPE_CODE["D"], _dp(dev.cM), _dp(dev.rP), dev.pitch, dev.pitch, _dp(dev.rS), _dp(dev.rIdx), dev.max_row,
                                   n, 0) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) >= 2 * dev.pitch * dev.max_row
    uses = capi.plan_counts(gpu)[0]
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    dev.spmv(gpu, dz, None, 1.0, dx, 0.0)
    torch.cuda.synchronize()
    assert capi.plan_counts(gpu)[0] == uses + 1
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(dev.rP)) == capi.SPGPU_SUCCESS
