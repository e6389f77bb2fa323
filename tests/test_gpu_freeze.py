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
def test_frozen_call_equals_unfrozen_call_and_oracle(gpu, letter, window, long_rows, aligned, hack, pattern, near, tuning):
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_FREEZE_MAX_ESCAPES_PCT=100)    # the escape paths are what some of these cases are for: freeze whatever the share of escapes
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


def test_negative_columns_and_base_index_one(gpu, tuning):
    """Slots whose stored column is below baseIndex are skipped by every kernel of the family (col >= 0); frozen they are escapes.
    baseIndex 1 with the packed words counted from the 0-based column."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_FREEZE_MAX_ESCAPES_PCT=100)            # (one slot in 97 is a hole here: just over the default share)
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
    """Complex fp64 (16-byte elements: one row per lane, no 2 048-row shape, no wide slab loads): a plan, no packed form --
    with and without a row order.  They stay what they were."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 4 * 2048
    hz = _matrix(gpu, n, "Z", 2048, 60, True)
    assert capi.spgpuHellSpmvFreeze(gpu, capi.TYPE_CODE["Z"], _dp(hz["cM"]), _dp(hz["rP"]), 32, _dp(hz["hack_offsets"]), _dp(hz["rS"]), None, n, 0) == capi.SPGPU_UNSUPPORTED
    assert _freeze(gpu, "Z", hz, n) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    x = synth.values_for("Z", 3, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(hz, "Z", n), x, None, 1.0, 0.0, r_idx=hz["rIdx"].cpu().numpy(), **O.slab_shape("Z", "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, "Z", hz, n, dz, None, dx, 1.0, 0.0)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == want.tobytes()


def test_ordered_matrix_with_scattered_columns_keeps_its_plan_but_gets_no_copy(gpu):
    """More than one escape in a hundred entries (columns all over x): Freeze says SPGPU_UNSUPPORTED, holds no memory, and the calls
    run from the plan as before."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, letter = 300 * 1024, "D"                         # (columns must reach beyond 16 bits of a block's lowest to be escapes)
    h = _matrix(gpu, n, letter, 2048, 60, True, longest=300, pattern="random")
    assert _freeze(gpu, letter, h, n) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    x = synth.values_for(letter, 3, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(h, letter, n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    uses0 = capi.plan_counts(gpu)[0]
    _call(gpu, letter, h, n, dz, None, dx, 1.0, 0.0)
    torch.cuda.synchronize()
    assert capi.plan_counts(gpu)[0] == uses0 + 1        # planned (Freeze prepared it), unfrozen
    assert dz.cpu().numpy().tobytes() == want.tobytes()


def test_frozen_ell(gpu, tuning):
    """spgpuEllSpmvFreeze: ELL with a row order (the arrays ellToOell leaves, reference ell.c:161-202) through the same kernels;
    a random order scatters a block's rows over the matrix -- offsets beside the tile and escapes."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_FREEZE_MAX_ESCAPES_PCT=100)
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


# ---- matrices WITHOUT a row order: the default kernels' frozen form (ellpack_spmv.hip freezeSlab, slabSpmvKernel<..., PACKED>) ----

def _hell_of(coo, letter, hack=32, base=0):
    from spgpu_amd import formats
    n, _, r, c, v = coo
    ell = formats.coo_to_ell(n, r, c, v, coo_base=base, ell_base=base)
    return ell, formats.ell_to_hell(ell, hack)


def _freeze_plain(gpu, dev):
    from spgpu_amd import capi
    return capi.spgpuHellSpmvFreeze(gpu, capi.TYPE_CODE[dev.letter], _dp(dev.cM), _dp(dev.rP), dev.hack_size, _dp(dev.hack_offsets), _dp(dev.rS),
                                    None, dev.rows, dev.base)


def _spmv(gpu, dev, x, y, alpha, beta, avg=0):
    import torch
    from spgpu_amd import formats
    dx = formats.to_device(x)
    dy = formats.to_device(y) if y is not None else None
    dz = torch.full((dev.rows,), float("nan"), dtype=dx.dtype, device="cuda")
    dev.spmv(gpu, dz, dy, alpha, dx, beta, avg)
    torch.cuda.synchronize()
    return dz.cpu().numpy()


@pytest.mark.parametrize("letter", ["D", "S", "C"])
@pytest.mark.parametrize("kind", ["band", "band_base1", "near", "ragged_near", "far_escapes"])
def test_frozen_matrix_without_row_order_same_bits(gpu, letter, kind):
    """BASELINE configs[1]'s kind of matrix (no rIdx: the default kernels): band columns run the strip form, columns near the row
    the gather form, ragged lengths the whole-wave tail rows (which read rP itself); `far_escapes`: one entry in 200 lies far
    away (0xFFFF -> rP).  Frozen == unfrozen == oracle, bit for bit; several calls (AUTO settles on its form on the way)."""
    from spgpu_amd import capi, formats, synth
    n = 40 * 128 + 50
    base = 1 if kind == "band_base1" else 0
    rng = np.random.default_rng(11)
    if kind.startswith("band"):
        coo = synth.banded_coo(n, 16, letter, seed=3, base=base)
    else:
        lengths = np.full(n, 24, np.int64) if kind != "ragged_near" else np.minimum(rng.zipf(1.6, size=n) + 3, 400)
        rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
        k = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lengths) - lengths, lengths)
        span = np.repeat(np.maximum(lengths * 3, 64), lengths)
        cols = (rows - span // 2 + (k * span) // np.repeat(lengths, lengths)) % n       # ascending inside a row, within +-span/2
        if kind == "far_escapes":
            far = rng.random(rows.size) < 0.005
            cols = np.where(far, (cols + n // 2 + 70000) % n, cols)
        coo = (n, n, rows.astype(np.int32), cols.astype(np.int32), synth.values_for(letter, 5, rows.size))
    ell, hell = _hell_of(coo, letter, base=base)
    dev = formats.DeviceHell(hell)
    x, y = synth.values_for(letter, 31, n), synth.values_for(letter, 32, n)
    want = O.default_spmv(hell, x, y, 1.5, -0.25)
    for _ in range(3):
        assert _spmv(gpu, dev, x, y, 1.5, -0.25).tobytes() == want.tobytes()
    assert _freeze_plain(gpu, dev) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) >= 2 * hell["values"].size
    uses0 = capi.plan_counts(gpu)[0]
    for _ in range(4):
        assert _spmv(gpu, dev, x, y, 1.5, -0.25).tobytes() == want.tobytes()
    used = capi.plan_counts(gpu)[0] - uses0
    # every call found the frozen record -- unless AUTO has settled on the LDS-tile form for this matrix (columns inside a window
    # an LDS tile holds, rows longer than a stage): that form has no packed variant and runs as before
    assert used == 4 if kind.startswith("band") else used in (0, 4), used
    want0 = O.default_spmv(hell, x, None, 2.0, 0.0)
    assert _spmv(gpu, dev, x, None, 2.0, 0.0).tobytes() == want0.tobytes()
    assert capi.spgpuSpmvThaw(gpu, _dp(dev.rP)) == capi.SPGPU_SUCCESS
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    uses1 = capi.plan_counts(gpu)[0]
    assert _spmv(gpu, dev, x, y, 1.5, -0.25).tobytes() == want.tobytes()
    assert capi.plan_counts(gpu)[0] == uses1


@pytest.mark.parametrize("letter", ["D", "S"])
def test_frozen_gather_form_without_row_order(gpu, letter):
    """The gather kernel's packed variant, asked for by the handle's form hint (AUTO takes the LDS tile for such columns): columns
    near the row, ragged lengths (whole-wave tail rows read rP), escapes."""
    from spgpu_amd import capi, formats, synth
    n = 30 * 128 + 9
    rng = np.random.default_rng(3)
    lengths = np.minimum(rng.zipf(1.7, size=n) + 5, 300)
    rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
    k = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    span = np.repeat(np.maximum(lengths * 4, 64), lengths)
    cols = (rows - span // 2 + (k * span) // np.repeat(lengths, lengths)) % n
    cols = np.where(rng.random(rows.size) < 0.004, (cols + 90000) % n, cols)
    coo = (n, n, rows.astype(np.int32), cols.astype(np.int32), synth.values_for(letter, 5, rows.size))
    ell, hell = _hell_of(coo, letter)
    dev = formats.DeviceHell(hell)
    x = synth.values_for(letter, 7, n)
    want = O.default_spmv(hell, x, None, 1.0, 0.0)
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER)
    try:
        assert _spmv(gpu, dev, x, None, 1.0, 0.0).tobytes() == want.tobytes()
        assert _freeze_plain(gpu, dev) == capi.SPGPU_SUCCESS
        uses0 = capi.plan_counts(gpu)[0]
        for _ in range(3):
            assert _spmv(gpu, dev, x, None, 1.0, 0.0).tobytes() == want.tobytes()
        assert capi.plan_counts(gpu)[0] - uses0 == 3
        capi.spgpuSetSpmvForm(gpu, capi.FORM_STRIPS)          # the strip-capable kernel on columns that are no strips
        assert _spmv(gpu, dev, x, None, 1.0, 0.0).tobytes() == want.tobytes()
        assert capi.plan_counts(gpu)[0] - uses0 == 4
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)
        capi.spgpuSpmvThaw(gpu, _dp(dev.rP))


def test_scattered_matrix_without_row_order_is_not_frozen(gpu):
    """Columns all over x: more than one entry in a hundred would be an escape -- nothing is frozen, nothing changes."""
    from spgpu_amd import capi, formats, synth
    n = 200_000
    coo = synth.random_rows_coo(n, n, np.full(n, 8), seed=4, letter="D")
    ell, hell = _hell_of(coo, "D")
    dev = formats.DeviceHell(hell)
    assert _freeze_plain(gpu, dev) == capi.SPGPU_UNSUPPORTED
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
    x = synth.values_for("D", 2, n)
    assert _spmv(gpu, dev, x, None, 1.0, 0.0).tobytes() == O.default_spmv(hell, x, None, 1.0, 0.0).tobytes()


def test_frozen_ell_without_row_order(gpu):
    from spgpu_amd import capi, formats, synth
    n = 30 * 128 + 7
    coo = synth.banded_coo(n, 12, "D", seed=9)
    ell, _ = _hell_of(coo, "D")
    dev = formats.DeviceEll(ell)
    x, y = synth.values_for("D", 1, n), synth.values_for("D", 2, n)
    want = O.default_spmv(ell, x, y, -1.0, 0.5)
    assert capi.spgpuEllSpmvFreeze(gpu, capi.TYPE_CODE["D"], _dp(dev.cM), _dp(dev.rP), dev.pitch, dev.pitch, _dp(dev.rS), None, dev.max_row, n, 0) == capi.SPGPU_SUCCESS
    uses0 = capi.plan_counts(gpu)[0]
    for _ in range(3):
        assert _spmv(gpu, dev, x, y, -1.0, 0.5).tobytes() == want.tobytes()
    assert capi.plan_counts(gpu)[0] - uses0 == 3
    assert capi.spgpuSpmvThaw(gpu, _dp(dev.rP)) == capi.SPGPU_SUCCESS


def test_captured_launches_run_unfrozen_and_survive_a_thaw(gpu):
    """A launch captured into a graph must not carry the frozen copy's address (a graph outlives a Thaw): captured calls run as
    unfrozen calls -- the record is not used -- and the graph replays with the oracle's bits after the matrix was thawed."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 24 * 128
    ell, hell = _hell_of(synth.banded_coo(n, 16, "D", seed=3), "D")
    dev = formats.DeviceHell(hell)
    assert _freeze_plain(gpu, dev) == capi.SPGPU_SUCCESS
    dx = formats.to_device(synth.values_for("D", 1, n))
    dz = torch.zeros(n, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()
    try:
        with torch.cuda.stream(side):
            dev.spmv(gpu, dz, None, 1.0, dx, 0.0)            # outside the capture: from the frozen record
        side.synchronize()
        uses = capi.plan_counts(gpu)[0]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            dev.spmv(gpu, dz, None, 1.0, dx, 0.0)
        assert capi.plan_counts(gpu)[0] == uses               # the captured launch did not look the record up
        assert capi.spgpuSpmvThaw(gpu, _dp(dev.rP)) == capi.SPGPU_SUCCESS
        for rep in range(3):
            x = synth.values_for("D", 50 + rep, n)
            dx.copy_(formats.to_device(x))
            dz.fill_(float("nan"))
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert dz.cpu().numpy().tobytes() == O.default_spmv(hell, x, None, 1.0, 0.0).tobytes(), rep
    finally:
        capi.spgpuSetStream(gpu, None)


def test_ninth_frozen_matrix_evicts_the_least_recently_used(gpu):
    """The handle keeps 8 records: a ninth frozen matrix takes the place of the least recently used one, whose calls simply run
    unfrozen again (same bits); the memory of its copy is given back."""
    from spgpu_amd import capi, formats, synth
    n = 16 * 128
    mats = []
    for i in range(9):
        ell, hell = _hell_of(synth.banded_coo(n, 8 + i, "D", seed=20 + i), "D")
        mats.append((formats.DeviceHell(hell), hell))
    x = synth.values_for("D", 77, n)
    for dev, hell in mats:
        assert _freeze_plain(gpu, dev) == capi.SPGPU_SUCCESS
        assert _spmv(gpu, dev, x, None, 1.0, 0.0).tobytes() == O.default_spmv(hell, x, None, 1.0, 0.0).tobytes()
    uses = capi.plan_counts(gpu)[0]
    dev0, hell0 = mats[0]                                     # the first one has lost its record
    assert _spmv(gpu, dev0, x, None, 1.0, 0.0).tobytes() == O.default_spmv(hell0, x, None, 1.0, 0.0).tobytes()
    assert capi.plan_counts(gpu)[0] == uses
    dev8, hell8 = mats[8]
    assert _spmv(gpu, dev8, x, None, 1.0, 0.0).tobytes() == O.default_spmv(hell8, x, None, 1.0, 0.0).tobytes()
    assert capi.plan_counts(gpu)[0] == uses + 1
    for dev, _ in mats:
        capi.spgpuSpmvThaw(gpu, _dp(dev.rP))
    assert capi.spgpuSpmvFrozenBytes(gpu) == 0
