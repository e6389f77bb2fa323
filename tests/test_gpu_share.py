"""GPU: the kernel a row order selects (csrc/share_spmv.hip.h: shares of equal work, (sub-group, chunk) items, chunk sums
added in chunk order) against the oracle in that order, bit for bit: every type, HELL with hack sizes that take the
equal-work partition (32, 64, 96) and that do not (48, 2), ELL, every workgroup shape, tile and gathers, beta != 0 and in
place -- and the corners of the pass logic: a hack deeper than the chunk sums a pass can park, a share with more sub-groups
than a pass holds, empty rows, one row."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

# Round 4: neither kernel is in the product any more (a stream without a deep list now runs the planned kernel without a plan:
# stateless too, and in the ONE order of additions of every other ordered path); they live on in the -DSPGPU_TUNING_VARIANTS
# build for A/B runs, where these tests keep them honest.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif("not config._lab_build", reason="shareSpmvKernel / pipeSpmvKernel: -DSPGPU_TUNING_VARIANTS build only")]

# SPGPU_RAGGED: 2 = shareSpmvKernel (a workgroup per share), 3 = pipeSpmvKernel (a resident workgroup per CU, blocks prepared
# beside the stream); both add in the same order.  ("pipe", groups): SPGPU_PIPE_GROUPS, fewer workgroups than CUs so that a
# small matrix runs many blocks per workgroup
KERNELS = [("share", 0), ("pipe", 0), ("pipe", 3)]


def _select(tuning, kernel, **more):
    name, groups = kernel
    tuning(SPGPU_RAGGED={"share": 2, "pipe": 3}[name], SPGPU_PIPE_GROUPS=groups, **more)


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host_hell(n, lengths, letter, hack, rng, cols_n=None, near=None):
    """HELL on the host from row lengths (any order of the rows): random columns (near the row if asked), random values."""
    from spgpu_amd import formats
    cols_n = cols_n or n
    lengths = np.minimum(np.asarray(lengths, np.int64), cols_n)
    rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
    nnz = int(rows.size)
    if near:
        cols = (rows + rng.integers(-near, near + 1, nnz)) % cols_n
    else:
        cols = rng.integers(0, cols_n, nnz)
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    vals = rng.standard_normal(nnz).astype(real)
    if letter in "CZ":
        vals = (vals + 1j * rng.standard_normal(nnz).astype(real)).astype(O.NP_DTYPE[letter])
    ell = formats.coo_to_ell(n, rows, cols, vals)
    return ell, formats.ell_to_hell(ell, hack)


def _vec(rng, letter, n):
    real = O.NP_DTYPE[{"S": "S", "C": "S", "D": "D", "Z": "D"}[letter]]
    v = rng.standard_normal(n).astype(real)
    if letter in "CZ":
        v = (v + 1j * rng.standard_normal(n).astype(real)).astype(O.NP_DTYPE[letter])
    return v


def _run_and_compare(gpu, letter, hell, r_idx, rng, form="auto", betas=((1.0, 0.0, False), (-0.75, 0.5, False), (2.0, 1.0, True))):
    import torch
    from spgpu_amd import capi, formats
    n = hell["rows"]
    cols_n = int(hell["indices"].max()) + 1 if hell["indices"].size else 1
    x, y = _vec(rng, letter, max(cols_n, 1)), _vec(rng, letter, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    d = formats.DeviceHell(hell, r_idx=np.ascontiguousarray(r_idx, np.int32))
    shape_args = O.slab_shape(letter, "share")
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER if form == "gather" else capi.FORM_AUTO)
    try:
        for alpha, beta, in_place in betas:
            dz = dy.clone() if in_place else torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            d.spmv(gpu, dz, dz if in_place else (dy if beta != 0 else None), alpha, dx, beta)
            torch.cuda.synchronize()
            want = O.spmv_tail(hell, x, y if beta != 0 else None, alpha, beta, r_idx=r_idx, **shape_args)
            got = dz.cpu().numpy()
            if got.tobytes() != want.tobytes():
                bad = np.flatnonzero(got != want)
                raise AssertionError(f"{bad.size} of {n} rows differ, first z[{bad[0]}]: got {got[bad[0]]!r} want {want[bad[0]]!r} "
                                     f"(alpha {alpha}, beta {beta}, in place {in_place})")
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


LAB = pytest.mark.skipif("not config._lab_build", reason="non-default kernel shape: -DSPGPU_TUNING_VARIANTS build")


@pytest.mark.parametrize("kernel,shape,form", [(KERNELS[0], 0, "auto"), pytest.param(KERNELS[0], 1, "auto", marks=LAB), pytest.param(KERNELS[0], 2, "auto", marks=LAB), (KERNELS[0], 0, "gather"),
                                               (KERNELS[1], 0, "auto"), (KERNELS[1], 0, "gather"), (KERNELS[2], 0, "auto"), (KERNELS[2], 0, "gather")])
@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("hack", [32, 64, 96, 48, 2])
def test_share_kernel_bit_exact(gpu, tuning, kernel, letter, shape, form, hack):
    """Power-law lengths (up to 400) ordered by length in windows, through rIdx; HELL."""
    from spgpu_amd import formats, synth
    _select(tuning, kernel, SPGPU_RAGGED_SHAPE=shape)
    if letter == "S" and hack == 2:
        hack = 4     # a lane's 16-byte strip (4 fp32 rows) must not straddle a hack: hack 2 takes the narrow slab kernel
    n = 7000 + 5
    rng = np.random.default_rng(hack * 7 + shape)
    lengths = np.minimum(synth.power_law_lengths(n, 12.0, 400, 9), 600)
    r_idx, sorted_len = formats.oell_order(lengths, 512, 40)
    _, hell = _host_hell(n, sorted_len, letter, hack, rng, near=300)
    _run_and_compare(gpu, letter, hell, r_idx, rng, form)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("form", ["auto", "gather"])
def test_share_kernel_ell(gpu, tuning, kernel, letter, form):
    """ELL with a row order: consecutive rows per workgroup."""
    import torch
    from spgpu_amd import capi, formats, synth
    _select(tuning, kernel)
    n = 5000 + 3
    rng = np.random.default_rng(5)
    lengths = np.minimum(synth.power_law_lengths(n, 10.0, 300, 4), 300)
    r_idx, sorted_len = formats.oell_order(lengths, 1024, 0)
    ell, _ = _host_hell(n, sorted_len, letter, 32, rng, near=200)
    x, y = _vec(rng, letter, n), _vec(rng, letter, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    d = formats.DeviceEll(ell, r_idx=np.ascontiguousarray(r_idx, np.int32))
    capi.spgpuSetSpmvForm(gpu, capi.FORM_GATHER if form == "gather" else capi.FORM_AUTO)
    try:
        for alpha, beta in ((1.0, 0.0), (-0.5, 2.0)):
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            d.spmv(gpu, dz, dy if beta != 0 else None, alpha, dx, beta)
            torch.cuda.synchronize()
            want = O.spmv_tail(ell, x, y if beta != 0 else None, alpha, beta, r_idx=r_idx, **O.slab_shape(letter, "share"))
            assert dz.cpu().numpy().tobytes() == want.tobytes(), (alpha, beta)
    finally:
        capi.spgpuSetSpmvForm(gpu, capi.FORM_AUTO)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("letter", ["S", "D", "Z"])
@pytest.mark.parametrize("form,deepest", [("gather", 5000), ("auto", 9000), ("gather", 1700)])
def test_a_hack_deeper_than_a_pass_can_park(gpu, tuning, kernel, letter, form, deepest):
    """One hack far deeper than the chunk sums a pass holds (gather form: 16 KiB of LDS; tile form: half of 64 KiB): the
    sub-group is cut by pass boundaries and its carry waits in LDS.  Other deep hacks around it, then short rows."""
    _select(tuning, kernel)
    n = 3000
    rng = np.random.default_rng(deepest)
    lengths = rng.integers(0, 30, n)
    lengths[:32] = rng.integers(deepest // 2, deepest, 32)
    lengths[0] = deepest
    lengths[32:96] = rng.integers(100, 700, 64)
    lengths[1500:1532] = rng.integers(deepest // 3, deepest // 2, 32)   # a deep hack that is not the first of its share
    _, hell = _host_hell(n, lengths, letter, 32, rng, cols_n=12000)
    _run_and_compare(gpu, letter, hell, rng.permutation(n).astype(np.int32), rng, form, betas=((1.0, 0.0, False), (0.5, -1.0, False)))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("letter", ["D", "S"])
@pytest.mark.parametrize("form", ["auto", "gather"])
def test_a_share_with_more_sub_groups_than_a_pass(gpu, tuning, kernel, letter, form):
    """Half the rows hold 100 entries, the other half none: the equal-work shares of the empty half span hundreds of
    sub-groups and run in passes of 64."""
    _select(tuning, kernel)
    n = 40000 + 7
    rng = np.random.default_rng(3)
    lengths = np.where(np.arange(n) < n // 2, 100, 0)
    lengths[n // 2 + 5000] = 3
    _, hell = _host_hell(n, lengths, letter, 32, rng, near=500)
    _run_and_compare(gpu, letter, hell, rng.permutation(n).astype(np.int32), rng, form, betas=((1.0, 0.0, False), (2.0, 0.5, False)))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("n,hack", [(1, 32), (31, 32), (33, 32), (64, 64), (1000, 32), (1025, 64), (5000, 96)])
def test_small_and_empty(gpu, tuning, kernel, n, hack):
    """Few rows, empty rows, an all-empty matrix: every row still gets z = beta * y."""
    _select(tuning, kernel)
    rng = np.random.default_rng(n)
    for lengths in (rng.integers(0, 4, n) * rng.integers(0, 20, n), np.zeros(n, np.int64), np.full(n, 49)):
        _, hell = _host_hell(n, lengths, "D", hack, rng, cols_n=max(n, 60))
        _run_and_compare(gpu, "D", hell, rng.permutation(n).astype(np.int32), rng, "auto", betas=((1.5, 0.0, False), (1.0, 1.0, True)))


@pytest.mark.parametrize("kernel", [KERNELS[0], KERNELS[1]])
def test_many_blocks_per_workgroup(gpu, tuning, kernel):
    """1.5 M rows of power-law lengths ordered in windows: every resident workgroup of the pipelined kernel walks several
    blocks, with deep hacks at the head of the order."""
    from spgpu_amd import formats, synth
    _select(tuning, kernel)
    n = 1_500_000 + 11
    rng = np.random.default_rng(8)
    lengths = synth.power_law_lengths(n, 8.0, 1500, 3)
    r_idx, sorted_len = formats.oell_order(lengths, 2048, 200)
    _, hell = _host_hell(n, sorted_len, "D", 32, rng, near=1000)
    _run_and_compare(gpu, "D", hell, r_idx, rng, "auto", betas=((1.0, 0.0, False), (0.5, 2.0, False)))
