"""GPU: the per-matrix plan of the ELL/HELL SpMV with a row order (spgpu_amd/csrc/planned_spmv.hip, include/spgpu/tuning.h).

The reference runs an ordered matrix thousands of times through spgpu?hellspmv with rIdx (hellPerf.cpp:333-378); here the
first call analyses the matrix behind itself and later calls on the same arrays run one launch with the deep sub-groups in
workgroups of their own.  What these tests pin: a call WITH a plan gives the bits of the call WITHOUT one and of the oracle
(tests/oracle_api.py: the queue kernel's order, chunks of a sub-group added in chunk order); so does a call with a plan that
has gone STALE (another matrix copied over the same device arrays); the library notices and rebuilds."""
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
                hack_offsets=h["hack_offsets"].cpu().numpy(), hack_size=hack, row_lengths=h["rS"][:n].cpu().numpy(), base=0)


def _call(gpu, letter, h, n, dz, dy, dx, alpha, beta, hack=32, arrays=None):
    from spgpu_amd import capi
    a = arrays or h
    capi.hellspmv[letter](gpu, _dp(dz), _dp(dy) if beta != 0 else None, capi.scalar(letter, alpha), _dp(a["cM"]), _dp(a["rP"]), hack,
                          _dp(a["hack_offsets"]), _dp(a["rS"]), _dp(a["rIdx"]), 12, n, _dp(dx), capi.scalar(letter, beta), 0)


def _until_planned(gpu, run, most=6):
    """Runs `run` (one SpMV + synchronise) until a call has used a plan; returns the number of calls made."""
    from spgpu_amd import capi
    before = capi.plan_counts(gpu)[0]
    for call in range(1, most + 1):
        run()
        if capi.plan_counts(gpu)[0] > before:
            return call
    raise AssertionError("no call used a plan")


@pytest.mark.parametrize("letter", ["S", "D", "C", "Z"])
@pytest.mark.parametrize("window,long_rows,aligned,hack", [(2048, 60, True, 32), (512, 40, False, 32), (0, 0, False, 64), (256, 100, False, 96)])
def test_planned_call_equals_unplanned_call_and_oracle(gpu, letter, window, long_rows, aligned, hack):
    """Every type, the aligned order (2 048-row staged shape once the probe has answered), drifting windows (1 024-row shape), one
    global sort, hack sizes 32 / 64 / 96, beta != 0: the first call (no plan), the calls while the analysis lands and the
    planned calls all give the oracle's bits."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 9 * 2048 + 77
    h = _matrix(gpu, n, letter, window, long_rows, aligned, hack=hack, longest=900, near=500)
    sub, r_idx = _host(h, letter, n, hack), h["rIdx"].cpu().numpy()
    x, y = synth.values_for(letter, 31, n), synth.values_for(letter, 32, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    want = O.spmv_tail(sub, x, y, -0.5, 2.0, r_idx=r_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    uses0 = capi.plan_counts(gpu)[0]
    seen = []
    for call in range(7):
        dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
        _call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)
        torch.cuda.synchronize()
        seen.append(capi.plan_counts(gpu)[0] - uses0)
        assert dz.cpu().numpy().tobytes() == want.tobytes(), (call, seen)
    assert seen[-1] >= 2, seen   # (the first call may already find a plan: the previous case's, stale, at the same addresses)


@pytest.mark.parametrize("letter", ["D", "S"])
def test_stale_plan_same_bits_then_rebuilt(gpu, letter):
    """Matrix A is planned; then matrix B -- same number of rows, other row lengths, other order -- is copied over the SAME device
    arrays.  The next call still has A's plan: sub-groups the plan lists are shallow now, deep ones are missing from it.  The
    result is B's product in the oracle's bits all the same, the kernels report the contradiction, and the call after that
    starts a new analysis; once it has landed B runs planned, same bits again."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 6 * 2048 + 300
    a = _matrix(gpu, n, letter, 2048, 60, True, longest=700, seed=3)
    b = _matrix(gpu, n, letter, 2048, 60, True, longest=1500, mean=40.0, seed=11, near=800)
    deep_of = lambda m: set(np.flatnonzero(np.maximum.reduceat(m["rS"][:n].cpu().numpy(), np.arange(0, n, 32)) > O.DEEP_CAP).tolist())
    assert deep_of(a) != deep_of(b)   # sub-groups the plan lists are shallow now, or deep ones are not in it
    slots = max(a["slots"], b["slots"])
    fixed = dict(cM=torch.zeros(slots, dtype=a["cM"].dtype, device="cuda"), rP=torch.zeros(slots, dtype=torch.int32, device="cuda"),
                 hack_offsets=torch.zeros_like(a["hack_offsets"]), rS=torch.zeros(n, dtype=torch.int32, device="cuda"),
                 rIdx=torch.zeros(n, dtype=torch.int32, device="cuda"))

    def load(m):
        fixed["cM"][:m["slots"]] = m["cM"][:m["slots"]]
        fixed["rP"][:m["slots"]] = m["rP"][:m["slots"]]
        fixed["hack_offsets"].copy_(m["hack_offsets"])
        fixed["rS"].copy_(m["rS"][:n])
        fixed["rIdx"].copy_(m["rIdx"])
        torch.cuda.synchronize()

    x = synth.values_for(letter, 41, n)
    dx = formats.to_device(x)
    shape = O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP)
    want = {name: O.spmv_tail(_host(m, letter, n), x, None, 1.5, 0.0, r_idx=m["rIdx"].cpu().numpy(), **shape) for name, m in (("a", a), ("b", b))}
    assert want["a"].tobytes() != want["b"].tobytes()
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")

    def run():
        dz.fill_(float("nan"))
        _call(gpu, letter, None, n, dz, None, dx, 1.5, 0.0, arrays=fixed)
        torch.cuda.synchronize()

    load(a)
    _until_planned(gpu, run)
    assert dz.cpu().numpy().tobytes() == want["a"].tobytes()
    uses, builds, stales = capi.plan_counts(gpu)
    load(b)                                   # B lives where A lived
    run()                                     # A's plan, B's matrix
    assert capi.plan_counts(gpu)[0] == uses + 1, "the stale plan was expected to be used once more"
    assert dz.cpu().numpy().tobytes() == want["b"].tobytes()
    run()                                     # the library has seen the kernels' report: no plan, a new analysis
    assert dz.cpu().numpy().tobytes() == want["b"].tobytes()
    after = capi.plan_counts(gpu)
    assert after[2] == stales + 1 and after[1] == builds + 1, (after, (uses, builds, stales))
    _until_planned(gpu, run)
    assert dz.cpu().numpy().tobytes() == want["b"].tobytes()
    assert capi.plan_counts(gpu)[2] == stales + 1   # the new plan fits


@pytest.mark.parametrize("keep,split,per_block,spread", [(-1, -1, 4, -1), (0, 48, 1, 0), (64, 0, 8, 100), (32, 150, 3, 37)])
def test_planned_knobs_keep_the_oracle_order(gpu, tuning, keep, split, per_block, spread):
    """SPGPU_DEEP_KEEP / SPGPU_RAGGED_SPLIT change the chunks of a sub-group (and so the bits) for the list path and the planned path
    alike -- the oracle restates them; SPGPU_PLAN_DEEP_PER_BLOCK / SPGPU_PLAN_DEEP_SPREAD only move work around."""
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_DEEP_KEEP=keep, SPGPU_RAGGED_SPLIT=split, SPGPU_PLAN_DEEP_PER_BLOCK=per_block, SPGPU_PLAN_DEEP_SPREAD=spread)
    n = 5 * 2048 + 9
    h = _matrix(gpu, n, "D", 2048, 100, True, longest=1200, seed=5, near=700)
    x = synth.values_for("D", 51, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(h, "D", n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(),
                       **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP, split=split, deep_keep=keep))
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")

    def run():
        dz.fill_(float("nan"))
        _call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == want.tobytes()

    _until_planned(gpu, run)
    run()


def test_planned_very_deep_rows_take_rounds(gpu):
    """A sub-group deeper than the chunk sums one workgroup can park at once (~170 chunks of 64 columns) is worked off in rounds;
    here three rows of 14 000 .. 20 000 entries among short ones."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 24000
    lengths = np.full(n, 7, np.int32)
    lengths[[5, 4000, 19999]] = [20000, 14000, 17001]
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, "band", 300, "D", seed=13)
    h = formats.coo_to_ordered_hell_device(gpu, n, rows_t, cols_t, vals_t, "D", 32, 2048, 256, aligned=True)
    x = synth.values_for("D", 61, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(h, "D", n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")

    def run():
        dz.fill_(float("nan"))
        _call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == want.tobytes()

    _until_planned(gpu, run)


def test_two_streams_one_planned_matrix(gpu):
    """Two SpMVs of one planned matrix in flight on two streams of ONE handle (the plan is read-only, there is no list): 10 rounds,
    both results the oracle's bits."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 40 * 2048
    h = _matrix(gpu, n, "D", 2048, 60, True, longest=900, seed=17, near=500)
    x1, x2 = synth.values_for("D", 71, n), synth.values_for("D", 72, n)
    d1, d2 = formats.to_device(x1), formats.to_device(x2)
    shape = O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP)
    sub, r_idx = _host(h, "D", n), h["rIdx"].cpu().numpy()
    w1, w2 = O.spmv_tail(sub, x1, None, 1.0, 0.0, r_idx=r_idx, **shape), O.spmv_tail(sub, x2, None, 1.0, 0.0, r_idx=r_idx, **shape)
    z1, z2 = torch.zeros(n, dtype=d1.dtype, device="cuda"), torch.zeros(n, dtype=d1.dtype, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        def warm():
            _call(gpu, "D", h, n, z1, None, d1, 1.0, 0.0)
            torch.cuda.synchronize()
        capi.spgpuSetStream(gpu, C.c_void_p(s1.cuda_stream))
        _until_planned(gpu, warm)
        for _ in range(10):
            z1.fill_(float("nan"))
            z2.fill_(float("nan"))
            torch.cuda.synchronize()
            capi.spgpuSetStream(gpu, C.c_void_p(s1.cuda_stream))
            _call(gpu, "D", h, n, z1, None, d1, 1.0, 0.0)
            capi.spgpuSetStream(gpu, C.c_void_p(s2.cuda_stream))
            _call(gpu, "D", h, n, z2, None, d2, 1.0, 0.0)
            torch.cuda.synchronize()
            assert z1.cpu().numpy().tobytes() == w1.tobytes()
            assert z2.cpu().numpy().tobytes() == w2.tobytes()
    finally:
        capi.spgpuSetStream(gpu, None)


def test_no_plan_when_switched_off(gpu, tuning):
    import torch
    from spgpu_amd import capi, formats, synth
    tuning(SPGPU_PLAN=0)
    n = 3 * 2048
    h = _matrix(gpu, n, "D", 2048, 60, True, seed=19)
    dx = formats.to_device(synth.values_for("D", 81, n))
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")
    before = capi.plan_counts(gpu)
    for _ in range(4):
        _call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
        torch.cuda.synchronize()
    assert capi.plan_counts(gpu) == before


def test_streams_come_and_go_lists_change_hands(gpu):
    """A program that creates and destroys streams as it goes: the handle keeps 8 deep lists, the ninth stream takes over the list
    of the least recently used stream whose work has finished instead of running the stateless kernel for good (its order of
    additions is another one).  Every call is the list path's bits (SPGPU_PLAN=0 would not matter: first calls have no plan)."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 4 * 2048 + 5
    h = _matrix(gpu, n, "D", 512, 40, False, seed=23)
    x = synth.values_for("D", 91, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(h, "D", n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
    recycled0, fallbacks0 = capi.spgpuDeepListsRecycled(gpu), capi.spgpuDeepListFallbacks(gpu)
    try:
        for _ in range(20):
            s = torch.cuda.Stream()
            capi.spgpuSetStream(gpu, C.c_void_p(s.cuda_stream))
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            torch.cuda.synchronize()   # the fill runs on torch's stream, the SpMV on the new (non-blocking) one: without this the fill may land on top of z
            before = (capi.plan_counts(gpu), capi.spgpuDeepListFallbacks(gpu), capi.spgpuDeepListsRecycled(gpu))
            _call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
            torch.cuda.synchronize()
            got = dz.cpu().numpy()
            if got.tobytes() != want.tobytes():      # say what a failure would need to be understood
                rows = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
                r_idx = h["rIdx"].cpu().numpy()
                where = np.empty(n, np.int64)
                where[r_idx] = np.arange(n)
                lengths = h["rS"][:n].cpu().numpy()
                raise AssertionError((_, rows[:8], where[rows][:8], lengths[where[rows]][:8], got[rows][:4], want[rows][:4], before,
                                      capi.plan_counts(gpu), capi.spgpuDeepListFallbacks(gpu), capi.spgpuDeepListsRecycled(gpu)))
            del s
    finally:
        capi.spgpuSetStream(gpu, None)
    assert capi.spgpuDeepListFallbacks(gpu) == fallbacks0
    assert capi.spgpuDeepListsRecycled(gpu) > recycled0


def test_no_error_left_behind_for_the_caller(gpu):
    """A caller in the reference's style asks cudaGetLastError after its launches (hellPerf.cpp:385-390).  While a matrix' analysis
    is still in flight the library polls its event; "not ready" is an answer, not an error, and must not be what the caller finds."""
    import torch
    from spgpu_amd import capi, formats, synth
    hip = C.CDLL("libamdhip64.so")
    n = 200 * 2048
    h = _matrix(gpu, n, "D", 2048, 60, True, seed=29)
    dx = formats.to_device(synth.values_for("D", 95, n))
    dz = torch.zeros(n, dtype=dx.dtype, device="cuda")
    torch.cuda.synchronize()
    assert hip.hipGetLastError() == 0
    for _ in range(6):                       # back to back: the later calls find the analysis (queued behind the first) not finished yet
        _call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
    assert hip.hipGetLastError() == 0
    torch.cuda.synchronize()
    assert hip.hipGetLastError() == 0


@pytest.mark.parametrize("letter", ["D", "S", "Z"])
def test_prepare_makes_the_first_call_a_planned_one(gpu, letter):
    """spgpuHellSpmvPrepare (include/spgpu/tuning.h): probe and analysis now, waited for -- the very first SpMV on the arrays runs
    from the plan (and gives the oracle's bits); without rIdx there is nothing to prepare (SPGPU_UNSUPPORTED, not an error)."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 7 * 2048 + 31
    h = _matrix(gpu, n, letter, 2048, 60, True, seed=31, longest=700, near=400)
    x = synth.values_for(letter, 97, n)
    dx = formats.to_device(x)
    want = O.spmv_tail(_host(h, letter, n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    torch.cuda.synchronize()
    code = capi.TYPE_CODE[letter]
    assert capi.spgpuHellSpmvPrepare(gpu, code, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), _dp(h["rIdx"]), n, 0) == capi.SPGPU_SUCCESS
    uses = capi.plan_counts(gpu)[0]
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    _call(gpu, letter, h, n, dz, None, dx, 1.0, 0.0)
    torch.cuda.synchronize()
    assert capi.plan_counts(gpu)[0] == uses + 1
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    assert capi.spgpuHellSpmvPrepare(gpu, code, _dp(h["cM"]), _dp(h["rP"]), 32, _dp(h["hack_offsets"]), _dp(h["rS"]), None, n, 0) == capi.SPGPU_UNSUPPORTED


def test_prepare_ell_with_a_row_order(gpu):
    """The ELL flavour: any rIdx selects the ordered path; prepared, the first spgpuDellspmv runs from the plan, oracle bits."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 3000
    lengths = np.minimum(np.random.default_rng(5).zipf(1.5, size=n), 300)
    _, _, r, c, v = synth.random_rows_coo(n, n, lengths, seed=6, letter="D")
    ell = formats.coo_to_ell(n, r, c, v)
    perm = np.random.default_rng(7).permutation(n).astype(np.int32)
    dev = formats.DeviceEll(ell, r_idx=perm)
    x = synth.values_for("D", 99, n)
    dx = formats.to_device(x)
    torch.cuda.synchronize()
    assert capi.spgpuEllSpmvPrepare(gpu, capi.TYPE_CODE["D"], _dp(dev.cM), _dp(dev.rP), dev.pitch, dev.pitch, _dp(dev.rS), _dp(dev.rIdx), dev.max_row,
                                    n, 0) == capi.SPGPU_SUCCESS
    uses = capi.plan_counts(gpu)[0]
    dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
    dev.spmv(gpu, dz, None, 1.0, dx, 0.0)
    torch.cuda.synchronize()
    assert capi.plan_counts(gpu)[0] == uses + 1
    assert dz.cpu().numpy().tobytes() == O.default_spmv(ell, x, None, 1.0, 0.0, r_idx=perm).tobytes()
