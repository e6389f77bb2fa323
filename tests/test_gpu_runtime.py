"""GPU: runtime properties of the C ABI that a solver relies on: calls are capturable into a HIP graph
(no allocation or synchronisation inside the launch path), independent handles/streams do not interfere,
and unusual hack sizes work."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _problem(seed=1, rows=3000, hs=32):
    from spgpu_amd import formats, synth
    lengths = synth.power_law_lengths(rows, 9.0, 70, seed=seed)
    n, m, r, c, v = synth.random_rows_coo(rows, rows, lengths, seed=seed + 1, letter="D")
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), hs)
    return hell, synth.values_for("D", seed + 2, m), synth.values_for("D", seed + 3, n)


def test_spmv_and_axpby_capture_into_a_hip_graph(gpu):
    """One CG-like step (SpMV, axpby, axpby) captured once and replayed: results equal the eager ones."""
    import torch
    from spgpu_amd import capi, formats
    hell, x, y = _problem()
    mat = formats.DeviceHell(hell)
    dx, dy = formats.to_device(x), formats.to_device(y)
    dz, dw = torch.empty_like(dy), torch.empty_like(dy)
    n = hell["rows"]

    def step():
        mat.spmv(gpu, dz, dy, 1.5, dx, -0.5)
        capi.axpby["D"](gpu, _p(dw), n, 2.0, _p(dz), 0.25, _p(dy))
        capi.axpby["D"](gpu, _p(dw), n, 1.0, _p(dw), -1.0, _p(dz))

    side = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()   # the buffers were set up on torch's stream; `side` does not wait for it by itself
    with torch.cuda.stream(side):
        step()
    side.synchronize()
    eager = dw.clone()
    dw.zero_(); dz.zero_()
    torch.cuda.synchronize()

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        # torch captures on `side`; the handle launches on the same stream, so the launches are recorded
        step()
    dw.zero_(); dz.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(dw, eager)
    dy.mul_(2.0)                       # the graph reads the buffers, not captured values
    graph.replay()
    torch.cuda.synchronize()
    want = O.default_spmv(hell, x, 2.0 * y, 1.5, -0.5)
    assert dz.cpu().numpy().tobytes() == want.tobytes()
    capi.spgpuSetStream(gpu, None)


def test_two_handles_two_streams_interleaved(gpu):
    """Independent handles with their own streams and reduction scratch give the single-stream answers."""
    import torch
    from spgpu_amd import capi, formats
    h2 = capi.create_handle(0)
    probs = [_problem(seed=s, rows=4000 + 500 * s) for s in (1, 2)]
    mats = [formats.DeviceHell(p[0]) for p in probs]
    dxs = [formats.to_device(p[1]) for p in probs]
    dzs = [torch.empty(p[0]["rows"], dtype=torch.float64, device="cuda:0") for p in probs]
    dots = [None, None]
    for rep in range(20):
        for i, handle in enumerate((gpu, h2)):
            mats[i].spmv(handle, dzs[i], None, 1.0, dxs[i], 0.0)
            dots[i] = capi.dot["D"](handle, probs[i][0]["rows"], _p(dzs[i]), _p(dzs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        want = O.default_spmv(probs[i][0], probs[i][1], None, 1.0, 0.0)
        assert dzs[i].cpu().numpy().tobytes() == want.tobytes()
        assert abs(dots[i] - float(np.dot(want, want))) <= 1e-11 * float(np.dot(want, want))
    capi.spgpuDestroy(h2)


@pytest.mark.parametrize("hs", [96, 160, 48, 8])
def test_unusual_hack_sizes(gpu, hs):
    """hackSize a multiple of 32 beyond 32/64, and values the header does not promise (48, 8): every lane
    derives its hack from its own row, so they work too (strips never straddle a hack when hs % RPL == 0)."""
    from spgpu_amd import formats
    import torch
    hell, x, y = _problem(seed=7, rows=1000, hs=hs)
    dz = torch.empty(hell["rows"], dtype=torch.float64, device="cuda:0")
    dx, dy = formats.to_device(x), formats.to_device(y)
    formats.DeviceHell(hell).spmv(gpu, dz, dy, 1.0, dx, 0.5)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == O.default_spmv(hell, x, y, 1.0, 0.5).tobytes()


@pytest.mark.parametrize("letter", "SDC")
def test_strip_and_gather_forms_agree_and_the_choice_is_learnt(gpu, letter, tuning):
    """ELL/HELL SpMV has a strip-load form (rows naming consecutive columns) and a gather form; by default the library
    learns per matrix, from its own kernel's feedback, which one to launch.  Whatever it launches -- first call, later
    calls, either form forced -- the result is the same bits (same order of additions), on a band matrix and on a
    scattered one."""
    import torch
    from spgpu_amd import capi, formats, synth
    rows = 40_000
    for kind in ("banded", "scattered"):
        if kind == "banded":
            n, m, r, c, v = synth.banded_coo(rows, 6, letter, seed=3)
        else:
            n, m, r, c, v = synth.random_rows_coo(rows, rows, synth.power_law_lengths(rows, 9.0, 60, seed=4), seed=5, letter=letter)
        hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
        x, y = synth.values_for(letter, 6, m), synth.values_for(letter, 7, n)
        want = O.default_spmv(hell, x, y, 1.5, -0.5).tobytes()
        mat = formats.DeviceHell(hell)
        dx, dy = formats.to_device(x), formats.to_device(y)
        dz = torch.empty_like(dy)
        results = []
        for _ in range(4):                       # unknown -> feedback read on the following calls
            mat.spmv(gpu, dz, dy, 1.5, dx, -0.5)
            torch.cuda.synchronize()
            results.append(dz.cpu().numpy().tobytes())
        for forced in (0, 1):
            tuning(SPGPU_X_STRIPS=forced)
            mat.spmv(gpu, dz, dy, 1.5, dx, -0.5)
            torch.cuda.synchronize()
            results.append(dz.cpu().numpy().tobytes())
        tuning(SPGPU_X_STRIPS=-1)
        assert all(res == want for res in results), kind


@pytest.mark.parametrize("letter", "SDC")
def test_strip_form_on_bands_with_holes_ell_and_hell(gpu, letter, tuning):
    """The strip form's hand-over points: a band matrix in which some rows miss entries (so only some stages of some
    wavefronts qualify), index base 1, ELL and HELL with hack sizes 32 and 64, beta != 0 --
    all bit-exact against the oracle with the strip kernel forced."""
    import torch
    from spgpu_amd import formats, synth
    tuning(SPGPU_X_STRIPS=1)
    rng = np.random.default_rng(ord(letter))
    rows, half = 20_000, 5
    i = np.arange(rows)[:, None]
    c = i + np.arange(-half, half + 1)[None, :]
    keep = (c >= 0) & (c < rows)
    holes = rng.random(c.shape) < 0.002            # a few missing entries: the rows behind them shift left in the slab
    holes[rows // 3: rows // 3 + 300] = False      # ... and a stretch without any
    keep &= ~holes
    r = np.broadcast_to(i, c.shape)[keep]
    cols = c[keep]
    v = synth.values_for(letter, 21, r.size)
    x, y = synth.values_for(letter, 22, rows), synth.values_for(letter, 23, rows)
    ell = formats.coo_to_ell(rows, r, cols, v, ell_base=1)
    dx, dy = formats.to_device(x), formats.to_device(y)
    dz = torch.empty_like(dy)
    for hs in (32, 64):
        hell = formats.ell_to_hell(ell, hs)
        formats.DeviceHell(hell).spmv(gpu, dz, dy, 0.75, dx, 2.0)
        torch.cuda.synchronize()
        assert dz.cpu().numpy().tobytes() == O.default_spmv(hell, x, y, 0.75, 2.0).tobytes()
    formats.DeviceEll(ell).spmv(gpu, dz, dy, 0.75, dx, 2.0)
    torch.cuda.synchronize()
    assert dz.cpu().numpy().tobytes() == O.default_spmv(ell, x, y, 0.75, 2.0).tobytes()
