"""GPU: fused steps of a Krylov iteration (include/spgpu/device_scalars.h, csrc/fused_solver.hip; SURVEY section 8
row f4).  The calls are the first stage of the dot with its second operand produced on the fly: z bit for bit the
oracle's one-phase HELL SpMV, *result bit for bit what spgpu?dot returns for the stored vectors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _ragged_hell(letter, n, hack, base, seed):
    from spgpu_amd import formats, synth
    lengths = np.minimum(synth.power_law_lengths(n, mean=6.0, max_len=60, seed=seed), n)
    lengths[3::7] = 0                                                # empty rows
    _, _, r, c, v = synth.random_rows_coo(n, n, lengths, seed=seed, letter=letter, base=base)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v, coo_base=base, ell_base=base), hack)
    return hell


@pytest.mark.parametrize("letter", "SD")
@pytest.mark.parametrize("n,hack,base,offset", [(1, 32, 0, 0), (33, 32, 1, 0), (4097, 32, 0, 0), (5000, 30, 1, 0),
                                                (70_001, 32, 0, 1), (2_100_000, 32, 0, 0)])
@pytest.mark.parametrize("with_beta", [False, True])
def test_hellspmv_dot_device(gpu, letter, n, hack, base, offset, with_beta):
    """Ragged rows incl. empty ones, hack sizes that are / are not a multiple of the pack, rows not a multiple of the
    pack, unaligned vectors (element mapping of the dot), more rows than one pass of the grid (2.1 M > 1024 x 2048)."""
    import torch
    import oracle_api as O
    from spgpu_amd import capi, formats, synth
    if n > 100_000:
        r_n, _, r, c, v = synth.laplacian_2d_5pt(1450, dtype=O.NP_DTYPE[letter])   # 2 102 500 rows
        hell = formats.ell_to_hell(formats.coo_to_ell(r_n, r, c, v), hack)
        n = r_n
    else:
        hell = _ragged_hell(letter, n, hack, base, seed=n)
    mat = formats.DeviceHell(hell)
    x = synth.values_for(letter, 3, n)
    w = synth.values_for(letter, 4, n + offset)[offset:]
    y = synth.values_for(letter, 5, n) if with_beta else None
    alpha, beta = (1.25, -0.5) if with_beta else (1.0, 0.0)
    dx, dy = formats.to_device(x), formats.to_device(y)
    dw = formats.to_device(synth.values_for(letter, 4, n + offset))[offset:]
    dz = torch.full((n + offset,), 7, dtype=dx.dtype, device="cuda:0")[offset:]
    out = torch.zeros(2, dtype=dx.dtype, device="cuda:0")
    capi.hellspmv_dot_device[letter](gpu, _p(out), _p(dw), _p(dz), _p(dy), capi.scalar(letter, alpha), _p(mat.cM), _p(mat.rP),
                                     hack, _p(mat.hack_offsets), _p(mat.rS), n, _p(dx), capi.scalar(letter, beta), base)
    torch.cuda.synchronize()
    want_z = O.hell_spmv(hell, x, y, alpha, beta, phases=1)
    assert dz.cpu().numpy().tobytes() == want_z.tobytes()
    want_dot = capi.dot[letter](gpu, n, _p(dw), _p(dz))
    got = out.cpu().numpy()[0]
    assert np.asarray(got).tobytes() == np.asarray(want_dot, dtype=got.dtype).tobytes()
    # and the value is a dot product: against longdouble within the accumulated rounding of n terms
    exact = np.sum(w.astype(np.longdouble) * want_z.astype(np.longdouble))
    scale = np.sum(np.abs(w.astype(np.longdouble) * want_z.astype(np.longdouble))) + 1e-300
    assert abs(np.longdouble(got) - exact) <= n * np.finfo(O.NP_DTYPE[letter]).eps * scale

    # w == NULL: the dot with x itself (p . Ap)
    capi.hellspmv_dot_device[letter](gpu, _p(out[1:]), None, _p(dz), _p(dy), capi.scalar(letter, alpha), _p(mat.cM), _p(mat.rP),
                                     hack, _p(mat.hack_offsets), _p(mat.rS), n, _p(dx), capi.scalar(letter, beta), base)
    want_dot = capi.dot[letter](gpu, n, _p(dx), _p(dz))
    assert np.asarray(out.cpu().numpy()[1]).tobytes() == np.asarray(want_dot, dtype=got.dtype).tobytes()


@pytest.mark.parametrize("letter", "SD")
def test_hellspmv_dot_device_equals_default_spmv_on_a_stencil(gpu, letter):
    """On rows of even length the default kernel adds in the same order: z equals spgpu?hellspmv's bit for bit."""
    import torch
    from spgpu_amd import capi, formats, synth
    import oracle_api as O
    n, _, r, c, v = synth.laplacian_2d_5pt(200, dtype=O.NP_DTYPE[letter])
    mat = formats.DeviceHell(formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32))
    dx = formats.to_device(synth.values_for(letter, 3, n))
    z1, z2 = torch.empty_like(dx), torch.empty_like(dx)
    out = torch.zeros(1, dtype=dx.dtype, device="cuda:0")
    mat.spmv(gpu, z1, None, 1.0, dx, 0.0)
    capi.hellspmv_dot_device[letter](gpu, _p(out), None, _p(z2), None, capi.scalar(letter, 1.0), _p(mat.cM), _p(mat.rP), 32,
                                     _p(mat.hack_offsets), _p(mat.rS), n, _p(dx), capi.scalar(letter, 0.0), 0)
    torch.cuda.synchronize()
    if letter == "D":       # fp32's default kernel splits a row over 8 phases: same value within rounding only
        assert torch.equal(z1, z2)
    else:
        assert torch.allclose(z1, z2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("letter", "SD")
@pytest.mark.parametrize("n,offset", [(1, 0), (1001, 0), (4097, 1), (1 << 20, 0), (2_500_003, 0)])
def test_axpby_pair_dot_device(gpu, letter, n, offset):
    """z1, z2 and *result bit for bit what the three separate calls leave; in place (the CG use) and out of place."""
    import torch
    from spgpu_amd import capi, formats, synth
    vec = lambda seed: formats.to_device(synth.values_for(letter, seed, n + offset))[offset:]
    x1, y1, x2, y2 = vec(1), vec(2), vec(3), vec(4)
    num, den = synth.values_for(letter, 5, 2) + 2
    scal = formats.to_device(np.array([num, den], dtype=synth.values_for(letter, 5, 1).dtype))
    want1, want2 = torch.empty_like(x1), torch.empty_like(x1)
    want = torch.zeros(1, dtype=x1.dtype, device="cuda:0")
    capi.axpby_quot_device[letter](gpu, _p(want1), n, None, None, _p(y1), _p(scal[0:]), _p(scal[1:]), 0, _p(x1))
    capi.axpby_quot_device[letter](gpu, _p(want2), n, None, None, _p(y2), _p(scal[0:]), _p(scal[1:]), 1, _p(x2))
    capi.dot_device[letter](gpu, _p(want), n, _p(want2), _p(want2))
    got1, got2 = torch.empty_like(x1), torch.empty_like(x1)
    got = torch.zeros(2, dtype=x1.dtype, device="cuda:0")
    capi.axpby_pair_dot_device[letter](gpu, _p(got), n, _p(got1), _p(y1), _p(x1), _p(got2), _p(y2), _p(x2), _p(scal[0:]),
                                       _p(scal[1:]))
    torch.cuda.synchronize()
    assert torch.equal(got1, want1) and torch.equal(got2, want2)
    assert got.cpu().numpy()[0].tobytes() == want.cpu().numpy()[0].tobytes()
    # in place: z1 = y1, z2 = y2
    a, b = y1.clone(), y2.clone()
    capi.axpby_pair_dot_device[letter](gpu, _p(got[1:]), n, _p(a), _p(a), _p(x1), _p(b), _p(b), _p(x2), _p(scal[0:]),
                                       _p(scal[1:]))
    torch.cuda.synchronize()
    assert torch.equal(a, want1) and torch.equal(b, want2)
    assert got.cpu().numpy()[1].tobytes() == want.cpu().numpy()[0].tobytes()


def test_fused_calls_on_empty_input(gpu):
    import torch
    from spgpu_amd import capi
    out = torch.full((2,), 5.0, dtype=torch.float64, device="cuda:0")
    capi.hellspmv_dot_device["D"](gpu, _p(out), None, None, None, 1.0, None, None, 32, None, None, 0, None, 0.0, 0)
    capi.axpby_pair_dot_device["D"](gpu, _p(out[1:]), 0, None, None, None, None, None, None, None, None)
    torch.cuda.synchronize()
    assert out.cpu().tolist() == [0.0, 0.0]


def test_fused_cg_iteration_equals_eager(gpu):
    """CG on a 2-D Laplacian: the iteration as 3 calls (SpMV+dot, two updates+dot, direction) replayed from one graph
    per parity, against the eager loop with host scalars -- identical iterates and |r|^2."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, _, r, c, v = synth.laplacian_2d_5pt(96)
    mat = formats.DeviceHell(formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32))
    b = formats.to_device(synth.values_for("D", 11, n))
    iters = 20

    x, rvec, p, ap = torch.zeros_like(b), b.clone(), b.clone(), torch.empty_like(b)
    rr = capi.dot["D"](gpu, n, _p(rvec), _p(rvec))
    for _ in range(iters):
        mat.spmv(gpu, ap, None, 1.0, p, 0.0)
        alpha = rr / capi.dot["D"](gpu, n, _p(p), _p(ap))
        capi.axpby["D"](gpu, _p(x), n, 1.0, _p(x), alpha, _p(p))
        capi.axpby["D"](gpu, _p(rvec), n, 1.0, _p(rvec), -alpha, _p(ap))
        rr_new = capi.dot["D"](gpu, n, _p(rvec), _p(rvec))
        capi.axpby["D"](gpu, _p(p), n, rr_new / rr, _p(p), 1.0, _p(rvec))
        rr = rr_new
    torch.cuda.synchronize()
    x_eager, rr_eager = x.clone(), rr

    x, rvec, p, ap = torch.zeros_like(b), b.clone(), b.clone(), torch.empty_like(b)
    s = torch.zeros(3, dtype=torch.float64, device="cuda:0")      # |r|^2 (two cells, alternating), p.Ap
    side = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()   # the vectors above were written on torch's stream; `side` does not wait for it by itself

    def iteration(rr_old, rr_new):
        capi.hellspmv_dot_device["D"](gpu, _p(s[2:]), None, _p(ap), None, 1.0, _p(mat.cM), _p(mat.rP), 32,
                                      _p(mat.hack_offsets), _p(mat.rS), n, _p(p), 0.0, 0)
        capi.axpby_pair_dot_device["D"](gpu, _p(rr_new), n, _p(x), _p(x), _p(p), _p(rvec), _p(rvec), _p(ap), _p(rr_old),
                                        _p(s[2:]))
        capi.axpby_quot_device["D"](gpu, _p(p), n, _p(rr_new), _p(rr_old), _p(p), None, None, 0, _p(rvec))

    try:
        with torch.cuda.stream(side):
            iteration(s[0:], s[1:])          # warm-up outside the capture (module load), then start over
        side.synchronize()
        x.zero_(), rvec.copy_(b), p.copy_(b)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            capi.dot_device["D"](gpu, _p(s), n, _p(rvec), _p(rvec))
        side.synchronize()
        graphs = []
        for parity in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                iteration(s[parity:], s[1 - parity:])
            graphs.append(g)
        for i in range(iters):
            graphs[i & 1].replay()
        torch.cuda.synchronize()
    finally:
        capi.spgpuSetStream(gpu, None)
    assert torch.equal(x, x_eager)
    assert s.cpu().numpy()[iters & 1] == rr_eager
