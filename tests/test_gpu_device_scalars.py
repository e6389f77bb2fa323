"""GPU: Level-1 calls with device-resident scalars (include/spgpu/device_scalars.h): the same
bits as the host-scalar calls, no host synchronisation (so a whole CG iteration replays from one captured graph)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
EPS = {"S": 2.0 ** -24, "D": 2.0 ** -53}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("letter", "SD")
@pytest.mark.parametrize("n,offset", [(1, 0), (1000, 0), (4097, 1), (1 << 20, 0), (3_000_001, 0)])
def test_dot_device_equals_dot(gpu, letter, n, offset):
    """Aligned and unaligned vectors, one block to the block cap: *result is bit for bit what spgpu?dot returns."""
    import torch
    from spgpu_amd import capi, formats, synth
    a = formats.to_device(synth.values_for(letter, 1, n + offset))[offset:]
    b = formats.to_device(synth.values_for(letter, 2, n + offset))[offset:]
    out = torch.zeros(1, dtype=a.dtype, device="cuda:0")
    capi.dot_device[letter](gpu, _p(out), n, _p(a), _p(b))
    want = capi.dot[letter](gpu, n, _p(a), _p(b))          # synchronises the handle's stream
    got = out.cpu().numpy()[0]
    assert np.asarray(got).tobytes() == np.asarray(want, dtype=got.dtype).tobytes()
    # and against the ORACLE (ddot.cu:37-150 restated, ascending order): another order of additions, so within the rounding bound
    ah, bh = a.cpu().numpy(), b.cpu().numpy()
    assert abs(float(got) - float(O.dot(letter, ah, bh))) <= 2 * n * EPS[letter] * float(np.sum(np.abs(ah.astype(np.float64) * bh)))
    capi.dot_device[letter](gpu, _p(out), 0, _p(a), _p(b))  # empty vectors: 0
    torch.cuda.synchronize()
    assert out.cpu().numpy()[0] == 0


@pytest.mark.parametrize("letter", "SD")
@pytest.mark.parametrize("n,offset", [(1, 0), (1000, 0), (4097, 1), (1 << 20, 0), (3_000_001, 0)])
def test_nrm2_device_equals_nrm2(gpu, letter, n, offset):
    import torch
    from spgpu_amd import capi, formats, synth
    a = formats.to_device(synth.values_for(letter, 1, n + offset))[offset:]
    out = torch.zeros(1, dtype=a.dtype, device="cuda:0")
    capi.nrm2_device[letter](gpu, _p(out), n, _p(a))
    want = capi.nrm2[letter](gpu, n, _p(a))
    got = out.cpu().numpy()[0]
    assert np.asarray(got).tobytes() == np.asarray(want, dtype=got.dtype).tobytes()
    oracle = float(O.nrm2(letter, a.cpu().numpy()))        # dnrm2.cu:52-53,146 restated
    assert abs(float(got) - oracle) <= 2 * n * EPS[letter] * oracle
    capi.nrm2_device[letter](gpu, _p(out), 0, _p(a))
    torch.cuda.synchronize()
    assert out.cpu().numpy()[0] == 0


@pytest.mark.parametrize("letter", "SD")
def test_axpby_and_div_device_equal_host_scalar_calls(gpu, letter):
    import torch
    from spgpu_amd import capi, formats, synth
    n = 100_003
    x = formats.to_device(synth.values_for(letter, 3, n))
    y = formats.to_device(synth.values_for(letter, 4, n))
    dt = x.dtype
    num, den = synth.values_for(letter, 5, 2) + 2
    scal = formats.to_device(np.array([num, den, 0, 0, 0], dtype=synth.values_for(letter, 5, 1).dtype))
    capi.div_device[letter](gpu, _p(scal[2:]), _p(scal[0:]), _p(scal[1:]), 0)   # alpha = num / den
    capi.div_device[letter](gpu, _p(scal[3:]), _p(scal[0:]), _p(scal[1:]), 1)   # beta = -num / den
    torch.cuda.synchronize()
    alpha, beta = scal.cpu().numpy()[2], scal.cpu().numpy()[3]
    assert alpha == num / den and beta == -(num / den)

    z_dev, z_host = torch.empty_like(x), torch.empty_like(x)
    capi.axpby_device[letter](gpu, _p(z_dev), n, _p(scal[3:]), _p(y), _p(scal[2:]), _p(x))
    capi.axpby[letter](gpu, _p(z_host), n, capi.scalar(letter, beta), _p(y), capi.scalar(letter, alpha), _p(x))
    torch.cuda.synchronize()
    assert torch.equal(z_dev, z_host)
    # and the oracle's axpby (daxpby.cu:31-45 restated) with the same coefficients: bit for bit
    assert z_dev.cpu().numpy().tobytes() == O.axpby(letter, n, beta, y.cpu().numpy(), alpha, x.cpu().numpy()).tobytes()
    # *beta == 0 and beta == NULL: y is not read (NaNs in it do not reach z), as with spgpu?axpby(beta = 0)
    y_nan = torch.full_like(y, float("nan"))
    for beta_ptr in (scal[4:], None):
        z_dev.fill_(7)
        capi.axpby_device[letter](gpu, _p(z_dev), n, _p(beta_ptr), _p(y_nan), _p(scal[2:]), _p(x))
        capi.axpby[letter](gpu, _p(z_host), n, capi.scalar(letter, 0.0), _p(y_nan), capi.scalar(letter, alpha), _p(x))
        torch.cuda.synchronize()
        assert torch.equal(z_dev, z_host) and not torch.isnan(z_dev).any()
    # coefficients as quotients: beta = den/num, alpha = -(num/den); NULL operands stand for 1
    capi.axpby_quot_device[letter](gpu, _p(z_dev), n, _p(scal[1:]), _p(scal[0:]), _p(y), _p(scal[0:]), _p(scal[1:]), 1, _p(x))
    capi.axpby[letter](gpu, _p(z_host), n, capi.scalar(letter, den / num), _p(y), capi.scalar(letter, -(num / den)), _p(x))
    torch.cuda.synchronize()
    assert torch.equal(z_dev, z_host)
    capi.axpby_quot_device[letter](gpu, _p(z_dev), n, None, None, _p(y), None, _p(scal[1:]), 0, _p(x))
    capi.axpby[letter](gpu, _p(z_host), n, capi.scalar(letter, 1.0), _p(y), capi.scalar(letter, type(num)(1) / den), _p(x))
    torch.cuda.synchronize()
    assert torch.equal(z_dev, z_host)
    # in place, unaligned
    z1, z2 = x.clone(), x.clone()
    capi.axpby_device[letter](gpu, _p(z1[1:]), n - 1, _p(scal[3:]), _p(z1[1:]), _p(scal[2:]), _p(y[1:]))
    capi.axpby[letter](gpu, _p(z2[1:]), n - 1, capi.scalar(letter, beta), _p(z2[1:]), capi.scalar(letter, alpha), _p(y[1:]))
    torch.cuda.synchronize()
    assert torch.equal(z1, z2)


def test_cg_iteration_replays_from_one_graph(gpu):
    """20 CG iterations on a 2-D Laplacian: eager with host scalars (2 synchronising dots per iteration) vs ONE captured
    iteration replayed 20 times with the scalars on the device -- identical iterates."""
    import torch
    from spgpu_amd import capi, formats, synth
    n, _, r, c, v = synth.laplacian_2d_5pt(96)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
    mat = formats.DeviceHell(hell)
    b = formats.to_device(synth.values_for("D", 11, n))
    iters = 20

    def fresh():
        return torch.zeros_like(b), b.clone(), b.clone(), torch.empty_like(b)

    # eager
    x, rvec, p, ap = fresh()
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

    # one captured iteration
    x, rvec, p, ap = fresh()
    s = torch.zeros(8, dtype=torch.float64, device="cuda:0")   # rr, rr', pAp, alpha, -alpha, beta, one
    s[6] = 1.0
    RR, RRN, PAP, AL, NAL, BE, ONE = (s[i:] for i in range(7))
    side = torch.cuda.Stream()
    capi.spgpuSetStream(gpu, C.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()   # the vectors and scalars above were written on torch's stream; `side` does not wait for it by itself
    try:
        with torch.cuda.stream(side):
            capi.dot_device["D"](gpu, _p(RR), n, _p(rvec), _p(rvec))

            def iteration():
                mat.spmv(gpu, ap, None, 1.0, p, 0.0)
                capi.dot_device["D"](gpu, _p(PAP), n, _p(p), _p(ap))
                capi.div_device["D"](gpu, _p(AL), _p(RR), _p(PAP), 0)
                capi.div_device["D"](gpu, _p(NAL), _p(RR), _p(PAP), 1)
                capi.axpby_device["D"](gpu, _p(x), n, _p(ONE), _p(x), _p(AL), _p(p))
                capi.axpby_device["D"](gpu, _p(rvec), n, _p(ONE), _p(rvec), _p(NAL), _p(ap))
                capi.dot_device["D"](gpu, _p(RRN), n, _p(rvec), _p(rvec))
                capi.div_device["D"](gpu, _p(BE), _p(RRN), _p(RR), 0)
                capi.axpby_device["D"](gpu, _p(p), n, _p(BE), _p(p), _p(ONE), _p(rvec))
                capi.div_device["D"](gpu, _p(RR), _p(RRN), _p(ONE), 0)

            iteration()                      # warm-up outside the capture (module load), then start over
        side.synchronize()
        x, rvec, p, ap = fresh()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            capi.dot_device["D"](gpu, _p(RR), n, _p(rvec), _p(rvec))
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            iteration()
        # capture does not execute: state is still the start state
        for _ in range(iters):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        capi.spgpuSetStream(gpu, None)
    assert torch.equal(x, x_eager)
    assert s.cpu().numpy()[0] == rr_eager
