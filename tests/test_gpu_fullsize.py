"""GPU: size-independent properties at (or near) BASELINE.json's full sizes, and the device-side
generators the full-size runs depend on, checked against the host converters on small instances."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("m", [32, 64])
def test_device_hdia_laplacian_equals_the_converter(gpu, m):
    """synth.hdia_laplacian7_on_device == cooToHdia(laplacian_3d_7pt) byte for byte."""
    from spgpu_amd import formats, synth
    n, nc, r, c, v = synth.laplacian_3d_7pt(m)
    host = formats.coo_to_hdia(n, nc, r, c, v, 32)
    dev = synth.hdia_laplacian7_on_device(m, "D", 32)
    assert dev["height"] == host["height"] and dev["nnz"] == r.size
    assert np.array_equal(dev["hack_offsets"].cpu().numpy(), host["hack_offsets"])
    assert np.array_equal(dev["offsets"].cpu().numpy(), host["offsets"])
    assert dev["dM"].cpu().numpy().tobytes() == host["values"].tobytes()


def test_device_uniform_hell_equals_the_converters(gpu):
    """synth.hell_uniform_on_device lays HELL out as cooToEll + ellToHell do."""
    from spgpu_amd import formats, synth
    h = synth.hell_uniform_on_device(640, 8, "random", "D", 32, seed=3)
    cols = h["rP"].view(20, 8, 32).permute(0, 2, 1).reshape(640, 8).cpu().numpy()     # [row][k]
    vals = h["cM"].view(20, 8, 32).permute(0, 2, 1).reshape(640, 8).cpu().numpy()
    rows = np.repeat(np.arange(640), 8)
    hell = formats.ell_to_hell(formats.coo_to_ell(640, rows, cols.reshape(-1), vals.reshape(-1)), 32)
    assert hell["indices"].tobytes() == h["rP"].cpu().numpy().tobytes()
    assert hell["values"].tobytes() == h["cM"].cpu().numpy().tobytes()
    assert np.array_equal(hell["hack_offsets"], h["hack_offsets"].cpu().numpy())


@pytest.mark.parametrize("pattern", ["banded", "random"])
def test_full_size_hell_fp64_properties(gpu, pattern):
    """BASELINE configs[1] size (10 M rows x 32): linearity in x, ELL == HELL bit for bit (one summation
    order), alpha/beta epilogue, and oracle parity on row windows."""
    import torch
    from spgpu_amd import capi, synth
    n, L = 10_000_000, 32
    h = synth.hell_uniform_on_device(n, L, pattern, "D", 32, seed=1)
    x1, x2, y = (synth.device_vector(n, "D", s) for s in (3, 4, 5))
    z1, z2, z12, zb = (torch.empty_like(y) for _ in range(4))
    torch.cuda.synchronize()

    def hell(z, yy, alpha, x, beta):
        capi.hellspmv["D"](gpu, _p(z), _p(yy), alpha, _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), None,
                           L, n, _p(x), beta, 0)

    hell(z1, None, 1.0, x1, 0.0)
    hell(z2, None, 1.0, x2, 0.0)
    x12 = x1 + 2.0 * x2
    torch.cuda.synchronize()
    hell(z12, None, 1.0, x12, 0.0)
    hell(zb, y, -0.5, x1, 2.0)
    torch.cuda.synchronize()
    # linearity A(x1 + 2 x2) = A x1 + 2 A x2 within rounding of 32-term sums of values in [0,1)
    err = (z12 - (z1 + 2.0 * z2)).abs().max().item()
    assert err <= 1e-12 * 3 * L
    # epilogue: zb = -0.5 * (A x1) + 2 y, with one rounding for the fma
    ref = torch.addcmul(2.0 * y, z1, torch.tensor(-0.5, dtype=torch.float64, device=y.device))
    assert (zb - ref).abs().max().item() <= 1e-14 * (L + 4)
    # the same slots read as ELL (uniform rows: pitch = n, slot (r,k) = r + k*n) give the same bits
    cM_ell = h["cM"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    rP_ell = h["rP"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    ze = torch.empty_like(y)
    torch.cuda.synchronize()
    capi.ellspmv["D"](gpu, _p(ze), None, 1.0, _p(cM_ell), _p(rP_ell), n, n, _p(h["rS"]), None, L, L, n, _p(x1), 0.0, 0)
    torch.cuda.synchronize()
    assert torch.equal(ze, z1)
    # dot(z,z) as the reference's harness prints it, against torch
    d = capi.dot["D"](gpu, n, _p(z1), _p(z1))
    assert abs(d - float(torch.dot(z1, z1))) <= 1e-10 * d
    # oracle parity on three windows
    xs = x1.cpu().numpy()
    for first in (0, 4_999_936, n - 2048):
        sub = synth.hell_rows_to_host(h, first, 2048)
        assert z1[first:first + 2048].cpu().numpy().tobytes() == O.default_spmv(sub, xs, None, 1.0, 0.0).tobytes()


# ---- independent of any kernel-shaped oracle: extended precision straight from the stored entries --------------------
TOL = {"S": 1e-4, "D": 1e-6}


def _exact_rows(values, cols, lens, x, alpha=1.0):
    """alpha * sum_k values[r][k] * x[cols[r][k]] over k < lens[r] in 64-bit-mantissa arithmetic, and the magnitude
    |alpha| * sum |a x| the north_star tolerance is relative to.  values/cols: [rows][depth] host arrays."""
    v = values.astype(np.longdouble)
    xs = x[np.clip(cols, 0, x.size - 1)].astype(np.longdouble)
    live = np.arange(values.shape[1])[None, :] < lens[:, None]
    prod = np.where(live, v * xs, np.longdouble(0))
    return np.longdouble(alpha) * prod.sum(axis=1), abs(alpha) * np.abs(prod).sum(axis=1)


def _hell_window(h, first, rows):
    """rows [first, first+rows) of a uniform device HELL dict as ([rows][L] values, columns, lengths) on the host"""
    hs, L = h["hack_size"], h["row_len"]
    s0, s1 = (first // hs) * hs * L, ((first + rows) // hs) * hs * L
    vals = h["cM"][s0:s1].view(rows // hs, L, hs).permute(0, 2, 1).reshape(rows, L).cpu().numpy()
    cols = h["rP"][s0:s1].view(rows // hs, L, hs).permute(0, 2, 1).reshape(rows, L).cpu().numpy()
    return vals, cols, np.full(rows, L)


@pytest.mark.parametrize("letter,pattern", [("D", "banded"), ("D", "random"), ("S", "window")])
def test_full_size_hell_and_ell_against_extended_precision(gpu, letter, pattern):
    """10 M rows x 32: HELL and ELL results of five row windows against sums formed in longdouble from the device-built
    arrays themselves -- no oracle, no assumption about the kernel's summation order -- within the north_star bound
    (1e-6 fp64 / 1e-4 fp32 of |alpha| sum |a x| + |beta y|)."""
    import torch
    from spgpu_amd import capi, synth
    n, L = 10_000_000, 32
    h = synth.hell_uniform_on_device(n, L, pattern, letter, 32, seed=1)
    x, y = synth.device_vector(n, letter, 3), synth.device_vector(n, letter, 4)
    z = torch.empty_like(y)
    torch.cuda.synchronize()
    alpha, beta = -1.25, 0.5
    one = capi.scalar(letter, alpha), capi.scalar(letter, beta)
    capi.hellspmv[letter](gpu, _p(z), _p(y), one[0], _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), None, L, n,
                          _p(x), one[1], 0)
    cM_ell = h["cM"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    rP_ell = h["rP"].view(n // 32, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    ze = torch.empty_like(y)
    torch.cuda.synchronize()
    capi.ellspmv[letter](gpu, _p(ze), _p(y), one[0], _p(cM_ell), _p(rP_ell), n, n, _p(h["rS"]), None, L, L, n, _p(x), one[1], 0)
    torch.cuda.synchronize()
    xs = x.cpu().numpy()
    for first in (0, 2_500_000 // 2048 * 2048, 4_999_936, 7_500_000 // 2048 * 2048, n - 2048):
        vals, cols, lens = _hell_window(h, first, 2048)
        exact, scale = _exact_rows(vals, cols, lens, xs, alpha)
        ys = y[first:first + 2048].cpu().numpy().astype(np.longdouble)
        exact = exact + np.longdouble(beta) * ys
        bound = TOL[letter] * (scale + abs(beta) * np.abs(ys)) + np.finfo(np.float64).tiny
        for got in (z, ze):
            err = np.abs(got[first:first + 2048].cpu().numpy().astype(np.longdouble) - exact)
            assert np.all(err <= bound), float(np.max(err / bound))


def test_full_size_hdia_512_cubed(gpu):
    """BASELINE configs[3] at full size (7-point Laplacian 512^3, 134 M rows): linearity, the epilogue identity,
    DIA == HDIA on a sub-cube (the same kernel over one all-rows hack must agree with the hacked form), and five windows
    against extended-precision sums formed from the stencil itself."""
    import torch
    from spgpu_amd import capi, formats, synth
    m = 512
    d = synth.hdia_laplacian7_on_device(m, "D", 32)
    n = d["rows"]
    x1, x2, y = (synth.device_vector(n, "D", s) for s in (3, 4, 5))
    z1, z2, z12, zb = (torch.empty_like(y) for _ in range(4))
    torch.cuda.synchronize()

    def hdia(z, yy, alpha, x, beta):
        capi.hdiaspmv["D"](gpu, _p(z), _p(yy), alpha, _p(d["dM"]), _p(d["offsets"]), 32, _p(d["hack_offsets"]), n, n, _p(x), beta)

    hdia(z1, None, 1.0, x1, 0.0)
    hdia(z2, None, 1.0, x2, 0.0)
    x12 = x1 + 2.0 * x2
    torch.cuda.synchronize()
    hdia(z12, None, 1.0, x12, 0.0)
    hdia(zb, y, -0.5, x1, 2.0)
    torch.cuda.synchronize()
    assert (z12 - (z1 + 2.0 * z2)).abs().max().item() <= 1e-12 * 30
    ref = torch.addcmul(2.0 * y, z1, torch.tensor(-0.5, dtype=torch.float64, device=y.device))
    assert (zb - ref).abs().max().item() <= 1e-13
    # extended precision from the stencil: (A x)_i = 6 x_i - sum of the existing neighbours
    xs = x1.cpu().numpy().astype(np.longdouble)
    for first in (0, n // 3 // 2048 * 2048, n // 2, n - 2048):
        i = np.arange(first, first + 2048)
        gx, gy, gz = i % m, (i // m) % m, i // (m * m)
        exact = 6 * xs[i]
        scale = 6 * np.abs(xs[i])
        for ok, off in ((gx > 0, -1), (gx < m - 1, 1), (gy > 0, -m), (gy < m - 1, m), (gz > 0, -m * m), (gz < m - 1, m * m)):
            nb = np.where(ok, xs[np.clip(i + off, 0, n - 1)], np.longdouble(0))
            exact = exact - nb
            scale = scale + np.abs(nb)
        err = np.abs(z1[first:first + 2048].cpu().numpy().astype(np.longdouble) - exact)
        assert np.all(err <= 1e-6 * scale + np.finfo(np.float64).tiny)
    del z2, z12, zb, x2, x12
    torch.cuda.empty_cache()
    # DIA == HDIA on a 64^3 sub-problem built by the converters (same diagonals, one all-rows hack vs hacks of 32)
    n3, nc, r, c, v = synth.laplacian_3d_7pt(64)
    xx, yy = synth.hashed_vector(n3), synth.hashed_vector(n3, multiplier=40503)
    dx, dy = formats.to_device(xx), formats.to_device(yy)
    za, zh = torch.empty_like(dy), torch.empty_like(dy)
    formats.DeviceDia(formats.coo_to_dia(n3, nc, r, c, v)).spmv(gpu, za, dy, 1.5, dx, -0.5)
    formats.DeviceHdia(formats.coo_to_hdia(n3, nc, r, c, v, 32)).spmv(gpu, zh, dy, 1.5, dx, -0.5)
    torch.cuda.synchronize()
    assert torch.equal(za, zh)


@pytest.mark.parametrize("aligned", [False, True])
def test_full_size_power_law_ordered_on_the_device(gpu, aligned):
    """(aligned: spgpuOellOrderAlignedDevice -- every window of the order is one 2 048-row block of the new numbering.)
    The north_star target at full size: 10 M rows, power-law lengths (mean 32, max 2048), fp64, columns near the row.
    Ordered on the device (windows of 2048 rows, rows longer than 256 set aside), built through the COO route, run
    through rIdx: slots per nonzero <= 1.1 (plain: ~5), the order is a permutation that keeps windowed rows inside
    their window, three ordered windows equal the oracle in the kernel's order bit for bit, the ordered product equals the
    plain product within the north_star bound on EVERY row, and windows of both agree with extended precision."""
    import torch
    from spgpu_amd import capi, formats, synth
    n = 10_000_000
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    coo = synth.ragged_coo_on_device(lengths, n, "near", 2048, "D", seed=5)
    x = synth.device_vector(n, "D", 3)
    torch.cuda.synchronize()
    plain = formats.coo_to_ordered_hell_device(gpu, n, *coo, "D", 32, order=False)
    ordered = formats.coo_to_ordered_hell_device(gpu, n, *coo, "D", 32, 2048, 256, aligned=aligned)
    assert plain["slots"] / plain["nnz"] > 4.5 and ordered["slots"] / ordered["nnz"] <= 1.1
    r_idx = ordered["rIdx"].to(torch.int64)
    assert torch.equal(torch.sort(r_idx).values, torch.arange(n, device=r_idx.device))
    long_rows = int((torch.from_numpy(lengths) > 256).sum())
    moved = (r_idx[long_rows:] - torch.arange(n - long_rows, device=r_idx.device)).abs().max().item()
    assert moved < 2048 + long_rows          # a windowed row stays inside its window (shifted by the rows set aside)
    if aligned:      # every 2 048-row block behind the set-aside rows holds 2 048 consecutive shorter rows of the original numbering
        first_block = (long_rows // 2048 + 1) * 2048
        blocks = r_idx[first_block:first_block + (n - first_block) // 2048 * 2048].view(-1, 2048)
        spans = blocks.max(dim=1).values - blocks.min(dim=1).values
        assert int(spans.max()) < 2048 + 512 and bool((blocks.min(dim=1).values[1:] > blocks.max(dim=1).values[:-1]).all())
    zp, zo = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
    for h, z in ((plain, zp), (ordered, zo)):
        capi.hellspmv["D"](gpu, _p(z), None, 1.0, _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), _p(h["rIdx"]), 32, n,
                           _p(x), 0.0, 0)
    torch.cuda.synchronize()
    # the two layouts add the same products in different orders: every row within 1e-6 of sum |a x| (all terms >= 0 here)
    assert ((zp - zo).abs() <= 1e-6 * zp.abs() + 1e-300).all()
    xs = x.cpu().numpy()
    shape = O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP)
    for first in (0, 5_000_000 // 2048 * 2048, n - 2048):
        sub = synth.hell_rows_to_host_general(ordered, first, 2048)
        want = O.spmv_tail(sub, xs, None, 1.0, 0.0, **shape)
        assert zo[r_idx[first:first + 2048]].cpu().numpy().tobytes() == want.tobytes()
    # extended precision from the COO triplets of three row ranges (independent of every layout)
    rows_t, cols_t, vals_t = coo
    starts = np.concatenate([[0], np.cumsum(lengths.astype(np.int64))])
    for first in (0, 3_333_333, n - 4096):
        e0, e1 = int(starts[first]), int(starts[first + 4096])
        rr = rows_t[e0:e1].cpu().numpy() - first
        prod = vals_t[e0:e1].cpu().numpy().astype(np.longdouble) * xs[cols_t[e0:e1].cpu().numpy()].astype(np.longdouble)
        exact = np.zeros(4096, np.longdouble)
        np.add.at(exact, rr, prod)
        for z in (zp, zo):
            err = np.abs(z[first:first + 4096].cpu().numpy().astype(np.longdouble) - exact)
            assert np.all(err <= 1e-6 * np.abs(exact) + np.finfo(np.float64).tiny)


def test_full_size_c3_hell_fp32_power_law(gpu):
    """BASELINE configs[2] at 10 M rows: HELL fp32 on power-law lengths with random columns -- windows against extended
    precision, footprint against the ELL the same rows would need."""
    import torch
    from spgpu_amd import capi, synth
    n = 10_000_000
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    h = synth.hell_ragged_on_device(lengths, n, "S", 32, seed=5)
    x = synth.device_vector(n, "S", 3)
    z = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    capi.hellspmv["S"](gpu, _p(z), None, 1.0, _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), None, 32, n, _p(x), 0.0, 0)
    torch.cuda.synchronize()
    hell_bytes = h["slots"] * 8 + n * 4 + (n // 32) * 4
    ell_bytes = ((n + 31) // 32 * 32) * int(lengths.max()) * 8 + n * 4
    assert hell_bytes < 0.1 * ell_bytes          # 12.8 GB against 164 GB
    xs = x.cpu().numpy()
    ho = h["hack_offsets"].cpu().numpy().astype(np.int64)
    for first in (0, 5_000_000 // 2048 * 2048, n - 2048):
        h0 = first // 32
        for hack in range(h0, h0 + 64, 9):
            depth = int(h["depth"][hack])
            s0 = int(ho[hack])
            vals = h["cM"][s0:s0 + depth * 32].view(depth, 32).t().cpu().numpy()
            cols = h["rP"][s0:s0 + depth * 32].view(depth, 32).t().cpu().numpy()
            lens = lengths[hack * 32:(hack + 1) * 32]
            exact, scale = _exact_rows(vals, cols, lens, xs)
            err = np.abs(z[hack * 32:(hack + 1) * 32].cpu().numpy().astype(np.longdouble) - exact)
            assert np.all(err <= 1e-4 * scale + np.finfo(np.float32).tiny)


def test_full_size_c3_ell_fp32_power_law(gpu):
    """The ELL half of BASELINE configs[2] at its full size: 10 M rows of power-law lengths (max 2048) as ELL are
    pitch x 2048 slots = 164 GB of coefficients and indices (the HELL of the same rows: 12.8 GB).  Built in HBM, run through
    spgpuSellspmv (slot index r + k * pitch exceeds 2^31: 64-bit index arithmetic), rows of the first, a middle and the last
    hack against sums in 64-bit-mantissa arithmetic formed from the stored slots."""
    import torch
    from spgpu_amd import capi, synth
    n = 10_000_000
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    free, _ = torch.cuda.mem_get_info()
    pitch, deepest = (n + 31) // 32 * 32, int(lengths.max())
    need = pitch * deepest * 8
    if free < need + (4 << 30):
        pytest.skip(f"needs {need / 1e9:.0f} GB of free device memory, {free / 1e9:.0f} GB free")
    e = synth.ell_ragged_on_device(lengths, n, "S", seed=6)
    assert e["pitch"] * e["max_row"] > 2**31
    x = synth.device_vector(n, "S", 3)
    z = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    capi.ellspmv["S"](gpu, _p(z), None, 1.0, _p(e["cM"]), _p(e["rP"]), e["pitch"], e["pitch"], _p(e["rS"]), None, 32, e["max_row"], n,
                      _p(x), 0.0, 0)
    torch.cuda.synchronize()
    xs = x.cpu().numpy()
    got = z.cpu().numpy()
    longest = int(np.argmax(lengths))
    for first in (0, longest // 32 * 32, 5_000_000 // 32 * 32, n - 32):
        rows = torch.arange(first, first + 32, device="cuda", dtype=torch.int64)
        depth = int(lengths[first:first + 32].max())
        slots = rows[:, None] + torch.arange(depth, device="cuda", dtype=torch.int64)[None, :] * e["pitch"]
        vals, cols = e["cM"][slots].cpu().numpy(), e["rP"][slots].cpu().numpy()
        exact, scale = _exact_rows(vals, cols, lengths[first:first + 32], xs)
        err = np.abs(got[first:first + 32].astype(np.longdouble) - exact)
        assert np.all(err <= TOL["S"] * scale + np.finfo(np.float32).tiny), first
    del e
    torch.cuda.empty_cache()


def test_full_size_spmm_shard_of_configs4(gpu):
    """The single-GPU slice of BASELINE configs[4]: 5 M rows x 32 nnz x 16 right-hand sides, fp64, through spgpuDhellspmm;
    windows of rows against sums in 64-bit-mantissa arithmetic formed from the stored slots (every right-hand side)."""
    import torch
    from spgpu_amd import capi, synth
    n, L, k = 5_000_000, 32, 16
    h = synth.hell_uniform_on_device(n, L, "banded", "D", 32, seed=11)
    X = synth.device_vector(n * k, "D", 21).view(n, k)
    Z = torch.empty_like(X)
    torch.cuda.synchronize()
    capi.hellspmm["D"](gpu, _p(Z), None, C.c_double(1.0), _p(h["cM"]), _p(h["rP"]), 32, _p(h["hack_offsets"]), _p(h["rS"]), None, L, n,
                       _p(X), C.c_double(0.0), 0, k, k, k)
    torch.cuda.synchronize()
    for first in (0, 2_500_000 // 2048 * 2048, n - 2048):
        vals, cols, _ = _hell_window(h, first, 2048)
        xs = X[torch.from_numpy(cols.astype(np.int64)).cuda()].cpu().numpy().astype(np.longdouble)      # [rows][L][k]
        prod = vals.astype(np.longdouble)[:, :, None] * xs
        exact, scale = prod.sum(axis=1), np.abs(prod).sum(axis=1)
        err = np.abs(Z[first:first + 2048].cpu().numpy().astype(np.longdouble) - exact)
        assert np.all(err <= TOL["D"] * scale), first


def test_configs0_laplacian_1024_through_the_host_converters(gpu):
    """BASELINE configs[0] on the GPU: 5-point Laplacian 1024 x 1024 (1 048 576 rows), COO -> cooToEll -> ellToHell on the
    host, uploaded, spgpuDhellspmv with alpha = 2, beta = -3 (the coefficients of the reference's ctest.c): every row
    against the oracle bit for bit, and against the stencil evaluated directly."""
    import torch
    from spgpu_amd import formats, synth
    n, m, r, c, v = synth.laplacian_2d_5pt(1024)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
    x, y = synth.hashed_vector(m), synth.hashed_vector(n, multiplier=40503)
    dx, dy = formats.to_device(x), formats.to_device(y)
    dz = torch.empty_like(dy)
    formats.DeviceHell(hell).spmv(gpu, dz, dy, 2.0, dx, -3.0)
    torch.cuda.synchronize()
    got = dz.cpu().numpy()
    assert got.tobytes() == O.default_spmv(hell, x, y, 2.0, -3.0).tobytes()
    acc = np.zeros(n, np.longdouble)
    np.add.at(acc, r, v.astype(np.longdouble) * x.astype(np.longdouble)[c])
    mag = np.zeros(n, np.longdouble)
    np.add.at(mag, r, np.abs(v.astype(np.longdouble) * x.astype(np.longdouble)[c]))
    want = 2 * acc - 3 * y.astype(np.longdouble)
    assert np.all(np.abs(got.astype(np.longdouble) - want) <= 1e-6 * (2 * mag + 3 * np.abs(y)))
