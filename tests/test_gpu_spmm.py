"""GPU: spgpu?hellspmm (new multi-vector product of the row-sharded path) against the oracle,
bit for bit (per (row, rhs) the kernel adds in ascending k, as the oracle does), plus the
multivector layout converters."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hell(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as f:
        g = {k: f[k] for k in f.files}
    letter = O.LETTER_OF[g["coo_vals"].dtype]
    hell = dict(letter=letter, rows=int(g["n_rows"]), values=g["hell_values"], indices=g["hell_indices"],
                hack_offsets=g["hell_hack_offsets"], hack_size=int(g["hack_size"]), row_lengths=g["row_lengths"],
                base=int(g["base"]), height=int(g["hell_height"]))
    return g, letter, hell


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("count", [1, 3, 4, 7, 8, 16, 21, 32])
@pytest.mark.parametrize("name", ["powerlaw_d_b0_h32", "powerlaw_s_b1_h64", "lap3d_16_d", "ctest_s"])
def test_spmm_matches_oracle(gpu, name, count):
    import torch
    from spgpu_amd import capi, formats, synth
    g, letter, hell = _hell(name)
    n_cols = int(g["n_cols"])
    X = synth.values_for(letter, 100 + count, n_cols * count).reshape(n_cols, count)
    Y = synth.values_for(letter, 200 + count, hell["rows"] * count).reshape(hell["rows"], count)
    mat = formats.DeviceHell(hell)
    dX, dY = formats.to_device(X), formats.to_device(Y)
    for beta in (0.0, -0.5):
        dZ = torch.full_like(dY, float("nan"))
        capi.hellspmm[letter](gpu, _p(dZ), _p(dY), capi.scalar(letter, 1.25), _p(mat.cM), _p(mat.rP), mat.hack_size,
                              _p(mat.hack_offsets), _p(mat.rS), None, 0, mat.rows, _p(dX), capi.scalar(letter, beta),
                              mat.base, count, count, count)
        torch.cuda.synchronize()
        want = O.hell_spmm(hell, X, Y if beta != 0 else None, 1.25, beta)
        assert dZ.cpu().numpy().tobytes() == want.tobytes()
    # every column of the product equals the single-vector SpMV in reference order
    z0 = O.hell_spmv(hell, np.ascontiguousarray(X[:, 0]), None, 1.25, 0.0, phases=1)
    assert np.array_equal(O.hell_spmm(hell, X, None, 1.25, 0.0)[:, 0], z0)


def test_spmm_leading_dimensions_row_reorder_and_in_place(gpu):
    import torch
    from spgpu_amd import capi, formats, synth
    g, letter, hell = _hell("powerlaw_d_b1_h64")
    n_cols, rows, count, ldx, ldz = int(g["n_cols"]), hell["rows"], 6, 9, 8
    Xp = synth.values_for("D", 1, n_cols * ldx).reshape(n_cols, ldx)
    Yp = synth.values_for("D", 2, rows * ldz).reshape(rows, ldz)
    perm = np.random.default_rng(4).permutation(rows).astype(np.int32)
    mat = formats.DeviceHell(hell, r_idx=perm)
    dX, dZ = formats.to_device(Xp), formats.to_device(Yp)   # Z aliases Y
    capi.hellspmm["D"](gpu, _p(dZ), _p(dZ), 2.0, _p(mat.cM), _p(mat.rP), mat.hack_size, _p(mat.hack_offsets), _p(mat.rS),
                       _p(mat.rIdx), 0, rows, _p(dX), 0.75, mat.base, count, ldx, ldz)
    torch.cuda.synchronize()
    got = dZ.cpu().numpy()
    want = O.hell_spmm(hell, np.ascontiguousarray(Xp[:, :count]), np.ascontiguousarray(Yp[:, :count]), 2.0, 0.75, r_idx=perm)
    assert got[:, :count].tobytes() == want.tobytes()
    assert np.array_equal(got[:, count:], Yp[:, count:])   # padding columns of Z untouched


@pytest.mark.parametrize("letter", "SD")
def test_multivector_layout_converters(gpu, letter):
    import torch
    from spgpu_amd import capi, formats, synth
    n, count, pitch, ld = 1003, 5, 1024, 7
    src = synth.values_for(letter, 9, count * pitch)
    d_src = formats.to_device(src)
    inter = torch.zeros(n * ld, dtype=d_src.dtype, device="cuda:0")
    capi.mv_interleave[letter](gpu, _p(inter), ld, _p(d_src), pitch, n, count)
    back = torch.zeros_like(d_src)
    capi.mv_deinterleave[letter](gpu, _p(back), pitch, _p(inter), ld, n, count)
    torch.cuda.synchronize()
    got = inter.cpu().numpy().reshape(n, ld)
    for j in range(count):
        assert np.array_equal(got[:, j], src[j * pitch:j * pitch + n])
        assert np.array_equal(back.cpu().numpy()[j * pitch:j * pitch + n], src[j * pitch:j * pitch + n])
    assert not np.any(got[:, count:])


@pytest.mark.parametrize("pattern", ["banded", "random"])
def test_own_rest_column_split_reproduces_the_block(gpu, pattern):
    """bench.py's overlap path: a row block cut by column ownership (own + rest) gives the block's product."""
    import torch
    from spgpu_amd import capi, synth
    rows, n_total, L, k, first = 3200, 12800, 32, 16, 6400
    block = synth.hell_uniform_on_device(rows, L, pattern, "D", 32, seed=5, n_cols=n_total, row_offset=first)
    own, rest = synth.split_uniform_hell_by_columns(block, first, rows)
    assert own["nnz"] + rest["nnz"] == block["nnz"]
    X = synth.device_vector(n_total * k, "D", 6).view(n_total, k)
    torch.cuda.synchronize()

    def product(part, Z, Y, Xt, beta):
        capi.hellspmm["D"](gpu, _p(Z), _p(Y), 1.0, _p(part["cM"]), _p(part["rP"]), 32, _p(part["hack_offsets"]),
                           _p(part["rS"]), None, L, rows, _p(Xt), beta, 0, k, k, k)

    z_whole = torch.empty(rows, k, dtype=torch.float64, device="cuda:0")
    z_split = torch.empty_like(z_whole)
    product(block, z_whole, None, X, 0.0)
    product(own, z_split, None, X[first:first + rows], 0.0)     # columns rebased to the owned X block
    product(rest, z_split, z_split, X, 1.0)
    torch.cuda.synchronize()
    a, b = z_whole.cpu().numpy(), z_split.cpu().numpy()
    assert np.max(np.abs(a - b) / (np.abs(a) + 1.0)) <= 1e-13
    # and the whole block against the oracle, bit for bit
    sub = synth.hell_rows_to_host(block, 0, rows)
    assert a.tobytes() == O.hell_spmm(sub, X.cpu().numpy(), None, 1.0, 0.0).tobytes()


@pytest.mark.parametrize("letter,hs,count,base", [("D", 32, 16, 0), ("D", 64, 12, 1), ("D", 96, 16, 1), ("S", 32, 16, 1),
                                                  ("S", 64, 10, 0), ("D", 32, 32, 0), ("D", 32, 8, 0), ("S", 64, 6, 1),
                                                  ("D", 96, 7, 1), ("S", 32, 5, 0)])
def test_spmm_window_tile_with_ragged_rows(gpu, letter, hs, count, base):
    """The LDS-tile path (columns of a workgroup's 256 rows inside a narrow window) on what it has to get right:
    rows of different lengths including empty ones, a last wavefront and a last hack that are partly filled,
    one-based indices, fewer than 16 right-hand sides, hack sizes 32/64/96 -- bit for bit against the oracle."""
    import torch
    from spgpu_amd import capi, formats, synth
    rng = np.random.default_rng(hs + count)
    rows = 1000 + hs // 2 + 5                      # not a multiple of 4, 32, 64 or the hack size
    lengths = rng.integers(0, 32, size=rows)         # a row near the border has 31 columns to choose from
    lengths[rng.integers(0, rows, size=40)] = 0
    lengths[:64] = 24                              # one wavefront with uniform rows: the unchecked loop only
    r = np.repeat(np.arange(rows), lengths)
    c = np.concatenate([np.sort(rng.choice(np.arange(max(0, i - 30), min(rows, i + 31)), size=n, replace=False))
                        for i, n in enumerate(lengths)]) if r.size else np.zeros(0, dtype=np.int64)
    v = synth.values_for(letter, 3, r.size)
    hell = formats.ell_to_hell(formats.coo_to_ell(rows, r, c, v, ell_base=base), hs)
    X = synth.values_for(letter, 4, rows * count).reshape(rows, count)
    Y = synth.values_for(letter, 5, rows * count).reshape(rows, count)
    mat = formats.DeviceHell(hell)
    dX, dY = formats.to_device(X), formats.to_device(Y)
    for beta in (0.0, 0.5):
        dZ = torch.full_like(dY, float("nan"))
        capi.hellspmm[letter](gpu, _p(dZ), _p(dY), capi.scalar(letter, -1.5), _p(mat.cM), _p(mat.rP), mat.hack_size,
                              _p(mat.hack_offsets), _p(mat.rS), None, 0, mat.rows, _p(dX), capi.scalar(letter, beta),
                              mat.base, count, count, count)
        torch.cuda.synchronize()
        want = O.hell_spmm(hell, X, Y if beta != 0 else None, -1.5, beta)
        assert dZ.cpu().numpy().tobytes() == want.tobytes()


def test_in_place_sum_leaves_rows_without_entries_untouched(gpu):
    """Z += alpha*A*X (Z == Y, beta == 1): rows with rS == 0 are neither read nor written (spmm.h) -- even a -0.0
    or a NaN sitting there survives -- and the other rows equal the oracle called the same way, bit for bit."""
    import torch
    from spgpu_amd import capi, formats, synth
    rng = np.random.default_rng(12)
    rows, count = 2000, 16
    lengths = np.where(rng.random(rows) < 0.8, 0, rng.integers(1, 9, size=rows))
    r = np.repeat(np.arange(rows), lengths)
    c = rng.integers(0, rows, size=r.size)
    v = synth.values_for("D", 7, r.size)
    hell = formats.ell_to_hell(formats.coo_to_ell(rows, r, c, v), 32)
    X = synth.values_for("D", 8, rows * count).reshape(rows, count)
    Z0 = synth.values_for("D", 9, rows * count).reshape(rows, count)
    empty = np.flatnonzero(lengths == 0)
    Z0[empty[0::2]] = -0.0
    Z0[empty[1::2]] = np.nan
    mat = formats.DeviceHell(hell)
    dX, dZ = formats.to_device(X), formats.to_device(Z0)
    capi.hellspmm["D"](gpu, _p(dZ), _p(dZ), 0.5, _p(mat.cM), _p(mat.rP), mat.hack_size, _p(mat.hack_offsets), _p(mat.rS),
                       None, 0, rows, _p(dX), 1.0, mat.base, count, count, count)
    torch.cuda.synchronize()
    got = dZ.cpu().numpy()
    assert got[empty].tobytes() == Z0[empty].tobytes()
    assert got.tobytes() == O.hell_spmm(hell, X, Z0, 0.5, 1.0, in_place=True).tobytes()


def test_spmm_randomized_shapes(gpu):
    """40 random problems around the strip kernel's branch points: rows longer than the 32 slab columns whose indices
    stay in registers, windows that fit the LDS tile and windows that do not (per workgroup: both kinds in one matrix),
    hack sizes 32..128, both index bases, 10..32 right-hand sides, leading dimensions larger than the count."""
    import torch
    from spgpu_amd import capi, formats, synth
    rng = np.random.default_rng(2024)
    for trial in range(40):
        letter = "D" if trial % 3 else "S"
        hs = int(rng.choice([32, 64, 96, 128]))
        base = int(rng.integers(0, 2))
        rows = int(rng.integers(1, 1500))
        cols = int(rng.integers(max(64, rows // 2), 2 * rows + 200))
        count = int(rng.choice([10, 12, 14, 16, 18, 32]))
        ld = count + int(rng.choice([0, 2, 6]))
        max_len = int(rng.choice([3, 20, 40, 90]))
        lengths = rng.integers(0, max_len + 1, size=rows)
        half = int(rng.choice([8, 40, 150]))                  # band half-width; some row blocks get scattered columns
        scattered = rng.random((rows + 255) // 256) < 0.3
        r_parts, c_parts = [], []
        for i, n in enumerate(lengths):
            if n == 0:
                continue
            centre = int(i * cols / max(rows, 1))
            lo, hi = (0, cols) if scattered[i // 256] else (max(0, centre - half), min(cols, centre + half + 1))
            n = min(int(n), hi - lo)
            c_parts.append(np.sort(rng.choice(np.arange(lo, hi), size=n, replace=False)))
            r_parts.append(np.full(n, i))
        r = np.concatenate(r_parts) if r_parts else np.zeros(0, dtype=np.int64)
        c = np.concatenate(c_parts) if c_parts else np.zeros(0, dtype=np.int64)
        v = synth.values_for(letter, 50 + trial, r.size)
        hell = formats.ell_to_hell(formats.coo_to_ell(rows, r, c, v, ell_base=base), hs)
        Xp = synth.values_for(letter, 90 + trial, cols * ld).reshape(cols, ld)
        Yp = synth.values_for(letter, 130 + trial, rows * ld).reshape(rows, ld)
        mat = formats.DeviceHell(hell)
        dX, dY = formats.to_device(Xp), formats.to_device(Yp)
        beta = float(rng.choice([0.0, 1.0, -0.75]))
        dZ = dY.clone()
        capi.hellspmm[letter](gpu, _p(dZ), _p(dY), capi.scalar(letter, 0.5), _p(mat.cM), _p(mat.rP), mat.hack_size,
                              _p(mat.hack_offsets), _p(mat.rS), None, 0, mat.rows, _p(dX), capi.scalar(letter, beta),
                              mat.base, count, ld, ld)
        torch.cuda.synchronize()
        got = dZ.cpu().numpy()
        want = O.hell_spmm(hell, np.ascontiguousarray(Xp[:, :count]), np.ascontiguousarray(Yp[:, :count]) if beta else None, 0.5, beta)
        assert got[:, :count].tobytes() == want.tobytes(), (trial, letter, hs, base, rows, cols, count, ld, max_len, half)
        assert np.array_equal(got[:, count:], Yp[:, count:]), trial


@pytest.mark.parametrize("hs", [8, 48, 80])
@pytest.mark.parametrize("letter", "SD")
def test_spmm_hack_sizes_off_the_strip_path(gpu, letter, hs):
    """hackSize not a multiple of 32: the one-row-per-lane kernels (LDS-tiled for 16 rhs, plain otherwise) take over."""
    import torch
    from spgpu_amd import capi, formats, synth
    rng = np.random.default_rng(hs)
    rows, cols = 777, 900
    lengths = rng.integers(0, 21, size=rows)                   # a border row has 21 columns to choose from
    r = np.repeat(np.arange(rows), lengths)
    c = np.concatenate([np.sort(rng.choice(np.arange(max(0, i - 20), min(cols, i + 21)), size=n, replace=False))
                        for i, n in enumerate(lengths)])
    v = synth.values_for(letter, 1, r.size)
    hell = formats.ell_to_hell(formats.coo_to_ell(rows, r, c, v), hs)
    mat = formats.DeviceHell(hell)
    for count in (16, 9, 3):
        X = synth.values_for(letter, 2, cols * count).reshape(cols, count)
        Y = synth.values_for(letter, 3, rows * count).reshape(rows, count)
        dX, dY = formats.to_device(X), formats.to_device(Y)
        dZ = torch.full_like(dY, float("nan"))
        capi.hellspmm[letter](gpu, _p(dZ), _p(dY), capi.scalar(letter, 2.0), _p(mat.cM), _p(mat.rP), mat.hack_size,
                              _p(mat.hack_offsets), _p(mat.rS), None, 0, mat.rows, _p(dX), capi.scalar(letter, 0.25),
                              mat.base, count, count, count)
        torch.cuda.synchronize()
        assert dZ.cpu().numpy().tobytes() == O.hell_spmm(hell, X, Y, 2.0, 0.25).tobytes()


@pytest.mark.parametrize("letter", ["D", "S"])
def test_band_wavefronts_and_their_neighbours(gpu, letter):
    """Round 4: wavefronts whose 64 rows form a band through all their columns take the sliding-window loop of the strip kernel
    (hell_spmm.hip, BAND).  The banded test matrix wraps its columns at both ends, so its first and last wavefronts are NOT bands and
    run the general loop beside band wavefronts of the same workgroup; then a matrix with scattered columns is copied over the same
    arrays (no wavefront qualifies).  Every call: the oracle's bits (beta != 0, 16 right-hand sides)."""
    import torch
    from spgpu_amd import capi, synth
    n, k, L = 40_000 // 32 * 32, 16, 32
    band = synth.hell_uniform_on_device(n, L, "banded", letter, 32, seed=5)
    scat = synth.hell_uniform_on_device(n, L, "random", letter, 32, seed=6)
    X = synth.device_vector(n * k, letter, 7).view(n, k)
    Y = synth.device_vector(n * k, letter, 8).view(n, k)
    torch.cuda.synchronize()
    xs, ys = X.cpu().numpy(), Y.cpu().numpy()
    want = {name: O.hell_spmm(synth.hell_rows_to_host(h, 0, n), xs, ys, 1.5, -0.25) for name, h in (("band", band), ("scat", scat))}
    live = dict(band)

    def run(expect):
        Z = torch.full_like(Y, float("nan"))
        capi.hellspmm[letter](gpu, _p(Z), _p(Y), capi.scalar(letter, 1.5), _p(live["cM"]), _p(live["rP"]), 32, _p(live["hack_offsets"]),
                              _p(live["rS"]), None, L, n, _p(X), capi.scalar(letter, -0.25), 0, k, k, k)
        torch.cuda.synchronize()
        assert Z.cpu().numpy().tobytes() == want[expect].tobytes(), expect

    for _ in range(2):
        run("band")
    live["cM"].copy_(scat["cM"])
    live["rP"].copy_(scat["rP"])
    torch.cuda.synchronize()
    run("scat")
