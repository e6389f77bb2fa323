"""Synthetic inputs of BASELINE.json / SURVEY.md section 8(d), all seeded.

Small/medium cases are produced as COO on the host (numpy) so that they run
through the same converters as a Matrix Market file would in the reference's
harness.  Generators are deterministic functions of their arguments.
"""
import numpy as np

_MASK64 = (1 << 64) - 1


def splitmix64(seed, idx):
    """Vectorised splitmix64 of (seed, idx) -> uint64 (column / value hashes)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(idx, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, dtype=np.float64):
    """U[0,1) vector from splitmix64(seed, i)."""
    u = (splitmix64(seed, np.arange(n, dtype=np.uint64)) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return u.astype(dtype)


def values_for(letter, seed, n):
    """Random coefficients/vectors of the API flavour S/D/C/Z in [-1, 1) (+ i[-1,1))."""
    real = {"S": np.float32, "D": np.float64, "C": np.float32, "Z": np.float64}[letter]
    re = (2.0 * uniform01(seed, n) - 1.0).astype(real)
    if letter in "SD":
        return re
    im = (2.0 * uniform01(seed + 7919, n) - 1.0).astype(real)
    return (re + 1j * im).astype(np.complex64 if letter == "C" else np.complex128)


def hashed_vector(n, multiplier=2654435761, dtype=np.float64):
    """x_i = ((i * multiplier) mod 2^32) / 2^32  (SURVEY 8(d), config C1)."""
    i = np.arange(n, dtype=np.uint64)
    return (((i * np.uint64(multiplier)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 4294967296.0).astype(dtype)


def laplacian_2d_5pt(n, dtype=np.float64, base=0):
    """5-point Laplacian on an n x n grid, natural order; per row ascending columns
    (-n, -1, 0, +1, +n); diagonal 4, off-diagonal -1 (SURVEY 8(a) generator)."""
    N = n * n
    i = np.arange(N, dtype=np.int64)
    gx, gy = i % n, i // n
    parts = [
        (gy > 0, i - n, -1.0), (gx > 0, i - 1, -1.0), (np.ones(N, bool), i, 4.0),
        (gx < n - 1, i + 1, -1.0), (gy < n - 1, i + n, -1.0),
    ]
    return _assemble_row_major(N, i, parts, dtype, base)


def laplacian_3d_7pt(m, dtype=np.float64, base=0):
    """7-point Laplacian on an m^3 grid, natural order; per row ascending columns
    (-m^2, -m, -1, 0, +1, +m, +m^2); diagonal 6, off-diagonal -1."""
    N = m * m * m
    i = np.arange(N, dtype=np.int64)
    gx, gy, gz = i % m, (i // m) % m, i // (m * m)
    parts = [
        (gz > 0, i - m * m, -1.0), (gy > 0, i - m, -1.0), (gx > 0, i - 1, -1.0),
        (np.ones(N, bool), i, 6.0),
        (gx < m - 1, i + 1, -1.0), (gy < m - 1, i + m, -1.0), (gz < m - 1, i + m * m, -1.0),
    ]
    return _assemble_row_major(N, i, parts, dtype, base)


def _assemble_row_major(N, i, parts, dtype, base):
    """Interleave per-stencil-point (mask, col, value) so that entries are row-major."""
    k = len(parts)
    mask = np.stack([p[0] for p in parts], axis=1)           # N x k
    cols = np.stack([p[1] for p in parts], axis=1)
    vals = np.broadcast_to(np.array([p[2] for p in parts], dtype=np.float64), (N, k))
    rows = np.broadcast_to(i[:, None], (N, k))
    sel = mask.reshape(-1)
    return (N, N, (rows.reshape(-1)[sel] + base).astype(np.int32), (cols.reshape(-1)[sel] + base).astype(np.int32),
            vals.reshape(-1)[sel].astype(dtype))


def ctest_matrix(dtype=np.float32):
    """The matrix of the reference's ctest.c:25-39: 100x100, 200 entries, rows[i]=cols[i]=i%100, value 1."""
    e = np.arange(200, dtype=np.int32)
    return 100, 100, (e % 100).astype(np.int32), (e % 100).astype(np.int32), np.ones(200, dtype=dtype)


def power_law_lengths(n_rows, mean=32.0, max_len=2048, seed=5, exponent=2.0):
    """Row lengths min(max_len, floor(c * u^(-1/exponent))), c tuned so that the mean is ~`mean`
    (SURVEY 8(d), config C3)."""
    u = np.maximum(uniform01(seed, n_rows), 1e-12)
    shape = u ** (-1.0 / exponent)
    lo, hi = 0.01, float(max_len)
    for _ in range(60):
        c = 0.5 * (lo + hi)
        m = np.minimum(max_len, np.floor(c * shape)).mean()
        lo, hi = (c, hi) if m < mean else (lo, c)
    return np.maximum(1, np.minimum(max_len, np.floor(0.5 * (lo + hi) * shape))).astype(np.int32)


def random_rows_coo(n_rows, n_cols, lengths, seed=1, letter="D", base=0, shuffle=False):
    """COO with `lengths[r]` entries in row r, columns splitmix64(seed, slot) mod n_cols,
    row-major order (optionally a seeded shuffle of the entry order)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    nnz = int(lengths.sum())
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lengths)
    cols = (splitmix64(seed, np.arange(nnz, dtype=np.uint64)) % np.uint64(max(n_cols, 1))).astype(np.int64)
    vals = values_for(letter, seed + 1, nnz)
    if shuffle:
        perm = np.argsort(splitmix64(seed + 2, np.arange(nnz, dtype=np.uint64)), kind="stable")
        rows, cols, vals = rows[perm], cols[perm], vals[perm]
    return n_rows, n_cols, (rows + base).astype(np.int32), (cols + base).astype(np.int32), vals


def banded_coo(n, half_width=16, letter="D", seed=2, base=0):
    """n x n band: row i has columns i-half_width .. i+half_width-1 clipped to [0, n)."""
    i = np.arange(n, dtype=np.int64)[:, None]
    c = i + np.arange(-half_width, half_width, dtype=np.int64)[None, :]
    ok = (c >= 0) & (c < n)
    rows = np.broadcast_to(i, c.shape)[ok]
    cols = c[ok]
    vals = values_for(letter, seed, rows.size)
    return n, n, (rows + base).astype(np.int32), (cols + base).astype(np.int32), vals


# ---- device-side generators for the BASELINE-size workloads (torch is plumbing) --------

def hell_uniform_on_device(n_rows, nnz_per_row, pattern="random", letter="D", hack_size=32, seed=1,
                           device="cuda:0", n_cols=None, band_offset=None, row_offset=0):
    """A HELL matrix with exactly `nnz_per_row` entries in every row, built directly in HBM
    (BASELINE config 2: 10 M rows x 32).  With a uniform row length L the HELL arrays of
    hell.c:46-104 are a dense [hacks][L][hack_size] block: slot(r, k) = (r // hs)*hs*L + r % hs + k*hs.

    pattern "banded": row i has columns i-L/2 .. i+L/2-1, wrapped into [0, n_cols)   
    pattern "random": L columns uniform over [0, n_cols), ascending within the row     
    pattern "window": like random but inside a window of 65 536 columns around the row (locality in between)

    Returns dict(letter, rows, cols, hack_size, nnz, cM, rP, hack_offsets, rS) of torch tensors.
    """
    import torch
    assert n_rows % hack_size == 0, "uniform generator wants whole hacks"
    n_cols = n_cols or n_rows
    hacks, L, hs = n_rows // hack_size, nnz_per_row, hack_size
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    tdt = {"S": torch.float32, "D": torch.float64, "C": torch.complex64, "Z": torch.complex128}[letter]
    rdt = {"S": torch.float32, "D": torch.float64, "C": torch.float32, "Z": torch.float64}[letter]

    # global row number of every slot; row_offset != 0 builds a row BLOCK of a larger matrix
    row = (torch.arange(hacks, device=device, dtype=torch.int64)[:, None, None] * hs
           + torch.arange(hs, device=device, dtype=torch.int64)[None, None, :] + row_offset)  # [hacks,1,hs]
    if pattern == "banded":
        k = torch.arange(L, device=device, dtype=torch.int64)[None, :, None]
        cols = (row + k - (L // 2 if band_offset is None else band_offset)) % n_cols
    elif pattern == "random":
        cols = torch.randint(0, n_cols, (hacks, L, hs), device=device, generator=gen, dtype=torch.int64)
        cols = torch.sort(cols, dim=1).values
    elif pattern.startswith("near"):
        # like "window" with a narrow window: columns random inside +-W of the row (pattern "near2048": W = 2048)
        w = min(2 * int(pattern[4:] or 2048), n_cols)
        cols = (row + torch.randint(-(w // 2), w // 2, (hacks, L, hs), device=device, generator=gen,
                                    dtype=torch.int64)) % n_cols
        cols = torch.sort(cols, dim=1).values
    elif pattern == "window":
        w = min(65536, n_cols)
        cols = (row + torch.randint(-(w // 2), w // 2, (hacks, L, hs), device=device, generator=gen,
                                    dtype=torch.int64)) % n_cols
        cols = torch.sort(cols, dim=1).values
    else:
        raise ValueError(pattern)
    rP = cols.to(torch.int32).reshape(-1).contiguous()
    del cols, row
    if letter in "SD":
        cM = torch.rand(hacks * L * hs, device=device, generator=gen, dtype=rdt)
    else:
        cM = torch.complex(torch.rand(hacks * L * hs, device=device, generator=gen, dtype=rdt),
                           torch.rand(hacks * L * hs, device=device, generator=gen, dtype=rdt)).to(tdt)
    hack_offsets = (torch.arange(hacks, device=device, dtype=torch.int64) * (hs * L)).to(torch.int32)
    rS = torch.full((n_rows,), L, device=device, dtype=torch.int32)
    return dict(letter=letter, rows=n_rows, cols=n_cols, hack_size=hs, nnz=n_rows * L, cM=cM, rP=rP,
                hack_offsets=hack_offsets, rS=rS, base=0, row_len=L)


def device_vector(n, letter="D", seed=3, device="cuda:0"):
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rdt = {"S": torch.float32, "D": torch.float64, "C": torch.float32, "Z": torch.float64}[letter]
    if letter in "SD":
        return torch.rand(n, device=device, generator=gen, dtype=rdt)
    return torch.complex(torch.rand(n, device=device, generator=gen, dtype=rdt),
                         torch.rand(n, device=device, generator=gen, dtype=rdt))


def hell_rows_to_host(h, first_row, n_rows):
    """Rows [first_row, first_row+n_rows) of a device HELL matrix from hell_uniform_on_device
    (first_row and n_rows multiples of hack_size) as a host HELL dict with rebased hackOffsets:
    the sub-range of hacks is a contiguous piece of cM / rP (cf. the reference's own chunk loop,
    hell_spmv_base.cuh:139-144)."""
    import numpy as np
    hs, L = h["hack_size"], h["row_len"]
    assert first_row % hs == 0 and n_rows % hs == 0
    s0, s1 = (first_row // hs) * hs * L, ((first_row + n_rows) // hs) * hs * L
    ho = h["hack_offsets"][first_row // hs:(first_row + n_rows) // hs].cpu().numpy().astype(np.int64) - s0
    return dict(letter=h["letter"], rows=n_rows, values=h["cM"][s0:s1].cpu().numpy(),
                indices=h["rP"][s0:s1].cpu().numpy(), hack_offsets=ho.astype(np.int32), hack_size=hs,
                row_lengths=h["rS"][first_row:first_row + n_rows].cpu().numpy(), base=h["base"])


def split_uniform_hell_by_columns(h, col_first, col_count):
    """Cut a uniform device HELL block (from hell_uniform_on_device) by column ownership into
    (own, rest): `own` keeps the entries whose column is in [col_first, col_first+col_count), with
    columns rebased to that block; `rest` keeps the others with global columns.  Entries keep their
    order within a row; both results are ragged HELL matrices built as hell.c:46-104 lays them out
    (per hack: hackSize * longest row slots, padding zero)."""
    import torch
    hs, L, rows = h["hack_size"], h["row_len"], h["rows"]
    hacks = rows // hs
    cols = h["rP"].view(hacks, L, hs).to(torch.int64)
    vals = h["cM"].view(hacks, L, hs)
    own_mask = (cols >= col_first) & (cols < col_first + col_count)

    def compact(mask, col_shift):
        lengths = mask.sum(dim=1)                                   # [hacks, hs]
        depth = lengths.max(dim=1).values                           # [hacks]
        hack_offsets = torch.cumsum(depth * hs, 0) - depth * hs     # exclusive prefix
        total = int((depth * hs).sum().item())
        pos = torch.cumsum(mask.to(torch.int64), dim=1) - 1         # k-position among kept entries
        lane = torch.arange(hs, device=cols.device, dtype=torch.int64)[None, None, :]
        dst = (hack_offsets[:, None, None] + lane + pos * hs)[mask]
        cM = torch.zeros(max(total, 1), dtype=vals.dtype, device=cols.device)
        rP = torch.zeros(max(total, 1), dtype=torch.int32, device=cols.device)
        cM[dst] = vals[mask]
        rP[dst] = (cols[mask] - col_shift).to(torch.int32)
        return dict(letter=h["letter"], rows=rows, cols=h["cols"], hack_size=hs, nnz=int(mask.sum().item()),
                    cM=cM, rP=rP, hack_offsets=hack_offsets.to(torch.int32),
                    rS=lengths.reshape(-1).to(torch.int32).contiguous(), base=0, slots=total)

    return compact(own_mask, col_first), compact(~own_mask, 0)


def hdia_laplacian7_on_device(m, letter="D", hack_size=32, device="cuda:0"):
    """HDIA arrays of the 7-point Laplacian on an m^3 grid (BASELINE config 4: m = 512), built in HBM.
    Equals what cooToHdia (hdia.cpp:230-349) produces from laplacian_3d_7pt(m) -- checked bit for bit on
    small m in tests/test_gpu_fullsize.py.  Needs m % hack_size == 0 so that a hack lies inside one grid
    line: then its diagonals are -m^2 (gz>0), -m (gy>0), -1, 0, +1, +m (gy<m-1), +m^2 (gz<m-1) and only the
    +-1 diagonals hold explicit zeros (at gx = 0 resp. gx = m-1)."""
    import torch
    assert m % hack_size == 0
    hs, n = hack_size, m * m * m
    hacks = n // hs
    tdt = {"S": torch.float32, "D": torch.float64}[letter]
    h = torch.arange(hacks, device=device, dtype=torch.int64)
    row0 = h * hs
    gx0, gy, gz = row0 % m, (row0 // m) % m, row0 // (m * m)
    cand = torch.tensor([-m * m, -m, -1, 0, 1, m, m * m], device=device, dtype=torch.int64)
    true = torch.ones(hacks, dtype=torch.bool, device=device)
    present = torch.stack([gz > 0, gy > 0, true, true, true, gy < m - 1, gz < m - 1], dim=1)   # [hacks, 7]
    counts = present.sum(dim=1)
    hack_offsets = torch.zeros(hacks + 1, dtype=torch.int64, device=device)
    hack_offsets[1:] = torch.cumsum(counts, 0)
    offsets = cand[None, :].expand(hacks, 7)[present]                                         # ascending per hack
    owner = torch.arange(hacks, device=device, dtype=torch.int64)[:, None].expand(hacks, 7)[present]
    H = int(offsets.numel())
    dM = torch.where(offsets == 0, 6.0, -1.0).to(tdt)[:, None].expand(H, hs).contiguous()
    # explicit zeros: left neighbour missing at gx == 0 (lane 0 of hacks starting a grid line),
    # right neighbour missing at gx == m-1 (last lane of hacks ending a grid line)
    left = (offsets == -1) & (gx0[owner] == 0)
    right = (offsets == 1) & (gx0[owner] + hs == m)
    dM[left, 0] = 0
    dM[right, hs - 1] = 0
    nnz = 7 * n - 6 * m * m
    return dict(letter=letter, rows=n, cols=n, hack_size=hs, height=H, nnz=nnz, dM=dM.reshape(-1),
                offsets=offsets.to(torch.int32), hack_offsets=hack_offsets.to(torch.int32))


def hell_ragged_on_device(lengths, n_cols, letter="S", hack_size=32, seed=5, device="cuda:0"):
    """HELL arrays for given row lengths (BASELINE config 3: power-law lengths), built in HBM.
    Every slot, padding included, gets a random valid column and coefficient: padding slots
    (k >= rS[row]) are never used by the kernels, exactly as with the reference's malloc'd HELL
    arrays (hellPerf.cpp:258-259)."""
    import torch
    hs = hack_size
    L = torch.as_tensor(lengths, dtype=torch.int64, device=device)
    rows = int(L.numel())
    hacks = (rows + hs - 1) // hs
    padded = torch.zeros(hacks * hs, dtype=torch.int64, device=device)
    padded[:rows] = L
    depth = padded.view(hacks, hs).max(dim=1).values
    hack_offsets = torch.cumsum(depth * hs, 0) - depth * hs
    slots = int((depth * hs).sum().item())
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rdt = {"S": torch.float32, "D": torch.float64}[letter]
    rP = torch.randint(0, n_cols, (max(slots, 1),), device=device, generator=gen, dtype=torch.int32)
    cM = torch.rand(max(slots, 1), device=device, generator=gen, dtype=rdt)
    return dict(letter=letter, rows=rows, cols=n_cols, hack_size=hs, nnz=int(L.sum().item()), cM=cM, rP=rP,
                hack_offsets=hack_offsets.to(torch.int32), rS=L.to(torch.int32), base=0, slots=slots,
                depth=depth)


def ell_ragged_on_device(lengths, n_cols, letter="S", seed=6, device="cuda:0"):
    """ELL arrays (pitch = rows rounded up to 32, max row length columns) for given row lengths, built in
    HBM; every slot random like hell_ragged_on_device.  Footprint = pitch * max_len * (sizeof(T)+4)."""
    import torch
    L = torch.as_tensor(lengths, dtype=torch.int32, device=device)
    rows = int(L.numel())
    pitch = (rows + 31) // 32 * 32
    max_len = int(L.max().item())
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rdt = {"S": torch.float32, "D": torch.float64}[letter]
    rP = torch.randint(0, n_cols, (pitch * max_len,), device=device, generator=gen, dtype=torch.int32)
    cM = torch.rand(pitch * max_len, device=device, generator=gen, dtype=rdt)
    return dict(letter=letter, rows=rows, cols=n_cols, pitch=pitch, max_row=max_len, nnz=int(L.sum().item()),
                cM=cM, rP=rP, rS=L, base=0)


def ragged_rows_to_host(h, first_row, n_rows):
    """Rows [first_row, first_row+n_rows) (hack-aligned) of a device HELL dict from hell_ragged_on_device
    as a host HELL dict with rebased hackOffsets."""
    import numpy as np
    hs = h["hack_size"]
    assert first_row % hs == 0 and n_rows % hs == 0
    h0, h1 = first_row // hs, (first_row + n_rows) // hs
    ho = h["hack_offsets"].cpu().numpy().astype(np.int64)
    s0 = int(ho[h0])
    s1 = int(ho[h1]) if h1 < ho.size else int(h["slots"])
    return dict(letter=h["letter"], rows=n_rows, values=h["cM"][s0:s1].cpu().numpy(), indices=h["rP"][s0:s1].cpu().numpy(),
                hack_offsets=(ho[h0:h1] - s0).astype(np.int32), hack_size=hs,
                row_lengths=h["rS"][first_row:first_row + n_rows].cpu().numpy(), base=0)


def ragged_coo_on_device(lengths, n_cols, pattern="near", near=2048, letter="D", seed=5, device="cuda:0"):
    """COO triplets in HBM, row-major, of a matrix with the given row lengths (the north_star target: power-law
    lengths, mean 32, max 2048).  A row's columns ascend and are distinct (stratified draw: the k-th of L entries lies
    in the k-th of L equal pieces of the row's column range):

    pattern "near":   the range is [row - near, row + near), wrapped into [0, n_cols) -- the locality of a mesh or a
                      banded problem, without consecutive columns
    pattern "random": the range is all of [0, n_cols)
    pattern "band":   consecutive columns centred on the row, [row - L/2, row + L - L/2) wrapped -- BASELINE
                      configs[1]'s band (i-16 .. i+15) with the row's own length

    Returns (rows int32, cols int32, vals) torch tensors; nnz = sum(lengths)."""
    import torch
    L = torch.as_tensor(np.asarray(lengths), dtype=torch.int64, device=device)
    n_rows = int(L.numel())
    start = torch.cumsum(L, 0) - L
    nnz = int(L.sum().item())
    row = torch.repeat_interleave(torch.arange(n_rows, device=device, dtype=torch.int64), L, output_size=nnz)
    k = torch.arange(nnz, device=device, dtype=torch.int64) - start[row]
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = torch.rand(nnz, device=device, generator=gen, dtype=torch.float64)
    length = L[row]
    if pattern == "near":
        width = 2 * near
        assert int(L.max().item()) <= width
        cols = (row - near + ((k.to(torch.float64) + u) * width / length.to(torch.float64)).to(torch.int64)) % n_cols
    elif pattern == "band":
        cols = (row - length // 2 + k) % n_cols
    elif pattern == "random":
        cols = torch.clamp(((k.to(torch.float64) + u) * n_cols / length.to(torch.float64)).to(torch.int64), max=n_cols - 1)
    else:
        raise ValueError(pattern)
    del u, k, length, start
    rdt = {"S": torch.float32, "D": torch.float64}[letter]
    vals = torch.rand(nnz, device=device, generator=gen, dtype=rdt)
    return row.to(torch.int32), cols.to(torch.int32), vals


def hell_rows_to_host_general(h, first_row, n_rows):
    """Rows [first_row, first_row + n_rows) (hack-aligned) of ANY device HELL dict (cM, rP, hack_offsets, rS, hack_size,
    slots) as a host HELL dict with rebased hackOffsets."""
    hs = h["hack_size"]
    assert first_row % hs == 0 and n_rows % hs == 0
    h0, h1 = first_row // hs, (first_row + n_rows) // hs
    ho = h["hack_offsets"].cpu().numpy().astype(np.int64)
    s0 = int(ho[h0])
    s1 = int(ho[h1]) if h1 < ho.size else int(h["slots"])
    return dict(letter=h["letter"], rows=n_rows, values=h["cM"][s0:s1].cpu().numpy(), indices=h["rP"][s0:s1].cpu().numpy(),
                hack_offsets=(ho[h0:h1] - s0).astype(np.int32), hack_size=hs,
                row_lengths=h["rS"][first_row:first_row + n_rows].cpu().numpy(), base=h.get("base", 0))
