#!/usr/bin/env python3
"""GPU box experiment: the ways the ELL/HELL SpMV fetches x (gathers / strips / LDS tile, spgpuSetSpmvForm) on column
patterns between "consecutive" and "scattered", and on power-law row lengths before and after ordering the rows by
length (spgpuOellOrderDevice).  Every timing is followed by an oracle check of three row windows.

    python tools/exp_tile.py [D|S] [rows] [cases: uniform,powerlaw]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402
import oracle_api as O  # noqa: E402

letter = sys.argv[1] if len(sys.argv) > 1 else "D"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
cases = (sys.argv[3] if len(sys.argv) > 3 else "uniform,powerlaw").split(",")
elem = {"S": 4, "D": 8}[letter]
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
one, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.0)
x = synth.device_vector(n, letter, 3)
z = torch.zeros(n, dtype=x.dtype, device="cuda:0")
xs = x.cpu().numpy()
FORMS = {"auto": 0, "gather": 1, "strips": 2, "tile0": 3, "tile1": 3, "tile2": 3, "tile3": 3}
# summation order of each form's kernel, for the oracle (tests/oracle_api.py spmv_tail)
GATHER_SHAPE = O.TAIL_SHAPE[letter]
TILE_SHAPES = {"D": {0: dict(group_rows=128, rows_per_lane=2, step=4, tail_lanes=16, phases=1),
                     1: dict(group_rows=32, rows_per_lane=2, step=8, tail_lanes=16, phases=4),
                     2: dict(group_rows=128, rows_per_lane=2, step=4, tail_lanes=16, phases=1),
                     3: dict(group_rows=128, rows_per_lane=2, step=4, tail_lanes=16, phases=1)},
               "S": {0: dict(group_rows=256, rows_per_lane=4, step=4, tail_lanes=16, phases=1),
                     1: dict(group_rows=32, rows_per_lane=4, step=16, tail_lanes=16, phases=8),
                     2: dict(group_rows=256, rows_per_lane=4, step=4, tail_lanes=16, phases=1),
                     3: dict(group_rows=256, rows_per_lane=4, step=4, tail_lanes=16, phases=1)}}[letter]


def check(h, form, windows=3, rows=2048):
    shape = TILE_SHAPES[int(form[4:])] if form.startswith("tile") else GATHER_SHAPE
    step = max(1, (h["rows"] - rows) // max(windows - 1, 1))
    for w in range(windows):
        first = min(w * step, h["rows"] - rows) // 2048 * 2048
        sub = synth.hell_rows_to_host_general(h, first, rows)
        want = O.spmv_tail(sub, xs, None, 1.0, 0.0, **shape)
        if h.get("rIdx") is not None:
            got = z[h["rIdx"][first:first + rows].to(torch.int64)].cpu().numpy()
        else:
            got = z[first:first + rows].cpu().numpy()
        if got.tobytes() != want.tobytes():
            bad = np.flatnonzero(got != want)
            return f"MISMATCH rows {first}+{bad[:4].tolist()} ({bad.size} of {rows}): got {got[bad[0]]!r} want {want[bad[0]]!r}"
    return "bit-exact"


def run(h, label, forms):
    rows = h["rows"]
    hacks = (rows + 31) // 32
    alg = h["nnz"] * (elem + 4) + rows * (4 + elem) + n * elem + hacks * 4 + (rows * 4 if h.get("rIdx") is not None else 0)
    call = lambda: capi.hellspmv[letter](handle, p(z), None, one, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                         p(h.get("rIdx")), 32, rows, p(x), zero, 0)
    for form in forms:
        os.environ["SPGPU_X_TILE_SHAPE"] = form[4:] if form.startswith("tile") else "0"
        capi.spgpuTuningReload()
        capi.spgpuSetSpmvForm(handle, FORMS[form])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            for _ in range(3):
                call()
            a.record(stream)
            for _ in range(20):
                call()
            b.record(stream)
        b.synchronize()
        t = a.elapsed_time(b) / 20
        z.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            call()
        stream.synchronize()
        print(f"{letter} {label:46s} {form:7s} slots/nnz {h['slots'] / h['nnz']:.3f}  {t:.4f} ms  {alg / t * 1e-6:7.1f} GB/s  "
              f"{alg / t * 1e-6 / 8000:.3f} of 8 TB/s  {check(h, form)}", flush=True)
    capi.spgpuSetSpmvForm(handle, 0)


if "uniform" in cases:
    for pattern in ("near2048", "near512", "window", "banded"):
        h = synth.hell_uniform_on_device(n // 32 * 32, 32, pattern, letter, 32, seed=1)
        h["slots"] = h["nnz"]
        torch.cuda.synchronize()
        run(h, f"uniform 32/row, columns {pattern}", ["gather", "strips", "tile0", "tile1", "tile2", "tile3"])
        del h
        torch.cuda.empty_cache()

if "powerlaw" in cases:
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    for pattern in ("near", "random"):
        rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, 2048, letter, seed=5)
        torch.cuda.synchronize()
        orders = [("plain", None), ("sorted all", (0, 0))]
        if pattern == "near":
            orders += [(f"sorted window {w} long>{t}", (w, t)) for w, t in ((4096, 0), (2048, 256), (4096, 256), (4096, 128), (8192, 256), (16384, 0))]
        for name, order in orders:
            h = formats.coo_to_ordered_hell_device(handle, n, rows_t, cols_t, vals_t, letter, 32, *(order or (0, 0)),
                                                   order=order is not None)
            forms = ["gather"] if (pattern == "random" or order is None or order == (0, 0)) else ["gather", "tile0", "tile1", "tile2"]
            run(h, f"power-law {pattern}, {name}", forms)
            del h
            torch.cuda.empty_cache()
        del rows_t, cols_t, vals_t
        torch.cuda.empty_cache()
capi.spgpuDestroy(handle)
