#!/usr/bin/env python3
"""GPU box: the other BASELINE.json configurations (bench.py carries configs[1] and the SpMM shard).
   C3  HELL fp32 vs ELL fp32, power-law row lengths (max 2048, mean ~32), random columns
   C4  HDIA fp64, 7-point Laplacian 512^3
One JSON line per measurement; each result is spot-checked against the oracle on a row window.
Usage: python tools/bench_configs.py [c3] [c4] [--rows N] [--ell-rows N] [--grid M]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402
import oracle_api as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="*", default=["c3", "c4"])
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--ell-rows", type=int, default=2_000_000)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()

handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
PEAK = 8000.0


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        fn(); fn()
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def report(name, t, nnz, alg, extra):
    print(json.dumps(dict(config=name, ms=round(t * 1e3, 4), gflops=round(2.0 * nnz / t * 1e-9, 1),
                          hbm_gbs=round(alg / t * 1e-9, 1), frac_of_8TBs=round(alg / t * 1e-9 / PEAK, 4), **extra)), flush=True)


if "c3" in args.which:
    n = args.rows // 32 * 32
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    h = synth.hell_ragged_on_device(lengths, n, "S", 32, seed=5)
    x, z = synth.device_vector(n, "S", 3), torch.empty(n, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    call = lambda: capi.hellspmv["S"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                      None, 32, n, p(x), 0.0, 0)
    t = timed(call, args.reps)
    sub = synth.ragged_rows_to_host(h, 0, 4096)
    want = O.default_spmv(sub, x.cpu().numpy(), None, 1.0, 0.0)
    parity = "bit-exact vs oracle on 4096 rows" if z[:4096].cpu().numpy().tobytes() == want.tobytes() else "MISMATCH"
    alg = h["nnz"] * 8 + n * (4 + 4) + n * 4 + (n // 32) * 4
    hell_bytes = h["slots"] * 8 + n * 4 + (n // 32) * 4
    ell_bytes_full = ((n + 31) // 32 * 32) * int(lengths.max()) * 8 + n * 4
    report("C3 HELL fp32 power-law", t, h["nnz"], alg,
           dict(rows=n, nnz=h["nnz"], mean_len=round(h["nnz"] / n, 2), max_len=int(lengths.max()), parity=parity,
                hell_footprint_GB=round(hell_bytes * 1e-9, 2), ell_footprint_GB_same_rows=round(ell_bytes_full * 1e-9, 2),
                padding_ratio=round(h["slots"] / h["nnz"], 3)))
    del h
    torch.cuda.empty_cache()
    # ELL needs pitch*2048*8 B: 164 GB at 10 M rows (fits the 288 GB of one MI355X); --ell-rows scales it down,
    # and an allocation failure falls back to a fifth of the rows
    ne = min(args.ell_rows, n) // 32 * 32
    le = lengths[:ne]
    try:
        e = synth.ell_ragged_on_device(le, ne, "S", seed=6)
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        ne = ne // 5 // 32 * 32
        le = lengths[:ne]
        e = synth.ell_ragged_on_device(le, ne, "S", seed=6)
    he = synth.hell_ragged_on_device(le, ne, "S", 32, seed=5)
    xe, ze = synth.device_vector(ne, "S", 3), torch.empty(ne, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    call_e = lambda: capi.ellspmv["S"](handle, p(ze), None, 1.0, p(e["cM"]), p(e["rP"]), e["pitch"], e["pitch"], p(e["rS"]),
                                       None, 32, e["max_row"], ne, p(xe), 0.0, 0)
    call_h = lambda: capi.hellspmv["S"](handle, p(ze), None, 1.0, p(he["cM"]), p(he["rP"]), 32, p(he["hack_offsets"]),
                                        p(he["rS"]), None, 32, ne, p(xe), 0.0, 0)
    te, th = timed(call_e, args.reps), timed(call_h, args.reps)
    alg_e = e["nnz"] * 8 + ne * 8 + ne * 4
    common = dict(rows=ne, nnz=e["nnz"], max_len=e["max_row"])
    report(f"C3 ELL fp32 power-law ({ne} rows)", te, e["nnz"], alg_e,
           dict(**common, footprint_GB=round((e["pitch"] * e["max_row"] * 8 + ne * 4) * 1e-9, 2)))
    report(f"C3 HELL fp32 power-law (same {ne} rows)", th, he["nnz"], alg_e + (ne // 32) * 4,
           dict(**common, footprint_GB=round((he["slots"] * 8 + ne * 4 + ne // 32 * 4) * 1e-9, 2)))
    del e, he
    torch.cuda.empty_cache()

if "c4" in args.which:
    m = args.grid
    d = synth.hdia_laplacian7_on_device(m, "D", 32)
    n = d["rows"]
    x, y = synth.device_vector(n, "D", 3), synth.device_vector(n, "D", 4)
    z = torch.empty_like(y)
    torch.cuda.synchronize()
    hacks = n // 32
    for beta in (0.0, 0.5):
        call = lambda: capi.hdiaspmv["D"](handle, p(z), p(y), 1.0, p(d["dM"]), p(d["offsets"]), 32, p(d["hack_offsets"]),
                                          n, n, p(x), beta)
        t = timed(call, args.reps)
        S = 32 * d["height"]   # every stored slot of this matrix has its column in range except two
        alg = S * 8 + d["height"] * 4 + (hacks + 1) * 4 + n * 8 + n * 8 * (2 if beta else 1)
        # parity on a window of hacks in the middle of the grid
        first = (hacks // 2) * 32
        ho = d["hack_offsets"][hacks // 2: hacks // 2 + 65].cpu().numpy().astype(np.int64)
        sub = dict(letter="D", rows=2048, cols=n, hack_size=32, hack_offsets=(ho - ho[0]).astype(np.int32),
                   offsets=(d["offsets"][ho[0]:ho[-1]].cpu().numpy().astype(np.int64) + first).astype(np.int32),
                   values=d["dM"][ho[0] * 32: ho[-1] * 32].cpu().numpy())
        # the oracle indexes x by offsets[d] + local row: shift x instead of the offsets
        sub["offsets"] = d["offsets"][ho[0]:ho[-1]].cpu().numpy()
        xs = x.cpu().numpy()
        xw = np.zeros(n + 2 * m * m + 4096)
        lo = first - m * m
        xw[:] = 0.0
        seg = xs[max(lo, 0): min(first + 2048 + m * m, n)]
        xw[max(lo, 0) - lo: max(lo, 0) - lo + seg.size] = seg
        sub_shift = dict(sub)
        sub_shift["offsets"] = (sub["offsets"].astype(np.int64) + m * m).astype(np.int32)   # local col = off + i + m^2 >= 0
        sub_shift["cols"] = xw.size
        ys = y[first:first + 2048].cpu().numpy()
        want = O.hdia_spmv(sub_shift, xw, ys if beta else None, 1.0, beta)
        parity = "bit-exact vs oracle on 2048 rows" if z[first:first + 2048].cpu().numpy().tobytes() == want.tobytes() else "MISMATCH"
        report(f"C4 HDIA fp64 7-pt Laplacian {m}^3 beta={beta}", t, d["nnz"], alg,
               dict(rows=n, nnz=d["nnz"], stored_diagonals=d["height"], dM_GB=round(S * 8e-9, 2), parity=parity))
