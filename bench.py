#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json measured on MI355X.

    python bench.py --gpus 1 --steps K --warmup W          # headline: HELL fp64 SpMV (configs[1])
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # row-sharded SpMM

A "step" is one pass of the hot path over the synthetic input, inputs resident in HBM:
  N == 1 : one spgpuDhellspmv on BASELINE configs[1] (10 M rows, 32 nnz/row uniform, hackSize 32, fp64)
  N  > 1 : one row-sharded HELL fp64 SpMM step (A x 16 rhs, 5 M rows per GPU; all-gather of X over RCCL,
           then spgpuDhellspmm on the local row block) -- the only part of the path that shards
           (north_star: "single-vector SpMV stays single-GPU").
Rank 0 prints ONE JSON line.  metric = GFLOP/s (2*nnz per SpMV, 64-bit; the reference's harness uses
2*nnz-1, hellPerf.cpp:316); roofline = algorithmic HBM bytes / measured kernel time
against 8 TB/s; cpu_baseline = the oracle (a port: the reference has no CPU SpMV) timed on this box's
host cores on a bounded sample.  Only that leg and the parity spot-check touch oracle/.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--rows", type=int, default=10_000_000, help="rows of the SpMV workload (configs[1]: 10 M)")
    p.add_argument("--nnz-per-row", type=int, default=32)
    p.add_argument("--pattern", default="banded", choices=["banded", "random", "window"],
                   help="column pattern of the headline workload (SURVEY 8(d) C2 defines banded and random)")
    p.add_argument("--no-extras", action="store_true", help="skip the untimed extra variants and the CPU baseline")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work budget of the cpu_baseline leg")
    p.add_argument("--workload", default="auto", choices=["auto", "spmv", "spmm"],
                   help="auto: spmv on 1 GPU (the headline metric), row-sharded spmm on more")
    p.add_argument("--ell-rows", type=int, default=2_000_000, help="rows of the ELL side of configs[2] (ELL at 10 M rows is 164 GB)")
    p.add_argument("--grid", type=int, default=512, help="grid edge of configs[3] (HDIA 7-point Laplacian grid^3)")
    p.add_argument("--spmm-rows-per-gpu", type=int, default=5_000_000)
    p.add_argument("--rhs", type=int, default=16)
    p.add_argument("--spmm-pattern", default="banded", choices=["banded", "random", "window"])
    p.add_argument("--force-split", action="store_true", help="spmm: cut by column ownership even on one rank (rehearsal)")
    p.add_argument("--exchange", default="allgather", choices=["allgather", "needed"],
                   help="spmm, N > 1: what `value` reports.  allgather (default): RCCL all-gather of the dense X per step, BASELINE "
                        "configs[4] word for word.  needed: only the X rows the off-block columns name (all_to_all).  The other "
                        "one is measured in the same process and reported under `variants`")
    p.add_argument("--driver", default="c", choices=["c", "torch"],
                   help="spmm: who issues a step -- libspgpu.so's sharded driver (RCCL through dlopen), or torch.distributed")
    p.add_argument("--placements", type=int, default=5,
                   help="north_star target: how many placements of the matrix' arrays (allocations of their own, one hipMalloc each) are timed")
    p.add_argument("--frozen", action="store_true",
                   help="spmv: the headline matrix is frozen first (spgpuHellSpmvFreeze: 16-bit column indices); the record says so.  "
                        "Default off: `value` is the plain call; the frozen call is reported beside it as `headline_frozen`")
    p.add_argument("--no-split", action="store_true",
                   help="spmm: do not cut the local block by column ownership (no compute/all-gather overlap)")
    return p.parse_args()


def hell_algorithmic_bytes(nnz, rows, cols, hacks, elem=8, beta_nonzero=False, rhs=1):
    """padding and cache-line over-fetch do not count; x counted once."""
    matrix = nnz * (elem + 4) + rows * 4 + hacks * 4
    vectors = cols * elem + rows * elem * (2 if beta_nonzero else 1)
    return matrix + rhs * vectors


def committed_traffic(rows, nnz_per_row, pattern, rhs=None):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE, separate runs, gfx950 x2 correction on FETCH_SIZE), if they were taken on
    this workload; None otherwise.  bench.py itself cannot collect counters.  rhs: the SpMM workload."""
    import glob
    best = None
    kind = "spmv" if rhs is None else "spmm"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_bench_{kind}_pmc.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        w = d.get("workload", dict(rows=10_000_000, nnz_per_row=32, pattern="banded"))
        if (w.get("rows"), w.get("nnz_per_row"), w.get("pattern"), w.get("rhs")) == (rows, nnz_per_row, pattern, rhs):
            best = int(d["hbm_traffic_bytes_per_launch"])
    return best


def live_traffic(args, kernel_words=("SpmvKernel",), extra=()):
    """HBM bytes per launch of the headline kernel, counted in THIS run: two child processes of this very file (--no-extras, 10
    steps) under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` -- separate passes, the counters
    corrected as MI355X_MICROARCH.md's HBM section says (gfx950: FETCH_SIZE reports half the bytes of wide streaming reads; KiB
    units).  The children are started as ordinary child processes with python3 itself behind `--` (no exec from a process that holds
    the GPU, no shell hop).  Returns (bytes, detail) or (None, reason): the caller then falls back to the committed summary."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="spgpu_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", SPGPU_BENCH_INNER="1")
    seen = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter.lower())
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", counter.lower(), "--", "python3",
                   os.path.abspath(__file__), "--no-extras", "--steps", "10", "--warmup", "2", "--rows", str(args.rows),
                   "--nnz-per-row", str(args.nnz_per_row), "--pattern", args.pattern, *extra]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)
            except subprocess.TimeoutExpired:
                return None, f"{counter} pass timed out"
            per_kernel = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") == counter and all(w in row["Kernel_Name"] for w in kernel_words):
                        per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not per_kernel:
                return None, f"{counter} pass: no dispatch of the kernel in the counter file (rc {r.returncode}: {r.stderr[-200:]!r})"
            name, values = max(per_kernel.items(), key=lambda kv: len(kv[1]))   # the kernel of the timed launches: the most dispatches
            seen[counter] = (name, sum(values) / len(values), len(values))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    read_bytes, write_bytes = 2.0 * seen["FETCH_SIZE"][1] * 1024, seen["WRITE_SIZE"][1] * 1024
    return int(read_bytes + write_bytes), dict(
        source="live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace, separate child runs of this bench.py (--no-extras --steps 10) "
               "during this run; read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 correction), write bytes = WRITE_SIZE KiB x 1024",
        kernel=seen["FETCH_SIZE"][0][:160], dispatches_counted=seen["FETCH_SIZE"][2], hbm_read_bytes=int(read_bytes), hbm_write_bytes=int(write_bytes))


def vendor():
    """tools/vendor_context.py (rocSPARSE through ctypes), or None: same-node context numbers, never on the path."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import vendor_context
        return vendor_context
    except Exception:  # noqa: BLE001 - context only
        return None


FORM_NAMES = {1: "gathers", 2: "strip x loads", 3: "x tile in LDS", 4: "sweep"}


def settle(stream, step, calls=3):
    """AUTO reads what the sample wavefronts of a COMPLETED launch on the same arrays reported (include/spgpu/tuning.h): a
    matrix that was just built -- possibly at the address of the previous one -- needs a completed call or two before
    its form is the one that is timed.  (Round 2 timed the window variant in the form of the banded matrix that had
    lived at that address: +8 %.)"""
    for _ in range(calls):
        step()
        stream.synchronize()


def form_ran(handle):
    from spgpu_amd import capi
    return FORM_NAMES.get(capi.spgpuGetLastSpmvForm(handle), "?")


def time_launches(stream, fn, steps):
    """HIP events on the stream the kernels are launched on; returns seconds for `steps` launches."""
    import torch
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        start.record(stream)
        for _ in range(steps):
            fn()
        stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3


def spot_check_hell(h, x, y, z, alpha, beta, rows_per_probe=2048):
    """Parity at full size: three hack-aligned row windows of the device result against the oracle."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    from spgpu_amd import synth
    xs = x.cpu().numpy()
    n = h["rows"]
    for first in (0, (n // 2) // h["hack_size"] * h["hack_size"], n - rows_per_probe):
        sub = synth.hell_rows_to_host(h, first, rows_per_probe)
        ys = y[first:first + rows_per_probe].cpu().numpy() if beta != 0 else None
        want = O.default_spmv(sub, xs, ys, alpha, beta)
        got = z[first:first + rows_per_probe].cpu().numpy()
        if got.tobytes() != want.tobytes():
            return f"MISMATCH in rows [{first},{first + rows_per_probe})"
    return "bit-exact vs oracle on 3 x %d rows" % rows_per_probe


def usable_cores():
    """Host cores this process may really use: cgroup CPU quota if there is one, else the affinity
    mask; the GPU boxes expose 128 hardware threads but give a 1-GPU job a 16-CPU share."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    if "OMP_NUM_THREADS" in os.environ:
        n = int(os.environ["OMP_NUM_THREADS"])
    elif n > 32:
        n = 16
    return n


def cpu_baseline(h, x, seconds):
    """cpu_baseline leg, on a bounded sample of the same matrix:
      SpMV        the oracle's orc_dhellspmv (C, OpenMP over rows; a PORT -- the reference has no CPU SpMV), on ALL the
                  cores this job may use and on ONE core;
      conversion  what the reference's CPU path really is: cooToEll -> ellToHell and cooToHdia on one thread
                  (ell.c:39-80, hell.c:46-104, hdia.cpp:230-324).  Timed on the reference's own objects where they
                  travelled with the tree (oracle/_ref, kind "reference"), else on the oracle's byte-identical restatement.
    value/cores are the all-cores SpMV figure."""
    import numpy as np
    import oracle_api as O
    from spgpu_amd import synth
    sample_rows = min(h["rows"], 2_000_000) // h["hack_size"] * h["hack_size"]
    sub = synth.hell_rows_to_host(h, 0, sample_rows)
    xs = np.ascontiguousarray(x.cpu().numpy())
    zs = np.zeros(sample_rows)
    nnz = int(sub["row_lengths"].sum(dtype=np.int64))
    cores = usable_cores()
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    call = lambda: O.orc.orc_dhellspmv(ptr(zs), None, C.c_double(1.0), ptr(sub["values"]), ptr(sub["indices"]),
                                       sub["hack_size"], ptr(sub["hack_offsets"]), ptr(sub["row_lengths"]), None,
                                       sample_rows, ptr(xs), C.c_double(0.0), 0, 1)

    def spmv_rate(threads, budget):
        O.orc.orc_set_threads(threads)
        call()  # warm the pages
        t0, passes = time.perf_counter(), 0
        while True:
            call()
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget or passes >= 1000:
                break
        return round(2.0 * nnz * passes / dt * 1e-9, 3), passes, dt

    all_rate, passes, dt = spmv_rate(cores, seconds * 0.35)
    one_rate, passes1, dt1 = spmv_rate(1, seconds * 0.35)
    O.orc.orc_set_threads(cores)

    # conversion: COO of the first rows of the same matrix (row-major triplets rebuilt from the HELL sample)
    conv_rows = min(sample_rows, 500_000)
    hs, L = sub["hack_size"], int(sub["row_lengths"][0])
    vals = sub["values"][:conv_rows * L].reshape(conv_rows // hs, L, hs).transpose(0, 2, 1).reshape(-1)
    cols = sub["indices"][:conv_rows * L].reshape(conv_rows // hs, L, hs).transpose(0, 2, 1).reshape(-1)
    rows = np.repeat(np.arange(conv_rows, dtype=np.int32), L)
    vals, cols = np.ascontiguousarray(vals), np.ascontiguousarray(cols, np.int32)
    conv = O.reference_converters() if O.reference_available() else None
    kind = "reference" if conv is not None else "port"
    conv = conv or O.oracle_converters
    t0 = time.perf_counter()
    ell = conv.coo_to_ell(conv_rows, rows, cols, vals)
    hell = conv.ell_to_hell(ell, hs)
    t_hell = time.perf_counter() - t0
    assert hell["values"].tobytes() == sub["values"][:conv_rows * L].tobytes(), "converter output differs from the device-built matrix"
    n_cols = int(xs.size)
    t0 = time.perf_counter()
    conv.coo_to_hdia(conv_rows, n_cols, rows, cols, vals, hs)
    t_hdia = time.perf_counter() - t0
    return dict(value=all_rate, unit="GFLOP/s", cores=cores, kind="port",
                sample=f"oracle orc_dhellspmv (C, OpenMP, {cores} threads) on the first {sample_rows} rows "
                       f"({nnz} nnz) of the same matrix, {passes} passes in {dt:.1f} s",
                all_cores=dict(gflops=all_rate, threads=cores), one_core=dict(gflops=one_rate, threads=1, passes=passes1, seconds=round(dt1, 2)),
                convert_single_thread_s=dict(kind=kind, rows=conv_rows, nnz=int(rows.size),
                                             coo_to_ell_to_hell=round(t_hell, 3), coo_to_hdia=round(t_hdia, 3),
                                             nnz_per_s_hell=round(rows.size / t_hell), nnz_per_s_hdia=round(rows.size / t_hdia),
                                             note="reference ell.c:39-80 + hell.c:46-104 and hdia.cpp:230-324, one thread, as the reference runs them"))


class OwnAllocations:
    """The device arrays of a matrix (and x, z) copied into allocations of their own, one hipMalloc each -- how a C caller of the
    reference holds them (hellPerf.cpp:176-190: one cudaMalloc per array).  The arrays that come out of this harness' conversion
    pipeline sit inside blocks of torch's caching allocator instead, and WHERE the arrays of a matrix lie moves the time of a
    memory-bound kernel on this hardware by several per cent (tools/exp_alloc.py: the same kernel on the same data in one
    process, 0.68 ... 0.76 ms from one set of allocations to the next) -- so the target is timed on
    several such placements and the record carries all of them."""

    def __init__(self, tensors):
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = {}
        for name, t in tensors.items():
            p, size = C.c_void_p(), t.numel() * t.element_size()
            if self.hip.hipMalloc(C.byref(p), C.c_size_t(size)) != 0 or self.hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), C.c_size_t(size), 3) != 0:
                self.free()
                raise MemoryError(f"hipMalloc / hipMemcpy of {size} bytes for {name}")
            self.ptr[name] = p

    def __getitem__(self, name):
        return self.ptr[name]

    def free(self):
        for p in self.ptr.values():
            self.hip.hipFree(p)
        self.ptr = {}


def timed_blocks(stream, call, blocks=3, launches=20):
    """ms per launch of `blocks` timed blocks of `launches` back-to-back launches, after the settling calls"""
    with __import__("torch").cuda.stream(stream):
        settle(stream, call, 4)
    return [time_launches(stream, call, launches) / launches * 1e3 for _ in range(blocks)]


def spread(values):
    v = sorted(values)
    return dict(min=round(v[0], 4), median=round(v[len(v) // 2], 4), max=round(v[-1], 4))


def check_windows(h, x, z, letter, shape, windows=3, rows=2048):
    """Parity at full size for ANY device HELL dict (ragged, ordered through rIdx or not): `windows` hack-aligned row
    windows of the device result against the oracle run in the kernel's summation order (`shape`: the spmv_tail
    parameters, tests/oracle_api.py slab_shape), bit for bit."""
    import torch
    import oracle_api as O
    from spgpu_amd import synth
    xs = x.cpu().numpy()
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
            return f"MISMATCH in ordered rows [{first},{first + rows})"
    return f"bit-exact vs oracle on {windows} x {rows} rows"


def bench_powerlaw(args, handle, stream, dev, rows):
    """The north_star target: HELL fp64 on power-law row lengths (mean 32, max 2048), as the rows come and after the
    device-side ordering by length (spgpuOellOrderDevice: windows of 2048 rows, rows longer than 256 set aside; the
    COO route through spgpuCooToHellDevice); columns near the row (+-2048, not consecutive: every row, however short,
    spreads over 4 096 columns), as a band (consecutive columns centred on the row, configs[1]'s pattern with the
    row's own length) and scattered."""
    import torch
    import oracle_api as O
    from spgpu_amd import capi, formats, synth
    letter, elem = "D", 8
    lengths = synth.power_law_lengths(rows, mean=32.0, max_len=2048, seed=5)
    x = synth.device_vector(rows, letter, 3, dev)
    z = torch.zeros(rows, dtype=torch.float64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    out = dict(rows=rows, mean_len=round(float(lengths.mean()), 2), max_len=int(lengths.max()),
               order="spgpuOellOrderDevice(window=2048, longRows=256) + spgpuCooPermuteRowsDevice + spgpuCooToHellDevice",
               timing="ordered layouts: ms = median over `placements` (the matrix' arrays, x and z in allocations of their own, one hipMalloc "
                      "each as hellPerf.cpp:176-190 holds them; 3 timed blocks of 20 launches per placement, each block listed) -- "
                      "as_built_ms: the arrays as this harness' pipeline leaves them inside torch's allocator; the placement of the arrays "
                      "moves a memory-bound kernel by several per cent on this hardware (tools/exp_alloc.py)")
    # the first ordering of a process pays rocPRIM's and the allocator's cold start (1.2 s measured): not the build time
    warm = synth.ragged_coo_on_device(lengths[:4096], 4096, "near", 2048, letter, seed=5, device=dev)
    formats.coo_to_ordered_hell_device(handle, 4096, *warm, letter, 32, 2048, 256, order=True)
    del warm
    for pattern in ("near", "band", "random"):
        coo = synth.ragged_coo_on_device(lengths, rows, pattern, 2048, letter, seed=5, device=dev)
        torch.cuda.synchronize()
        if vendor() is not None:   # rocSPARSE CSR (adaptive) on the same rows, as they come: context, never on the path
            try:
                out[f"{pattern}_vendor_context"] = vendor().coo_context(stream, rows, rows, lengths, coo[1], coo[2], x, torch.empty_like(z))
            except Exception as error:  # noqa: BLE001
                out[f"{pattern}_vendor_context"] = repr(error)
        # "sorted_aligned": spgpuOellOrderAlignedDevice -- every window of the order is one 2 048-row workgroup (include/spgpu/ell_conv.h)
        # "sorted_global": the reference's own order -- ellToOell, ONE sort of all rows by length (ell.c:85-202, hellPerf.cpp:333-378)
        # "*_frozen": the same arrays after spgpuHellSpmvFreeze (include/spgpu/tuning.h: the caller promises that the index arrays
        # stay as they are; the library keeps a 16-bit copy of the column indices with the matrix' plan) -- the same ABI call,
        # 10 instead of 12 bytes per nonzero streamed; `frac` is still ALGORITHMIC bytes (12 per nonzero) / time / peak
        # "plain_adopted": the rows as they come, NO rIdx in the call -- after spgpuHellSpmvAdopt (the caller's promise not to touch any of
        # the matrix' arrays: the library keeps an ordered, frozen copy of its own and runs the call on it; z in the caller's row order)
        z_aligned = None
        for name, ordered in ({"band": (("plain", False), ("sorted_global", True), ("sorted", True), ("sorted_aligned", True), ("sorted_aligned_frozen", True),
                                        ("plain_adopted", False)),
                               "near": (("plain", False), ("sorted_global", True), ("sorted", True), ("sorted_aligned", True), ("sorted_aligned_frozen", True),
                                        ("plain_adopted", False)),
                               "random": (("plain", False), ("sorted", True))}[pattern]):
            t0 = time.perf_counter()
            window, long_rows = (0, 0) if name == "sorted_global" else (2048, 256)
            frozen = name.endswith("_frozen")
            adopted = name.endswith("_adopted")
            h = formats.coo_to_ordered_hell_device(handle, rows, *coo, letter, 32, window, long_rows, order=ordered, aligned="aligned" in name)
            build_s = time.perf_counter() - t0
            freeze = lambda a: capi.spgpuHellSpmvFreeze(handle, capi.TYPE_CODE[letter], a["cM"], a["rP"], 32, a["hack_offsets"], a["rS"], a["rIdx"], rows, 0)
            if frozen:
                t0 = time.perf_counter()
                said = freeze({k: p(h[k]) for k in ("cM", "rP", "hack_offsets", "rS", "rIdx")})
                torch.cuda.synchronize()
                freeze_s, frozen_bytes = time.perf_counter() - t0, capi.spgpuSpmvFrozenBytes(handle)
                assert said == capi.SPGPU_SUCCESS, said
            adopt = lambda a: capi.spgpuHellSpmvAdopt(handle, capi.TYPE_CODE[letter], a["cM"], a["rP"], 32, a["hack_offsets"], a["rS"], rows, 0)
            if adopted:
                t0 = time.perf_counter()
                said = adopt({k: p(h[k]) for k in ("cM", "rP", "hack_offsets", "rS")})
                torch.cuda.synchronize()
                adopt_s, adopted_bytes = time.perf_counter() - t0, capi.spgpuSpmvFrozenBytes(handle)
                assert said == capi.SPGPU_SUCCESS, said
            # form AUTO throughout: through rIdx the tile form falls back to gathers column by column, and on scattered
            # columns it runs within 1 % of the plain gather form (tools/exp_tile.py, ragged0 vs raggedg)
            capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
            call = lambda: capi.hellspmv[letter](handle, p(z), None, C.c_double(1.0), p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]),
                                                 p(h["rS"]), p(h["rIdx"]), 32, rows, p(x), C.c_double(0.0), 0)
            with torch.cuda.stream(stream):
                settle(stream, call, 4)
            as_built = [time_launches(stream, call, 20) / 20 * 1e3 for _ in range(3)]
            t = sorted(as_built)[1] * 1e-3
            placements = None
            if (ordered or adopted) and pattern != "random" and name != "sorted_global" and args.placements > 0:
                placements = []
                for _ in range(args.placements):
                    arrays = dict(cM=h["cM"], rP=h["rP"], hack_offsets=h["hack_offsets"], rS=h["rS"], x=x, z=z)
                    if h["rIdx"] is not None:
                        arrays["rIdx"] = h["rIdx"]
                    own = OwnAllocations(arrays)
                    try:
                        torch.cuda.synchronize()
                        if frozen:
                            assert freeze(own) == capi.SPGPU_SUCCESS
                        if adopted:
                            assert adopt(own) == capi.SPGPU_SUCCESS
                        placed = lambda own=own: capi.hellspmv[letter](handle, own["z"], None, C.c_double(1.0), own["cM"], own["rP"], 32, own["hack_offsets"],
                                                                       own["rS"], own.ptr.get("rIdx"), 32, rows, own["x"], C.c_double(0.0), 0)
                        placements.append(timed_blocks(stream, placed))
                    finally:
                        torch.cuda.synchronize()
                        if frozen or adopted:
                            capi.spgpuSpmvThaw(handle, own["rP"])
                        own.free()
                t = sorted(sorted(blocks)[len(blocks) // 2] for blocks in placements)[len(placements) // 2] * 1e-3
            z.zero_()
            torch.cuda.synchronize()
            time_launches(stream, call, 1)
            ran = form_ran(handle)
            capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
            hacks = (rows + 31) // 32
            alg = h["nnz"] * (elem + 4) + rows * (4 + elem) + rows * elem + hacks * 4 + (rows * 4 if ordered else 0)
            shape = O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP) if ordered else O.slab_shape(letter)
            if name == "sorted_aligned":
                z_aligned = z.clone()
            if adopted:   # z in the caller's row order, the ordered kernel's sums: the explicitly ordered call's bits
                same = z_aligned is not None and bool(torch.equal(z.view(torch.int64), z_aligned.view(torch.int64)))
                adopted_parity = ("bit-identical to the explicitly ordered call (sorted_aligned), itself " + out[f"{pattern}_sorted_aligned"]["parity"]) if same \
                    else "MISMATCH vs the explicitly ordered call"
            out[f"{pattern}_{name}"] = dict(slots_per_nnz=round(h["slots"] / h["nnz"], 3), ms=round(t * 1e3, 4),
                                            gflops=round(2.0 * h["nnz"] / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1),
                                            frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4), algorithmic_bytes=alg,
                                            hell_GB=round(h["slots"] * (elem + 4) * 1e-9, 2), build_ms=round(build_s * 1e3, 1), form=ran,
                                            parity=adopted_parity if adopted else check_windows(h, x, z, letter, shape),
                                            as_built_ms=[round(v, 4) for v in as_built])
            if placements:
                entry = out[f"{pattern}_{name}"]
                entry["placements_ms"] = [[round(v, 4) for v in blocks] for blocks in placements]
                every = [v for blocks in placements for v in blocks]
                entry["ms_spread"] = spread(every)
                entry["frac_spread"] = {k: round(alg / (v * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4) for k, v in (("best", min(every)), ("worst", max(every)))}
            if ordered:
                entry = out[f"{pattern}_{name}"]
                entry["plan_counts_uses_builds_stales"] = list(capi.plan_counts(handle))
            if adopted:
                entry = out[f"{pattern}_{name}"]
                entry["adopt_ms"] = round(adopt_s * 1e3, 1)
                entry["adopted_copy_GB"] = round(adopted_bytes * 1e-9, 3)
                entry["calls_on_the_copy"] = capi.spgpuSpmvAdoptedUses(handle)
                entry["slots_per_nnz_of_the_copy"] = out[f"{pattern}_sorted_aligned"]["slots_per_nnz"]
                capi.spgpuSpmvThaw(handle, p(h["rP"]))
            if frozen:
                entry = out[f"{pattern}_{name}"]
                entry["freeze_ms"] = round(freeze_s * 1e3, 1)
                entry["frozen_copy_GB"] = round(frozen_bytes * 1e-9, 3)
                entry["streamed_bytes_per_nnz"] = elem + 2
                capi.spgpuSpmvThaw(handle, p(h["rP"]))
            del h
            torch.cuda.empty_cache()
        del coo
        torch.cuda.empty_cache()
    return out


def bench_c1(handle, stream, dev):
    """BASELINE configs[0]: spgpuDhellspmv on the 5-point Laplacian 1024 x 1024, built by the HOST converters exactly as
    the reference's harness does (hellPerf.cpp:127-264); converter output checked against the checksums the reference's
    own objects produced (tests/golden/checksums.json), the product against the oracle on ALL rows; and the CG solver of
    tools/cg_amd.c (plain C on the ABI) for the time per iteration of a captured iteration."""
    import subprocess
    import numpy as np
    import torch
    import oracle_api as O
    from spgpu_amd import capi, formats, synth
    n, m, r, c, v = synth.laplacian_2d_5pt(1024)
    hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
    with open(os.path.join(ROOT, "tests", "golden", "checksums.json")) as f:
        want = json.load(f)["lap2d_1024_d"]["fnv"]
    converters = all(O.fnv(hell[k]) == want[w] for k, w in (("values", "hell_values"), ("indices", "hell_indices"),
                                                            ("hack_offsets", "hell_hack_offsets"), ("row_lengths", "row_lengths")))
    x = synth.hashed_vector(m)
    dx, dz = formats.to_device(x, dev), torch.empty(n, dtype=torch.float64, device=dev)
    mat = formats.DeviceHell(hell, dev)
    torch.cuda.synchronize()
    call = lambda: mat.spmv(handle, dz, None, 1.0, dx, 0.0, avg_nnz=5)
    time_launches(stream, call, 20)
    t = time_launches(stream, call, 200) / 200
    torch.cuda.synchronize()
    ok = dz.cpu().numpy().tobytes() == O.default_spmv(hell, x, None, 1.0, 0.0).tobytes()
    alg = hell_algorithmic_bytes(mat.nnz, n, m, n // 32)
    out = dict(workload="spgpuDhellspmv, 5-point Laplacian 1024 x 1024 (1 048 576 rows, 5 238 784 nnz), host-converted",
               us=round(t * 1e6, 2), gflops=round(2.0 * mat.nnz / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1),
               frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4), algorithmic_bytes=alg,
               converters="byte-identical to the reference build (checksums)" if converters else "MISMATCH",
               parity="bit-exact vs oracle on all rows" if ok else "MISMATCH")
    cg = os.path.join(ROOT, "tools", "cg_amd.bin")
    if os.path.exists(cg):
        try:
            run = subprocess.run([cg, "1024", "60", "1e-30", "timing"], capture_output=True, text=True, timeout=120)
            for line in run.stdout.splitlines():
                if line.startswith("graph replay:"):
                    words = line.replace("(", " ").replace(")", " ").replace(";", " ").split()
                    out["cg_graph_us_per_iteration"] = float(words[words.index("us") - 1])
                    out["cg_eager_host_scalars_us_per_iteration"] = float(words[words.index("scalars:") + 1])
                    out["cg_iterates"] = line.split("; iterate")[-1].strip()
                if line.startswith("fused replay:"):
                    words = line.replace("(", " ").split()
                    out["cg_fused_graph_us_per_iteration"] = float(words[words.index("us") - 1])
                    out["cg_fused_iterates"] = line.split("; iterate")[-1].strip()
        except (OSError, subprocess.SubprocessError, ValueError) as error:
            out["cg"] = f"not run: {error!r}"
    return out


def bench_c3(handle, stream, dev, rows, ell_rows):
    """BASELINE configs[2]: HELL fp32 vs ELL fp32 on power-law row lengths (max 2048, mean 32), random columns: time and
    FOOTPRINT (the memory win of HELL).  ELL at 10 M rows needs 164 GB: measured at `ell_rows` rows beside HELL on the
    same rows; HELL also at the full size."""
    import torch
    import oracle_api as O
    from spgpu_amd import capi, synth
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    n = rows // 32 * 32
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    out = {}

    def hell_case(count, adopt=False):
        h = synth.hell_ragged_on_device(lengths[:count], count, "S", 32, seed=5, device=dev)
        h["letter"] = "S"
        x, z = synth.device_vector(count, "S", 3, dev), torch.zeros(count, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        call = lambda: capi.hellspmv["S"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                          None, 32, count, p(x), 0.0, 0)
        time_launches(stream, call, 3)
        t = time_launches(stream, call, 20) / 20
        torch.cuda.synchronize()
        alg = h["nnz"] * 8 + count * 8 + count * 4 + (count // 32) * 4
        entry = dict(rows=count, nnz=h["nnz"], ms=round(t * 1e3, 4), gflops=round(2.0 * h["nnz"] / t * 1e-9, 1),
                     hbm_gbs=round(alg / t * 1e-9, 1), frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4),
                     footprint_GB=round((h["slots"] * 8 + count * 4 + count // 32 * 4) * 1e-9, 2),
                     slots_per_nnz=round(h["slots"] / h["nnz"], 3), parity=check_windows(h, x, z, "S", O.slab_shape("S")))
        if adopt:
            # the same call after spgpuHellSpmvAdopt (include/spgpu/tuning.h): the library's ordered copy of the caller's matrix
            z_plain = z.clone()
            t0 = time.perf_counter()
            said = capi.spgpuHellSpmvAdopt(handle, capi.TYPE_CODE["S"], p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), count, 0)
            torch.cuda.synchronize()
            adopt_ms = (time.perf_counter() - t0) * 1e3
            if said == capi.SPGPU_SUCCESS:
                time_launches(stream, call, 3)
                ta = time_launches(stream, call, 20) / 20
                torch.cuda.synchronize()
                worst = float(((z - z_plain).abs().max() / z_plain.abs().max()).item())
                entry["adopted"] = dict(ms=round(ta * 1e3, 4), gflops=round(2.0 * h["nnz"] / ta * 1e-9, 1), frac=round(alg / ta * 1e-9 / HBM_PEAK_GBS, 4),
                                        adopt_ms=round(adopt_ms, 1), copy_GB=round(capi.spgpuSpmvFrozenBytes(handle) * 1e-9, 2),
                                        parity=f"max |z - z_plain| / max |z_plain| = {worst:.2e} (another order of additions; bar 1e-4)")
                capi.spgpuSpmvThaw(handle, p(h["rP"]))
            else:
                entry["adopted"] = dict(status=said)
        return entry

    out["hell_fp32"] = hell_case(n, adopt=True)
    torch.cuda.empty_cache()
    ne = min(ell_rows, n) // 32 * 32
    out["hell_fp32_same_rows_as_ell"] = hell_case(ne)
    torch.cuda.empty_cache()
    e = synth.ell_ragged_on_device(lengths[:ne], ne, "S", seed=6, device=dev)
    x, z = synth.device_vector(ne, "S", 3, dev), torch.zeros(ne, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    call = lambda: capi.ellspmv["S"](handle, p(z), None, 1.0, p(e["cM"]), p(e["rP"]), e["pitch"], e["pitch"], p(e["rS"]),
                                     None, 32, e["max_row"], ne, p(x), 0.0, 0)
    time_launches(stream, call, 3)
    t = time_launches(stream, call, 20) / 20
    alg = e["nnz"] * 8 + ne * 8 + ne * 4
    out["ell_fp32"] = dict(rows=ne, nnz=e["nnz"], ms=round(t * 1e3, 4), gflops=round(2.0 * e["nnz"] / t * 1e-9, 1),
                           hbm_gbs=round(alg / t * 1e-9, 1), frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4),
                           footprint_GB=round((e["pitch"] * e["max_row"] * 8 + ne * 4) * 1e-9, 2))
    # the same spgpuSellspmv call after spgpuEllSpmvAdopt: the library's ordered HELL copy of the caller's ELL matrix
    try:
        z_plain = z.clone()
        t0 = time.perf_counter()
        said = capi.spgpuEllSpmvAdopt(handle, capi.TYPE_CODE["S"], p(e["cM"]), p(e["rP"]), e["pitch"], e["pitch"], p(e["rS"]), e["max_row"], ne, 0)
        torch.cuda.synchronize()
        adopt_ms = (time.perf_counter() - t0) * 1e3
        if said == capi.SPGPU_SUCCESS:
            time_launches(stream, call, 3)
            ta = time_launches(stream, call, 20) / 20
            torch.cuda.synchronize()
            worst = float(((z - z_plain).abs().max() / z_plain.abs().max()).item())
            out["ell_fp32"]["adopted"] = dict(ms=round(ta * 1e3, 4), gflops=round(2.0 * e["nnz"] / ta * 1e-9, 1), adopt_ms=round(adopt_ms, 1),
                                              copy_GB=round(capi.spgpuSpmvFrozenBytes(handle) * 1e-9, 2),
                                              parity=f"max |z - z_plain| / max |z_plain| = {worst:.2e} (another order of additions; bar 1e-4)")
            capi.spgpuSpmvThaw(handle, p(e["rP"]))
        else:
            out["ell_fp32"]["adopted"] = dict(status=said)
        del z_plain
    except Exception as error:  # noqa: BLE001
        out["ell_fp32"]["adopted"] = dict(error=repr(error))
    out["ell_footprint_GB_at_full_rows"] = round(((n + 31) // 32 * 32) * int(lengths.max()) * 8e-9 + n * 4e-9, 1)
    del e, x, z
    torch.cuda.empty_cache()
    # the same lengths with the rows ordered by length on the device (the reference's third format, OELL/OHELL:
    # hellPerf.cpp:324-378): what is left of HELL's footprint, and the time (random columns: still gather-bound)
    from spgpu_amd import formats
    coo = synth.ragged_coo_on_device(lengths, n, "random", 2048, "S", seed=5, device=dev)
    h = formats.coo_to_ordered_hell_device(handle, n, *coo, "S", 32, 2048, 256)
    x, z = synth.device_vector(n, "S", 3, dev), torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    if vendor() is not None:
        try:
            out["vendor_context"] = vendor().coo_context(stream, n, n, lengths, coo[1], coo[2], x, torch.empty_like(z), letter="S")
        except Exception as error:  # noqa: BLE001
            out["vendor_context"] = repr(error)
    del coo
    torch.cuda.empty_cache()
    capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
    call = lambda: capi.hellspmv["S"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                      p(h["rIdx"]), 32, n, p(x), 0.0, 0)
    time_launches(stream, call, 3)
    t = time_launches(stream, call, 20) / 20
    z.zero_()
    torch.cuda.synchronize()
    time_launches(stream, call, 1)
    capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
    alg = h["nnz"] * 8 + n * 8 + n * 4 + (n // 32) * 4 + n * 4
    out["hell_fp32_rows_ordered"] = dict(rows=n, nnz=h["nnz"], ms=round(t * 1e3, 4), gflops=round(2.0 * h["nnz"] / t * 1e-9, 1),
                                         hbm_gbs=round(alg / t * 1e-9, 1), frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4),
                                         footprint_GB=round((h["slots"] * 8 + n * 8 + n // 32 * 4) * 1e-9, 2),
                                         slots_per_nnz=round(h["slots"] / h["nnz"], 3),
                                         parity=check_windows(h, x, z, "S", O.slab_shape("S", "ragged", deep_cap=O.DEEP_CAP)))
    return out


def bench_c4(handle, stream, dev, grid):
    """BASELINE configs[3]: HDIA fp64, 7-point Laplacian grid^3 (512^3: 134 M rows, 938 M nnz), oracle check on a
    2048-row window in the middle of the grid."""
    import numpy as np
    import torch
    import oracle_api as O
    from spgpu_amd import capi, synth
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    m = grid
    d = synth.hdia_laplacian7_on_device(m, "D", 32, device=dev)
    n = d["rows"]
    hacks = n // 32
    x, y = synth.device_vector(n, "D", 3, dev), synth.device_vector(n, "D", 4, dev)
    z = torch.empty_like(y)
    torch.cuda.synchronize()
    out = dict(rows=n, nnz=d["nnz"], stored_diagonals=d["height"])
    for beta in (0.0, 0.5):
        call = lambda: capi.hdiaspmv["D"](handle, p(z), p(y), 1.0, p(d["dM"]), p(d["offsets"]), 32, p(d["hack_offsets"]), n, n, p(x), beta)
        time_launches(stream, call, 3)
        t = time_launches(stream, call, 20) / 20
        torch.cuda.synchronize()
        slots = 32 * d["height"]
        alg = slots * 8 + d["height"] * 4 + (hacks + 1) * 4 + n * 8 + n * 8 * (2 if beta else 1)
        first = (hacks // 2) * 32
        ho = d["hack_offsets"][hacks // 2: hacks // 2 + 65].cpu().numpy().astype(np.int64)
        # the oracle indexes x by offsets[d] + local row: hand it the slice of x the window can reach, shifted by m^2
        xs = x[max(first - m * m, 0): min(first + 2048 + m * m, n)].cpu().numpy()
        xw = np.zeros(2048 + 2 * m * m)
        lead = max(first - m * m, 0) - (first - m * m)
        xw[lead:lead + xs.size] = xs
        sub = dict(letter="D", rows=2048, cols=xw.size, hack_size=32, hack_offsets=(ho - ho[0]).astype(np.int32),
                   offsets=(d["offsets"][ho[0]:ho[-1]].cpu().numpy().astype(np.int64) + m * m).astype(np.int32),
                   values=d["dM"][ho[0] * 32: ho[-1] * 32].cpu().numpy())
        ys = y[first:first + 2048].cpu().numpy()
        want = O.hdia_spmv(sub, xw, ys if beta else None, 1.0, beta)
        ok = z[first:first + 2048].cpu().numpy().tobytes() == want.tobytes()
        out[f"beta{beta:g}"] = dict(ms=round(t * 1e3, 4), gflops=round(2.0 * d["nnz"] / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1),
                                    frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4), algorithmic_bytes=alg,
                                    parity="bit-exact vs oracle on 2048 rows" if ok else "MISMATCH")
    return out


def run_spmv(args, rank, world):
    import torch
    from spgpu_amd import capi, synth

    dist = collectives()
    dev = local_device()
    torch.cuda.set_device(dev)
    handle = capi.create_handle(torch.cuda.current_device())
    stream = torch.cuda.Stream()
    capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))

    def build(pattern):
        h = synth.hell_uniform_on_device(args.rows, args.nnz_per_row, pattern, "D", 32, seed=1, device=dev)
        # torch fills these arrays on ITS stream; the library launches on the handle's stream.
        # Without this the SpMV could read column indices that are still being written.
        torch.cuda.synchronize()
        return h

    def launcher(h, x, y, z, alpha, beta):
        p = lambda t: C.c_void_p(t.data_ptr())
        a = (handle, p(z), p(y), C.c_double(alpha), p(h["cM"]), p(h["rP"]), h["hack_size"], p(h["hack_offsets"]),
             p(h["rS"]), None, args.nnz_per_row, h["rows"], p(x), C.c_double(beta), 0)
        return lambda: capi.hellspmv["D"](*a)

    h = build(args.pattern)
    x = synth.device_vector(h["cols"], "D", 3, dev)
    y = synth.device_vector(h["rows"], "D", 4, dev)
    z = torch.empty_like(y)
    hacks = h["rows"] // h["hack_size"]
    step = launcher(h, x, y, z, 1.0, 0.0)  # alpha = 1, beta = 0 as the reference's harness (hellPerf.cpp:27-28)
    if args.frozen:
        pf = lambda t: C.c_void_p(t.data_ptr())
        said = capi.spgpuHellSpmvFreeze(handle, capi.TYPE_CODE["D"], pf(h["cM"]), pf(h["rP"]), h["hack_size"], pf(h["hack_offsets"]), pf(h["rS"]), None, h["rows"], 0)
        assert said == capi.SPGPU_SUCCESS, f"--frozen: spgpuHellSpmvFreeze said {said}"

    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        settle(stream, step)
        for _ in range(args.warmup):
            step()
    stream.synchronize()
    headline_form = form_ran(handle)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_s = time_launches(stream, step, args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    flops = 2.0 * h["nnz"]
    alg = hell_algorithmic_bytes(h["nnz"], h["rows"], h["cols"], hacks)
    per_launch = kernel_s / args.steps
    # the same launches once more in five blocks, with what the card reports about itself sampled beside them: a memory-bound
    # kernel on this pool runs at speeds several per cent apart from card to card and from one set of allocations to the next
    #  -- the record says which card, in which state, and how much the blocks differ
    from spgpu_amd import gpu_state
    card = gpu_state.Card(torch.cuda.current_device())
    with gpu_state.Sampler(card) as sampler:
        blocks = [time_launches(stream, step, max(args.steps // 5, 10)) / max(args.steps // 5, 10) * 1e3 for _ in range(5)]
    device = dict(card.identity(), name=torch.cuda.get_device_name(dev), during_timed_blocks=sampler.summary(),
                  note="sysfs of the card as this container sees it (clocks / power sampled every 4 ms during the five blocks; inside this "
                       "pool's containers the busy counters read 0 and the clock files do not follow the load: recorded for what they are worth)")
    out = dict(
        metric="HELL fp64 SpMV GFLOP/s + achieved HBM GB/s (% of roofline), 1 GPU",
        value=round(flops * args.steps * world / wall * 1e-9, 2), unit="GFLOP/s", n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=round(wall / args.steps * 1e3, 5),
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
        config=dict(workload=f"HELL fp64 SpMV (spgpuDhellspmv), {h['rows']} rows x {args.nnz_per_row} nnz/row uniform, "
                             f"hackSize 32, columns {args.pattern}, alpha=1 beta=0 (BASELINE configs[1])"
                             + (" -- FROZEN first (spgpuHellSpmvFreeze: 16-bit column indices, 10 bytes per nonzero streamed)" if args.frozen else ""),
                    rows=h["rows"], nnz=h["nnz"], hack_size=32, pattern=args.pattern,
                    parallelism="single GPU" if world == 1 else f"{world} independent replicas"),
        roofline=dict(bound="hbm", achieved=round(alg / per_launch * 1e-9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                      frac=round(alg / per_launch * 1e-9 / HBM_PEAK_GBS, 4),
                      traffic=None if args.frozen else committed_traffic(h["rows"], args.nnz_per_row, args.pattern),
                      kernel=("sweepSpmvKernel<double, 2 rows x 16 packs per lane, HELL, the default kernel's tail order>: the form "
                              "spgpuGetLastSpmvForm reports: sweep" if headline_form == "sweep" else
                              "slabSpmvKernel<double, RPL 2, 1 phase, HELL, nt, 8 columns/stage, prefetch after gathers, tail> in the form "
                              f"spgpuGetLastSpmvForm reports: {headline_form}"), algorithmic_bytes_per_launch=alg,
                      kernel_ms=round(per_launch * 1e3, 5), kernel_ms_blocks=spread(blocks)),
        device=device,
    )

    if rank == 0:
        step()
        torch.cuda.synchronize()
        out["parity"] = spot_check_hell(h, x, y, z, 1.0, 0.0)
        if not args.no_extras:
            extras = {}
            # beta != 0 on the headline matrix
            s2 = launcher(h, x, y, z, 1.0, 0.5)
            time_launches(stream, s2, 5)
            t = time_launches(stream, s2, 50) / 50
            a2 = hell_algorithmic_bytes(h["nnz"], h["rows"], h["cols"], hacks, beta_nonzero=True)
            extras[f"{args.pattern}_beta0.5"] = dict(gflops=round(flops / t * 1e-9, 1), hbm_gbs=round(a2 / t * 1e-9, 1),
                                                      frac=round(a2 / t * 1e-9 / HBM_PEAK_GBS, 4), ms=round(t * 1e3, 4), form=form_ran(handle))
            # the headline matrix FROZEN (include/spgpu/tuning.h spgpuHellSpmvFreeze: the caller's promise that the index arrays stay as
            # they are; the library streams its 16-bit copy of the column indices -- 10 instead of 12 bytes per nonzero): the same
            # spgpuDhellspmv call, bit-identical result.  `frac` is still ALGORITHMIC bytes (12 per nonzero) / time / peak; the
            # headline `value` above is the unfrozen call.
            try:
                p_ = lambda t: C.c_void_p(t.data_ptr())
                t0 = time.perf_counter()
                said = capi.spgpuHellSpmvFreeze(handle, capi.TYPE_CODE["D"], p_(h["cM"]), p_(h["rP"]), h["hack_size"], p_(h["hack_offsets"]), p_(h["rS"]),
                                                None, h["rows"], 0)
                torch.cuda.synchronize()
                freeze_ms = (time.perf_counter() - t0) * 1e3
                if said == capi.SPGPU_SUCCESS:
                    uses0 = capi.plan_counts(handle)[0]
                    blocks_f = timed_blocks(stream, step, blocks=5, launches=max(args.steps // 5, 10))
                    t = sorted(blocks_f)[len(blocks_f) // 2] * 1e-3
                    z.zero_()
                    step()
                    torch.cuda.synchronize()
                    out["headline_frozen"] = dict(
                        ms=round(t * 1e3, 5), gflops=round(flops / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1), frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4),
                        ms_blocks=spread(blocks_f), streamed_bytes_per_nnz=10, streamed_bytes_per_launch=alg - 2 * h["nnz"],
                        frac_of_streamed_bytes=round((alg - 2 * h["nnz"]) / t * 1e-9 / HBM_PEAK_GBS, 4),
                        frozen_copy_GB=round(capi.spgpuSpmvFrozenBytes(handle) * 1e-9, 3), freeze_ms=round(freeze_ms, 1),
                        calls_from_the_frozen_record=capi.plan_counts(handle)[0] - uses0, form=form_ran(handle),
                        parity=spot_check_hell(h, x, y, z, 1.0, 0.0),
                        what="the headline matrix and call after spgpuHellSpmvFreeze: frac = algorithmic bytes (12 per nonzero) / time / 8 TB/s as for "
                             "`roofline`; frac_of_streamed_bytes = what the frozen kernel actually streams (10 per nonzero) / time / 8 TB/s")
                else:
                    out["headline_frozen"] = dict(status=said)
                capi.spgpuSpmvThaw(handle, p_(h["rP"]))
            except Exception as error:  # noqa: BLE001 - an extra must not take the headline record down
                out["headline_frozen"] = dict(error=repr(error))
            # the headline launches once more with the arrays in allocations of their own, `--placements` times: how far does the
            # placement of the caller's arrays move THIS kernel on this card?  (`value` stays what the contract
            # says: the K timed steps above, on the arrays as they were built.)
            if args.placements > 0:
                moved = []
                for _ in range(args.placements):
                    own = OwnAllocations(dict(cM=h["cM"], rP=h["rP"], hack_offsets=h["hack_offsets"], rS=h["rS"], x=x, z=z))
                    try:
                        torch.cuda.synchronize()
                        placed = lambda own=own: capi.hellspmv["D"](handle, own["z"], None, C.c_double(1.0), own["cM"], own["rP"], h["hack_size"],
                                                                    own["hack_offsets"], own["rS"], None, args.nnz_per_row, h["rows"], own["x"],
                                                                    C.c_double(0.0), 0)
                        moved.append([round(v, 4) for v in timed_blocks(stream, placed)])
                    finally:
                        torch.cuda.synchronize()
                        own.free()
                every = [v for blocks in moved for v in blocks]
                out["roofline"]["placements_kernel_ms"] = moved
                out["roofline"]["placements_frac"] = {k: round(alg / (v * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4) for k, v in (("best", min(every)), ("worst", max(every)))}
            out["cpu_baseline"] = cpu_baseline(h, x, args.cpu_seconds)
            if vendor() is not None:
                try:
                    extras[f"{args.pattern}_vendor_context"] = vendor().uniform_hell_context(stream, h, x, torch.empty_like(y))
                except Exception as error:  # noqa: BLE001
                    extras[f"{args.pattern}_vendor_context"] = repr(error)
            for pattern in ("banded", "window", "random"):
                if pattern == args.pattern:
                    continue
                del h, step, s2
                torch.cuda.empty_cache()
                h = build(pattern)
                step = s2 = launcher(h, x, y, z, 1.0, 0.0)
                with torch.cuda.stream(stream):
                    settle(stream, step)
                time_launches(stream, step, 5)
                t = time_launches(stream, step, 50) / 50
                extras[pattern] = dict(gflops=round(flops / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1),
                                       frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4), ms=round(t * 1e3, 4), form=form_ran(handle),
                                       parity=spot_check_hell(h, x, y, z, 1.0, 0.0))
                if vendor() is not None:   # the vendor library on the same matrix, for context (never on the path)
                    try:
                        extras[pattern]["vendor_context"] = vendor().uniform_hell_context(stream, h, x, torch.empty_like(y))
                    except Exception as error:  # noqa: BLE001
                        extras[pattern]["vendor_context"] = repr(error)
                if pattern == "random":
                    # AUTO takes the SWEEP form here since round 4 (scattered columns that ascend inside the rows: 32 rows per lane
                    # carried through the columns in step, include/spgpu/tuning.h; the same bits as the gather kernel).  For the
                    # record, the gather kernel on the same arrays:
                    capi.spgpuSetSpmvForm(handle, capi.FORM_GATHER)
                    time_launches(stream, step, 3)
                    t = time_launches(stream, step, 20) / 20
                    z.zero_()
                    torch.cuda.synchronize()
                    time_launches(stream, step, 1)
                    extras["random_form_gather"] = dict(gflops=round(flops / t * 1e-9, 1), hbm_gbs=round(alg / t * 1e-9, 1),
                                                        frac=round(alg / t * 1e-9 / HBM_PEAK_GBS, 4), ms=round(t * 1e3, 4), form=form_ran(handle),
                                                        parity=spot_check_hell(h, x, y, z, 1.0, 0.0))
                    capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
            out["variants"] = extras
            # the same banded workload in the other three value types (untimed extras, each checked against the oracle)
            types = {}
            for letter, elem in (("S", 4), ("C", 8), ("Z", 16)):
                del h, step, s2
                torch.cuda.empty_cache()
                h = synth.hell_uniform_on_device(args.rows, args.nnz_per_row, "banded", letter, 32, seed=1, device=dev)
                xt, yt = synth.device_vector(h["cols"], letter, 3, dev), synth.device_vector(h["rows"], letter, 4, dev)
                zt = torch.empty_like(yt)
                torch.cuda.synchronize()
                pt = lambda t: C.c_void_p(t.data_ptr())
                at = (handle, pt(zt), pt(yt), capi.scalar(letter, 1.0), pt(h["cM"]), pt(h["rP"]), 32, pt(h["hack_offsets"]),
                      pt(h["rS"]), None, args.nnz_per_row, h["rows"], pt(xt), capi.scalar(letter, 0.0), 0)
                step = s2 = lambda at=at, letter=letter: capi.hellspmv[letter](*at)
                time_launches(stream, step, 5)
                t = time_launches(stream, step, 50) / 50
                bytes_t = hell_algorithmic_bytes(h["nnz"], h["rows"], h["cols"], hacks, elem=elem)
                flops_t = (2.0 if letter == "S" else 8.0) * h["nnz"]
                types[letter] = dict(ms=round(t * 1e3, 4), hbm_gbs=round(bytes_t / t * 1e-9, 1),
                                     frac=round(bytes_t / t * 1e-9 / HBM_PEAK_GBS, 4), gflops=round(flops_t / t * 1e-9, 1),
                                     parity=spot_check_hell(h, xt, yt, zt, 1.0, 0.0))
                # frozen (16-bit column indices; complex fp64 has no packed form: SPGPU_UNSUPPORTED)
                if capi.spgpuHellSpmvFreeze(handle, capi.TYPE_CODE[letter], pt(h["cM"]), pt(h["rP"]), 32, pt(h["hack_offsets"]), pt(h["rS"]), None,
                                            h["rows"], 0) == capi.SPGPU_SUCCESS:
                    time_launches(stream, step, 5)
                    tf = time_launches(stream, step, 50) / 50
                    zt.zero_()
                    time_launches(stream, step, 1)
                    types[letter]["frozen"] = dict(ms=round(tf * 1e3, 4), frac=round(bytes_t / tf * 1e-9 / HBM_PEAK_GBS, 4), gflops=round(flops_t / tf * 1e-9, 1),
                                                   streamed_bytes_per_nnz=elem + 2, parity=spot_check_hell(h, xt, yt, zt, 1.0, 0.0))
                    capi.spgpuSpmvThaw(handle, pt(h["rP"]))
                del xt, yt, zt
            out["other_types_banded"] = types
            # the other BASELINE configurations and the north_star target, each with its oracle check
            del h, step, s2
            torch.cuda.empty_cache()
            h = step = s2 = None
            configs = {}
            for name, fn in (("powerlaw_fp64", lambda: bench_powerlaw(args, handle, stream, dev, args.rows)),
                             ("c1_laplacian_1024", lambda: bench_c1(handle, stream, dev)),
                             ("c3_hell_vs_ell_fp32", lambda: bench_c3(handle, stream, dev, args.rows, args.ell_rows)),
                             ("c4_hdia_7pt", lambda: bench_c4(handle, stream, dev, args.grid))):
                try:
                    configs[name] = fn()
                except Exception as error:  # noqa: BLE001 - an extra must not take the headline record down
                    configs[name] = dict(error=repr(error))
                torch.cuda.empty_cache()
            out["configs"] = configs
            pl = configs.get("powerlaw_fp64", {})
            target = {name: {key: pl[name][key] for key in ("ms", "frac", "frac_spread", "ms_spread", "slots_per_nnz", "streamed_bytes_per_nnz", "parity") if key in pl[name]}
                      for name in ("band_sorted_aligned", "band_sorted_aligned_frozen", "band_sorted", "near_sorted_aligned", "near_sorted_aligned_frozen",
                                   "near_sorted", "band_sorted_global", "band_plain", "band_plain_adopted", "near_plain", "near_plain_adopted")
                      if isinstance(pl.get(name), dict)}
            target["what"] = ("north_star target: spgpuDhellspmv, fp64, 10 M rows, power-law lengths (mean 32, max 2048), rows ordered "
                              "on the device (windows of 2048, rows > 256 set aside) and run through rIdx; *_aligned: the order whose windows "
                              "coincide with the kernel's 2048-row workgroups (spgpuOellOrderAlignedDevice); *_global: the reference's own order, "
                              "one sort of all rows (ellToOell, ell.c:85-202); *_plain: the rows as they come, no rIdx; *_plain_adopted: the same arrays and the same call WITHOUT rIdx after "
                              "spgpuHellSpmvAdopt (the caller's promise not to touch the matrix: the library orders and freezes a copy of its own and runs the "
                              "call on it; z in the caller's row order, the explicitly ordered call's bits); *_frozen: the same arrays and the same "
                              "spgpuDhellspmv call after spgpuHellSpmvFreeze (the caller's promise that the index arrays stay as they are: the library "
                              "streams its 16-bit copy of the column indices, 10 instead of 12 bytes per nonzero; bit-identical results); frac = "
                              "algorithmic bytes (12 per nonzero, frozen or not) / time / 8 TB/s, time = median over placements of the arrays (ms_spread / frac_spread: every timed block); bar 0.70")
            out["config"]["north_star_target"] = target
            out["target"] = target
            # the headline kernel's HBM traffic, counted during THIS run (two short child runs under rocprofv3 --pmc; the value read
            # from the committed summary under profiles/ stays as the fallback and is kept beside it)
            if world == 1 and not os.environ.get("SPGPU_BENCH_INNER") and not os.environ.get("SPGPU_BENCH_NO_PMC"):
                committed = out["roofline"]["traffic"]
                try:
                    counted, detail = live_traffic(args)
                except Exception as error:  # noqa: BLE001 - the record does not depend on the profiler
                    counted, detail = None, repr(error)
                if counted is not None and isinstance(out.get("headline_frozen"), dict) and "ms" in out["headline_frozen"] and not args.frozen:
                    try:
                        counted_f, detail_f = live_traffic(args, extra=("--frozen",))
                    except Exception as error:  # noqa: BLE001
                        counted_f, detail_f = None, repr(error)
                    out["headline_frozen"]["traffic"] = counted_f
                    out["headline_frozen"]["traffic_source"] = detail_f
                if counted is not None:
                    out["roofline"]["traffic"] = counted
                    out["roofline"]["traffic_source"] = dict(detail, committed_summary_value=committed,
                                                             over_algorithmic=round(counted / out["roofline"]["algorithmic_bytes_per_launch"], 4))
                else:
                    out["roofline"]["traffic_source"] = dict(source="profiles/*_bench_spmv_pmc.json (committed PMC summary of the same workload)",
                                                             live_attempt=detail)
            if world == 1:
                # the N = 1 point of the curve `--gpus N` (N > 1) measures: same sharded SpMM step on one rank
                one = measure_spmm(args, 0, 1, handle, stream, dev, 50, 5)
                out["spmm_1gpu"] = dict(value=one["value"], unit=one["unit"], ms_per_step=one["ms_per_step"],
                                        workload=one["config"]["workload"], roofline_frac=one["roofline"]["frac"],
                                        kernel_ms=one["roofline"]["kernel_ms"], parity=one["parity"])
        elif "cpu_baseline" not in out:
            out["cpu_baseline"] = None
        first = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "target", "headline_frozen", "cpu_baseline", "parity", "device"]
        out = {**{key: out[key] for key in first if key in out}, **{key: value for key, value in out.items() if key not in first}}
        print(json.dumps(out), flush=True)
    capi.spgpuDestroy(handle)


def run_spmm(args, rank, world):
    """Row-sharded HELL fp64 SpMM (BASELINE configs[4]): weak scaling, 5 M rows x 32 nnz per GPU, 16 rhs;
    one step = all-gather of the X row blocks over RCCL + the local product(s)."""
    import torch
    from spgpu_amd import capi

    dev = local_device()
    torch.cuda.set_device(dev)
    handle = capi.create_handle(torch.cuda.current_device())
    stream = torch.cuda.Stream()
    capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
    out = measure_spmm(args, rank, world, handle, stream, dev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(out), flush=True)
    capi.spgpuDestroy(handle)


def measure_spmm(args, rank, world, handle, stream, dev, steps, warmup):
    """The sharded SpMM step on `world` ranks (world == 1: the single-GPU point of the same curve).  Returns the
    JSON record on rank 0, None elsewhere."""
    import torch
    from spgpu_amd import capi, sharded, synth

    dist = collectives()
    rows_local, k, L = args.spmm_rows_per_gpu // 32 * 32, args.rhs, args.nnz_per_row
    n_total = rows_local * world
    first = rank * rows_local
    blocks = [(r * rows_local, rows_local) for r in range(world)]
    block = synth.hell_uniform_on_device(rows_local, L, args.spmm_pattern, "D", 32, seed=11 + rank, device=dev,
                                         n_cols=n_total, row_offset=first)
    split = (world > 1 or args.force_split) and not args.no_split
    if split:
        own, rest = synth.split_uniform_hell_by_columns(block, first, rows_local)
        del block
        torch.cuda.empty_cache()
    else:
        own, rest = block, None
    x_local = synth.device_vector(rows_local * k, "D", 21 + rank, dev).view(rows_local, k)
    y_local = synth.device_vector(rows_local * k, "D", 31 + rank, dev).view(rows_local, k)
    z_local = torch.empty_like(y_local)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def local_product(part, Z, Y, alpha, X, beta):
        capi.hellspmm["D"](handle, p(Z), p(Y), C.c_double(alpha), p(part["cM"]), p(part["rP"]), 32, p(part["hack_offsets"]),
                           p(part["rS"]), None, L, part["rows"], p(X), C.c_double(beta), 0, k, k, k)

    new_rows = lambda rows: torch.empty(rows, k, dtype=torch.float64, device=dev)
    needed_mode = split and args.exchange == "needed"   # what `value` reports
    both = split                                          # the other exchange is measured beside it
    distributed = world > 1 or (dist.is_available() and dist.is_initialized())

    # ---- the driver of a step: the C ABI (include/spgpu/sharded.h: packing kernel, RCCL and both products issued by the
    # library) unless --driver torch or the gloo rehearsal (two ranks on one GPU cannot share an RCCL communicator)
    driver = "torch" if (rehearsing() or args.driver == "torch") else "c"
    plans, comm = {}, None
    if driver == "c":
        ok = 1.0

        def everyone(mine):
            """True iff `mine` is true on every rank (one all_reduce: called by all ranks at the same point)."""
            if not distributed:
                return bool(mine)
            flag = torch.tensor([1.0 if mine else 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return float(flag.item()) > 0.5

        try:
            if distributed:
                # Every step that could fail on ONE rank is followed by an agreement of all ranks before the next
                # collective: a rank that raised on its own would leave the others waiting in a collective for ever.
                if not everyone(capi.spgpuCommAvailable()):
                    raise RuntimeError("RCCL cannot be loaded on every rank")
                ident = torch.zeros(129, dtype=torch.uint8, device=dev)       # 128 bytes of id + "valid"
                if rank == 0:
                    raw = (C.c_char * 128)()
                    if capi.spgpuCommGetUniqueId(raw) == capi.SPGPU_SUCCESS:
                        ident[:128].copy_(torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8))
                        ident[128] = 1
                dist.broadcast(ident, 0)                                      # always, valid or not
                torch.cuda.synchronize()
                host_ident = ident.cpu().numpy()
                if host_ident[128] != 1:
                    raise RuntimeError("spgpuCommGetUniqueId failed on rank 0")
                raw = (C.c_char * 128).from_buffer_copy(bytes(host_ident[:128].tobytes()))
                comm = C.c_void_p()
                if not everyone(capi.spgpuCommInitRank(C.byref(comm), world, raw, rank) == capi.SPGPU_SUCCESS):
                    raise RuntimeError("spgpuCommInitRank failed on some rank")
            first_rows = (C.c_longlong * (world + 1))(*[r * rows_local for r in range(world + 1)])
            own_block = capi.hell_block(own, L)
            rest_block = capi.hell_block(rest, L) if rest is not None else None
            kinds = [("allgather", capi.EXCHANGE_ALLGATHER)] + ([("needed", capi.EXCHANGE_NEEDED)] if both else [])
            for name, kind in kinds:
                plan = capi.ShardedPlan()
                status = capi.spgpuDhellspmmShardedCreate(C.byref(plan), handle, comm, rank, world, first_rows, C.byref(own_block),
                                                          C.byref(rest_block) if rest_block is not None else None, k, kind)
                if status == capi.SPGPU_SUCCESS:
                    plans[name] = plan
                if not everyone(status == capi.SPGPU_SUCCESS):
                    raise RuntimeError(f"spgpuDhellspmmShardedCreate({name}) returned {status} (or failed on another rank)")
        except Exception as error:  # noqa: BLE001 - raised by every rank at the same point (see `everyone`)
            print(f"rank {rank}: C driver set-up failed ({error!r}); falling back to the torch.distributed driver", file=sys.stderr, flush=True)
            ok = 0.0
        if ok < 0.5:
            for plan in plans.values():
                capi.spgpuDhellspmmShardedDestroy(plan)
            plans, driver = {}, "torch"

    if driver == "c":
        plan_step = plans["needed" if needed_mode else "allgather"]
        one, zero = C.c_double(1.0), C.c_double(0.0)

        def step():
            capi.spgpuDhellspmmShardedStep(plan_step, p(z_local), p(y_local), one, p(x_local), zero)

        def step_allgather():
            capi.spgpuDhellspmmShardedStep(plans["allgather"], p(z_local), p(y_local), one, p(x_local), zero)

        def step_needed():
            capi.spgpuDhellspmmShardedStep(plans["needed"], p(z_local), p(y_local), one, p(x_local), zero)

        def products_only():
            capi.spgpuDhellspmmShardedProducts(plan_step, p(z_local), p(y_local), one, p(x_local), zero)

        def gather_only():
            capi.spgpuDhellspmmShardedExchange(plans["allgather"], p(x_local))
            capi.spgpuDhellspmmShardedExchangeWait(plans["allgather"])

        def needed_only():
            capi.spgpuDhellspmmShardedExchange(plans["needed"], p(x_local))
            capi.spgpuDhellspmmShardedExchangeWait(plans["needed"])

        rows_received = lambda: int(capi.spgpuDhellspmmShardedRowsReceived(plans["needed"]))
    else:
        if both:
            # only the X rows A_rest names travel: its columns are renumbered into that sorted list.  The local part of
            # the set-up runs first and all ranks agree that it worked before any of them enters the set-up collectives
            # (otherwise: everyone falls back to the all-gather).
            try:
                needed, compact = sharded.needed_rows_of(rest["rP"], 0)
                rest_compact = dict(rest, rP=compact.contiguous())
                ready = 1.0
            except Exception as error:  # noqa: BLE001 - any local failure means "use the other exchange"
                print(f"rank {rank}: needed-rows set-up failed ({error!r}); falling back to all-gather", file=sys.stderr, flush=True)
                ready = 0.0
            if distributed:
                flag = torch.tensor([ready], device=dev, dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ready = float(flag.item())
            both = ready > 0.5
            needed_mode = needed_mode and both
        products_stream = stream.cuda_stream
        op_allgather = sharded.ShardedSpmm(dist, rank, world, blocks, own, rest, local_product, new_rows, products_stream=products_stream)
        op_needed = (sharded.ShardedSpmm(dist, rank, world, blocks, own, rest_compact, local_product, new_rows, needed=needed,
                                         products_stream=products_stream) if both else None)
        op = op_needed if needed_mode else op_allgather

        def step():
            op.step(z_local, y_local, 1.0, x_local, 0.0)

        def step_allgather():
            op_allgather.step(z_local, y_local, 1.0, x_local, 0.0)

        def step_needed():
            op_needed.step(z_local, y_local, 1.0, x_local, 0.0)

        def products_only():
            if rest is None:
                local_product(own, z_local, y_local, 1.0, op.x_full, 0.0)
            else:
                local_product(own, z_local, y_local, 1.0, x_local, 0.0)
                if needed_mode:
                    local_product(rest_compact, z_local, z_local, 1.0, op.needed.x_needed, 1.0)
                else:
                    local_product(rest, z_local, z_local, 1.0, op.x_full, 1.0)

        def gather_only():
            w = op_allgather.gather_x(x_local, async_op=False)
            if w is not None:
                w.wait()

        def needed_only():
            w = op_needed.needed.start(x_local, async_op=False)
            if w is not None:
                w.wait()

        rows_received = lambda: (sum(op_needed.needed.recv_splits) - op_needed.needed.recv_splits[rank]) if both else (world - 1) * rows_local
    torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # untimed: the local products alone (on whatever the exchange buffer holds) and the exchanges alone
    def timed(fn, reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            fn()
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
        b.synchronize()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    t_compute = sorted(timed(products_only, 10) for _ in range(3))[1]   # median of three blocks of 10
    t_gather = timed(gather_only, 10) if distributed and (split or world > 1) else 0.0
    t_needed = timed(needed_only, 10) if both and distributed else 0.0
    # the other exchange's step, measured in the same process (same products; what differs is what moves)
    t_step_other = 0.0
    if both and distributed:
        t_step_other = timed(step_allgather if needed_mode else step_needed, max(3, steps // 10))
        if world > 1:
            torch.cuda.synchronize()
            t = torch.tensor([t_step_other], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_step_other = float(t.item())
    t_step_allgather = t_step_other if needed_mode else 0.0
    # the step under test last, for the parity check below
    with torch.cuda.stream(stream):
        step()
    torch.cuda.synchronize()
    received = rows_received() if both else None
    for plan in plans.values():
        capi.spgpuDhellspmmShardedDestroy(plan)
    if comm is not None:
        capi.spgpuCommDestroy(comm)
    nnz_local = rows_local * L
    hacks = rows_local // 32
    alg = hell_algorithmic_bytes(nnz_local, rows_local, n_total, hacks, rhs=k)
    flops_total = 2.0 * nnz_local * k * world

    # parity, on EVERY rank: a window of its own rows against the oracle (bit for bit when the block is not split), with the X of
    # every rank regenerated from the ranks' seeds -- the first rows of a rank's block name columns of its neighbour's block, so a
    # rank whose exchange delivered the wrong rows (or none) fails here.  The first real RCCL run of this code is the driver's
    # scale run: the line says how many ranks were seen and how many of them checked out.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_api as O
    whole = synth.hell_uniform_on_device(rows_local, L, args.spmm_pattern, "D", 32, seed=11 + rank, device=dev,
                                         n_cols=n_total, row_offset=first) if split else own
    x_everywhere = torch.cat([synth.device_vector(rows_local * k, "D", 21 + r, dev).view(rows_local, k) for r in range(world)])
    torch.cuda.synchronize()
    sub = synth.hell_rows_to_host(whole, 0, 1024)
    want = O.hell_spmm(sub, x_everywhere.cpu().numpy(), None, 1.0, 0.0)
    del x_everywhere
    got = z_local[:1024].cpu().numpy()
    if split:
        mine_ok = bool(np.max(np.abs(got - want) / (np.abs(want) + 1.0)) <= 1e-12)
        parity = "within 1e-12 of oracle on 1024 rows (own+rest regroup the sums)" if mine_ok else "MISMATCH"
    else:
        mine_ok = got.tobytes() == want.tobytes()
        parity = "bit-exact vs oracle on 1024 rows" if mine_ok else "MISMATCH"
    ranks_seen, ranks_ok = 1, int(mine_ok)
    if world > 1:
        # one small all-gather of (rank, did my window check out) over the same backend as the run: who is there, who agrees
        mine = torch.tensor([float(rank), 1.0 if mine_ok else 0.0], device=dev, dtype=torch.float64)
        everyone_said = torch.zeros(2 * world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(everyone_said, mine)
        torch.cuda.synchronize()
        said = everyone_said.cpu().numpy().reshape(world, 2)
        ranks_seen = int(np.unique(said[:, 0]).size)
        ranks_ok = int(said[:, 1].sum())
        if ranks_ok != world:
            parity += f"; {world - ranks_ok} of {world} ranks MISMATCH"

    if rank == 0:
        out = dict(
            metric=(f"row-sharded HELL fp64 SpMM GFLOP/s (A x {k} rhs, weak scaling, {world} GPU" + ("s" if world > 1 else "") + ", "
                    + ("needed X rows by RCCL all_to_all" if needed_mode else "RCCL all-gather(X)") + " per step) + achieved HBM GB/s of the "
                    "local product (% of roofline)"),
            value=round(flops_total * steps / wall * 1e-9, 2), unit="GFLOP/s", n_gpus=world, steps=steps,
            warmup=warmup, ms_per_step=round(wall / steps * 1e3, 5), higher_is_better=True, scaling="weak",
            vs_baseline=None, dtype="f64", data="synthetic",
            config=dict(workload=f"row-sharded HELL fp64 SpMM (spgpuDhellspmm), {rows_local} rows/GPU x {L} nnz/row x {k} rhs, "
                                 f"{n_total} rows total, columns {args.spmm_pattern}, "
                                 + ("needed X rows per step by RCCL all_to_all" if needed_mode else "RCCL all-gather(X) per step")
                                 + " (BASELINE configs[4] at 8 GPUs)",
                        rows_per_gpu=rows_local, rows_total=n_total, rhs=k, pattern=args.spmm_pattern,
                        exchange="needed rows (all_to_all)" if needed_mode else "all-gather",
                        parallelism=f"row partition x{world}, " + ("needed-rows exchange" if needed_mode else "all-gather of X")
                                    + (", own/rest column split (overlap)" if split else "")),
            roofline=dict(bound="hbm", achieved=round(alg / t_compute * 1e-9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                          frac=round(alg / t_compute * 1e-9 / HBM_PEAK_GBS, 4),
                          traffic=None if split else committed_traffic(rows_local, L, args.spmm_pattern, rhs=k),
                          kernel="hellSpmmStripKernel<double, 2, 2>",
                          algorithmic_bytes_per_launch=alg, kernel_ms=round(t_compute * 1e3, 4)),
            spmm=dict(compute_only_ms=round(t_compute * 1e3, 4), allgather_only_ms=round(t_gather * 1e3, 4),
                      compute_only_gflops_total=round(flops_total / t_compute * 1e-9, 1),
                      allgather_GBps_per_rank=round((world - 1) * rows_local * k * 8 / t_gather * 1e-9, 1) if t_gather else None,
                      needed_rows_only_ms=round(t_needed * 1e3, 4) if both and t_needed else None,
                      needed_rows_received_per_rank=received, driver=driver,
                      rccl_ranks_seen=ranks_seen, ranks_whose_window_matches_the_oracle=ranks_ok,
                      exchange_bytes_per_rank=dict(allgather=(world - 1) * rows_local * k * 8,
                                                   needed_rows=(received * k * 8 if received is not None else None)),
                      allgather_step_ms=round(t_step_allgather * 1e3, 4) if t_step_allgather else None,
                      allgather_step_gflops_total=round(flops_total / t_step_allgather * 1e-9, 1) if t_step_allgather else None),
            parity=parity, cpu_baseline=None)
        if world > 1:
            out["scaling_base"] = ("this line is the row-sharded SpMM of BASELINE configs[4]; its N = 1 point is `spmm_1gpu.value` of the "
                                   "`--gpus 1` line (same 5 M rows x 32 x 16 rhs per GPU, no exchange), NOT that line's `value`, which is the "
                                   "single-GPU SpMV of configs[1] -- single-vector SpMV does not shard (replicas only, DESIGN.md section 6); "
                                   "`spmm.compute_only_gflops_total` here is the same products on all ranks without the exchange")
        out["config"]["driver"] = ("C ABI (spgpuDhellspmmShardedStep: packing kernel, RCCL and products issued by libspgpu.so)" if driver == "c"
                                   else "torch.distributed collectives + C-ABI products")
        if t_step_allgather:
            # BASELINE configs[4] word for word (all-gather of the dense X), measured in this same process
            out["as_named_allgather"] = dict(value=round(flops_total / t_step_allgather * 1e-9, 2), unit="GFLOP/s",
                                             ms_per_step=round(t_step_allgather * 1e3, 5),
                                             bytes_received_per_rank=(world - 1) * rows_local * k * 8,
                                             note="same products, whole X all-gathered per step; `value` above moves only the X rows "
                                                  "the off-block columns name (DESIGN.md section 6)")
        if t_step_other and not needed_mode:
            # the better engineering, not the named configuration: only the X rows the off-block columns name travel
            out["variants"] = dict(needed_rows=dict(
                value=round(flops_total / t_step_other * 1e-9, 2), unit="GFLOP/s", ms_per_step=round(t_step_other * 1e3, 5),
                rows_received_per_rank=received, bytes_received_per_rank=(received or 0) * k * 8,
                note="same products as `value`; one all_to_all of the needed X rows instead of the all-gather of all of X "
                     f"({(world - 1) * rows_local * k * 8} bytes per rank and step)"))
        return out
    return None


def rehearsing():
    return os.environ.get("SPGPU_BENCH_BACKEND") == "gloo"


def local_device():
    return "cuda:0" if rehearsing() else f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"


def collectives():
    import torch.distributed as dist
    return HostStagedCollectives(dist) if rehearsing() else dist


class HostStagedCollectives:
    """Rehearsal only (SPGPU_BENCH_BACKEND=gloo): the collectives bench.py uses, staged through host memory over gloo,
    so that several ranks can share the one GPU of a test box (RCCL refuses two ranks on one device).  Same call
    signatures as torch.distributed; nothing here is timed for the record."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def barrier(self):
        self._d.barrier()

    def all_reduce(self, t, op=None):
        import torch
        torch.cuda.synchronize()
        h = t.cpu()
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp, async_op=False):
        import torch
        torch.cuda.synchronize()
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self._d.get_world_size())]
        self._d.all_gather(parts, inp.cpu())
        out.copy_(torch.cat(parts, 0).view(out.shape))
        return None

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        import torch
        torch.cuda.synchronize()
        h = torch.empty(out.shape, dtype=out.dtype)
        self._d.all_to_all_single(h, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
        out.copy_(h)
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU call,
    let the child's rank 0 print the JSON line on the shared stdout, and exit with the child's code.  Nothing is
    exec'd and this parent never touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:       # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        # a record that says n_gpus = world while the caller asked for --gpus N would be void: refuse
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # SPGPU_BENCH_FORCE_DIST=1: initialise RCCL and run the collectives even with one rank (rehearsal of the
    # multi-GPU code path on a 1-GPU box)
    force_dist = os.environ.get("SPGPU_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if not rehearsing() and torch.cuda.device_count() < world:
            sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_device())
        dist.init_process_group("gloo" if rehearsing() else "nccl")
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        world = dist.get_world_size()
    workload = args.workload if args.workload != "auto" else ("spmv" if world == 1 else "spmm")
    try:
        (run_spmv if workload == "spmv" else run_spmm)(args, rank, world)
    finally:
        if world > 1 or force_dist:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
