#!/usr/bin/env python3
"""GPU box experiment: the ways the ELL/HELL SpMV fetches x (gathers / strips / LDS tile, spgpuSetSpmvForm) on column
patterns between "consecutive" and "scattered", and on power-law row lengths before and after ordering the rows by
length (spgpuOellOrderDevice).  Every timing is followed by an oracle check of three row windows.

    python tools/exp_tile.py [D|S] [rows] [cases: uniform,mild,powerlaw]

Environment: EXP_FORMS = comma list of  auto | gather | strips | tileN (x-tile shape N) | raggedN (the kernel for ordered
rows, workgroup shape N) | raggedg (the same with plain gathers); a suffix xR (ragged0x4) runs it with R consecutive row
blocks per XCD.  EXP_PATTERNS = near,band,random (power-law) or near2048,near512,window,banded (uniform);
EXP_ORDERS = window:longRows pairs (2048:256,...); EXP_ONLY_WINDOWED=1 skips the plain and globally sorted layouts;
EXP_WINDOWS_FOR_ALL=1 runs the windowed orders on scattered columns too; EXP_ALIGNED=1 orders the rows with spgpuOellOrderAlignedDevice
(windows counted among the short rows: one window = one workgroup);
EXP_DROP_RIDX=1 runs the ordered matrix without its row order (timing only); EXP_FREEZE=1 freezes every ordered matrix first
(spgpuHellSpmvFreeze: 16-bit column indices); EXP_ADOPT=1 adopts every matrix that comes without an order (spgpuHellSpmvAdopt: the check
then reports a mismatch -- it compares with the plain kernel's bits; timing and kernel names are what this is for).  SPGPU_* knobs pass through.
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
FORMS = {"auto": 0, "gather": 1, "strips": 2, "tile0": 3, "tile1": 3, "tile2": 3, "tile3": 3, "sweep": 4}
DEEP_CAP = int(os.environ.get("SPGPU_DEEP_CAP", "256"))


def spgpu_lab():
    """a -DSPGPU_TUNING_VARIANTS build keeps the older kernels for ordered rows selectable (SPGPU_RAGGED=0)"""
    return bool(capi.spgpuTuningVariantsBuilt()) and os.environ.get("SPGPU_RAGGED", "1") == "0"


def shape_of(form, ordered):
    """spmv_tail parameters of the kernel that runs (tests/oracle_api.py slab_shape); a row order switches the deep split on"""
    deep = DEEP_CAP if (ordered and os.environ.get("SPGPU_DEEP_SPLIT", "-1") != "0") or os.environ.get("SPGPU_DEEP_SPLIT") == "1" else 0
    if form == "sweep":
        return dict(group_rows=64, rows_per_lane=1, step=1, tail_lanes=0, phases=1)      # ascending k, nothing else
    if form.startswith(("share", "pipe")):
        return O.slab_shape(letter, "share")
    if form.startswith("ragged") or (form in ("auto", "gather") and ordered and not spgpu_lab()):     # with a row order the product runs the queue kernel in every form
        return O.slab_shape(letter, "ragged", 0, deep_cap=deep, split=int(os.environ.get("SPGPU_RAGGED_SPLIT", "-1")))
    if form.startswith("tile"):
        return O.slab_shape(letter, "xtile", int(form[4:]), deep_cap=deep)
    return O.slab_shape(letter, "gather", 0, deep_cap=deep)


def check(h, form, windows=3, rows=2048):
    shape = shape_of(form, h.get("rIdx") is not None)
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
    r_idx = h.get("rIdx")
    if os.environ.get("EXP_IDENTITY_RIDX") and r_idx is not None:      # timing only: same kernel, z written in row order
        r_idx = torch.arange(rows, dtype=torch.int32, device="cuda")
    if os.environ.get("EXP_IDENTITY_FOR_PLAIN") and r_idx is None:     # timing only: rows as they come through the kernels for ordered rows
        r_idx = torch.arange(rows, dtype=torch.int32, device="cuda")
    call = lambda: capi.hellspmv[letter](handle, p(z), None, one, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                         None if os.environ.get("EXP_DROP_RIDX") else p(r_idx), 32, rows, p(x), zero, 0)
    if os.environ.get("EXP_FREEZE") and r_idx is not None:             # spgpuHellSpmvFreeze first: 16-bit column indices (include/spgpu/tuning.h)
        said = capi.spgpuHellSpmvFreeze(handle, capi.TYPE_CODE[letter], p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), p(r_idx), rows, 0)
        label = f"{label} FROZEN({said})"
    if os.environ.get("EXP_ADOPT") and r_idx is None:                   # spgpuHellSpmvAdopt first: the library's ordered, frozen copy (include/spgpu/tuning.h)
        said = capi.spgpuHellSpmvAdopt(handle, capi.TYPE_CODE[letter], p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), rows, 0)
        label = f"{label} ADOPTED({said})"
    for full in forms:
        name = full
        full, _, keep = full.partition("%")         # "auto%32": SPGPU_DEEP_KEEP for this run (same matrix, same process)
        os.environ["SPGPU_DEEP_KEEP"] = keep or os.environ.get("EXP_DEEP_KEEP", "64")
        full, _, split = full.partition("@")        # "ragged0@0": sub-groups never cut; "ragged4@48": chunks of 48 columns
        os.environ["SPGPU_RAGGED_SPLIT"] = split or "-1"
        form, _, xcd = full.partition("x")          # "ragged0x4": shape 0 with runs of 4 row blocks per XCD
        os.environ["SPGPU_XCD_ORDER"] = xcd or os.environ.get("EXP_XCD_ORDER", "0")
        os.environ["SPGPU_X_TILE_SHAPE"] = form[4:] if form.startswith("tile") else "0"
        os.environ["SPGPU_RAGGED"] = "3" if form.startswith("pipe") else "2" if form.startswith("share") else "1" if form.startswith("ragged") else "0"      # raggedN / shareN: shape N with the tile; raggedg / shareg: gathers
        os.environ["SPGPU_RAGGED_SHAPE"] = form[6:] if form.startswith("ragged") and form[6:].isdigit() else form[5:] if form.startswith("share") and form[5:].isdigit() else form[4:] if form.startswith("pipe") and form[4:].isdigit() else "0"
        capi.spgpuTuningReload()
        capi.spgpuSetSpmvForm(handle, 1 if form in ("raggedg", "shareg", "pipeg") else 0 if form.startswith(("ragged", "share", "pipe")) else 3 if form.startswith("tile") else FORMS[form])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            for _ in range(3):
                call()
                stream.synchronize()   # AUTO reads what the previous launch's sample wavefronts reported
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
        print(f"{letter} {label:46s} {name:10s} slots/nnz {h['slots'] / h['nnz']:.3f}  {t:.4f} ms  {alg / t * 1e-6:7.1f} GB/s  "
              f"{alg / t * 1e-6 / 8000:.3f} of 8 TB/s  {check(h, form)}  plans(uses,builds,stales)={capi.plan_counts(handle)}", flush=True)
    capi.spgpuSetSpmvForm(handle, 0)
    # EXP_SWEEP="SPGPU_PLAN_DEEP_SPREAD=0,30,60;SPGPU_PLAN_DEEP_PER_BLOCK=2,4": every combination on THIS matrix in THESE allocations
    # (the time of a kernel moves by several per cent with the placement of the arrays: A/B only inside one process), round-robin,
    # EXP_SWEEP_REPS times
    if os.environ.get("EXP_SWEEP"):
        import itertools
        knobs = [(part.split("=")[0], part.split("=")[1].split(",")) for part in os.environ["EXP_SWEEP"].split(";")]
        combos = list(itertools.product(*[values for _, values in knobs]))
        seen = {c: [] for c in combos}
        before = {name: os.environ.get(name) for name, _ in knobs}
        for rep in range(int(os.environ.get("EXP_SWEEP_REPS", "3"))):
            for combo in combos:
                for (name, _), value in zip(knobs, combo):
                    os.environ[name] = value
                capi.spgpuTuningReload()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(stream):
                    for _ in range(3):
                        call()
                        stream.synchronize()
                    a.record(stream)
                    for _ in range(20):
                        call()
                    b.record(stream)
                b.synchronize()
                seen[combo].append(a.elapsed_time(b) / 20)
        for name, value in before.items():      # the next matrix is run (and checked) with the knobs as they were
            if value is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = value
        capi.spgpuTuningReload()
        for combo in combos:
            ts = seen[combo]
            print(f"    sweep {label[:40]:40s} " + " ".join(f"{n}={v}" for (n, _), v in zip(knobs, combo)) + "  " + " ".join(f"{t:.4f}" for t in ts)
                  + f"  median {sorted(ts)[len(ts) // 2]:.4f} ms = {alg / sorted(ts)[len(ts) // 2] * 1e-6 / 8000:.3f}", flush=True)


if "uniform" in cases:
    for pattern in os.environ.get("EXP_PATTERNS", "near2048,near512,window,banded").split(","):
        h = synth.hell_uniform_on_device(n // 32 * 32, 32, pattern, letter, 32, seed=1)
        h["slots"] = h["nnz"]
        torch.cuda.synchronize()
        run(h, f"uniform 32/row, columns {pattern}", os.environ.get("EXP_FORMS", "gather,strips,tile0,tile2,tile3,tile4,tile5").split(","))
        del h
        torch.cuda.empty_cache()

if "mild" in cases:
    # rows of 24..40 entries near the row, ordered by length inside windows: the permutation and the scattered z writes of
    # the ordered power-law case without its long rows
    rng = np.random.default_rng(1)
    lengths = rng.integers(24, 41, size=n).astype(np.int32)
    mild_pattern = os.environ.get("EXP_MILD_PATTERN", "near")
    rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, mild_pattern, 2048, letter, seed=5)
    torch.cuda.synchronize()
    for name, order in ([] if os.environ.get("EXP_ONLY_WINDOWED") else [("plain", None)]) + [(f"sorted window {w}", (w, 0)) for w in (int(v) for v in os.environ.get("EXP_MILD_WINDOWS", "1024,2048,4096").split(","))]:
        h = formats.coo_to_ordered_hell_device(handle, n, rows_t, cols_t, vals_t, letter, 32, *(order or (0, 0)), order=order is not None)
        run(h, f"24..40/row {mild_pattern}, {name}", os.environ.get("EXP_FORMS", "gather,tile0,tile2,tile3").split(","))
        del h
        torch.cuda.empty_cache()
    del rows_t, cols_t, vals_t
    torch.cuda.empty_cache()

def aligned_order(lengths, window, long_rows):
    """Experiment: the long rows first (windows of 32 * window original rows), the others in windows of `window` rows
    COUNTED AMONG THE SHORT ROWS and placed so that window boundaries are multiples of `window` in the new order."""
    L = np.asarray(lengths, dtype=np.int64)
    rows = np.arange(L.size, dtype=np.int64)
    is_long = L > long_rows if long_rows > 0 else np.zeros(L.size, bool)
    P = int(is_long.sum())
    group = np.empty(L.size, np.int64)
    group[is_long] = rows[is_long] // (32 * window)
    long_groups = int(group[is_long].max()) + 1 if P else 0
    rank = np.cumsum(~is_long) - 1
    in_class = (P + rank[~is_long]) // window - P // window
    group[~is_long] = long_groups + in_class
    sign = np.where((np.where(is_long, group, group - long_groups) % 2) == 0, -1, 1)
    return np.lexsort((rows * sign, L * sign, group)).astype(np.int32)


if "powerlaw" in cases:
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=int(os.environ.get("EXP_MAXLEN", "2048")), seed=5)   # EXP_MAXLEN: where do the long rows' costs start?
    for pattern in os.environ.get("EXP_PATTERNS", "near,random").split(","):
        rows_t, cols_t, vals_t = synth.ragged_coo_on_device(lengths, n, pattern, int(os.environ.get("EXP_NEAR", "2048")), letter, seed=5)   # EXP_NEAR: half-width of the "near" pattern
        torch.cuda.synchronize()
        orders = [("plain", None), ("sorted all", (0, 0))]
        if pattern in ("near", "band") or os.environ.get("EXP_WINDOWS_FOR_ALL"):
            pairs = [tuple(int(v) for v in item.split(":")) for item in os.environ.get("EXP_ORDERS", "2048:128,2048:256,4096:256,8192:256,16384:0").split(",")]
            # "window:long" or "window:long:cap" (SPGPU_DEEP_CAP for the runs on that order; without it the environment's)
            orders += [(f"sorted window {p[0]} long>{p[1]}" + (f" cap {p[2]}" if len(p) > 2 else ""), p) for p in pairs]
        if os.environ.get("EXP_ONLY_WINDOWED"):
            orders = orders[2:]
        if os.environ.get("EXP_ONLY_PLAIN"):
            orders = orders[:1]
        for name, order in orders:
            if order and len(order) > 2:
                os.environ["SPGPU_DEEP_CAP"] = str(order[2])
                DEEP_CAP = order[2]
                order = order[:2]
            aligned = bool(os.environ.get("EXP_ALIGNED") and order and order[0] > 0)
            if aligned:
                name += " ALIGNED"
            h = formats.coo_to_ordered_hell_device(handle, n, rows_t, cols_t, vals_t, letter, 32, *(order or (0, 0)),
                                                   order=order is not None, aligned=aligned)
            if aligned and os.environ.get("EXP_ALIGNED") == "check":    # the device order against this file's numpy statement of it
                assert h["rIdx"].cpu().numpy().tolist() == aligned_order(lengths, *order).tolist()
            forms = os.environ.get("EXP_GLOBAL_FORMS", "gather").split(",") if ((pattern == "random" and not os.environ.get("EXP_WINDOWS_FOR_ALL")) or order is None or order == (0, 0)) else os.environ.get("EXP_FORMS", "gather,tile0,tile2,tile3").split(",")
            run(h, f"power-law {pattern}, {name}", forms)
            del h
            torch.cuda.empty_cache()
        del rows_t, cols_t, vals_t
        torch.cuda.empty_cache()
capi.spgpuDestroy(handle)
