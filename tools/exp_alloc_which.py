#!/usr/bin/env python3
"""GPU box experiment (follows exp_alloc.py): WHICH array's placement moves the time of the ordered power-law SpMV?  All arrays in
allocations of their own (hipMalloc); then one array at a time is moved to a new allocation (the old one is freed afterwards, so the
new one cannot land on the same memory) and the same 3 x 20 launches are timed.  EXP_KIND=uniform: the banded 32-per-row headline matrix
through the default kernel instead (is that kernel as sensitive?).
    python tools/exp_alloc_which.py [rows] [rounds]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = os.environ.get("EXP_KIND", "powerlaw")
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
if kind == "powerlaw":
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    coo = synth.ragged_coo_on_device(lengths, n, os.environ.get("EXP_PATTERN", "band"), 2048, "D", seed=5)
    h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, 2048, 256, aligned=True)
    del coo
else:
    h = synth.hell_uniform_on_device(n // 32 * 32, 32, "banded", "D", 32, seed=1)
    h["rIdx"] = None
x0 = synth.device_vector(n, "D", 3)
torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")
src = {k: h[k] for k in ("cM", "rP", "hack_offsets", "rS", "rIdx") if h.get(k) is not None}
src["x"], src["z"] = x0, x0


def fresh(t):
    p, size = C.c_void_p(), t.numel() * t.element_size()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(size)) == 0
    assert hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), C.c_size_t(size), 3) == 0
    return p


at = {k: fresh(t) for k, t in src.items()}
P = lambda k: at[k] if k in at else None
call = lambda: capi.hellspmv["D"](handle, P("z"), None, 1.0, P("cM"), P("rP"), 32, P("hack_offsets"), P("rS"), P("rIdx"), 32, n, P("x"), 0.0, 0)


def measure():
    torch.cuda.synchronize()
    ts = []
    with torch.cuda.stream(stream):
        for _ in range(4):
            call()
            stream.synchronize()
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(20):
                call()
            b.record(stream)
            b.synchronize()
            ts.append(a.elapsed_time(b) / 20)
    return sorted(ts)[1]


last = measure()
print(f"{kind}: all arrays placed: {last:.4f} ms", flush=True)
groups = [("cM",), ("rP",), ("x",), ("z",), tuple(k for k in ("hack_offsets", "rS", "rIdx") if k in at)]
for r in range(rounds):
    for group in groups:
        old = {k: at[k] for k in group}
        for k in group:
            at[k] = fresh(src[k])
        now = measure()
        print(f"round {r}: moved {'+'.join(group):24s} {last:.4f} -> {now:.4f} ms  ({(now / last - 1) * 100:+.1f} %)   new @ " + " ".join(f"{at[k].value:#x}" for k in group), flush=True)
        last = now
        for p in old.values():
            hip.hipFree(p)
