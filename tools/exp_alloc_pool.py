#!/usr/bin/env python3
"""GPU box experiment (follows exp_alloc.py): ONE hipMalloc pool, the matrix' arrays placed inside it at chosen offsets -- do the
offsets between the arrays (their alignment to 2 MiB, the distance between the coefficient and the index stream) move the time,
or only WHICH physical memory the allocation got (exp_alloc.py: the same data in another allocation: 0.68 .. 0.76 ms)?
    python tools/exp_alloc_pool.py [rows]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
coo = synth.ragged_coo_on_device(lengths, n, os.environ.get("EXP_PATTERN", "band"), 2048, "D", seed=5)
h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, 2048, 256, aligned=True)
del coo
x0 = synth.device_vector(n, "D", 3)
torch.cuda.synchronize()
alg = h["nnz"] * 12 + n * 12 + n * 8 + (n // 32) * 4 + n * 4
hip = C.CDLL("libamdhip64.so")
MB2 = 2 << 20
names = ("cM", "rP", "hack_offsets", "rS", "rIdx")
src = {k: h[k] for k in names}
src["x"], src["z"] = x0, x0
size = {k: t.numel() * t.element_size() for k, t in src.items()}
up = lambda v, a: (v + a - 1) // a * a


def layout(shift):
    """byte offset of every array in the pool: 2 MiB-aligned starts, plus shift[name]"""
    at, off = 0, {}
    for k in ("cM", "rP", "hack_offsets", "rS", "rIdx", "x", "z"):
        at = up(at, MB2)
        off[k] = at + shift.get(k, 0)
        at = off[k] + size[k]
    return off, at


cases = [("all 2 MiB aligned", {}), ("rP + 0x9b000", {"rP": 0x9b000}), ("rP + 64 KiB", {"rP": 65536}), ("rP + 4 KiB", {"rP": 4096}),
         ("rP + 256 B", {"rP": 256}), ("cM + 1 MiB", {"cM": 1 << 20}), ("x + 0x18c00", {"x": 0x18c00}), ("all 2 MiB aligned (again)", {})]
pool_bytes = max(layout(s)[1] for _, s in cases) + MB2
for attempt in range(int(os.environ.get("EXP_POOLS", "2"))):
    pool = C.c_void_p()
    assert hip.hipMalloc(C.byref(pool), C.c_size_t(pool_bytes)) == 0
    base = up(pool.value, MB2)
    for label, shift in cases:
        off, _ = layout(shift)
        for k, t in src.items():
            assert hip.hipMemcpy(C.c_void_p(base + off[k]), C.c_void_p(t.data_ptr()), C.c_size_t(size[k]), 3) == 0
        torch.cuda.synchronize()
        P = lambda k: C.c_void_p(base + off[k])
        call = lambda: capi.hellspmv["D"](handle, P("z"), None, 1.0, P("cM"), P("rP"), 32, P("hack_offsets"), P("rS"), P("rIdx"), 32, n, P("x"), 0.0, 0)
        times = []
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
                times.append(a.elapsed_time(b) / 20)
        print(f"pool {attempt} @ {pool.value:#x}  {label:28s} " + " ".join(f"{t:.4f}" for t in times) + f" ms  best {alg / min(times) * 1e-6 / 8000:.3f}", flush=True)
    hip.hipFree(pool)
