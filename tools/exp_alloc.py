#!/usr/bin/env python3
"""GPU box experiment: does the time of ONE kernel on ONE matrix depend on which allocations hold the matrix?  The ordered
power-law target (band columns, aligned order) is built once; its arrays are then copied into fresh allocations several times
(EXP_ALLOC=torch: torch.empty_like, the old copies freed and the cache emptied in between; EXP_ALLOC=hip: hipMalloc through
ctypes, one allocation per array; EXP_ALLOC=contig: hipExtMallocWithFlags(hipDeviceMallocContiguous); EXP_KIND=uniform: the headline
matrix through the default kernel) and the same 20 calls are timed on every copy, 3 timed blocks each.
    python tools/exp_alloc.py [rows] [copies]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mode = os.environ.get("EXP_ALLOC", "torch")
pattern = os.environ.get("EXP_PATTERN", "band")
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
if os.environ.get("EXP_KIND", "powerlaw") == "uniform":
    h = synth.hell_uniform_on_device(n // 32 * 32, 32, "banded", "D", 32, seed=1)
    h["rIdx"] = torch.zeros(1, dtype=torch.int32, device="cuda")   # (not passed)
else:
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    coo = synth.ragged_coo_on_device(lengths, n, pattern, 2048, "D", seed=5)
    h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, 2048, 256, aligned=True)
    del coo
uniform = os.environ.get("EXP_KIND", "powerlaw") == "uniform"
x0 = synth.device_vector(n, "D", 3)
torch.cuda.synchronize()
alg = h["nnz"] * 12 + n * 12 + n * 8 + (n // 32) * 4 + n * 4
hip = C.CDLL("libamdhip64.so") if mode in ("hip", "contig") else None


def fresh(t):
    if mode in ("hip", "contig"):
        p = C.c_void_p()
        if mode == "contig":   # hipDeviceMallocContiguous (0x4): physically contiguous device memory
            assert hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(t.numel() * t.element_size()), C.c_uint(4)) == 0
        else:
            assert hip.hipMalloc(C.byref(p), C.c_size_t(t.numel() * t.element_size())) == 0
        assert hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), C.c_size_t(t.numel() * t.element_size()), 3) == 0
        return p
    c = torch.empty_like(t)
    c.copy_(t)
    return c


addr = lambda v: v.value if isinstance(v, C.c_void_p) else v.data_ptr()
held = []
for k in range(copies):
    arrays = {name: fresh(h[name]) for name in ("cM", "rP", "hack_offsets", "rS", "rIdx")}
    x = fresh(x0)
    z = fresh(x0)
    torch.cuda.synchronize()
    call = lambda: capi.hellspmv["D"](handle, C.c_void_p(addr(z)), None, 1.0, C.c_void_p(addr(arrays["cM"])), C.c_void_p(addr(arrays["rP"])), 32,
                                      C.c_void_p(addr(arrays["hack_offsets"])), C.c_void_p(addr(arrays["rS"])), None if uniform else C.c_void_p(addr(arrays["rIdx"])), 32, n,
                                      C.c_void_p(addr(x)), 0.0, 0)
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
    print(f"copy {k} ({mode}): " + " ".join(f"{t:.4f}" for t in times) + f" ms  best {alg / min(times) * 1e-6 / 8000:.3f} of 8 TB/s   cM @ {addr(arrays['cM']):#x} rP @ {addr(arrays['rP']):#x} "
          f"x @ {addr(x):#x}  plans {capi.plan_counts(handle)}", flush=True)
    if os.environ.get("EXP_KEEP"):
        held.append((arrays, x, z))      # keep the copies alive: every new copy lands somewhere else
    elif mode == "torch":
        del arrays, x, z
        torch.cuda.empty_cache()
    elif os.environ.get("EXP_FREE_LATER"):
        held.append((arrays, x, z))
        if len(held) > 1:      # free the PREVIOUS copy only now: the new one cannot have landed on its memory
            for v in list(held[0][0].values()) + [held[0][1], held[0][2]]:
                hip.hipFree(v)
            held.pop(0)
    else:
        for v in list(arrays.values()) + [x, z]:
            hip.hipFree(v)
