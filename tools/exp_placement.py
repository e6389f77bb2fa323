#!/usr/bin/env python3
"""GPU box experiment: does the time of the ordered power-law SpMV depend on WHERE its arrays lie?  One matrix, the
coefficient and index arrays copied to different offsets inside one big buffer, 20 launches each."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

n = 10_000_000
pattern = sys.argv[1] if len(sys.argv) > 1 else "band"
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
coo = synth.ragged_coo_on_device(lengths, n, pattern, 2048, "D", seed=5)
h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, 2048, 256)
del coo
torch.cuda.empty_cache()
x = synth.device_vector(n, "D", 3)
z = torch.zeros(n, dtype=torch.float64, device="cuda")
slots = h["slots"]
big = torch.empty(slots * 12 + (64 << 20), dtype=torch.uint8, device="cuda")


def timed(cM, rP):
    call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(cM), p(rP), 32, p(h["hack_offsets"]), p(h["rS"]), p(h["rIdx"]), 32, n,
                                      p(x), 0.0, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(3):
            call()
        a.record(stream)
        for _ in range(20):
            call()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / 20


print(f"{pattern}: as allocated  cM {h['cM'].data_ptr():#x} rP {h['rP'].data_ptr():#x}  {timed(h['cM'], h['rP']):.4f} ms", flush=True)
for base in (0,):
    for gap in (0, 1 << 20):
        cM = big[base:base + slots * 8].view(torch.float64)
        start = base + slots * 8 + gap
        rP = big[start:start + slots * 4].view(torch.int32)
        cM.copy_(h["cM"][:slots])
        rP.copy_(h["rP"][:slots])
        torch.cuda.synchronize()
        print(f"  base +{base:>8}  gap {gap:>9}  {timed(cM, rP):.4f} ms", flush=True)
# z (then x) at different offsets inside one 512 MiB buffer, everything else fixed
pool = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
offsets = [0, 128, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 16 << 20, 17 << 20, 33 << 20, 64 << 20, 96 << 20, 130 << 20, 200 << 20, 256 << 20]


def timed_with(xv, zv):
    call = lambda: capi.hellspmv["D"](handle, p(zv), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), p(h["rIdx"]),
                                      32, n, p(xv), 0.0, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(3):
            call()
        a.record(stream)
        for _ in range(20):
            call()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / 20


for what in ("z", "x"):
    for off in offsets:
        view = pool[off:off + n * 8].view(torch.float64)
        if what == "x":
            view.copy_(x)
        torch.cuda.synchronize()
        t = timed_with(view if what == "x" else x, view if what == "z" else z)
        print(f"  {what} at pool + {off:>10} ({view.data_ptr():#x})  {t:.4f} ms", flush=True)
capi.spgpuDestroy(handle)
