#!/usr/bin/env python3
"""GPU box experiment (needs the -DSPGPU_TRACE_BLOCKS build, SPGPU_LIB=...): when do the workgroups of raggedSpmvKernel
start and end on the ordered power-law matrix?  Prints the number of resident workgroups over time, and the duration of
the workgroups by the kind of rows they hold."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
window, long_rows = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2048:256").split(":"))
case = sys.argv[3] if len(sys.argv) > 3 else "powerlaw"
handle = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
if case == "powerlaw":
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
else:
    lengths = np.random.default_rng(1).integers(24, 41, size=n).astype(np.int32)
coo = synth.ragged_coo_on_device(lengths, n, os.environ.get("EXP_PATTERN", "near"), 2048, "D", seed=5)
h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, window, long_rows)
x = synth.device_vector(n, "D", 3)
z = torch.zeros(n, dtype=torch.float64, device="cuda")
rows_per_block = 1024 * (2 if os.environ.get("SPGPU_RAGGED_SHAPE") == "1" else 1)
blocks = (n + rows_per_block - 1) // rows_per_block
trace = torch.zeros(3 * blocks + 16, dtype=torch.int64, device="cuda")
capi.lib.spgpuDebugSetTrace.argtypes = [C.c_void_p]
call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), p(h["rIdx"]), 32, n,
                                  p(x), 0.0, 0)
for _ in range(3):
    call()
torch.cuda.synchronize()
trace.zero_()
torch.cuda.synchronize()
capi.lib.spgpuDebugSetTrace(p(trace))
call()
torch.cuda.synchronize()
capi.lib.spgpuDebugSetTrace(None)
t = trace[:3 * blocks].view(blocks, 3).cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # microseconds (100 MHz clock)
print(f"kernel span {end.max():.1f} us, {blocks} workgroups, sum of workgroup times {np.sum(end - start) / 1e3:.1f} ms "
      f"= {np.sum(end - start) / end.max():.1f} resident on average")
edges = np.linspace(0, end.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    print(f"  t = {mid:7.1f} us  resident workgroups {int(np.sum((start <= mid) & (end > mid))):4d}")
dur = end - start
prologue = (t[:, 2] - t0) / 100.0 - start
print(f"  prologue (start -> tile in place): mean {prologue.mean():.1f} us, median {np.median(prologue):.1f}, 90 % {np.percentile(prologue, 90):.1f}; "
      f"workgroup life mean {dur.mean():.1f} us -> {prologue.sum() / dur.sum():.2f} of all workgroup time")
lens_sorted = h["rS"][:n].cpu().numpy()
first = np.arange(blocks) * rows_per_block
work = np.add.reduceat(lens_sorted.astype(np.int64), first)
deepest = np.maximum.reduceat(lens_sorted, first)
for name, mask in (("deepest row > 256", deepest > 256), ("128 < deepest <= 256", (deepest > 128) & (deepest <= 256)),
                   ("64 < deepest <= 128", (deepest > 64) & (deepest <= 128)), ("deepest <= 64", deepest <= 64)):
    if mask.any():
        print(f"  {name:22s} {int(mask.sum()):5d} workgroups, mean {dur[mask].mean():6.1f} us, max {dur[mask].max():6.1f} us, "
              f"{work[mask].sum() * 12 / dur[mask].sum() * 1e-3:6.1f} GB/s per workgroup, nnz share {work[mask].sum() / work.sum():.3f}")
order = np.argsort(end)[-5:]
print("  last to finish:", [(int(b), round(float(start[b]), 1), round(float(end[b]), 1), int(deepest[b])) for b in order])
