#!/usr/bin/env python3
"""GPU box experiment (needs the -DSPGPU_TRACE_BLOCKS build, SPGPU_LIB=...): the queue kernel (raggedSpmvKernel) on the ordered
power-law matrix: start, tile-in-place and end of every workgroup -> how much of the launch is prologue, stream, ramp and tail.
  python tools/exp_ragged_trace.py [rows] [window:long] [powerlaw|even] ; EXP_PATTERN=near|band ; SPGPU_RAGGED_SHAPE=0|4|5 ; EXP_ALIGNED=1: spgpuOellOrderAlignedDevice"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
os.environ["SPGPU_RAGGED"] = "1"
from spgpu_amd import capi, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
window, long_rows = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2048:256").split(":"))
case = sys.argv[3] if len(sys.argv) > 3 else "powerlaw"
shape = int(os.environ.get("SPGPU_RAGGED_SHAPE", "0"))
rows_per_group = 2048 if shape == 4 else 1024
handle = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
if case == "powerlaw":
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
else:
    lengths = np.random.default_rng(1).integers(24, 41, size=n).astype(np.int32)
coo = synth.ragged_coo_on_device(lengths, n, os.environ.get("EXP_PATTERN", "band"), 2048, "D", seed=5)
h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, window, long_rows, aligned=bool(os.environ.get("EXP_ALIGNED")))
x = synth.device_vector(n, "D", 3)
z = torch.zeros(n, dtype=torch.float64, device="cuda")
groups = (n + rows_per_group - 1) // rows_per_group
extra = 40000      # workgroups of deep sub-groups behind (or in front of) the blocks of rows when the matrix has a plan
trace = torch.zeros(8 * (groups + extra) + 16, dtype=torch.int64, device="cuda")
capi.lib.spgpuDebugSetTrace.argtypes = [C.c_void_p]
call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), p(h["rIdx"]), 32, n,
                                  p(x), 0.0, 0)
for _ in range(4):
    call()
    torch.cuda.synchronize()
trace.zero_()
torch.cuda.synchronize()
capi.lib.spgpuDebugSetTrace(p(trace))
call()
torch.cuda.synchronize()
capi.lib.spgpuDebugSetTrace(None)
everything = trace[:8 * (groups + extra)].view(groups + extra, 8).cpu().numpy().astype(np.float64)
ran = everything[:, 0] > 0
spread = int(os.environ.get("SPGPU_PLAN_DEEP_SPREAD", "60"))
deep_blocks = int(ran.sum()) - groups
print(f"plan counts {capi.plan_counts(handle)}; workgroups that ran: {int(ran.sum())} = {groups} blocks of rows + {deep_blocks} of deep sub-groups (spread {spread})")
grid = groups + deep_blocks
stride = 0 if spread < 0 or deep_blocks == 0 else max(1, (grid * min(spread, 100) // 100) // deep_blocks) | 1
ids = np.arange(grid)
is_deep = (ids >= groups) if stride == 0 else ((ids % stride == 0) & (ids // stride < deep_blocks))
t = everything[:grid][~is_deep]
t0 = everything[ran, 0].min()
if deep_blocks > 0:
    d = everything[:grid][is_deep]
    ds, de = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0
    print(f"  deep workgroups: start {ds.min():.1f} .. {ds.max():.1f} us, end max {de.max():.1f} us, life min {np.min(de - ds):.1f} median {np.median(de - ds):.1f} "
          f"p90 {np.percentile(de - ds, 90):.1f} max {np.max(de - ds):.1f} us, slot-time {np.sum(de - ds):.0f} us ({np.sum(de - ds) / 512:.1f} us of the chip)")
start, end, tiled = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
span = end.max()
life, prologue = end - start, tiled - start
slots = 512.0
print(f"{case} {os.environ.get('EXP_PATTERN', 'band')}{' ALIGNED' if os.environ.get('EXP_ALIGNED') else ''} shape {shape}: {groups} workgroups, main kernel span {span:.1f} us")
print(f"  workgroup life: min {life.min():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f} us;"
      f" prologue (start -> tile in place): median {np.median(prologue):.1f} p90 {np.percentile(prologue, 90):.1f} us")
print(f"  occupancy of the {slots:.0f} workgroup slots over the span: {life.sum() / (slots * span):.3f} resident,"
      f" {(end - tiled).sum() / (slots * span):.3f} streaming, {prologue.sum() / (slots * span):.3f} in the prologue")
last_start = start.max()
print(f"  last workgroup starts at {last_start:.1f} us ({span - last_start:.1f} us before the end); resident workgroups at that moment "
      f"{int(((start <= last_start) & (end > last_start)).sum())}; area idle after it: "
      f"{(slots * (span - last_start) - np.clip(end - np.maximum(start, last_start), 0, None).sum()) / (slots * span):.3f} of the launch")
names = ["lengths here (round trip 1)", "probes here, tables written (round trip 2)", "destinations staged (two barriers)", "item table, first stages requested", "tile in place (round trip 3)"]
marks = [t[:, 3], t[:, 4], t[:, 5], t[:, 6], t[:, 2]]
previous = t[:, 0]
for name, mark in zip(names, marks):
    step = (mark - previous) / 100.0
    ok = (mark > 0) & (step >= 0)
    print(f"    prologue step -> {name}: median {np.median(step[ok]):.2f} us, p90 {np.percentile(step[ok], 90):.2f}")
    previous = mark
edges = np.linspace(0, span, 21)
res = [int(((start <= (a + b) / 2) & (end > (a + b) / 2)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("  resident workgroups at 20 moments:", res)
heavy = np.argsort(life)[-5:]
print("  longest-lived workgroups:", [(int(b), round(float(start[b]), 1), round(float(life[b]), 1)) for b in heavy])
