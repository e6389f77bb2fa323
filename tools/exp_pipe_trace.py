#!/usr/bin/env python3
"""GPU box experiment (needs the -DSPGPU_TRACE_BLOCKS build, SPGPU_LIB=...): pipeSpmvKernel on the ordered power-law matrix
(or on evenly filled rows): when does every resident workgroup start and end, and how long do its streamers poll for the
scout?   python tools/exp_pipe_trace.py [rows] [window:long] [powerlaw|even] ; EXP_PATTERN=near|band"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
os.environ["SPGPU_RAGGED"] = "3"
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
coo = synth.ragged_coo_on_device(lengths, n, os.environ.get("EXP_PATTERN", "band"), 2048, "D", seed=5)
h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, window, long_rows)
x = synth.device_vector(n, "D", 3)
z = torch.zeros(n, dtype=torch.float64, device="cuda")
groups = 256
trace = torch.zeros(3 * groups + 16, dtype=torch.int64, device="cuda")
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
t = trace[:3 * groups].view(groups, 3).cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
raw = trace[:3 * groups].view(groups, 3).cpu().numpy()
start, end, polls, scout_polls = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (raw[:, 2] & 0xffffffff).astype(np.float64), (raw[:, 2] >> 32).astype(np.float64)
dur = end - start
print(f"{case} {os.environ.get('EXP_PATTERN', 'band')}: kernel span {end.max():.1f} us; workgroup life min {dur.min():.1f} median {np.median(dur):.1f} "
      f"max {dur.max():.1f} us; start spread {start.max():.1f} us")
print(f"  streamers' polls for the scout per workgroup: median {np.median(polls):.0f}, max {polls.max():.0f} "
      f"(15 streamers, ~0.1 us per poll -> median {np.median(polls) * 0.1 / 15:.1f} us per streamer)")
print(f"  the scout's polls for a free buffer per workgroup: median {np.median(scout_polls):.0f}, max {scout_polls.max():.0f}")
order = np.argsort(end)
print("  first to finish:", [(int(b), round(float(end[b]), 1)) for b in order[:5]])
print("  last to finish: ", [(int(b), round(float(end[b]), 1)) for b in order[-5:]])
