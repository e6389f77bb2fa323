#!/usr/bin/env python3
"""GPU box: interleaved in-process A/B of HDIA kernel knobs on the 7-point Laplacian (default 512^3).
Usage: python tools/ab_hdia.py [grid] ; settings = (SPGPU_XCD_ORDER, SPGPU_NT_LOADS, SPGPU_HDIA_VARIANT)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
d = synth.hdia_laplacian7_on_device(m, "D", 32)
n = d["rows"]
x, z = synth.device_vector(n, "D", 3), torch.empty(n, dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
alg = 32 * d["height"] * 8 + d["height"] * 4 + (n // 32 + 1) * 4 + 2 * n * 8
call = lambda: capi.hdiaspmv["D"](handle, p(z), None, 1.0, p(d["dM"]), p(d["offsets"]), 32, p(d["hack_offsets"]), n, n, p(x), 0.0)
settings = [(0, 1, 0, b) for b in (256, 512, 1024)] + [(0, 1, 2, 256), (0, 0, 0, 256), (1, 1, 0, 256), (4, 1, 0, 256)]
if os.environ.get("SETTINGS"):   # "xcd,nt,variant,block;..." overrides the default list
    settings = [tuple(int(v) for v in one.split(",")) for one in os.environ["SETTINGS"].split(";")]
times = {s: [] for s in settings}
for rnd in range(5):
    for s in settings:
        (os.environ["SPGPU_XCD_ORDER"], os.environ["SPGPU_NT_LOADS"], os.environ["SPGPU_HDIA_VARIANT"],
         os.environ["SPGPU_HDIA_BLOCK"]) = map(str, s)
        capi.spgpuTuningReload()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            call()
            a.record(stream)
            for _ in range(10):
                call()
            b.record(stream)
        b.synchronize()
        times[s].append(a.elapsed_time(b) / 10)
for s in settings:
    t = sorted(times[s])[len(times[s]) // 2]
    print(f"xcd_order={s[0]:3d} nt={s[1]} variant={s[2]} block={s[3]}  median {t:.4f} ms  min {min(times[s]):.4f}  {alg / t * 1e-6:7.1f} GB/s  {alg / t * 1e-6 / 8000:6.1%}", flush=True)
