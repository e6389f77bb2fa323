#!/usr/bin/env python3
"""GPU box experiment: the fused SpMV + dot kernel (fused_solver.hip: 1 024 workgroups walking the rows with a tile
stride, 8 rows per lane, no prefetch) and the SWEEP form of the SpMV (32 rows per lane) against the default gather kernel
on scattered columns.
    python tools/exp_fused_random.py [rows] [nnz per row] [pattern]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pat = sys.argv[3] if len(sys.argv) > 3 else "random"
h = capi.create_handle(0)
s = torch.cuda.Stream()
capi.spgpuSetStream(h, C.c_void_p(s.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
m = synth.hell_uniform_on_device(n, L, pat, "D", 32, seed=1)
x = synth.device_vector(n, "D", 3)
z = torch.empty_like(x)
out = torch.zeros(1, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
fused = lambda: capi.hellspmv_dot_device["D"](h, p(out), None, p(z), None, 1.0, p(m["cM"]), p(m["rP"]), 32, p(m["hack_offsets"]), p(m["rS"]), n,
                                              p(x), 0.0, 0)
plain = lambda: capi.hellspmv["D"](h, p(z), None, 1.0, p(m["cM"]), p(m["rP"]), 32, p(m["hack_offsets"]), p(m["rS"]), None, L, n, p(x), 0.0, 0)
def sweep():
    capi.spgpuSetSpmvForm(h, capi.FORM_SWEEP)
    plain()
    capi.spgpuSetSpmvForm(h, capi.FORM_AUTO)


plain()
torch.cuda.synchronize()
want = z.clone()
cases = [("fused spmv+dot", fused), ("spmv alone", plain), ("sweep form (32 rows per lane)", sweep)]
for name, fn in cases:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
            s.synchronize()
        a.record(s)
        for _ in range(10):
            fn()
        b.record(s)
    b.synchronize()
    same = "" if not name.startswith("sweep") else ("  same z" if torch.equal(z, want) else f"  max |dz| {float((z - want).abs().max()):.2e}")
    print(f"{n} rows x {L} {pat}: {name}: {a.elapsed_time(b) / 10 * 1e3:.1f} us{same}", flush=True)
capi.spgpuDestroy(h)
