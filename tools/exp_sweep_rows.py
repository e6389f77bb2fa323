#!/usr/bin/env python3
"""GPU box: from how many rows on does the SWEEP form beat the gather kernel on scattered ascending columns (AUTO's threshold,
kAutoSweepRows in csrc/ellpack_spmv.hip)?   python tools/exp_sweep_rows.py [nnz]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

nnz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
handle = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for n in (1 << 20, 1 << 21, 3 << 20, 1 << 22, 6 << 20, 10_000_000, 1 << 24):
    h = synth.hell_uniform_on_device(n, nnz, "random", "D", 32, seed=1)
    x = synth.device_vector(n, "D", 3)
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    out = {}
    for name, form in (("gather", capi.FORM_GATHER), ("sweep", capi.FORM_SWEEP)):
        capi.spgpuSetSpmvForm(handle, form)
        call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), None, nnz, n, p(x), 0.0, 0)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            call()
        b.record()
        torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) / 10
    print(f"rows {n:>9d} x {nnz}: gather {out['gather']:.4f} ms  sweep {out['sweep']:.4f} ms  sweep/gather {out['sweep'] / out['gather']:.3f}", flush=True)
    capi.spgpuSetSpmvForm(handle, capi.FORM_AUTO)
    del h, x, z
    torch.cuda.empty_cache()
