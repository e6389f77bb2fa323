#!/usr/bin/env python3
"""GPU box experiment (lab build: SPGPU_LIB=spgpu_amd/lib_lab/libspgpu.so): the moving x tile (csrc/slide_spmv.hip.h, SPGPU_SLIDE=1)
against the form AUTO picks and against the SWEEP form, whose order of additions it shares; uniform rows, fp64 HELL.
  python tools/exp_slide.py [rows] [nnz] ; EXP_PATTERNS=window,banded,random,near8192"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nnz = int(sys.argv[2]) if len(sys.argv) > 2 else 32
handle = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
x = synth.device_vector(n, "D", 3)


def timed(call, reps=20, blocks=3):
    out = []
    for _ in range(blocks):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(reps):
            call()
        stop.record()
        torch.cuda.synchronize()
        out.append(start.elapsed_time(stop) / reps)
    return out


for pattern in os.environ.get("EXP_PATTERNS", "window,banded").split(","):
    h = synth.hell_uniform_on_device(n, nnz, pattern, "D", 32, seed=1)
    algorithmic = h["nnz"] * 12 + n * (4 + 8) + n * 8 + (n // 32) * 4
    results = {}
    for name, form, slide in (("auto", capi.FORM_AUTO, "0"), ("slide", capi.FORM_AUTO, "1")):
        os.environ["SPGPU_SLIDE"] = slide
        capi.spgpuTuningReload()
        capi.spgpuSetSpmvForm(handle, form)
        z = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), None, nnz, n,
                                          p(x), 0.0, 0)
        for _ in range(6):
            call()
        torch.cuda.synchronize()
        ms = timed(call)
        results[name] = z.clone()
        best = min(ms)
        print(f"{pattern:10s} {name:6s} {' '.join(f'{v:.4f}' for v in ms)} ms   {algorithmic / best / 1e6:8.1f} GB/s  {algorithmic / best / 8e9:.3f} of 8 TB/s", flush=True)
    same = torch.equal(results["auto"], results["slide"])
    worst = (results["auto"] - results["slide"]).abs().max().item()
    print(f"{pattern:10s} slide == auto bit for bit: {same}  (largest difference {worst:.3e})", flush=True)
    os.environ["SPGPU_SLIDE"] = "0"
    capi.spgpuTuningReload()
    del h
