#!/usr/bin/env python3
"""Tuning aid (GPU box): interleaved timing of the HELL fp64 kernel variants on BASELINE configs[1].
Usage: python tools/sweep_hell.py [rows] [patterns...]   -> one line per (pattern, variant, nt)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
patterns = sys.argv[2:] or ["banded", "window", "random"]
L = int(os.environ.get("SWEEP_NNZ", 32))
letter = os.environ.get("SWEEP_TYPE", "D")
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
elem = {"S": 4, "D": 8, "C": 8, "Z": 16}[letter]

for pattern in patterns:
    h = synth.hell_uniform_on_device(rows, L, pattern, letter, 32, seed=1)
    x, y = synth.device_vector(rows, letter, 3), synth.device_vector(rows, letter, 4)
    z = torch.empty_like(y)
    torch.cuda.synchronize()  # inputs are produced on torch's stream, consumed on the handle's
    alg = h["nnz"] * (elem + 4) + rows * (4 + elem) + rows * elem + rows // 32 * 4
    one, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.0)
    call = lambda: capi.hellspmv[letter](handle, p(z), p(y), one, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]),
                                         p(h["rS"]), None, L, rows, p(x), zero, 0)
    variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "1,2,3,4").split(",")]
    # the second knob is SPGPU_NT_LOADS, or SPGPU_X_STRIPS with SWEEP_KNOB=xstrips
    knob = "SPGPU_X_STRIPS" if os.environ.get("SWEEP_KNOB") == "xstrips" else "SPGPU_NT_LOADS"
    configs = [(v, nt) for v in variants for nt in ([1, 0] if os.environ.get("SWEEP_NT", "both") == "both" else [1])]
    best = {}
    for rnd in range(int(os.environ.get("SWEEP_ROUNDS", 3))):
        for v, nt in configs:
            os.environ["SPGPU_SPMV_VARIANT"], os.environ[knob] = str(v), str(nt)
            capi.spgpuTuningReload()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                call(); call()
                a.record(stream)
                for _ in range(20):
                    call()
                b.record(stream)
            b.synchronize()
            t = a.elapsed_time(b) / 20
            best.setdefault((v, nt), []).append(t)
    for (v, nt), ts in sorted(best.items()):
        t = sorted(ts)[len(ts) // 2]
        print(f"{letter} {pattern:7s} variant={v} {knob[6:].lower()}={nt}  median {t:.4f} ms  min {min(ts):.4f} ms  "
              f"{alg / t * 1e-6:8.1f} GB/s  {alg / t * 1e-6 / 8000:6.1%} of 8 TB/s  {2 * h['nnz'] / t * 1e-6:8.1f} GFLOP/s",
              flush=True)
    del h
    torch.cuda.empty_cache()
