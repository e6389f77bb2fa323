#!/usr/bin/env python3
"""GPU box: interleaved in-process A/B of SpMM kernel variants (SPGPU_SPMM_VARIANT) on 5 M rows x 32 x 16 rhs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

rows, L, k = 5_000_000 // 32 * 32, 32, int(os.environ.get("RHS", 16))
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
for pattern in sys.argv[1:] or ["banded", "random"]:
    h = synth.hell_uniform_on_device(rows, L, pattern, "D", 32, seed=11)
    X = synth.device_vector(rows * k, "D", 21).view(rows, k)
    Z = torch.empty_like(X)
    torch.cuda.synchronize()
    call = lambda: capi.hellspmm["D"](handle, p(Z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), None,
                                      L, rows, p(X), 0.0, 0, k, k, k)
    alg = h["nnz"] * 12 + rows * 4 + rows // 32 * 4 + k * (rows + rows) * 8
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3").split(",")]
    ts = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            os.environ["SPGPU_SPMM_VARIANT"] = str(v)
            capi.spgpuTuningReload()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                call()
                a.record(stream)
                for _ in range(10):
                    call()
                b.record(stream)
            b.synchronize()
            ts[v].append(a.elapsed_time(b) / 10)
    for v in variants:
        t = sorted(ts[v])[len(ts[v]) // 2]
        print(f"{pattern:7s} spmm variant={v}  {t:.4f} ms  {alg / t * 1e-6:7.1f} GB/s ({alg / t * 1e-6 / 8000:5.1%})  {2 * h['nnz'] * k / t * 1e-6:8.1f} GFLOP/s", flush=True)
    del h
    torch.cuda.empty_cache()
