#!/usr/bin/env python3
"""GPU box: streaming rates of the level-1 kernels (calibrates what 'achievable HBM' is on this card).
One line per op: GB/s of algorithmic bytes (axpby n*8*(2+[beta!=0]); dot 2*n*8; nrm2 n*8)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
h = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(h, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
for letter, dt, es in (("D", torch.float64, 8), ("S", torch.float32, 4)):
    x, y, z = (torch.rand(n, dtype=dt, device="cuda:0") for _ in range(3))
    torch.cuda.synchronize()
    one, half, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.5), capi.scalar(letter, 0.0)
    ops = {
        "axpby beta=0 (1R+1W)": (lambda: capi.axpby[letter](h, p(z), n, zero, p(y), one, p(x)), 2 * n * es),
        "axpby beta!=0 (2R+1W)": (lambda: capi.axpby[letter](h, p(z), n, half, p(y), one, p(x)), 3 * n * es),
        "axpby in place z=y (2R+1W)": (lambda: capi.axpby[letter](h, p(y), n, half, p(y), one, p(x)), 3 * n * es),
        "dot (2R, host sync)": (lambda: capi.dot[letter](h, n, p(x), p(y)), 2 * n * es),
        "nrm2 (1R, host sync)": (lambda: capi.nrm2[letter](h, n, p(x)), n * es),
    }
    for name, (fn, nbytes) in ops.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            fn(); fn()
            a.record(stream)
            for _ in range(20):
                fn()
            b.record(stream)
        b.synchronize()
        t = a.elapsed_time(b) / 20
        print(f"{letter} n={n} {name:24s} {t:.4f} ms  {nbytes / t * 1e-6:8.1f} GB/s  {nbytes / t * 1e-6 / 8000:6.1%} of 8 TB/s", flush=True)
    del x, y, z
