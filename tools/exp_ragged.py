#!/usr/bin/env python3
"""GPU box experiment: cost of ragged rows (power-law lengths, max 2048) vs uniform rows at equal nnz and an
x small enough to sit in L2 (65 536 columns), so that the difference is the kernel's, not the gathers'."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

letter = sys.argv[1] if len(sys.argv) > 1 else "D"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
ncols = 65536
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
elem = {"S": 4, "D": 8}[letter]
one, zero = capi.scalar(letter, 1.0), capi.scalar(letter, 0.0)


def run(h, label):
    x = synth.device_vector(ncols, letter, 3)
    z = torch.empty(n, dtype=x.dtype, device="cuda:0")
    torch.cuda.synchronize()
    call = lambda: capi.hellspmv[letter](handle, p(z), None, one, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]),
                                         None, 32, n, p(x), zero, 0)
    for variant in os.environ.get("VARIANTS", "0").split(","):
        variant, _, tl = variant.partition(":")
        os.environ["SPGPU_SPMV_VARIANT"] = variant
        os.environ["SPGPU_TAIL_LANES"] = tl or "8"
        capi.spgpuTuningReload()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            call(); call()
            a.record(stream)
            for _ in range(10):
                call()
            b.record(stream)
        b.synchronize()
        t = a.elapsed_time(b) / 10
        alg = h["nnz"] * (elem + 4) + n * (4 + elem)
        print(f"{letter} {label:28s} variant={variant:>2s} tail_lanes={(tl or '8'):>2s} nnz={h['nnz']} {t:.4f} ms {alg / t * 1e-6:7.1f} GB/s {2 * h['nnz'] / t * 1e-6:7.1f} GFLOP/s", flush=True)


h = synth.hell_uniform_on_device(n, 32, "random", letter, 32, seed=1, n_cols=ncols)
run(h, "uniform 32/row")
del h
torch.cuda.empty_cache()
lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
h = synth.hell_ragged_on_device(lengths, ncols, letter, 32, seed=5)
print(f"ragged: slots/nnz = {h['slots'] / h['nnz']:.2f}, deepest hack {int(h['depth'].max())}")
run(h, "power-law mean 32 max 2048")
