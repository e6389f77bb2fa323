#!/usr/bin/env python3
"""GPU box experiment: bench.py's north_star target block on its own (every layout of the power-law matrix, the ordered ones on
--placements sets of allocations), for A/B runs of two builds (SPGPU_LIB=...) or of knobs.  Prints one line per layout.

    python tools/exp_powerlaw_only.py [placements] [rows]"""
import ctypes as C
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from spgpu_amd import capi  # noqa: E402

placements = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
out = bench.bench_powerlaw(types.SimpleNamespace(placements=placements), handle, stream, "cuda:0", rows)
print("lib", capi.LIB_PATH)
for k, v in out.items():
    if isinstance(v, dict) and "ms" in v:
        print(f"{k:22s} slots/nnz {v['slots_per_nnz']:.3f}  {v['ms']:.4f} ms  frac {v['frac']:.4f}  {v.get('ms_spread', '')}  {v['parity']}")
