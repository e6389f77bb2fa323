#!/usr/bin/env python3
"""GPU box experiment: bench.py's configs[2] block on its own (HELL fp32 vs ELL fp32 on power-law rows with scattered columns, the
plain calls and the same calls after spgpuHellSpmvAdopt / spgpuEllSpmvAdopt).

    python tools/exp_c3_only.py [rows] [ell_rows]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from spgpu_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ell_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
out = bench.bench_c3(handle, stream, "cuda:0", rows, ell_rows)
for key in ("hell_fp32", "ell_fp32", "hell_fp32_rows_ordered"):
    print(key, json.dumps(out[key]))
