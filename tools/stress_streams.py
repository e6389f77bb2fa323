#!/usr/bin/env python3
"""GPU box: the scenario of tests/test_gpu_plan.py::test_streams_come_and_go_lists_change_hands many times in one process -- a new
stream for every ordered SpMV of one matrix -- with, for every call that does not give the oracle's bytes, which rows differ and
what the handle's counters say.   python tools/stress_streams.py [rounds] ; SPGPU_LIB=... for the lab build"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_api as O  # noqa: E402
import test_gpu_plan as T  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gpu = capi.create_handle(0)
n = 4 * 2048 + 5
bad = 0
for rnd in range(rounds):
    h = T._matrix(gpu, n, "D", 512, 40, False, seed=23 + rnd)
    x = synth.values_for("D", 91, n)
    dx = formats.to_device(x)
    r_idx = h["rIdx"].cpu().numpy()
    want = O.spmv_tail(T._host(h, "D", n), x, None, 1.0, 0.0, r_idx=r_idx, **O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP))
    lengths = h["rS"][:n].cpu().numpy()
    where = np.empty(n, np.int64)
    where[r_idx] = np.arange(n)
    for call in range(20):
        s = torch.cuda.Stream()
        capi.spgpuSetStream(gpu, C.c_void_p(s.cuda_stream))
        dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
        before = (capi.plan_counts(gpu), capi.spgpuDeepListFallbacks(gpu), capi.spgpuDeepListsRecycled(gpu))
        T._call(gpu, "D", h, n, dz, None, dx, 1.0, 0.0)
        torch.cuda.synchronize()
        got = dz.cpu().numpy()
        if got.tobytes() != want.tobytes():
            bad += 1
            rows = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
            after = (capi.plan_counts(gpu), capi.spgpuDeepListFallbacks(gpu), capi.spgpuDeepListsRecycled(gpu))
            print(f"round {rnd} call {call}: {rows.size} rows differ; z rows {rows[:6]} at ordered positions {where[rows][:6]} lengths "
                  f"{lengths[where[rows]][:6]}; got {got[rows][:3]} want {want[rows][:3]}; counters before {before} after {after}; "
                  f"form {capi.spgpuGetLastSpmvForm(gpu)}", flush=True)
        del s
    capi.spgpuSetStream(gpu, None)
print(f"{rounds} rounds x 20 calls: {bad} calls off; plans {capi.plan_counts(gpu)} fallbacks {capi.spgpuDeepListFallbacks(gpu)} "
      f"recycled {capi.spgpuDeepListsRecycled(gpu)}")
