#!/usr/bin/env python3
"""GPU box, lab build (SPGPU_LIB=spgpu_amd/lib_lab/libspgpu.so): does any ordered SpMV path read LDS it has not written?
spgpuDebugFillLds (csrc/debug_lds.hip) fills every CU's LDS with one word (EXP_LDS_WORD=0 | 0xffffffff) in front of every call; the result must still be the oracle's bytes.
Paths: the first call (list), the planned call, SPGPU_PLAN=0, a stream without a list (stateless), stale plans (matrices swapped in
place), all four types, the shapes of tests/test_gpu_plan.py.   python tools/stress_lds.py [matrices]"""
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

count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gpu = capi.create_handle(0)
fill_lds = capi.lib.spgpuDebugFillLds
fill_lds.argtypes = [C.c_void_p, C.c_uint]
fill_lds.restype = C.c_int
word = int(os.environ.get("EXP_LDS_WORD", "0"), 0)      # 0, or 0xffffffff: -1 / NaN
rng = np.random.default_rng(11)
streams = [torch.cuda.Stream() for _ in range(10)]
bad = calls = 0
for i in range(count):
    letter = "DSCZ"[i % 4] if i >= 4 else "D"
    n = int(rng.choice([4 * 2048 + 5, 9 * 2048 + 77, 6 * 2048 + 300, 40 * 2048 + 1]))
    window, long_rows, aligned, hack = [(512, 40, False, 32), (2048, 60, True, 32), (0, 0, False, 64), (256, 100, False, 96)][i % 4]
    longest = int(rng.choice([600, 900, 1500]))
    h = T._matrix(gpu, n, letter, window, long_rows, aligned, hack=hack, mean=float(rng.choice([12.0, 30.0])), longest=longest, seed=200 + i,
                  near=max(int(rng.choice([300, 800])), longest // 2 + 50))
    x, y = synth.values_for(letter, 91 + i, n), synth.values_for(letter, 92 + i, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    shape = O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP)
    r_idx = h["rIdx"].cpu().numpy()
    want = O.spmv_tail(T._host(h, letter, n, hack), x, y, -0.5, 2.0, r_idx=r_idx, **shape)
    lengths = h["rS"][:n].cpu().numpy()
    where = np.empty(n, np.int64)
    where[r_idx] = np.arange(n)
    for plan in ("1", "0"):
        os.environ["SPGPU_PLAN"] = plan
        capi.spgpuTuningReload()
        for k in range(8):
            s = streams[(i * 3 + k) % len(streams)]        # 10 streams for 8 lists: some calls recycle a list
            capi.spgpuSetStream(gpu, C.c_void_p(s.cuda_stream))
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            s.wait_stream(torch.cuda.current_stream())
            assert fill_lds(gpu, word) == 0
            T._call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)
            torch.cuda.synchronize()
            calls += 1
            got = dz.cpu().numpy()
            if got.tobytes() != want.tobytes():
                bad += 1
                rows = np.nonzero(got.view(np.uint8).reshape(n, -1) != want.view(np.uint8).reshape(n, -1))[0]
                rows = np.unique(rows)
                print(f"matrix {i} {letter} n {n} window {window}:{long_rows} aligned {aligned} hack {hack} plan {plan} call {k}: {rows.size} rows differ; "
                      f"ordered positions {where[rows][:8]} lengths {lengths[where[rows]][:8]} got {got[rows][:3]} want {want[rows][:3]} "
                      f"plans {capi.plan_counts(gpu)} fallbacks {capi.spgpuDeepListFallbacks(gpu)}", flush=True)
    capi.spgpuSetStream(gpu, None)
os.environ.pop("SPGPU_PLAN", None)
capi.spgpuTuningReload()
print(f"{calls} calls with the LDS filled with {word:#x} in front: {bad} off; plans {capi.plan_counts(gpu)} fallbacks {capi.spgpuDeepListFallbacks(gpu)} "
      f"recycled {capi.spgpuDeepListsRecycled(gpu)}")
