#!/usr/bin/env python3
"""GPU box: is every ordered SpMV path deterministic to the bit?  Small ordered matrices (the sizes of tests/test_gpu_plan.py), each
run `reps` times back to back without and with the plan, on the handle's stream and on another one; every result compared with the
oracle's bytes on the device.   python tools/stress_determinism.py [matrices] [reps]"""
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

count = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
gpu = capi.create_handle(0)
rng = np.random.default_rng(7)
shape = O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP)
other = torch.cuda.Stream()
bad = 0
for i in range(count):
    n = int(rng.choice([4 * 2048 + 5, 9 * 2048 + 77, 6 * 2048 + 300, 40 * 2048 + 1]))
    window, long_rows, aligned = [(512, 40, False), (2048, 60, True), (0, 0, False), (256, 100, False)][i % 4]
    longest = int(rng.choice([600, 900, 1500]))
    h = T._matrix(gpu, n, "D", window, long_rows, aligned, mean=float(rng.choice([12.0, 30.0])), longest=longest, seed=100 + i,
                  near=max(int(rng.choice([300, 800])), longest // 2 + 50))
    x = synth.values_for("D", 91 + i, n)
    dx = formats.to_device(x)
    want = torch.from_numpy(O.spmv_tail(T._host(h, "D", n), x, None, 1.0, 0.0, r_idx=h["rIdx"].cpu().numpy(), **shape)).cuda()
    lengths = h["rS"][:n].cpu().numpy()
    r_idx = h["rIdx"].cpu().numpy()
    where = np.empty(n, np.int64)
    where[r_idx] = np.arange(n)
    for plan in ("0", "1"):
        os.environ["SPGPU_PLAN"] = plan
        capi.spgpuTuningReload()
        for stream in (None, other):
            capi.spgpuSetStream(gpu, C.c_void_p(stream.cuda_stream) if stream else None)
            outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(reps)]
            torch.cuda.synchronize()
            for r in range(reps):
                T._call(gpu, "D", h, n, outs[r], None, dx, 1.0, 0.0)
                if r % 7 == 3:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            for r in range(reps):
                if not torch.equal(outs[r].view(torch.int64), want.view(torch.int64)):
                    bad += 1
                    rows = torch.nonzero(outs[r].view(torch.int64) != want.view(torch.int64)).flatten().cpu().numpy()
                    print(f"matrix {i} n {n} window {window}:{long_rows} aligned {aligned} plan {plan} stream {'other' if stream else 'handle'} rep {r}: "
                          f"{rows.size} rows differ; ordered positions {where[rows][:8]} lengths {lengths[where[rows]][:8]} "
                          f"got {outs[r][rows[:3]].cpu().numpy()} want {want[rows[:3]].cpu().numpy()}", flush=True)
            del outs
    capi.spgpuSetStream(gpu, None)
os.environ.pop("SPGPU_PLAN", None)
capi.spgpuTuningReload()
print(f"{count} matrices x 2 x 2 x {reps} calls: {bad} off; plans {capi.plan_counts(gpu)} overflows {capi.spgpuDeepListOverflows(gpu) if hasattr(capi, 'spgpuDeepListOverflows') else '?'}")
