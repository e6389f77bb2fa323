#!/usr/bin/env python3
"""GPU box: matrices swapped IN PLACE under the plan (csrc/planned_spmv.hip) -- K ordered matrices of one size copied over the same
device arrays in random order, 1-4 SpMV calls each, with and without a synchronisation between the calls, on changing streams;
every result against the oracle of the matrix that is loaded.   python tools/stress_stale_plan.py [rounds] [seed]
EXP_MORE_CALLS=18: so many more calls per matrix (a plan that has been used 16 times is rebuilt when it goes stale, a younger one given up after the third time)"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
gpu = capi.create_handle(0)
n = 4 * 2048 + 5
specs = [(512, 40, False, 12.0, 600, 300), (2048, 60, True, 12.0, 700, 300), (512, 40, False, 40.0, 1500, 800), (2048, 60, True, 30.0, 900, 500),
         (0, 0, False, 12.0, 600, 300), (256, 100, False, 20.0, 400, 200)]
mats = [T._matrix(gpu, n, "D", w, lr, al, mean=mean, longest=longest, seed=5 + 7 * i, near=near) for i, (w, lr, al, mean, longest, near) in enumerate(specs)]
slots = max(m["slots"] for m in mats)
fixed = dict(cM=torch.zeros(slots, dtype=torch.float64, device="cuda"), rP=torch.zeros(slots, dtype=torch.int32, device="cuda"),
             hack_offsets=torch.zeros_like(mats[0]["hack_offsets"]), rS=torch.zeros(n, dtype=torch.int32, device="cuda"),
             rIdx=torch.zeros(n, dtype=torch.int32, device="cuda"))
x = synth.values_for("D", 41, n)
dx = formats.to_device(x)
shape = O.slab_shape("D", "ragged", deep_cap=O.DEEP_CAP)
want = [O.spmv_tail(T._host(m, "D", n), x, None, 1.5, 0.0, r_idx=m["rIdx"].cpu().numpy(), **shape) for m in mats]
streams = [torch.cuda.Stream() for _ in range(12)]
bad = 0
for rnd in range(rounds):
    which = int(rng.integers(0, len(mats)))
    m = mats[which]
    torch.cuda.synchronize()
    fixed["cM"][:m["slots"]] = m["cM"][:m["slots"]]
    fixed["rP"][:m["slots"]] = m["rP"][:m["slots"]]
    fixed["hack_offsets"].copy_(m["hack_offsets"])
    fixed["rS"].copy_(m["rS"][:n])
    fixed["rIdx"].copy_(m["rIdx"])
    torch.cuda.synchronize()
    s = streams[int(rng.integers(0, len(streams)))]
    capi.spgpuSetStream(gpu, C.c_void_p(s.cuda_stream))
    outs = []
    for call in range(int(rng.integers(1, 5)) + int(os.environ.get("EXP_MORE_CALLS", "0"))):
        dz = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        with torch.cuda.stream(s):
            pass
        T._call(gpu, "D", None, n, dz, None, dx, 1.5, 0.0, arrays=fixed)
        synced = bool(rng.integers(0, 2))
        if synced:
            torch.cuda.synchronize()
        outs.append((dz, synced, capi.plan_counts(gpu)))
    torch.cuda.synchronize()
    for call, (dz, synced, counts) in enumerate(outs):
        got = dz.cpu().numpy()
        if got.tobytes() != want[which].tobytes():
            bad += 1
            rows = np.nonzero(got.view(np.uint64) != want[which].view(np.uint64))[0]
            r_idx = m["rIdx"].cpu().numpy()
            where = np.empty(n, np.int64)
            where[r_idx] = np.arange(n)
            lengths = m["rS"][:n].cpu().numpy()
            print(f"round {rnd} matrix {which} {specs[which]} call {call} (synced after: {synced}): {rows.size} rows differ; ordered positions "
                  f"{where[rows][:8]} lengths {lengths[where[rows]][:8]} got {got[rows][:3]} want {want[which][rows][:3]} plan counts then {counts}", flush=True)
capi.spgpuSetStream(gpu, None)
print(f"{rounds} rounds: {bad} calls off; plans {capi.plan_counts(gpu)} fallbacks {capi.spgpuDeepListFallbacks(gpu)} recycled {capi.spgpuDeepListsRecycled(gpu)}")
