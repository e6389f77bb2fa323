#!/usr/bin/env python3
"""GPU box: do the ordered SpMV paths ever USE a padding slot of the HELL arrays?  (The converters leave them as they find them: in a
long-lived process they hold whatever lived there before.)  Every slot (r, k) with k >= rowLength[r] gets a NaN coefficient and a
random valid column; the first call (list), planned calls, SPGPU_PLAN=0 and all four types must still give the oracle's bytes.
  python tools/stress_padding.py [matrices]"""
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
rng = np.random.default_rng(13)
bad = calls = 0
for i in range(count):
    letter = "DSCZ"[i % 4]
    n = int(rng.choice([4 * 2048 + 5, 9 * 2048 + 77, 6 * 2048 + 300]))
    window, long_rows, aligned, hack = [(512, 40, False, 32), (2048, 60, True, 32), (0, 0, False, 64), (256, 100, False, 96)][(i // 4) % 4]
    longest = int(rng.choice([600, 900, 1500]))
    h = T._matrix(gpu, n, letter, window, long_rows, aligned, hack=hack, mean=float(rng.choice([12.0, 30.0])), longest=longest, seed=300 + i,
                  near=max(int(rng.choice([300, 800])), longest // 2 + 50))
    slots = h["slots"]
    lengths = h["rS"][:n].cpu().numpy().astype(np.int64)
    offsets = h["hack_offsets"].cpu().numpy().astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), lengths)
    ks = np.arange(lengths.sum(), dtype=np.int64) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    used = np.zeros(slots, bool)
    used[offsets[rows // hack] + rows % hack + ks * hack] = True
    padding = torch.from_numpy(~used).cuda()
    cM, rP = h["cM"][:slots], h["rP"][:slots]
    cM[padding] = float("nan")
    rP[padding] = torch.randint(0, n, (int(padding.sum().item()),), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    x, y = synth.values_for(letter, 91 + i, n), synth.values_for(letter, 92 + i, n)
    dx, dy = formats.to_device(x), formats.to_device(y)
    r_idx = h["rIdx"].cpu().numpy()
    want = O.spmv_tail(T._host(h, letter, n, hack), x, y, -0.5, 2.0, r_idx=r_idx, **O.slab_shape(letter, "ragged", deep_cap=O.DEEP_CAP))
    assert not np.isnan(want.view(np.float32 if letter in "SC" else np.float64)).any()
    for plan in ("1", "0"):
        os.environ["SPGPU_PLAN"] = plan
        capi.spgpuTuningReload()
        for k in range(5):
            dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
            torch.cuda.synchronize()
            T._call(gpu, letter, h, n, dz, dy, dx, -0.5, 2.0, hack)
            torch.cuda.synchronize()
            calls += 1
            got = dz.cpu().numpy()
            if got.tobytes() != want.tobytes():
                bad += 1
                rows_off = np.unique(np.nonzero(got.view(np.uint8).reshape(n, -1) != want.view(np.uint8).reshape(n, -1))[0])
                print(f"matrix {i} {letter} n {n} window {window}:{long_rows} aligned {aligned} hack {hack} plan {plan} call {k}: {rows_off.size} rows differ "
                      f"(padding slots: {int(padding.sum().item())} of {slots}); got {got[rows_off][:3]} want {want[rows_off][:3]}", flush=True)
os.environ.pop("SPGPU_PLAN", None)
capi.spgpuTuningReload()
print(f"{calls} calls on matrices whose padding slots hold NaN and random columns: {bad} off; plans {capi.plan_counts(gpu)}")
