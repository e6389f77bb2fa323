#!/usr/bin/env python3
"""GPU box: COO -> HELL construction time, device (spgpuCoo*Device) vs the host converters (cooToEll + ellToHell,
single thread, as the reference does it).  usage: bench_convert.py [rows=10000000] [nnz_per_row=32] [host_rows=1000000]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, formats  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
host_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
h = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr())
nnz = rows * L
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
r = torch.arange(rows, device="cuda:0", dtype=torch.int32).repeat_interleave(L)
perm = torch.randperm(nnz, device="cuda:0", generator=g)          # arbitrary COO order
r = r[perm].contiguous()
c = torch.randint(0, rows, (nnz,), device="cuda:0", generator=g, dtype=torch.int32)
v = torch.rand(nnz, device="cuda:0", generator=g, dtype=torch.float64)
del perm
work = torch.empty(capi.spgpuCooConvertWorkBytes(rows, nnz), dtype=torch.uint8, device="cuda:0")
rs = torch.empty(rows, dtype=torch.int32, device="cuda:0")
ho = torch.empty(rows // 32 + 1, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    mx, height = C.c_int(0), C.c_int(0)
    capi.spgpuCooRowLengthsDevice(h, p(rs), C.byref(mx), rows, nnz, p(r), 0, p(work))
    capi.spgpuHellPlanDevice(h, C.byref(height), p(ho), 32, rows, p(rs), p(work))
    slots = 32 * height.value
    hv = torch.zeros(slots, dtype=torch.float64, device="cuda:0")
    hi = torch.zeros(slots, dtype=torch.int32, device="cuda:0")
    capi.spgpuCooToHellDevice(h, p(hv), p(hi), p(ho), 32, 0, rows, nnz, p(r), p(c), p(v), 0, capi.TYPE_DOUBLE, p(rs), p(work))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"device: COO({nnz} nnz, shuffled) -> HELL({rows} rows, {slots} slots) in {dt * 1e3:.1f} ms ({nnz / dt * 1e-9:.2f} G nnz/s)", flush=True)
# host converters on the first host_rows rows' worth of entries (row-sorted sample), single thread
hr = np.repeat(np.arange(host_rows, dtype=np.int32), L)
hc = np.random.default_rng(1).integers(0, host_rows, hr.size).astype(np.int32)
hvv = np.random.default_rng(2).random(hr.size)
t0 = time.perf_counter()
ell = formats.coo_to_ell(host_rows, hr, hc, hvv)
hell = formats.ell_to_hell(ell, 32)
dt_h = time.perf_counter() - t0
print(f"host  : COO({hr.size} nnz) -> ELL -> HELL in {dt_h * 1e3:.1f} ms ({hr.size / dt_h * 1e-9:.3f} G nnz/s, 1 thread); "
      f"extrapolated to {nnz} nnz: {dt_h * nnz / hr.size:.1f} s + {nnz * 12 / 63e9:.2f} s PCIe upload", flush=True)
