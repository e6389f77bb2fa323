#!/usr/bin/env python3
"""GPU box: COO -> HDIA construction time for the 7-point Laplacian m^3 (BASELINE configs[3]'s family), device
(spgpuCooHdiaPlanDevice + spgpuCooToHdiaDevice) vs the host converter (single thread, as the reference does it).
usage: bench_convert_hdia.py [m=256] [host_m=96]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
host_m = int(sys.argv[2]) if len(sys.argv) > 2 else 96
h = capi.create_handle(0)
p = lambda t: C.c_void_p(t.data_ptr())
dev = "cuda:0"
N = m ** 3
i = torch.arange(N, device=dev, dtype=torch.int64)
gx, gy, gz = i % m, (i // m) % m, i // (m * m)
parts = [(gz > 0, -m * m, -1.0), (gy > 0, -m, -1.0), (gx > 0, -1, -1.0), (torch.ones_like(gx, dtype=torch.bool), 0, 6.0),
         (gx < m - 1, 1, -1.0), (gy < m - 1, m, -1.0), (gz < m - 1, m * m, -1.0)]
r = torch.cat([i[mask] for mask, _, _ in parts]).to(torch.int32)
c = torch.cat([i[mask] + d for mask, d, _ in parts]).to(torch.int32)
v = torch.cat([torch.full((int(mask.sum()),), val, device=dev, dtype=torch.float64) for mask, _, val in parts])
nnz = r.numel()
g = torch.Generator(device=dev); g.manual_seed(1)
perm = torch.randperm(nnz, device=dev, generator=g)              # arbitrary COO order
r, c, v = r[perm].contiguous(), c[perm].contiguous(), v[perm].contiguous()
del perm, i, gx, gy, gz, parts
hacks = capi.getHdiaHacksCount(32, N)
work = torch.empty(capi.spgpuCooHdiaPlanWorkBytes(N, nnz), dtype=torch.uint8, device=dev)
ho = torch.empty(hacks + 1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    height = C.c_int(0)
    assert capi.spgpuCooHdiaPlanDevice(h, C.byref(height), p(ho), 32, N, N, nnz, p(r), p(c), 0, p(work)) == 0
    t1 = time.perf_counter()
    values = torch.zeros(32 * height.value, dtype=torch.float64, device=dev)
    offsets = torch.empty(height.value, dtype=torch.int32, device=dev)
    scratch = torch.empty(capi.spgpuCooToHdiaScratchBytes(32, height.value), dtype=torch.uint8, device=dev)
    assert capi.spgpuCooToHdiaDevice(h, p(values), p(offsets), p(ho), 32, N, N, nnz, p(r), p(c), p(v), 0, capi.TYPE_DOUBLE,
                                     height.value, p(work), p(scratch)) == 0
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"device: COO({nnz} nnz, shuffled) -> HDIA({N} rows, {height.value} diagonals) plan {1e3 * (t1 - t0):.1f} ms + fill "
      f"{1e3 * (t2 - t1):.1f} ms ({nnz / (t2 - t0) * 1e-9:.2f} G nnz/s)", flush=True)
n, _, hr, hc, hv = synth.laplacian_3d_7pt(host_m)
t0 = time.perf_counter()
host = formats.coo_to_hdia(n, n, hr, hc, hv, 32)
dt = time.perf_counter() - t0
print(f"host  : COO({hr.size} nnz, row-major) -> HDIA in {dt * 1e3:.1f} ms ({hr.size / dt * 1e-9:.3f} G nnz/s, 1 thread); "
      f"extrapolated to {nnz} nnz: {dt * nnz / hr.size:.1f} s", flush=True)
# spot check: the device result for host_m equals the host one is covered by tests/test_gpu_convert_device.py
