#!/usr/bin/env python3
"""GPU box experiment: BASELINE configs[0] (5-point Laplacian 1024 x 1024, host-converted HELL) per kernel shape
(SPGPU_SPMV_VARIANT, lab build) -- microseconds per SpMV in a stream of 300 launches."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, formats, synth  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, m, r, c, v = synth.laplacian_2d_5pt(g)
hell = formats.ell_to_hell(formats.coo_to_ell(n, r, c, v), 32)
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
mat = formats.DeviceHell(hell)
x = formats.to_device(synth.hashed_vector(m))
z = torch.empty(n, dtype=torch.float64, device="cuda")
ref = None
for variant in [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "0,2,9,12,17,21,4,13").split(",")]:
    for strips in ("-1", "0"):
        os.environ["SPGPU_SPMV_VARIANT"], os.environ["SPGPU_X_STRIPS"] = str(variant), strips
        capi.spgpuTuningReload()
        with torch.cuda.stream(stream):
            for _ in range(5):
                mat.spmv(handle, z, None, 1.0, x, 0.0, avg_nnz=5)
                stream.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(300):
                mat.spmv(handle, z, None, 1.0, x, 0.0, avg_nnz=5)
            b.record(stream)
        b.synchronize()
        same = "" if ref is None else ("same values" if torch.allclose(z, ref, rtol=1e-13, atol=0) else "DIFFERENT")
        ref = z.clone() if ref is None else ref
        print(f"grid {g}: variant {variant:2d} strips {strips:>2s}: {a.elapsed_time(b) / 300 * 1e3:7.2f} us  form {capi.spgpuGetLastSpmvForm(handle)} {same}", flush=True)
