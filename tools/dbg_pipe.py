import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ["SPGPU_RAGGED"]="3"; os.environ["SPGPU_PIPE_GROUPS"]=sys.argv[1] if len(sys.argv)>1 else "3"
import numpy as np, torch
import oracle_api as O
from spgpu_amd import capi, formats
import test_gpu_share as T
gpu = capi.create_handle(0)
letter="D"
n = 40000 + 7
rng = np.random.default_rng(3)
lengths = np.where(np.arange(n) < n // 2, 100, 0)
lengths[n // 2 + 5000] = 3
_, hell = T._host_hell(n, lengths, letter, 32, rng, near=500)
r_idx = rng.permutation(n).astype(np.int32)
cols_n = int(hell["indices"].max()) + 1
x = T._vec(rng, letter, cols_n)
dx = formats.to_device(x)
y = T._vec(rng, letter, n)
dy = formats.to_device(y)
d = formats.DeviceHell(hell, r_idx=r_idx)
import ctypes as C
trace = torch.zeros(4096 + 32 * 8 + 64, dtype=torch.int64, device="cuda")
if hasattr(capi.lib, "spgpuDebugSetTrace"):
    capi.lib.spgpuDebugSetTrace.argtypes = [C.c_void_p]
    capi.lib.spgpuDebugSetTrace(C.c_void_p(trace.data_ptr()))
for form in (capi.FORM_GATHER, capi.FORM_AUTO):
    capi.spgpuSetSpmvForm(gpu, form)
    for rep in range(3):
        dz = torch.full((n,), float("nan"), dtype=dx.dtype, device="cuda")
        d.spmv(gpu, dz, dy, 2.0, dx, 0.5)
        torch.cuda.synchronize()
        got = dz.cpu().numpy()
        inv = np.empty(n, np.int64); inv[r_idx] = np.arange(n)
        bad_rows = np.sort(inv[np.flatnonzero(np.isnan(got))])   # ordered-row numbers never written
        want = O.spmv_tail(hell, x, y, 2.0, 0.5, r_idx=r_idx, **O.slab_shape(letter, "share"))
        wrong = np.sort(inv[np.flatnonzero((got != want) & ~np.isnan(got))])
        subs = np.unique(bad_rows // 32)
        runs = np.split(subs, np.flatnonzero(np.diff(subs) != 1) + 1) if subs.size else []
        print("   runs of unwritten sub-groups:", [(int(r[0]), int(r[-1])) for r in runs[:12]], "of", len(runs))
        dbg = trace[4096:4096 + 32 * 3].cpu().numpy().reshape(3, 32)
        for g in range(3):
            if dbg[g].any():
                print(f"   workgroup {g}: scout {hex(int(dbg[g][16]))}; streamers", [hex(int(v)) for v in dbg[g][:15]])
        trace.zero_()
        print(f"form {form} rep {rep}: {bad_rows.size} rows unwritten in {subs.size} sub-groups; first {subs[:12].tolist()} last {subs[-5:].tolist() if subs.size else []}; wrong values {wrong.size} first {wrong[:5].tolist()}")
