import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spgpu_amd import capi, synth
rows, L, k = 5_000_000 // 32 * 32, 32, 16
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr())
h = synth.hell_uniform_on_device(rows, L, "banded", "D", 32, seed=11)
X = synth.device_vector(rows * k, "D", 21).view(rows, k)
Z = torch.empty_like(X)
clock = torch.zeros(8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
call = lambda: capi.hellspmm["D"](handle, p(Z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), None, L, rows, p(X), 0.0, 0, k, k, k)
for v in (0,):
    os.environ["SPGPU_SPMM_VARIANT"] = str(v)
    os.environ.pop("SPGPU_SPMM_CLOCK", None)
    with torch.cuda.stream(stream):
        call()
    torch.cuda.synchronize()
    clock.zero_(); torch.cuda.synchronize()
    os.environ["SPGPU_SPMM_CLOCK"] = str(clock.data_ptr())
    n = 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        a.record(stream)
        for _ in range(n): call()
        b.record(stream)
    torch.cuda.synchronize()
    blocks = (rows + 255) // 256
    c = clock.cpu().numpy()[:6] / (n * blocks)
    print(f"variant {v}: {a.elapsed_time(b)/n:.3f} ms/launch; cycles per block: setup {c[0]:.0f} probe {c[1]:.0f} scan {c[2]:.0f} fill {c[3]:.0f} accumulate {c[4]:.0f} store {c[5]:.0f} total {c.sum():.0f}", flush=True)
