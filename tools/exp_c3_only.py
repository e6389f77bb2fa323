import ctypes as C, sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, bench
from spgpu_amd import capi
h=capi.create_handle(0); s=torch.cuda.Stream(); capi.spgpuSetStream(h, C.c_void_p(s.cuda_stream))
out=bench.bench_c3(h, s, "cuda:0", 10_000_000, 2_000_000)
print(json.dumps(out["hell_fp32"])); print(json.dumps(out["ell_fp32"])); print(json.dumps(out["hell_fp32_rows_ordered"]))
