#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
SPGPU_LIB=/root/repo/spgpu_amd/lib_lab/libspgpu.so timeout -k 10 1100 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_share.py tests/test_gpu_spmm.py -q -m gpu -x > gpurun_out/gpu_tests_lab.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_lab.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 800 python3 tools/profile_bench.py r03 spmv 2>&1 | tail -1 || exit 1
timeout -k 10 600 python3 tools/profile_bench.py r03 spmm 2>&1 | tail -1 || exit 1
find gpurun_out/profile_r03 -name "*.csv" -size +3M -delete
timeout -k 10 900 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log || { tail -5 gpurun_out/bench_err.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step']}, d['roofline']['frac'])
for k,v in d['target'].items(): print(k, v)
print(d['spmm_1gpu']['ms_per_step'], d['spmm_1gpu']['roofline_frac'])
PY
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
