#!/bin/bash
# GPU box, end of a round: the whole -m gpu suite on the product build, the shape / lab tests on the -DSPGPU_TUNING_VARIANTS build,
# the rocprofv3 artefacts of the headline and the SpMM line, the bench line, the smoke test.  (tag: $1, default r04)
cd /root/repo
tag=${1:-r04}
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
SPGPU_LIB=/root/repo/spgpu_amd/lib_lab/libspgpu.so timeout -k 10 1100 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_share.py tests/test_gpu_spmm.py tests/test_gpu_plan.py tests/test_gpu_fuzz.py tests/test_gpu_padding.py tests/test_gpu_freeze.py tests/test_gpu_adopt.py -q -m gpu -x > gpurun_out/gpu_tests_lab.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_lab.log
[ $rc -eq 0 ] || exit 1
# the ordered paths once more with the library's uninitialised scratch filled with 0xFF, and with the LDS of every CU filled with -1 / NaN in front of every call
SPGPU_POISON_SCRATCH=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_plan.py tests/test_gpu_oell_device.py tests/test_gpu_padding.py tests/test_gpu_freeze.py tests/test_gpu_adopt.py -q -m gpu -x > gpurun_out/gpu_tests_poison.log 2>&1; rc=$?
tail -1 gpurun_out/gpu_tests_poison.log
[ $rc -eq 0 ] || exit 1
SPGPU_LIB=/root/repo/spgpu_amd/lib_lab/libspgpu.so EXP_LDS_WORD=0xffffffff timeout -k 10 300 python3 tools/stress_lds.py 16 2>&1 | tail -1
timeout -k 10 800 python3 tools/profile_bench.py $tag spmv 2>&1 | tail -1 || exit 1
timeout -k 10 600 python3 tools/profile_bench.py $tag spmv_frozen 2>&1 | tail -1 || exit 1
timeout -k 10 600 python3 tools/profile_bench.py $tag spmm 2>&1 | tail -1 || exit 1
find gpurun_out/profile_$tag -name "*.csv" -size +3M -delete
timeout -k 10 900 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log || { tail -5 gpurun_out/bench_err.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step']}, d['roofline']['frac'], d['roofline'].get('kernel_ms_blocks'))
for k,v in d['target'].items(): print(k, v)
print(d['device'].get('unique_id'))
print(d['spmm_1gpu']['ms_per_step'], d['spmm_1gpu']['roofline_frac'])
for k,v in d['configs'].items():
    if k != 'powerlaw_fp64': print(k, json.dumps(v)[:600])
PY
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
