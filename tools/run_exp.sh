#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
SPGPU_LIB=/root/repo/spgpu_amd/lib_lab/libspgpu.so timeout -k 10 1100 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_share.py tests/test_gpu_spmm.py -q -m gpu -x > gpurun_out/gpu_tests_lab.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_lab.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log || { tail -5 gpurun_out/bench_err.log; exit 1; }
tail -c 1500 gpurun_out/bench_line.json
