#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
SPGPU_RAGGED_SHAPE=4 timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_c_harness.py -q -m gpu -x 2>&1 | tail -3
