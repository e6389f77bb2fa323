#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
SPGPU_LIB=/root/repo/spgpu_amd/lib_ab/libspgpu.so timeout -k 10 120 python tools/dbg_pipe.py 3 2>&1 | grep -v amdgpu.ids | grep "^form" | tee gpurun_out/r3_dbg.log
timeout -k 10 600 python -m pytest tests/test_gpu_share.py -q -m gpu -k "pipe or kernel1 or kernel2" > gpurun_out/r3_pipe_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3_pipe_tests.log
[ $rc -eq 124 ] && exit 1
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests/test_gpu_share.py -q -m gpu -k "kernel2" 2>&1 | tail -1; done
