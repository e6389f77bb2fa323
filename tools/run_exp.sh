#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_spmm.py tests/test_gpu_sharded_c.py tests/test_gpu_share.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3 || exit 1
VARIANTS=0,10 timeout -k 10 300 python3 tools/ab_spmm.py banded window random 2>&1 | grep spmm | tee gpurun_out/ab_spmm.log
RHS=8 VARIANTS=0,10 timeout -k 10 300 python3 tools/ab_spmm.py banded 2>&1 | grep spmm | tee -a gpurun_out/ab_spmm.log
