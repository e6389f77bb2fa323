#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_spmm.py -q -m gpu -x 2>&1 | tail -5 || exit 1
VARIANTS=0,14,13 timeout -k 10 300 python3 tools/ab_spmm.py banded window random 2>&1 | grep spmm | tee gpurun_out/ab_spmm_rows.log
RHS=8 VARIANTS=0,13 timeout -k 10 300 python3 tools/ab_spmm.py banded 2>&1 | grep spmm | tee -a gpurun_out/ab_spmm_rows.log
