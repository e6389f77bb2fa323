#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_spmm.py -q -m gpu -x 2>&1 | tail -3 || exit 1
for round in 1 2; do
for L in lib_ab lib; do
echo "== $L"
SPGPU_LIB=/root/repo/spgpu_amd/$L/libspgpu.so VARIANTS=0 timeout -k 10 300 python3 tools/ab_spmm.py banded 2>&1 | grep spmm || exit 1
SPGPU_LIB=/root/repo/spgpu_amd/$L/libspgpu.so RHS=8 VARIANTS=0 timeout -k 10 300 python3 tools/ab_spmm.py banded 2>&1 | grep spmm || exit 1
done
done
