#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export SPGPU_LIB=/root/repo/spgpu_amd/lib_ab/libspgpu.so
for shape in 4 0; do
SPGPU_RAGGED_SHAPE=$shape EXP_PATTERN=band timeout -k 10 300 python3 tools/exp_ragged_trace.py 2>&1 | grep -v amdgpu.ids || exit 1
done
SPGPU_RAGGED_SHAPE=4 EXP_PATTERN=band timeout -k 10 300 python3 tools/exp_ragged_trace.py 10000000 2048:256 even 2>&1 | grep -v amdgpu.ids || exit 1
