#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3 || exit 1
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_PATTERNS=band,near EXP_FORMS=auto EXP_ALIGNED=1
for round in 1 2 3; do
for L in lib_ab lib; do
echo "== $L"
SPGPU_LIB=/root/repo/spgpu_amd/$L/libspgpu.so timeout -k 10 300 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" | cut -c1-140 || exit 1
done
done
