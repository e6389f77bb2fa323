#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_PATTERNS=band,near EXP_FORMS=auto EXP_ALIGNED=1
for round in 1 2 3; do
for L in lib_ab lib; do
echo "== $L"
SPGPU_LIB=/root/repo/spgpu_amd/$L/libspgpu.so timeout -k 10 300 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" | cut -c1-140 || exit 1
done
done
