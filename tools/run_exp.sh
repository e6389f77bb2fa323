#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_PATTERNS=band,near EXP_FORMS=ragged0,ragged4
for cap in 256 512 1024 128; do
echo "== SPGPU_DEEP_CAP=$cap"
SPGPU_DEEP_CAP=$cap timeout -k 10 300 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
done
