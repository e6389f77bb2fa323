#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256,1024:256 EXP_PATTERNS=near EXP_FORMS=ragged0,ragged0@0,ragged5,ragged4
for al in "" 1; do
echo "== EXP_ALIGNED=$al"
EXP_ALIGNED=$al timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
done
