#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:0,2048:256 EXP_PATTERNS=band EXP_FORMS=ragged4,ragged0
timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
