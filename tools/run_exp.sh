#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_PATTERNS=band,near EXP_ALIGNED=1 EXP_ORDERS=2048:256
EXP_FORMS=auto%64,auto%32,auto%0,auto%128,auto%64,auto%32,auto%0,auto%128 timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
