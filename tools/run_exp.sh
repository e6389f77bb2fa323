#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_PATTERNS=band,near EXP_ALIGNED=1 EXP_FORMS=auto
EXP_ORDERS=2048:256:256,2048:320:320,2048:384:384,2048:512:512,2048:256:256 timeout -k 10 900 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
