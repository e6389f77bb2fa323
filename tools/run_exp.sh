#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_oell_device.py -q -m gpu -x -k "order" 2>&1 | tail -4 || exit 1
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_PATTERNS=band EXP_FORMS=ragged4,auto
EXP_ALIGNED=check timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH|Error|assert" || exit 1
EXP_ALIGNED= timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
