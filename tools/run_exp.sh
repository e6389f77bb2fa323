#!/bin/bash
# Scratch wrapper for one-off runs on the GPU box:  gpurun -- 'bash tools/run_exp.sh'
# (edit the commands below; everything that is kept lives in tools/*.py and profiles/)
cd /root/repo
export EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=ragged0
timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D "
