#!/bin/bash
cd /root/repo
timeout -k 10 300 python tools/exp_small.py 1024 0 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/exp_small.py 2048 0 2>&1 | grep -v amdgpu.ids
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
