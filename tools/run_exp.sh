#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -x --durations=5 2>&1 | tail -12
