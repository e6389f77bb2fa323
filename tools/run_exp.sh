#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_fuzz.py tests/test_gpu_runtime.py -q -m gpu -x 2>&1 | tail -4
