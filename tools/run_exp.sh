#!/bin/bash
cd /root/repo
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
SPGPU_RAGGED_SHAPE=4 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_oell_device.py tests/test_gpu_spmv.py -q -m gpu -x 2>&1 | tail -3
