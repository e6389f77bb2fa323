#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r3_gpu_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r3_gpu_tests.log
