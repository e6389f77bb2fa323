#!/bin/bash
cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_gpu_oell_device.py -q -m gpu -x -k "aligned" 2>&1 | tail -4
