#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -4
