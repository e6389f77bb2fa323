cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_c_harness.py -x -q 2>&1 | tail -5
./tools/hellperf_amd.bin 2000000 32 banded 50 d | tail -6
