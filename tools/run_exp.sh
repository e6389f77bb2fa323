cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_c_harness.py -x -q 2>&1 | tail -5
./tools/diaperf_amd.bin 256 7 50 d | tail -4
