cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15
