cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_oell_device.py tests/test_gpu_fuzz.py tests/test_gpu_c_harness.py tests/test_gpu_spmv.py tests/test_gpu_f3.py -x -q 2>&1 | tail -4
