cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_device_scalars.py tests/test_gpu_level1.py tests/test_gpu_level1_rest.py tests/test_gpu_fused_solver.py -x -q 2>&1 | tail -4
