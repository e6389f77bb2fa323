cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_fused_solver.py tests/test_gpu_device_scalars.py tests/test_gpu_level1.py -x -q 2>&1 | tail -4 &&
timeout -k 10 120 ./tools/cg_amd.bin 1024 60 1e-30 2>&1 | tail -4
