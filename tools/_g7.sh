mkdir -p gpurun_out/r04g
ls /sys/class/drm/ > gpurun_out/r04g/sysfs_ls.txt 2>&1
EXP_IDLE=3 timeout -k 10 300 python tools/exp_clocks.py 10000000 4 > gpurun_out/r04g/clocks_powerlaw.txt 2>&1; cat gpurun_out/r04g/clocks_powerlaw.txt
EXP_KIND=uniform timeout -k 10 300 python tools/exp_clocks.py 10000000 3 > gpurun_out/r04g/clocks_uniform.txt 2>&1; grep -v "^files\|hwmon files" gpurun_out/r04g/clocks_uniform.txt
timeout -k 10 600 python -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_share.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04g/gpu_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04g/gpu_tests.txt
tail -n 5 gpurun_out/r04g/gpu_tests.txt
