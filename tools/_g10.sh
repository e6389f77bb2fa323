mkdir -p gpurun_out/r04j
timeout -k 10 600 python -m pytest tests/test_gpu_convert_device.py tests/test_gpu_device_scalars.py tests/test_gpu_sharded_c.py tests/test_gpu_bench_rehearsal.py tests/test_gpu_spmv.py -x -q -m gpu > gpurun_out/r04j/gpu_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04j/gpu_tests.txt
tail -n 5 gpurun_out/r04j/gpu_tests.txt
timeout -k 10 900 python tools/profile_powerlaw.py r04 aligned > gpurun_out/r04j/profile_powerlaw.log 2>&1; tail -n 40 gpurun_out/r04j/profile_powerlaw.log
SPGPU_LIB=spgpu_amd/lib_trace/libspgpu.so SPGPU_RAGGED_SHAPE=4 EXP_ALIGNED=1 EXP_PATTERN=band timeout -k 10 200 python tools/exp_ragged_trace.py 10000000 2048:256 powerlaw > gpurun_out/r04j/trace_plan_aligned.txt 2>&1; cat gpurun_out/r04j/trace_plan_aligned.txt
timeout -k 10 900 python tools/profile_hdia.py r04 > gpurun_out/r04j/profile_hdia.log 2>&1; tail -n 60 gpurun_out/r04j/profile_hdia.log
(rocprofv3 --list-avail 2>/dev/null | grep -i "EA0_RDREQ\|MALL\|DRAM\|HBM" | head -40) > gpurun_out/r04j/counters_avail.txt; cat gpurun_out/r04j/counters_avail.txt | head -30
