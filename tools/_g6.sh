mkdir -p gpurun_out/r04f
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04f/gpu_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04f/gpu_tests.txt
tail -n 8 gpurun_out/r04f/gpu_tests.txt
EXP_POOLS=2 timeout -k 10 400 python tools/exp_alloc_pool.py 10000000 > gpurun_out/r04f/alloc_pool.txt 2>&1; cat gpurun_out/r04f/alloc_pool.txt
export EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1
EXP_SWEEP="SPGPU_PLAN_DEEP_SPREAD=-1,0,15,30,60;SPGPU_PLAN_DEEP_PER_BLOCK=2,4,8" EXP_SWEEP_REPS=3 timeout -k 10 600 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04f/sweep_aligned.txt 2>&1; cat gpurun_out/r04f/sweep_aligned.txt
EXP_ALIGNED= EXP_SWEEP="SPGPU_PLAN=0,1" EXP_SWEEP_REPS=3 timeout -k 10 600 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04f/sweep_drift.txt 2>&1; cat gpurun_out/r04f/sweep_drift.txt
