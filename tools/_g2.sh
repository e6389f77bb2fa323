set -x
mkdir -p gpurun_out/r04b
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r04b/test_plan.txt 2>&1; echo "rc=$?" >> gpurun_out/r04b/test_plan.txt
tail -n 15 gpurun_out/r04b/test_plan.txt
EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04b/aligned_plan.txt 2>&1
SPGPU_PLAN=0 EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04b/aligned_noplan.txt 2>&1
cat gpurun_out/r04b/aligned_plan.txt gpurun_out/r04b/aligned_noplan.txt
