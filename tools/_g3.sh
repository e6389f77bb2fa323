mkdir -p gpurun_out/r04c
timeout -k 10 900 python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r04c/test_plan.txt 2>&1; echo "rc=$?" >> gpurun_out/r04c/test_plan.txt
tail -n 12 gpurun_out/r04c/test_plan.txt
export EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1
for first in 0 1; do for g in 1 2 4 8; do
  echo "== deep first $first per block $g" >> gpurun_out/r04c/variants.txt
  SPGPU_PLAN_DEEP_FIRST=$first SPGPU_PLAN_DEEP_PER_BLOCK=$g timeout -k 10 200 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep power-law >> gpurun_out/r04c/variants.txt
done; done
echo "== no plan" >> gpurun_out/r04c/variants.txt
SPGPU_PLAN=0 timeout -k 10 200 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep power-law >> gpurun_out/r04c/variants.txt
cat gpurun_out/r04c/variants.txt
for first in 0 1; do
SPGPU_LIB=spgpu_amd/lib_trace/libspgpu.so SPGPU_PLAN_DEEP_FIRST=$first SPGPU_RAGGED_SHAPE=4 EXP_PATTERN=band timeout -k 10 200 python tools/exp_ragged_trace.py 10000000 2048:256 powerlaw > gpurun_out/r04c/trace_plan_first$first.txt 2>&1
done
SPGPU_LIB=spgpu_amd/lib_trace/libspgpu.so SPGPU_PLAN=0 SPGPU_RAGGED_SHAPE=4 EXP_PATTERN=band timeout -k 10 200 python tools/exp_ragged_trace.py 10000000 2048:256 powerlaw > gpurun_out/r04c/trace_noplan.txt 2>&1
cat gpurun_out/r04c/trace_*.txt
