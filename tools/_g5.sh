mkdir -p gpurun_out/r04e
timeout -k 10 900 python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r04e/test_plan.txt 2>&1; echo "rc=$?" >> gpurun_out/r04e/test_plan.txt
tail -n 6 gpurun_out/r04e/test_plan.txt
timeout -k 10 300 python tools/exp_alloc.py 10000000 6 > gpurun_out/r04e/alloc_torch.txt 2>&1; cat gpurun_out/r04e/alloc_torch.txt
EXP_KEEP=1 timeout -k 10 300 python tools/exp_alloc.py 10000000 6 > gpurun_out/r04e/alloc_torch_keep.txt 2>&1; cat gpurun_out/r04e/alloc_torch_keep.txt
EXP_ALLOC=hip timeout -k 10 300 python tools/exp_alloc.py 10000000 6 > gpurun_out/r04e/alloc_hip.txt 2>&1; cat gpurun_out/r04e/alloc_hip.txt
export EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1
for rep in 1 2 3; do
for spread in 0 30 60 100; do for g in 2 4; do
  echo "== rep $rep spread $spread per block $g" >> gpurun_out/r04e/variants.txt
  SPGPU_PLAN_DEEP_SPREAD=$spread SPGPU_PLAN_DEEP_PER_BLOCK=$g timeout -k 10 200 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep power-law >> gpurun_out/r04e/variants.txt
done; done
done
grep -B1 power-law gpurun_out/r04e/variants.txt | grep -v "^--" | paste - - | awk '{print $2,$3,$4,$5,$6,$7,$8, $(NF-11), $(NF-10)}'
