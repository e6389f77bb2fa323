mkdir -p gpurun_out/r04d
timeout -k 10 900 python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r04d/test_plan.txt 2>&1; echo "rc=$?" >> gpurun_out/r04d/test_plan.txt
tail -n 6 gpurun_out/r04d/test_plan.txt
export EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1
for rep in 1 2; do
for spread in -1 0 25 50 100; do for g in 2 4; do
  echo "== rep $rep spread $spread per block $g" >> gpurun_out/r04d/variants.txt
  SPGPU_PLAN_DEEP_SPREAD=$spread SPGPU_PLAN_DEEP_PER_BLOCK=$g timeout -k 10 200 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep power-law >> gpurun_out/r04d/variants.txt
done; done
echo "== rep $rep no plan" >> gpurun_out/r04d/variants.txt
SPGPU_PLAN=0 timeout -k 10 200 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep power-law >> gpurun_out/r04d/variants.txt
done
grep -B1 power-law gpurun_out/r04d/variants.txt | grep -v "^--" | paste - - | awk '{print $2,$3,$4,$5,$6,$7,$8, $(NF-11), $(NF-10)}'
for spread in 50; do
SPGPU_LIB=spgpu_amd/lib_trace/libspgpu.so SPGPU_PLAN_DEEP_SPREAD=$spread SPGPU_RAGGED_SHAPE=4 EXP_PATTERN=band timeout -k 10 200 python tools/exp_ragged_trace.py 10000000 2048:256 powerlaw > gpurun_out/r04d/trace_plan_spread$spread.txt 2>&1
done
cat gpurun_out/r04d/trace_*.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04d/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/exp_tile.py D 10000000 powerlaw > $GRAFT_REPO_ROOT/gpurun_out/r04d/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r04d raggedSpmvKernel deepItemsKernel > gpurun_out/r04d/pmc_summary.txt 2>&1
cat gpurun_out/r04d/pmc_summary.txt
find gpurun_out/r04d -name "*.csv" -size +1M -delete
