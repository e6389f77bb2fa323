#!/bin/bash
R=/root/repo
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_ALIGNED=1
for pat in band near; do
export EXP_PATTERNS=$pat
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r03_al_${pat}_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r03_al_${pat}_$C -- python3 $R/tools/exp_tile.py D 10000000 powerlaw > $R/gpurun_out/r03_al_${pat}_$C.log 2>&1 || exit 1
  echo "== $pat"; python3 $R/tools/pmc_summary.py $R/gpurun_out/r03_al_${pat}_$C raggedSpmvKernel deepItemsKernel
done
done
find $R/gpurun_out/r03_al_* -name "*.csv" -size +1M -delete
