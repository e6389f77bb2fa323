#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 800 python3 tools/profile_bench.py r03 spmv 2>&1 | tail -3 || exit 1
timeout -k 10 600 python3 tools/profile_bench.py r03 spmm 2>&1 | tail -3 || exit 1
# the ordered power-law target: kernel durations and HBM counters of one exp_tile run (AUTO: the library's own choice)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_pl_stats -o pl -- python3 $R/tools/exp_tile.py D 10000000 powerlaw > $R/gpurun_out/r03_pl_stats.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r03_pl_$C -- python3 $R/tools/exp_tile.py D 10000000 powerlaw > $R/gpurun_out/r03_pl_$C.log 2>&1 || exit 1
done
cd $R
grep "^D " gpurun_out/r03_pl_stats.log
python3 - <<'PY'
import csv, glob
for path in glob.glob('/root/repo/gpurun_out/r03_pl_stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if any(k in r['Name'] for k in ('raggedSpmv','deepItems','deepFinish','orderedProbe')):
            print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
for C in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_summary.py gpurun_out/r03_pl_$C raggedSpmvKernel deepItemsKernel deepFinishKernel; done
find gpurun_out/r03_pl_* gpurun_out/profile_r03 -name "*.csv" -size +3M -delete
