#!/bin/bash
R=/root/repo
cd /tmp; export TMPDIR=/tmp
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_ALIGNED=1
for L in lib_ab lib; do
for pat in band near; do
export SPGPU_LIB=$R/spgpu_amd/$L/libspgpu.so EXP_PATTERNS=$pat
rm -rf $R/gpurun_out/ks_$L_$pat
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_${L}_$pat -o pl -- python3 $R/tools/exp_tile.py D 10000000 powerlaw > $R/gpurun_out/ks_${L}_$pat.log 2>&1 || exit 1
echo "== $L $pat"; grep "^D " $R/gpurun_out/ks_${L}_$pat.log
python3 - <<PY
import csv, glob
for path in glob.glob('$R/gpurun_out/ks_${L}_$pat/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if any(k in r['Name'] for k in ('raggedSpmv','deepItems','deepFinish')):
            print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
done
find $R/gpurun_out/ks_* -name "*.csv" -size +1M -delete
