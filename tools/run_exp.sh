cd /tmp && export TMPDIR=/tmp
export EXP_PATTERNS=near,band EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=ragged0
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/t1prof2 -- python3 /root/repo/tools/exp_tile.py D 10000000 powerlaw > /root/repo/gpurun_out/t1prof2.log 2>&1
cd /root/repo
grep "^D " gpurun_out/t1prof2.log
python - <<'PY'
import glob, csv
for f in glob.glob("gpurun_out/t1prof2/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "Spmv" in row["Name"] or "deep" in row["Name"]:
            print(row["Name"][:70], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"])
PY
