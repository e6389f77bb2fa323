#!/usr/bin/env python3
"""GPU box: counters of the HDIA kernel on BASELINE configs[3] (7-point Laplacian 512^3, fp64) -> gpurun_out/profile_<tag>/<tag>_hdia_pmc.json.
    python tools/profile_hdia.py <tag> [grid]
Separate rocprofv3 --pmc passes (a pass that names a counter the card does not have is reported and skipped):
FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum |
TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum; the question they answer: is the x traffic beyond "once" served by L2, by the Infinity Cache or by HBM,
and would an LDS slice of x (reference: hdia_spmv_base_template.cuh:94-148 stages offsets per warp) take anything off the fabric?"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
grid = sys.argv[2] if len(sys.argv) > 2 else "512"
out = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp", SETTINGS="0,1,0,512")
cmd = ["python3", os.path.join(ROOT, "tools", "ab_hdia.py"), grid]
passes = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum"], ["TCC_EA0_RDREQ_32B_sum"], ["TA_BUSY_avr", "TCP_PENDING_STALL_CYCLES_sum"], ["TCC_REQ_sum", "TCC_READ_sum"]]
result = {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/ab_hdia.py " + grid + "  (SETTINGS=0,1,0,512: the default kernel shape)", "counters_mean_per_launch": {}, "skipped": {}}
said = None
for counters in passes:
    d = os.path.join(out, "hdia_" + "_".join(c.lower() for c in counters))
    r = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd, cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if "median" in ln]
    said = said or (lines[0] if lines else None)
    got = {}
    for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for row in csv.DictReader(open(path)):
            if "hdiaSpmvKernel" in row["Kernel_Name"]:
                got.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not got:
        result["skipped"][" ".join(counters)] = (r.stderr.strip().splitlines() or ["no rows"])[-1][:300]
    for name, values in got.items():
        result["counters_mean_per_launch"][name] = dict(mean=sum(values) / len(values), launches=len(values))
    for path in glob.glob(os.path.join(d, "*.csv")):
        if os.path.getsize(path) > (1 << 20):
            os.remove(path)
c = result["counters_mean_per_launch"]
n = int(grid) ** 3
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    read_b, write_b = 2 * c["FETCH_SIZE"]["mean"] * 1024, c["WRITE_SIZE"]["mean"] * 1024
    result["fabric_read_bytes_x2_corrected"] = read_b
    result["fabric_write_bytes"] = write_b
    result["x_and_z_once_bytes"] = 2 * n * 8
result["timing_under_the_profiler"] = said
with open(os.path.join(out, f"{tag}_hdia_pmc.json"), "w") as f:
    json.dump(result, f, indent=1)
print(json.dumps(result, indent=1))
