#!/usr/bin/env python3
"""GPU box: the rocprofv3 artefacts bench.py's roofline block is judged against, written under gpurun_out/profile_<tag>/
(copy what is to be kept into profiles/).

    python tools/profile_bench.py <tag> spmv|spmv_frozen|spmm

  1. rocprofv3 --kernel-trace --stats -- python3 bench.py ... --no-extras      -> <tag>_bench_<kind>_kernel_stats.csv
  2. rocprofv3 --pmc FETCH_SIZE ... and --pmc WRITE_SIZE ... (separate passes)  -> <tag>_bench_<kind>_pmc.json
The program after `--` is python3 itself (no env / shell hop: the profiler's preloaded library initialises the GPU)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, kind = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
os.makedirs(out, exist_ok=True)
bench = [os.path.join(ROOT, "bench.py"), "--no-extras"] + (["--workload", "spmm"] if kind == "spmm" else []) + (["--frozen"] if kind == "spmv_frozen" else [])
# spmv_frozen: the headline matrix after spgpuHellSpmvFreeze (bench.py --frozen): the PACKED instantiation of the same kernel
kernel_key = ("hellSpmmStripKernel" if kind == "spmm" else
              "slabSpmvKernel<double, 2, 1, true, true, 8, 2, true, 0, true, 256, 0, false, 1, 0, true>" if kind == "spmv_frozen" else
              "slabSpmvKernel<double, 2, 1, true, true, 8, 2, true, 0, true, 256, 0, false, 1, 0, false>")
env = dict(os.environ, TMPDIR="/tmp")


def run(name, prof_args, bench_args):
    d = os.path.join(out, name)
    cmd = ["rocprofv3"] + prof_args + ["--output-format", "csv", "-d", d, "-o", name, "--", "python3"] + bench + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return d, (json.loads(line[-1]) if line else None), r


d, rec, r = run("stats", ["--kernel-trace", "--stats"], ["--steps", "100", "--warmup", "10"])
if rec is None:
    sys.exit("bench.py printed no record:\n" + r.stdout[-2000:] + r.stderr[-2000:])
stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
rows = [x for x in csv.DictReader(open(stats)) if kernel_key in x["Name"]]
assert rows, f"kernel {kernel_key} not in {stats}"
k = rows[0]
os.replace(stats, os.path.join(out, f"{tag}_bench_{kind}_kernel_stats.csv"))


def counter(name):
    d, _, r = run(name.lower(), ["--pmc", name, "--kernel-trace"], ["--steps", "10", "--warmup", "2"])
    vals = [float(x["Counter_Value"]) for path in glob.glob(os.path.join(d, "*counter_collection.csv"))
            for x in csv.DictReader(open(path)) if kernel_key in x["Kernel_Name"] and x["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)


fetch, n = counter("FETCH_SIZE")
write, _ = counter("WRITE_SIZE")
alg = rec["roofline"]["algorithmic_bytes_per_launch"]
read_bytes, write_bytes = 2 * fetch * 1024, write * 1024
summary = dict(
    workload=dict(rows=rec["config"].get("rows", rec["config"].get("rows_per_gpu")), nnz_per_row=32, pattern=rec["config"]["pattern"],
                  **({"rhs": rec["config"]["rhs"]} if kind == "spmm" else {})),
    command="rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-extras"
            + (" --workload spmm" if kind == "spmm" else "") + (" --frozen" if kind == "spmv_frozen" else "")
            + " ; counters in separate runs: --pmc FETCH_SIZE --kernel-trace, --pmc WRITE_SIZE --kernel-trace (--steps 10 --warmup 2)",
    kernel=k["Name"], calls=int(k["Calls"]), average_ns=float(k["AverageNs"]), min_ns=int(k["MinNs"]), max_ns=int(k["MaxNs"]),
    bench_kernel_ms_same_run=rec["roofline"]["kernel_ms"], bench_value_gflops_same_run=rec["value"],
    FETCH_SIZE_KiB_mean=fetch, WRITE_SIZE_KiB_mean=write, dispatches_counted=n,
    correction="MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads "
               "-> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE exact (KiB)",
    hbm_read_bytes_corrected=read_bytes, hbm_write_bytes=write_bytes, hbm_traffic_bytes_per_launch=read_bytes + write_bytes,
    algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(read_bytes + write_bytes) / alg)
with open(os.path.join(out, f"{tag}_bench_{kind}_pmc.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps({key: summary[key] for key in ("kernel", "average_ns", "bench_kernel_ms_same_run", "traffic_over_algorithmic")}))
