#!/usr/bin/env python3
"""GPU box: the rocprofv3 evidence for the north_star target (ordered power-law HELL fp64 through rIdx), written under
gpurun_out/profile_<tag>/<tag>_powerlaw_kernel_stats.txt (copy into profiles/).

    python tools/profile_powerlaw.py <tag> [aligned|drift] [plain] [frozen] [adopted]

  1. rocprofv3 --kernel-trace --stats -- python3 tools/exp_tile.py D 10000000 powerlaw   (EXP_PATTERNS=band,near, AUTO)
  2. per pattern, separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE (KiB per launch; FETCH_SIZE x 2 on gfx950 for wide reads)
  `plain`: the rows as they come (no order, slabSpmvKernel) as well -- what does the natural order move?
The program after `--` is python3 itself (no env / shell hop: the profiler's preloaded library initialises the GPU)."""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
order = sys.argv[2] if len(sys.argv) > 2 else "aligned"
plain = "plain" in sys.argv[3:]
frozen = "frozen" in sys.argv[3:]      # spgpuHellSpmvFreeze before the timed calls (EXP_FREEZE=1)
adopted = "adopted" in sys.argv[3:]    # the rows as they come, spgpuHellSpmvAdopt before the timed calls (EXP_ADOPT=1; implies `plain`, no ordered layouts)
out = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
os.makedirs(out, exist_ok=True)
base_env = dict(os.environ, TMPDIR="/tmp", EXP_ORDERS="2048:256", EXP_FORMS="auto")
if order == "aligned":
    base_env["EXP_ALIGNED"] = "1"
if not plain:
    base_env["EXP_ONLY_WINDOWED"] = "1"
if frozen:
    base_env["EXP_FREEZE"] = "1"
if adopted:
    base_env.pop("EXP_ONLY_WINDOWED", None)
    base_env["EXP_ADOPT"] = "1"
    base_env["EXP_ONLY_PLAIN"] = "1"
    base_env["EXP_GLOBAL_FORMS"] = "auto"
exp = ["python3", os.path.join(ROOT, "tools", "exp_tile.py"), "D", "10000000", "powerlaw"]
KERNELS = ("raggedSpmvKernel", "deepItemsKernel", "deepFinishKernel", "slabSpmvKernel", "planBlocksKernel", "planListKernel", "planPackKernel", "planSlotsKernel",
           "orderedProbeKernel")
lines = []


def run(name, prof_args, patterns):
    d = os.path.join(out, name)
    env = dict(base_env, EXP_PATTERNS=patterns)
    r = subprocess.run(["rocprofv3"] + prof_args + ["--output-format", "csv", "-d", d, "-o", name, "--"] + exp, cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=900)
    return d, [ln for ln in r.stdout.splitlines() if "power-law" in ln], r


def short(name):
    for k in KERNELS:
        if k in name:
            args = name[name.index(k):]
            return args if len(args) < 150 else args[:150] + "..."
    return None


def ragged_kind(name):
    """raggedSpmvKernel<T, RPL, IS_HELL, UNROLL, WAVES, TILE, SUBS, DEEP, ZBYTES, PLAN, PACKED>: which of the three it is"""
    if "raggedSpmvKernel<" not in name:
        return ""
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    if len(args) >= 11 and args[10].startswith("true"):
        return "<PLAN, PACKED>"
    return "<PLAN>" if len(args) >= 10 and args[9].startswith("true") else ""


lines.append(f"The north_star target as tools/exp_tile.py runs it (AUTO; order: {order}{', and the rows as they come' if plain else ''}{'; FROZEN: spgpuHellSpmvFreeze first, EXP_FREEZE=1' if frozen else ''}{'; the rows as they come, ADOPTED first: spgpuHellSpmvAdopt, EXP_ADOPT=1 (the bit check compares with the PLAIN kernel and says MISMATCH: another order of additions)' if adopted else ''}), profiled with")
lines.append("  rocprofv3 --kernel-trace --stats -- python3 tools/exp_tile.py D 10000000 powerlaw   (EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near)")
lines.append("and, per pattern, in separate passes, --pmc FETCH_SIZE / --pmc WRITE_SIZE (KiB per launch; FETCH_SIZE x 2 on gfx950 for wide reads).")
d, said, r = run("stats", ["--kernel-trace", "--stats"], "band,near")
lines.append("")
lines += said
lines.append("")
lines.append(f"{'kernel':150s} {'calls':>6s} {'average ns':>12s} {'min':>10s} {'max':>10s}")
for path in glob.glob(os.path.join(d, "*kernel_stats.csv")):
    for row in csv.DictReader(open(path)):
        name = short(row["Name"])
        if name:
            lines.append(f"{name:150s} {row['Calls']:>6s} {float(row['AverageNs']):12.0f} {row['MinNs']:>10s} {row['MaxNs']:>10s}")
for pattern in ("band", "near"):
    sums, counts = defaultdict(float), defaultdict(int)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d, said, r = run(f"{counter.lower()}_{pattern}", ["--pmc", counter, "--kernel-trace"], pattern)
        for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for row in csv.DictReader(open(path)):
                name = short(row["Kernel_Name"])
                if name and row["Counter_Name"] == counter:
                    key = (name.split("<")[0] + ragged_kind(name), counter)
                    sums[key] += float(row["Counter_Value"])
                    counts[key] += 1
    lines.append("")
    lines.append(f"{pattern}: counters per launch (mean)")
    for (name, counter), total in sorted(sums.items()):
        mean = total / counts[(name, counter)]
        gb = mean * 1024 * (2 if counter == "FETCH_SIZE" else 1) * 1e-9
        lines.append(f"  {name:40s} {counter:11s} {mean:12.4g} KiB  ({'x2 = ' if counter == 'FETCH_SIZE' else ''}{gb:.3f} GB)   {counts[(name, counter)]} launches")
with open(os.path.join(out, f"{tag}_powerlaw_kernel_stats{'_' + order if order != 'aligned' else ''}{'_frozen' if frozen else ''}{'_adopted' if adopted else ''}.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
for path in glob.glob(os.path.join(out, "**", "*.csv"), recursive=True):
    if os.path.getsize(path) > (1 << 20):
        os.remove(path)
