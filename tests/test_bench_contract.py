"""CPU: the parts of bench.py's contract that need no GPU -- the algorithmic byte counts the roofline fractions are
computed from (matrix once, x once, z once, no padding, no over-fetch) and the committed HBM-counter
summaries they are compared with."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_of_the_named_workloads():
    rows, per_row = 10_000_000, 32
    nnz, hacks = rows * per_row, rows // 32
    # configs[1]: 12 B per nonzero + rS + hackOffsets + x + z
    assert bench.hell_algorithmic_bytes(nnz, rows, rows, hacks) == nnz * 12 + rows * 4 + hacks * 4 + 2 * rows * 8 == 4_041_250_000
    assert bench.hell_algorithmic_bytes(nnz, rows, rows, hacks, beta_nonzero=True) - bench.hell_algorithmic_bytes(nnz, rows, rows, hacks) == rows * 8
    # the SpMM point: the matrix once, 16 columns of X and Z
    r5 = 5_000_000
    assert bench.hell_algorithmic_bytes(r5 * 32, r5, r5, r5 // 32, rhs=16) == r5 * 32 * 12 + r5 * 4 + r5 // 32 * 4 + 16 * 2 * r5 * 8 == 3_220_625_000
    # fp32: 8 B per nonzero
    assert bench.hell_algorithmic_bytes(nnz, rows, rows, hacks, elem=4) == nnz * 8 + rows * 4 + hacks * 4 + 2 * rows * 4


def test_committed_counter_summaries_match_their_workloads():
    spmv = bench.committed_traffic(10_000_000, 32, "banded")
    spmm = bench.committed_traffic(5_000_000, 32, "banded", rhs=16)
    assert spmv is not None and spmm is not None
    assert 1.0 <= spmv / 4_041_250_000 < 1.01          # HBM traffic = algorithmic bytes within 1 %
    assert 1.0 <= spmm / 3_220_625_000 < 1.05
    assert bench.committed_traffic(10_000_000, 32, "random") is None   # no counters committed for that pattern
    for name in os.listdir(os.path.join(ROOT, "profiles")):
        if name.endswith("_pmc.json") and "_bench_" in name:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if "bench_kernel_ms_same_run" not in d:      # the first summaries of the round predate that field
                continue
            assert abs(d["average_ns"] * 1e-6 - d["bench_kernel_ms_same_run"]) / d["bench_kernel_ms_same_run"] < 0.05, name
            assert d["hbm_traffic_bytes_per_launch"] == d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"], name


def test_gpus_flag_is_honoured_or_refused(monkeypatch):
    """`--gpus N` must never silently run fewer ranks: with a launcher around it WORLD_SIZE has to equal N, and without one
    bench.py starts the N ranks itself as a child process (no GPU call, no exec in the parent)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert run.returncode != 0 and "WORLD_SIZE=1" in run.stderr

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    code = bench.launch_ranks(bench.parse())
    assert code == 7                                                   # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
