"""GPU: bench.py's multi-rank path end to end with two ranks sharing the one GPU of the test box.  RCCL refuses two ranks on
one device, so the collectives are staged through host memory over gloo (SPGPU_BENCH_BACKEND=gloo, bench.py's
HostStagedCollectives); everything else -- partition, own/rest split, needed-rows request lists, the products through
the C ABI, the oracle check on rank 0 -- is the code the driver runs on 2/4/8 GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("exchange,pattern", [("needed", "banded"), ("allgather", "window")])
def test_two_ranks_on_one_gpu(gpu, exchange, pattern):
    env = dict(os.environ, SPGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300 + (1 if exchange == "needed" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--spmm-rows-per-gpu", "200000", "--spmm-pattern", pattern, "--exchange", exchange]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "GFLOP/s"
    assert "MISMATCH" not in out["parity"] and "oracle" in out["parity"]
    assert out["config"]["rows_total"] == 400000
    assert "SpMM" in out["metric"] and "2 GPUs" in out["metric"]
    # first-contact diagnostics (round 4): every rank checked its own rows against the oracle, and said so to the others
    assert out["spmm"]["rccl_ranks_seen"] == 2 and out["spmm"]["ranks_whose_window_matches_the_oracle"] == 2
    assert out["spmm"]["exchange_bytes_per_rank"]["allgather"] == 200000 * 16 * 8
    if exchange == "needed":
        assert out["spmm"]["needed_rows_received_per_rank"] == 31        # 16 rows below the block, 15 above (wrapped band)
        assert out["spmm"]["allgather_step_ms"] > 0 and out["as_named_allgather"]["value"] > 0
    else:   # the named configuration leads; the needed-rows exchange is measured beside it
        assert "all-gather" in out["config"]["exchange"] and out["variants"]["needed_rows"]["value"] > 0


def test_gpus_two_direct_form(gpu):
    """`python bench.py --gpus 2` as the driver types it for N = 1 (no launcher around it): bench.py has to start the two
    ranks itself and the record has to say n_gpus = 2."""
    env = dict(os.environ, SPGPU_BENCH_BACKEND="gloo")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--spmm-rows-per-gpu", "100000"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rows_total"] == 200000 and "MISMATCH" not in out["parity"]
    assert out["config"]["exchange"] == "all-gather"      # BASELINE configs[4] as named is the default
