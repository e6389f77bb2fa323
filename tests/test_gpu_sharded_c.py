"""GPU: the row-sharded SpMM driver in C (include/spgpu/sharded.h) on the one GPU of the test box.

One rank with a real RCCL communicator (ncclCommInitRank, world 1; RCCL refuses two ranks on one device): the library
opens RCCL by dlopen, the needed-rows set-up on the device (sorted unique columns, renumbering, the cut at block
boundaries, the request lists), the packing kernel, the stream ordering of a step, both exchanges, against the oracle's
product of the UNSPLIT matrix.  And 2 ... 8 ranks as THREADS over an in-process stand-in for RCCL (tests/mock_rccl.c):
the send/recv legs between different ranks, equal and unequal blocks, each rank's rows against the oracle's product of
the UNSPLIT matrix."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def comm(gpu):
    from spgpu_amd import capi
    assert capi.spgpuCommAvailable() == 1, "libspgpu.so could not open RCCL"
    ident = (C.c_char * 128)()
    assert capi.spgpuCommGetUniqueId(ident) == capi.SPGPU_SUCCESS
    handle = C.c_void_p()
    assert capi.spgpuCommInitRank(C.byref(handle), 1, ident, 0) == capi.SPGPU_SUCCESS
    yield handle
    capi.spgpuCommDestroy(handle)


@pytest.mark.parametrize("exchange", ["needed", "allgather"])
@pytest.mark.parametrize("pattern,with_comm", [("banded", True), ("window", True), ("random", False)])
def test_one_rank_split_product_equals_oracle(gpu, comm, exchange, pattern, with_comm):
    """The rank's block cut by column ownership (columns below n/2 are "own", the others "rest"): every row the second
    product reads travels through the plan's exchange buffer -- for one rank the packing kernel and the self leg."""
    import torch
    from spgpu_amd import capi, synth
    n, L, k = 64000, 24, 16
    block = synth.hell_uniform_on_device(n, L, pattern, "D", 32, seed=5)
    own, rest = synth.split_uniform_hell_by_columns(block, 0, n // 2)
    x = synth.device_vector(n * k, "D", 7).view(n, k)
    y = synth.device_vector(n * k, "D", 8).view(n, k)
    first = (C.c_longlong * 2)(0, n)
    plan = capi.ShardedPlan()
    ob, rb = capi.hell_block(own, L), capi.hell_block(rest, L)
    torch.cuda.synchronize()
    kind = capi.EXCHANGE_NEEDED if exchange == "needed" else capi.EXCHANGE_ALLGATHER
    assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, comm if with_comm else None, 0, 1, first, C.byref(ob), C.byref(rb), k,
                                            kind) == capi.SPGPU_SUCCESS
    try:
        hold = synth.hell_rows_to_host(block, 0, n)
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        for alpha, beta, in_place in ((1.0, 0.0, False), (-0.5, 0.75, False), (2.0, 1.0, True)):
            z = y.clone() if in_place else torch.full((n, k), float("nan"), dtype=torch.float64, device="cuda")
            assert capi.spgpuDhellspmmShardedStep(plan, _dp(z), _dp(z if in_place else y), alpha, _dp(x), beta) == capi.SPGPU_SUCCESS
            torch.cuda.synchronize()
            want = O.hell_spmm(hold, xs, ys if beta != 0 else None, alpha, beta)
            assert np.max(np.abs(z.cpu().numpy() - want) / (np.abs(want) + 1.0)) <= 1e-12     # own + rest regroup the sums
        rows = C.c_longlong(0)
        assert capi.spgpuDhellspmmShardedExchanged(plan, C.byref(rows))
        distinct = np.unique(rest["rP"][:rest["slots"]].cpu().numpy()[_real(rest)]).size
        assert rows.value == (n if exchange == "allgather" else distinct)
        assert capi.spgpuDhellspmmShardedRowsReceived(plan) == 0          # one rank: nothing comes from other ranks
    finally:
        capi.spgpuDhellspmmShardedDestroy(plan)


def _real(part):
    """mask of the index slots that hold a real entry (slot k of a row is real iff k < rS[row])"""
    hs = part["hack_size"]
    ho = part["hack_offsets"].cpu().numpy().astype(np.int64)
    rs = part["rS"].cpu().numpy()
    slots = int(part["slots"])
    mask = np.zeros(slots, bool)
    ends = np.append(ho[1:], slots)
    for h in range(ho.size):
        depth = (ends[h] - ho[h]) // hs
        lens = rs[h * hs:(h + 1) * hs]
        mask[ho[h]:ends[h]] = (np.arange(depth)[:, None] < lens[None, :]).reshape(-1)
    return mask


def test_step_with_true_ownership(gpu, comm):
    """world 1, the block owns all columns: own == the matrix, rest empty; Step == spgpuDhellspmm bit for bit, with and
    without a communicator, both exchanges; and a block WITH foreign columns through Step: rest's rows all come from the
    rank itself (the self leg of the exchange)."""
    import torch
    from spgpu_amd import capi, synth
    n, L, k = 32000, 32, 16
    block = synth.hell_uniform_on_device(n, L, "window", "D", 32, seed=9)
    block["slots"] = block["nnz"]
    x = synth.device_vector(n * k, "D", 7).view(n, k)
    y = synth.device_vector(n * k, "D", 8).view(n, k)
    direct = torch.empty_like(y)
    capi.hellspmm["D"](gpu, _dp(direct), _dp(y), C.c_double(1.5), _dp(block["cM"]), _dp(block["rP"]), 32, _dp(block["hack_offsets"]),
                       _dp(block["rS"]), None, L, n, _dp(x), C.c_double(-0.25), 0, k, k, k)
    torch.cuda.synchronize()
    first = (C.c_longlong * 2)(0, n)
    ob = capi.hell_block(block, L)
    for kind in (capi.EXCHANGE_NEEDED, capi.EXCHANGE_ALLGATHER):
        for c in (comm, None):
            plan = capi.ShardedPlan()
            assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, c, 0, 1, first, C.byref(ob), None, k, kind) == capi.SPGPU_SUCCESS
            z = torch.full_like(y, float("nan"))
            assert capi.spgpuDhellspmmShardedStep(plan, _dp(z), _dp(y), 1.5, _dp(x), -0.25) == capi.SPGPU_SUCCESS
            torch.cuda.synchronize()
            assert z.cpu().numpy().tobytes() == direct.cpu().numpy().tobytes()
            capi.spgpuDhellspmmShardedDestroy(plan)
    # own = columns < n/2 (no rebasing needed: the block starts at 0), rest = the others, fetched from the rank itself
    own, rest = synth.split_uniform_hell_by_columns(block, 0, n // 2)
    ob, rb = capi.hell_block(own, L), capi.hell_block(rest, L)
    hold = synth.hell_rows_to_host(block, 0, n)
    want = O.hell_spmm(hold, x.cpu().numpy(), y.cpu().numpy(), 1.5, -0.25)
    for kind in (capi.EXCHANGE_NEEDED, capi.EXCHANGE_ALLGATHER):
        plan = capi.ShardedPlan()
        assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, comm, 0, 1, first, C.byref(ob), C.byref(rb), k, kind) == capi.SPGPU_SUCCESS
        for _ in range(3):      # repeated steps reuse the buffers: ordering of exchange against the previous second product
            z = torch.full_like(y, float("nan"))
            assert capi.spgpuDhellspmmShardedStep(plan, _dp(z), _dp(y), 1.5, _dp(x), -0.25) == capi.SPGPU_SUCCESS
        torch.cuda.synchronize()
        assert np.max(np.abs(z.cpu().numpy() - want) / (np.abs(want) + 1.0)) <= 1e-12
        capi.spgpuDhellspmmShardedDestroy(plan)


def test_create_rejects_bad_arguments(gpu):
    from spgpu_amd import capi, synth
    block = synth.hell_uniform_on_device(3200, 8, "banded", "D", 32, seed=1)
    block["slots"] = block["nnz"]
    ob = capi.hell_block(block, 8)
    plan = capi.ShardedPlan()
    first = (C.c_longlong * 3)(0, 3200, 6400)
    # two ranks without a communicator, a rank outside the world, a block whose size contradicts the partition
    assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, None, 0, 2, first, C.byref(ob), None, 4, 0) != capi.SPGPU_SUCCESS
    assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, None, 1, 1, first, C.byref(ob), None, 4, 0) != capi.SPGPU_SUCCESS
    bad = (C.c_longlong * 2)(0, 3000)
    assert capi.spgpuDhellspmmShardedCreate(C.byref(plan), gpu, None, 0, 1, bad, C.byref(ob), None, 4, 0) != capi.SPGPU_SUCCESS
    assert not plan


# ---- world > 1 on one GPU: the ranks as threads over an in-process stand-in for RCCL ------------------------------
@pytest.fixture(scope="module")
def mock_rccl(tmp_path_factory):
    """tests/mock_rccl.c built with gcc: rendezvous + device copies ordered by events, and checks that every receive
    meets a send of its size (what real RCCL would turn into a hang)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    cmd = ["gcc", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(here, "mock_rccl.c"),
           "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib", "-o", lib]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    return lib


@pytest.mark.parametrize("world,pattern,exchange,shape", [
    (2, "banded", "needed", "even"), (2, "banded", "allgather", "even"),
    (3, "window", "needed", "uneven"), (3, "window", "allgather", "uneven"),      # unequal blocks: grouped send/recv all-gather
    (4, "random", "needed", "even"), (4, "banded", "needed", "uneven"),
    (8, "banded", "needed", "even"), (8, "random", "allgather", "even"),          # the size of the driver's scale run
])
def test_many_ranks_as_threads(mock_rccl, world, pattern, exchange, shape):
    """The C driver with world > 1: who needs which rows, the request exchange at Create, the packed sends and receives
    of every step, the stream ordering -- each rank's rows against the oracle's product of the whole matrix.  The
    communication library is the stand-in (no second GPU here); everything else is the code the multi-GPU run executes."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SPGPU_RCCL_LIBRARY=mock_rccl)
    run = subprocess.run([sys.executable, os.path.join(here, "run_sharded_ranks.py"), str(world), pattern, exchange, shape],
                         capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "ALL RANKS OK" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]


def test_a_rank_that_fails_its_set_up_takes_every_rank_out_together(gpu, mock_rccl):
    """The needed-rows set-up is collective.  One of three ranks fails a LOCAL step (injected: SPGPU_TEST_FAIL_SETUP_RANK);
    the ranks agree before every collective, so all three come back from spgpuDhellspmmShardedCreate with an error and
    none is left waiting inside RCCL (the helper reports ranks that did not finish)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SPGPU_RCCL_LIBRARY=mock_rccl, SPGPU_TEST_FAIL_SETUP_RANK="1")
    run = subprocess.run([sys.executable, os.path.join(here, "run_sharded_ranks.py"), "3", "window", "needed", "uneven"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "ALL RANKS OK" in run.stdout and "did not finish" not in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
    assert run.stdout.count("a peer's set-up failed") == 3


def test_a_rank_with_a_wrong_partition_takes_every_rank_out_together(gpu, mock_rccl):
    """A failure BEFORE the set-up proper -- here: one of three ranks is handed a row partition that does not fit its own block, the
    check at the top of spgpuDhellspmmShardedCreate -- is agreed on like the set-up's own failures (sharded.h): every rank
    returns an error, none waits in the set-up's first collective for a peer that has already left."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SPGPU_RCCL_LIBRARY=mock_rccl, SPGPU_TEST_BAD_PARTITION_RANK="2")
    run = subprocess.run([sys.executable, os.path.join(here, "run_sharded_ranks.py"), "3", "window", "needed", "uneven"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "ALL RANKS OK" in run.stdout and "did not finish" not in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
    assert run.stdout.count("a peer's set-up failed") == 3
