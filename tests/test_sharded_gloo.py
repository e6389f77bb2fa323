"""CPU, world_size 2 over gloo: the row-sharded SpMM driver (spgpu_amd/sharded.py) -- partition,
hackOffsets rebasing, all-gather of the interleaved X blocks, own/rest column split -- with the
oracle standing in for the GPU kernel as the local product (it is the checker here, not a product
path: on GPUs local_product is spgpu?hellspmm through the C ABI, see bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_problem():
    from spgpu_amd import synth
    n = 32 * 23 + 11  # not a multiple of hack_size * world
    lengths = synth.power_law_lengths(n, mean=7.0, max_len=40, seed=77)
    n, m, r, c, v = synth.random_rows_coo(n, n, lengths, seed=78, letter="D", shuffle=True)
    rng = np.random.default_rng(79)
    X = rng.standard_normal((n, 5))
    Y = rng.standard_normal((n, 5))
    return n, r, c, v, X, Y


def _worker(rank, world, port, split, out_dir, exchange="allgather"):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_api as O
    from spgpu_amd import sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, r, c, v, X, Y = _build_problem()
        blocks = sharded.partition_rows(n, world, 32)
        first, count = blocks[rank]
        mine = (r >= first) & (r < first + count)
        rr, cc, vv = r[mine] - first, c[mine], v[mine]

        def to_hell(rows_, cols_, vals_, n_rows):
            ell = O.oracle_converters.coo_to_ell(n_rows, rows_, cols_, vals_)
            return O.oracle_converters.ell_to_hell(ell, 32)

        if split:
            (ro, co, vo), (rx, cx, vx) = sharded.split_by_column_owner(rr, cc, vv, first, count)
            own = to_hell(ro, co - first, vo, count)   # columns relative to the local X block
            rest = to_hell(rx, cx, vx, count)
        else:
            own, rest = to_hell(rr, cc, vv, count), None

        def local_product(part, Z, Yt, alpha, Xt, beta):
            res = O.hell_spmm(part, Xt.numpy(), None if beta == 0 else Yt.numpy(), alpha, beta)
            Z.copy_(torch.from_numpy(res))

        needed = None
        if exchange == "needed":
            needed, renumbered = sharded.needed_rows_of(torch.from_numpy(rest["indices"]), rest["base"])
            rest = dict(rest, indices=renumbered.numpy(), base=0)
        op = sharded.ShardedSpmm(dist, rank, world, blocks, own, rest, local_product,
                                 lambda rows: torch.zeros(rows, X.shape[1], dtype=torch.float64), needed=needed)
        z = torch.zeros(count, X.shape[1], dtype=torch.float64)
        for _ in range(2):   # a second step reuses the exchange plan and its buffers
            op.step(z, torch.from_numpy(Y[first:first + count].copy()), 1.5, torch.from_numpy(X[first:first + count].copy()), -0.5)
        if exchange == "needed":
            assert np.array_equal(op.needed.x_needed.numpy(), X[needed.numpy()]), "exchange did not deliver the rows asked for"
            others = ~((needed.numpy() >= first) & (needed.numpy() < first + count))
            assert op.needed.bytes_received(8 * X.shape[1]) == int(others.sum()) * 8 * X.shape[1]
        else:
            assert np.array_equal(op.x_full.numpy(), X), "all-gather did not reassemble X"
        np.save(os.path.join(out_dir, f"z{rank}.npy"), z.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split", [False, True])
def test_two_rank_sharded_spmm_matches_single_process(tmp_path, split):
    import oracle_api as O
    world, port = 2, 29500 + (os.getpid() % 1000) + (7 if split else 0)
    mp.spawn(_worker, args=(world, port, split, str(tmp_path)), nprocs=world, join=True)
    n, r, c, v, X, Y = _build_problem()
    ell = O.oracle_converters.coo_to_ell(n, r, c, v)
    hell = O.oracle_converters.ell_to_hell(ell, 32)
    want = O.hell_spmm(hell, X, Y, 1.5, -0.5)
    got = np.concatenate([np.load(tmp_path / f"z{k}.npy") for k in range(world)])
    if split:   # own + rest regroup the additions of a row: tolerance, not bits
        scale = np.abs(want) + 1.0
        assert np.max(np.abs(got - want) / scale) <= 1e-12
    else:
        assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("world", [2, 3])
def test_needed_rows_exchange_matches_single_process(tmp_path, world):
    """exchange="needed": only the X rows A_rest names travel (one all_to_all per step); three ranks so that the
    request lists have different lengths per pair."""
    import oracle_api as O
    port = 29500 + (os.getpid() % 1000) + 20 + world
    mp.spawn(_worker, args=(world, port, True, str(tmp_path), "needed"), nprocs=world, join=True)
    n, r, c, v, X, Y = _build_problem()
    hell = O.oracle_converters.ell_to_hell(O.oracle_converters.coo_to_ell(n, r, c, v), 32)
    want = O.hell_spmm(hell, X, Y, 1.5, -0.5)
    got = np.concatenate([np.load(tmp_path / f"z{k}.npy") for k in range(world)])
    assert np.max(np.abs(got - want) / (np.abs(want) + 1.0)) <= 1e-12


def test_partition_and_shard_are_consistent():
    import oracle_api as O
    from spgpu_amd import sharded, synth
    for n, world in ((1000, 3), (64, 4), (33, 2), (32 * 8, 8)):
        blocks = sharded.partition_rows(n, world, 32)
        assert sum(c for _, c in blocks) == n and all(f % 32 == 0 for f, _ in blocks)
        assert [f for f, _ in blocks] == list(np.cumsum([0] + [c for _, c in blocks])[:-1])
    n, m, r, c, v = synth.random_rows_coo(500, 400, synth.power_law_lengths(500, 6.0, 50, seed=1), seed=2, letter="D")
    hell = O.oracle_converters.ell_to_hell(O.oracle_converters.coo_to_ell(n, r, c, v), 32)
    x = np.random.default_rng(0).standard_normal(m)
    whole = O.hell_spmv(hell, x, None, 1.0, 0.0)
    for first, count in sharded.partition_rows(n, 3, 32):
        part = sharded.shard_hell(hell, first, count)
        assert O.hell_spmv(part, x, None, 1.0, 0.0).tobytes() == whole[first:first + count].tobytes()
