#!/usr/bin/env python3
"""TEST HELPER (run as a child process by tests/test_gpu_sharded_c.py with SPGPU_RCCL_LIBRARY = the in-process stand-in
of tests/mock_rccl.c): the C sharded-SpMM driver with `world` ranks as THREADS on one GPU.  Every rank builds its row
block of one global matrix, joins the communicator, creates its plan (a collective for the needed-rows exchange) and
runs steps; each rank's result is compared with the oracle's product of its rows of the WHOLE matrix and the X of all
ranks.  Prints one line per rank and "ALL RANKS OK" / exits non-zero.

    usage: run_sharded_ranks.py world pattern needed|allgather even|uneven
"""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_api as O  # noqa: E402
from spgpu_amd import capi, synth  # noqa: E402

world, pattern, exchange, shape = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
L, k = 12, 16
sizes = [2048 + (640 * ((r * 7) % 3) if shape == "uneven" else 0) for r in range(world)]      # multiples of 32
first = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
n_total = int(first[-1])
assert capi.spgpuCommAvailable() == 1
ident = (C.c_char * 128)()
assert capi.spgpuCommGetUniqueId(ident) == capi.SPGPU_SUCCESS
assert bytes(ident.raw[8:12]) == b"mock", "this helper must run on the stand-in, not on a real RCCL"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
x_all = np.concatenate([synth.device_vector(sizes[r] * k, "D", 21 + r).view(sizes[r], k).cpu().numpy() for r in range(world)])
results = [None] * world


def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        handle = capi.create_handle(0)
        stream = torch.cuda.Stream()
        capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
        rows = sizes[rank]
        block = synth.hell_uniform_on_device(rows, L, pattern, "D", 32, seed=11 + rank, n_cols=n_total, row_offset=int(first[rank]))
        own, rest = synth.split_uniform_hell_by_columns(block, int(first[rank]), rows)
        x = synth.device_vector(rows * k, "D", 21 + rank).view(rows, k)
        y = synth.device_vector(rows * k, "D", 31 + rank).view(rows, k)
        torch.cuda.synchronize()
        comm = C.c_void_p()
        assert capi.spgpuCommInitRank(C.byref(comm), world, ident, rank) == capi.SPGPU_SUCCESS
        firsts = (C.c_longlong * (world + 1))(*[int(v) for v in first])
        if os.environ.get("SPGPU_TEST_BAD_PARTITION_RANK") == str(rank):
            firsts[rank + 1] += 32      # this rank is told a row partition that does not fit its own block: Create fails BEFORE the set-up
        ob, rb = capi.hell_block(own, L), capi.hell_block(rest, L)
        plan = capi.ShardedPlan()
        kind = capi.EXCHANGE_NEEDED if exchange == "needed" else capi.EXCHANGE_ALLGATHER
        status = capi.spgpuDhellspmmShardedCreate(C.byref(plan), handle, comm, rank, world, firsts, C.byref(ob), C.byref(rb), k, kind)
        if os.environ.get("SPGPU_TEST_FAIL_SETUP_RANK") or os.environ.get("SPGPU_TEST_BAD_PARTITION_RANK"):
            # one rank's local set-up is made to fail: EVERY rank must come back from Create with an error (none may hang)
            results[rank] = (status != capi.SPGPU_SUCCESS, f"rank {rank}: Create returned {status} (a peer's set-up failed)")
            capi.spgpuCommDestroy(comm)
            capi.spgpuDestroy(handle)
            return
        assert status == capi.SPGPU_SUCCESS, f"Create returned {status}"
        hold = synth.hell_rows_to_host(block, 0, rows)
        ys = y.cpu().numpy()
        worst = 0.0
        for alpha, beta, in_place in ((1.0, 0.0, False), (-0.5, 0.75, False), (2.0, 1.0, True)):
            z = y.clone() if in_place else torch.full((rows, k), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            status = capi.spgpuDhellspmmShardedStep(plan, p(z), p(z if in_place else y), alpha, p(x), beta)
            assert status == capi.SPGPU_SUCCESS, f"Step returned {status}"
            torch.cuda.synchronize()
            want = O.hell_spmm(hold, x_all, ys if beta != 0 else None, alpha, beta)
            worst = max(worst, float(np.max(np.abs(z.cpu().numpy() - want) / (np.abs(want) + 1.0))))
        received = int(capi.spgpuDhellspmmShardedRowsReceived(plan))
        capi.spgpuDhellspmmShardedDestroy(plan)
        capi.spgpuCommDestroy(comm)
        capi.spgpuDestroy(handle)
        foreign = rest["rP"][:rest["slots"]].cpu().numpy()
        real = np.zeros(rest["slots"], bool)
        ho, rs = rest["hack_offsets"].cpu().numpy().astype(np.int64), rest["rS"].cpu().numpy()
        ends = np.append(ho[1:], rest["slots"])
        for hck in range(ho.size):
            depth = (ends[hck] - ho[hck]) // 32
            real[ho[hck]:ends[hck]] = (np.arange(depth)[:, None] < rs[hck * 32:(hck + 1) * 32][None, :]).reshape(-1)
        needed = np.unique(foreign[real])
        outside = int(np.count_nonzero((needed < first[rank]) | (needed >= first[rank + 1])))
        expect = outside if exchange == "needed" else n_total - rows
        results[rank] = (worst <= 1e-12 and received == expect, f"rank {rank}: rows {rows}, worst relative deviation {worst:.2e}, "
                         f"rows received from other ranks {received} (expected {expect})")
    except BaseException as error:  # noqa: BLE001 - reported by the parent thread
        results[rank] = (False, f"rank {rank}: {error!r}")


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=240)
stuck = [r for r, t in enumerate(threads) if t.is_alive()]
for r in range(world):
    print(results[r][1] if results[r] else f"rank {r}: no result")
if stuck:
    print(f"ranks {stuck} did not finish", flush=True)
    os._exit(3)
ok = all(res and res[0] for res in results)
print("ALL RANKS OK" if ok else "FAILED", flush=True)
os._exit(0 if ok else 1)
