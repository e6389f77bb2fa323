"""Row-sharded HELL SpMM across the GPUs of one node (BASELINE configs[4]).

Rows of Z = alpha*A*X + beta*Y are independent: A, Y and Z are partitioned by contiguous,
hack-aligned row blocks, one block per rank (one process per GPU); only the dense X is needed
everywhere, so a step is ONE exchange -- an all-gather of the ranks' X row blocks over RCCL
(`torch.distributed`, backend "nccl" on GPUs, "gloo" in the CPU tests) -- followed by the local
`spgpu?hellspmm`.  Interleaved multivectors make each rank's X block one contiguous buffer.

Overlap: with `split=True` a rank's block of A is cut by column ownership into A_own (columns it
already holds) and A_rest; A_own * X_own runs on the compute stream while the all-gather is in
flight on the communication stream, then A_rest * X_full is accumulated (beta = 1, Z aliasing Y).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-gather in which every rank sends its
block to its 7 peers uses all links at once; per link it moves one block.

Needed-rows exchange (`exchange="needed"`): A_rest only reads the X rows its column indices name.  At set-up
every rank lists those rows (sorted, unique), tells each owner which of its rows it wants (one all_to_all of
counts, one of indices) and renumbers A_rest's columns into that compact list; a step then moves exactly the
needed rows with ONE all_to_all (RCCL send/recv pairs, so only links between ranks that share columns carry
data) instead of the whole of X.  For a banded matrix that is a halo of a few rows per neighbour; for scattered
columns it degenerates to the all-gather's volume.  Same arithmetic as the all-gather path: A_own*X_own first,
then Z += A_rest*X_needed.

Nothing here computes on the CPU: `local_product` is injected (the C ABI on GPUs, the oracle in
the gloo test).
"""
import numpy as np


def partition_rows(n_rows, world, hack_size=32):
    """Contiguous row blocks, boundaries multiples of hack_size, sizes as equal as hacks allow.
    Returns [(first_row, n_rows_in_block)] * world."""
    hacks = (n_rows + hack_size - 1) // hack_size
    base, extra = divmod(hacks, world)
    out, first = [], 0
    for r in range(world):
        h = base + (1 if r < extra else 0)
        rows = min(h * hack_size, n_rows - first)
        out.append((first, max(rows, 0)))
        first += max(rows, 0)
    return out


def shard_hell(hell, first_row, n_rows):
    """Rows [first_row, first_row + n_rows) of a host HELL dict as a HELL dict of their own.
    A hack-aligned block is a contiguous piece of cM / rP; its hackOffsets are the global ones minus
    the first (the reference's own chunk loop advances hackOffsets the same way,
    hell_spmv_base.cuh:139-144).  hackOffsets has no trailing total (hell.c:64,75): the end of the
    last hack is the next block's first offset or the end of the arrays."""
    hs = hell["hack_size"]
    assert first_row % hs == 0
    h0 = first_row // hs
    h1 = (first_row + n_rows + hs - 1) // hs
    ho = np.asarray(hell["hack_offsets"], dtype=np.int64)
    s0 = int(ho[h0]) if h0 < ho.size else int(hell["values"].size)
    s1 = int(ho[h1]) if h1 < ho.size else int(hell["values"].size)
    return dict(letter=hell["letter"], rows=n_rows, values=hell["values"][s0:s1], indices=hell["indices"][s0:s1],
                hack_offsets=(ho[h0:h1] - s0).astype(np.int32), hack_size=hs,
                row_lengths=np.asarray(hell["row_lengths"][first_row:first_row + n_rows], dtype=np.int32),
                base=hell["base"])


def split_by_column_owner(coo_rows, coo_cols, coo_vals, col_first, col_count, base=0):
    """COO entries of a row block -> (own, rest): entries whose column lies in
    [col_first, col_first + col_count) and the others, COO order kept in both."""
    c0 = np.asarray(coo_cols, dtype=np.int64) - base
    own = (c0 >= col_first) & (c0 < col_first + col_count)
    pick = lambda m: (np.asarray(coo_rows)[m], np.asarray(coo_cols)[m], np.asarray(coo_vals)[m])
    return pick(own), pick(~own)


def needed_rows_of(rest_indices, base=0):
    """Sorted unique X rows (global, zero-based) the column indices of A_rest name -- a torch tensor on the device
    of `rest_indices` -- and the indices renumbered into that list (padding slots of the HELL slab included: they
    hold a valid index with a zero coefficient)."""
    import torch
    cols = rest_indices.to(torch.int64) - base
    needed = torch.unique(cols)                      # sorted
    return needed, torch.searchsorted(needed, cols).to(rest_indices.dtype)


class NeededRows:
    """The X rows of other ranks one rank needs, and the per-step all_to_all that fetches them."""

    def __init__(self, dist, rank, world, col_blocks, needed, new_rows):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        dev = needed.device
        starts = torch.tensor([f for f, _ in col_blocks], dtype=torch.int64, device=dev)
        owner = torch.bucketize(needed, starts[1:], right=True)            # block r holds starts[r] .. starts[r+1]-1
        want = torch.bincount(owner, minlength=world).to(torch.int64)       # rows I want from each rank
        give = torch.empty_like(want)
        dist.all_to_all_single(give, want)                                  # rows each rank wants from me
        self.recv_splits = [int(v) for v in want.tolist()]
        self.send_splits = [int(v) for v in give.tolist()]
        ask = needed - starts[owner]                                        # row numbers inside the owner's block
        self.send_index = torch.empty(sum(self.send_splits), dtype=torch.int64, device=dev)
        dist.all_to_all_single(self.send_index, ask, output_split_sizes=self.send_splits,
                               input_split_sizes=self.recv_splits)
        self.x_needed = new_rows(int(needed.numel()))
        self.send_buffer = None

    def start(self, x_local, async_op=True):
        """Gathers the rows the other ranks asked for and starts the exchange into x_needed."""
        import torch
        if self.send_buffer is None:
            self.send_buffer = torch.empty((self.send_index.numel(),) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                                           device=x_local.device)
        torch.index_select(x_local, 0, self.send_index, out=self.send_buffer)
        return self.dist.all_to_all_single(self.x_needed, self.send_buffer, output_split_sizes=self.recv_splits,
                                           input_split_sizes=self.send_splits, async_op=async_op)

    def bytes_received(self, row_bytes):
        return (sum(self.recv_splits) - self.recv_splits[self.rank]) * row_bytes


class ShardedSpmm:
    """One rank's state for the sharded product.

    local_product(part, Z, Y, alpha, X, beta) must compute Z = alpha*part*X + beta*Y for the HELL
    block `part` (whatever object the caller built: a device-resident matrix for the C ABI, a host
    dict for the oracle) with interleaved X / Y / Z.
    """

    def __init__(self, dist, rank, world, col_blocks, own, rest, local_product, new_full_x, comm_stream=None,
                 needed=None, products_stream=None):
        """needed: sorted unique X rows `rest` reads (needed_rows_of), with rest's columns already renumbered into
        that list -> a step exchanges only those rows.  None: a step all-gathers X (rest indexes the whole of X).

        Stream contract: the torch ops of a step (row packing, the collective, work.wait()) go to torch's CURRENT stream;
        `local_product` launches wherever the caller's library launches (the C ABI: the handle's stream).  A step is only
        correct when the two are the same stream, i.e. when the caller runs step() under `with torch.cuda.stream(s)` with s
        the handle's stream.  products_stream: that stream's raw address (stream.cuda_stream); when given, every step
        checks it and raises instead of computing on data that may not have landed.  (The C driver,
        include/spgpu/sharded.h, orders its own streams with events and has no such requirement.)"""
        self.products_stream = products_stream
        self.dist, self.rank, self.world = dist, rank, world
        self.needed = NeededRows(dist, rank, world, col_blocks, needed, new_full_x) if needed is not None else None
        self.col_blocks = col_blocks                  # [(first, count)] ownership of X rows per rank
        self.own, self.rest = own, rest               # rest is None when the block is not split
        self.local_product = local_product
        self.x_full = new_full_x(sum(c for _, c in col_blocks)) if needed is None else None
        self.comm_stream = comm_stream
        counts = {c for _, c in col_blocks}
        self.equal_blocks = len(counts) == 1

    def gather_x(self, x_local, async_op=False):
        """All-gather of the X row blocks into x_full (rows in rank order)."""
        if self.world == 1 and not (self.dist.is_available() and self.dist.is_initialized()):
            self.x_full[: x_local.shape[0]].copy_(x_local)
            return None
        if self.equal_blocks:
            return self.dist.all_gather_into_tensor(self.x_full, x_local, async_op=async_op)
        # unequal blocks: pad every contribution to the largest block
        big = max(c for _, c in self.col_blocks)
        import torch
        pad = torch.zeros((big,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        pad[: x_local.shape[0]].copy_(x_local)
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        work = self.dist.all_gather(parts, pad, async_op=async_op)
        if work is not None:
            work.wait()
        for (first, count), p in zip(self.col_blocks, parts):
            self.x_full[first:first + count].copy_(p[:count])
        return None

    def step(self, z_local, y_local, alpha, x_local, beta):
        """One sharded product.  With a split block the own-column part overlaps the all-gather."""
        if self.products_stream is not None:
            import torch
            if torch.cuda.current_stream().cuda_stream != self.products_stream:
                raise RuntimeError("ShardedSpmm.step: torch's current stream is not the stream the products run on; "
                                   "wrap the step in `with torch.cuda.stream(handle_stream)` (see the class docstring)")
        if self.rest is None:
            work = self.gather_x(x_local)
            if work is not None:
                work.wait()
            self.local_product(self.own, z_local, y_local, alpha, self.x_full, beta)
            return
        if self.needed is not None:
            work, x_rest = self.needed.start(x_local), self.needed.x_needed
        else:
            work, x_rest = self.gather_x(x_local, async_op=True), self.x_full
        # columns of `own` are indexed relative to this rank's X block
        self.local_product(self.own, z_local, y_local, alpha, x_local, beta)
        if work is not None:
            work.wait()
        self.local_product(self.rest, z_local, z_local, alpha, x_rest, 1.0)
