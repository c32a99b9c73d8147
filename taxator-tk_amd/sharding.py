"""Multi-GPU plumbing of the RPA path: shard record sets over ranks, gather the 32-byte prediction rows to rank 0.

Record sets are independent (core/taxator.cpp:69-74; README FAQ 10), so there is no data-path collective: each rank
(one process per GPU) runs whole record sets of a contiguous shard and the only exchange is the gather of
rpa_prediction rows (RCCL over xGMI on the GPU box, gloo in the CPU tests) -- the replacement of the reference's
mutex-guarded stdout (core/src/concurrentoutstream.hh:56-75).  Backend-agnostic: uses whatever torch.distributed
group is initialised.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_sets, world_size):
    """Contiguous blocks of record sets per rank (keeps input order for the merge); returns world_size+1 bounds."""
    return [(n_sets * r) // world_size for r in range(world_size + 1)]


def set_costs(batch):
    """Estimated alignment cells per record set: the cost of the path is quadratic in segment length (every demanded
    realignment is ~ query range x reference range cells), so a set costs ~ sum over its hits of
    (query range length) x (reference range length)."""
    so = np.asarray(batch["set_offsets"]).astype(np.int64)
    qspan = np.abs(np.asarray(batch["qstop"]).astype(np.int64) - np.asarray(batch["qstart"]).astype(np.int64)) + 1
    rspan = np.abs(np.asarray(batch["rstop"]).astype(np.int64) - np.asarray(batch["rstart"]).astype(np.int64)) + 1
    cells = np.concatenate([[0], np.cumsum(qspan * rspan)])
    return cells[so[1:]] - cells[so[:-1]]


def shard_bounds_by_cost(costs, world_size):
    """Contiguous blocks of record sets with (nearly) equal summed cost per rank -- for batches of mixed lengths such as
    BASELINE config 5, where equal counts leave ranks unevenly loaded.  Returns world_size+1 non-decreasing bounds; a set
    goes to the rank whose cost interval contains the midpoint of the set's own cost span."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    total = float(costs.sum())
    if n == 0 or total <= 0.0:
        return shard_bounds(n, world_size)
    mid = np.cumsum(costs) - 0.5 * costs
    cuts = total * np.arange(1, world_size) / world_size
    return [0] + [int(x) for x in np.searchsorted(mid, cuts, side="left")] + [n]


def slice_batch(batch, lo, hi):
    """Record sets [lo, hi) of a host batch (dict of numpy arrays in rpa_batch layout), offsets rebased to 0."""
    so = np.asarray(batch["set_offsets"])
    r0, r1 = int(so[lo]), int(so[hi])
    out = {}
    for k, v in batch.items():
        if v is None:
            out[k] = None
        elif k == "set_offsets":
            out[k] = (so[lo:hi + 1] - so[lo]).astype(np.uint32)
        elif k in ("query_seq", "query_len"):
            out[k] = v[lo:hi]
        else:
            out[k] = v[r0:r1]
    return out


def gather_rows(rows, dst=0, group=None, sizes=None, out=None, force=False, async_op=False):
    """rows: (n_local, 8) int32 tensor (any device).  Returns the rows of all ranks concatenated in rank order on `dst`
    (None elsewhere).  Shards may differ in size: sizes are exchanged first, rows are padded to the largest shard.
    Shards need not be equal (config 5 is cut by cost).  A caller that gathers the same shapes every step passes `sizes` (rows per rank, known up front) to skip the size
    exchange, and on `dst` `out` = a list of world_size preallocated (max(sizes), 8) tensors to skip the allocation."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force:  # force: run the collective anyway (a one-rank rehearsal of the N > 1 path on hardware)
        return rows
    if sizes is None:
        n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
        gathered = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(gathered, n_local, group=group)
        sizes = [int(s.item()) for s in gathered]
    n_max = max(sizes)
    padded = rows
    if rows.shape[0] != n_max:
        padded = torch.zeros((n_max,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        padded[: rows.shape[0]] = rows
    bufs = None
    if rank == dst:
        bufs = out if out is not None else [torch.empty_like(padded) for _ in range(world)]
    if async_op:
        # the caller overlaps the exchange with its next step: it gets (work, what to keep alive until work.wait()) and reads
        # `out` (its own preallocated receive buffers, required here) afterwards
        assert rank != dst or out is not None
        send = padded.contiguous()
        return dist.gather(send, bufs, dst=dst, group=group, async_op=True), send
    dist.gather(padded.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if all(n == n_max for n in sizes) and out is not None:
        return bufs  # equal shards, caller-owned buffers: no concatenation copy
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


class RowGatherRing:
    """The exchange of a stream of steps: every step's rows go to `dst` with an asynchronous gather that runs beside the
    steps that follow, through a ring of `n_buf` receive-buffer sets so that a set is not reused before the gather that
    reads or fills it is done.  bench.py posts the rows of step k into buffer set k % n_buf; on RCCL the tensors are device
    tensors and the gather runs on RCCL's stream over xGMI, in the gloo rehearsals they are host tensors -- same calls.

        ring = RowGatherRing(sizes, n_buf, device)      # sizes: rows per rank (may differ: config 5 is cut by cost)
        ring.reuse(k)                                   # before buffer set k is written again: its gather has finished
        ring.post(k, rows)                              # start the gather of this step's rows
        ring.drain()                                    # every gather posted so far has finished
        ring.result(k)                                  # on dst: the rows of all ranks in rank order (a tensor)
    """

    def __init__(self, sizes, n_buf, device, dst=0, group=None, width=8, dtype=torch.int32):
        self.sizes, self.n_buf, self.dst, self.group = [int(s) for s in sizes], int(n_buf), dst, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert len(self.sizes) == self.world
        n_max = max(self.sizes) if self.sizes else 0
        self.recv = [([torch.empty((n_max, width), dtype=dtype, device=device) for _ in range(self.world)] if self.rank == dst else None)
                     for _ in range(self.n_buf)]
        self.pending = [None] * self.n_buf
        self.posted = 0

    def reuse(self, k):
        if self.pending[k] is not None:
            self.pending[k][0].wait()
            self.pending[k] = None

    def post(self, k, rows):
        self.reuse(k)
        assert rows.shape[0] == self.sizes[self.rank], (rows.shape, self.sizes, self.rank)
        self.pending[k] = gather_rows(rows, dst=self.dst, group=self.group, sizes=self.sizes, out=self.recv[k], force=True, async_op=True)
        self.posted += 1

    def drain(self):
        for k in range(self.n_buf):
            self.reuse(k)

    def result(self, k):
        self.reuse(k)
        if self.rank != self.dst:
            return None
        return torch.cat([b[:n] for b, n in zip(self.recv[k], self.sizes)], dim=0)
