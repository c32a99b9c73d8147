"""N > 1 path on CPU: world_size 2, gloo.  Each rank runs its contiguous shard of record sets (the CPU checker stands
in for the GPU engine here -- the thing under test is the sharding + gather plumbing that bench.py uses), rank 0
gathers the 32-byte rows and must reproduce the single-process result in input order."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as graft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, uneven):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    from helpers import oracle as O
    ds = synth.make_dataset(n_sets=301 if uneven else 300, n_hits=8, seg_len=120, branching=2, genome_len=1500, seed=3, variety=True)
    bounds = sharding.shard_bounds(ds.n_sets, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    sub = sharding.slice_batch(ds.batch, lo, hi)
    tax = O.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = O.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = O.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    rc, rows, st = O.predict_batch(tax, q, db, 0, sub["set_offsets"], sub["query_seq"], sub["query_len"],
                                   synth.records_aos(sub, O.RECORD_DTYPE))
    assert rc == 0
    t = torch.from_numpy(rows.view(np.int32).reshape(-1, 8).copy())
    allrows = sharding.gather_rows(t, dst=0)
    dist.barrier()
    # the path bench.py takes: shard sizes known up front (no size exchange), receive buffers allocated once
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    outb = [torch.empty((max(sizes), 8), dtype=torch.int32) for _ in range(world)] if rank == 0 else None
    again = sharding.gather_rows(t, dst=0, sizes=sizes, out=outb)
    if rank == 0:
        again = torch.cat(list(again), dim=0) if isinstance(again, list) else again
        assert again.numpy().tobytes() == allrows.numpy().tobytes()
    dist.barrier()
    if rank == 0:
        rc, full, _ = O.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                      synth.records_aos(ds.batch, O.RECORD_DTYPE))
        ok = allrows.numpy().tobytes() == full.tobytes() and allrows.shape[0] == ds.n_sets
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert allrows is None
    dist.destroy_process_group()


@pytest.mark.parametrize("uneven", [False, True])
def test_shard_and_gather_world2(tmp_path, uneven):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out, uneven), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _ring_worker(rank, world, port, out_path):
    """Eight ranks, shards of different sizes, a stream of steps through the ring of receive buffers exactly as bench.py's
    default N > 1 path drives it: at most IN_FLIGHT steps started and not yet posted, a buffer set reused every N_BUF steps."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load_package()
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    costs = (np.arange(1, 1001, dtype=np.float64) % 37 + 1) ** 2          # one workload, cut where the summed cost is equal
    bounds = sharding.shard_bounds_by_cost(costs, world)
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    assert len(set(sizes)) > 1 and sum(sizes) == 1000
    IN_FLIGHT, STEPS = 3, 23
    n_buf = IN_FLIGHT + 2
    ring = sharding.RowGatherRing(sizes, n_buf, "cpu")

    def rows_of(r, step):  # what rank r computes in step `step`: a function of (global set index, step)
        idx = torch.arange(bounds[r], bounds[r + 1], dtype=torch.int32)
        return torch.stack([idx * 7 + step + c for c in range(8)], dim=1).contiguous()

    import collections
    started = collections.deque()
    ok = True
    last = None
    for step in range(STEPS):
        k = step % n_buf
        ring.reuse(k)                      # the buffer set this step will post into
        started.append((k, step))
        if len(started) == IN_FLIGHT:
            kk, st = started.popleft()
            ring.post(kk, rows_of(rank, st)); last = (kk, st)
    while started:
        kk, st = started.popleft()
        ring.post(kk, rows_of(rank, st)); last = (kk, st)
    ring.drain()
    if rank == 0:
        # the last n_buf steps are still in their buffer sets: every one must hold every rank's rows of its own step
        for step in range(STEPS - n_buf, STEPS):
            got = ring.result(step % n_buf)
            want = torch.cat([rows_of(r, step) for r in range(world)], dim=0)
            ok = ok and got.shape == want.shape and bool(torch.equal(got, want))
        ok = ok and ring.posted == STEPS and last == ((STEPS - 1) % n_buf, STEPS - 1)
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert ring.result(0) is None
    dist.barrier()
    dist.destroy_process_group()


def test_row_gather_ring_world8(tmp_path):
    """The exchange of bench.py's default multi-GPU path (asynchronous gathers through a ring of receive buffers) with the
    eight ranks the driver launches, on gloo: uneven shards (a workload cut by cost), 23 steps, three in flight."""
    out = str(tmp_path / "ring.txt")
    mp.spawn(_ring_worker, args=(8, _free_port(), out), nprocs=8, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_and_slices():
    graft.load_package()
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    assert sharding.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    b = dict(set_offsets=np.array([0, 2, 5, 6], np.uint32), query_seq=np.arange(3, dtype=np.uint32), query_len=np.array([9, 8, 7], np.uint32),
             qstart=np.arange(6, dtype=np.uint32), filtered=None)
    s = sharding.slice_batch(b, 1, 3)
    assert list(s["set_offsets"]) == [0, 3, 4] and list(s["qstart"]) == [2, 3, 4, 5] and list(s["query_len"]) == [8, 7]


def test_shard_bounds_by_cost():
    """batches of mixed lengths are cut by summed alignment cells, not by count."""
    graft.load_package()
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    # equal costs -> the same cut as by count (up to rounding at the boundary set)
    assert sharding.shard_bounds_by_cost(np.ones(8), 4) == [0, 2, 4, 6, 8]
    # one heavy set gets a rank to itself, the light ones share the others
    b = sharding.shard_bounds_by_cost([1, 1, 1, 1, 100, 1, 1, 1, 1], 3)
    assert b[0] == 0 and b[-1] == 9 and b == sorted(b) and b[1] <= 4 < b[2]
    rng = np.random.default_rng(5)
    for world in (2, 3, 8):
        costs = rng.integers(1, 1000, 5000).astype(np.float64) ** 2
        b = sharding.shard_bounds_by_cost(costs, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 5000 and b == sorted(b)
        loads = np.array([costs[b[r]:b[r + 1]].sum() for r in range(world)])
        assert loads.max() <= costs.sum() / world + costs.max()          # never worse than one set over the even share
        by_count = sharding.shard_bounds(5000, world)
        assert loads.max() <= max(costs[by_count[r]:by_count[r + 1]].sum() for r in range(world)) + costs.max()
    assert sharding.shard_bounds_by_cost([], 4) == [0, 0, 0, 0, 0] and sharding.shard_bounds_by_cost([0, 0, 0], 2) == sharding.shard_bounds(3, 2)
    # costs of a batch: query range x reference range, summed per record set
    batch = dict(set_offsets=np.array([0, 2, 3], np.uint32), qstart=np.array([1, 1, 5], np.uint32), qstop=np.array([10, 20, 14], np.uint32),
                 rstart=np.array([100, 50, 9], np.uint32), rstop=np.array([91, 69, 18], np.uint32))
    assert list(sharding.set_costs(batch)) == [10 * 10 + 20 * 20, 10 * 10]


# ---- `python bench.py --gpus N` with no launcher around it (the shape of the driver's 1-GPU command): the reference's parallel
# mode is one command (`taxator -p N`, core/taxator.cpp:181-210), so bench.py starts its N ranks itself -- before it imports
# torch or touches the GPU -- and relays rank 0's line and the launcher's exit code.  Here the ranks run a stub (no GPU).
def _run_launcherless(tmp_path, n, extra=()):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RPA_BENCH_RANK_SCRIPT"] = os.path.join(ROOT, "tests", "helpers", "rank_stub.py")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--mark-dir", str(tmp_path)] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=240, cwd=ROOT)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_without_a_launcher_starts_its_own_ranks(tmp_path):
    n = 4
    p, lines = _run_launcherless(tmp_path, n)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == n and lines[0]["dist"]["world_size"] == n and lines[0]["sum_of_ranks"] == n * (n + 1) // 2
    marks = sorted(os.listdir(tmp_path))
    assert [m.split(".")[0] for m in marks] == ["rank%d" % r for r in range(n)]                 # N children ...
    assert len({m.split(".pid")[1] for m in marks}) == n                                        # ... N different processes
    assert "starting %d ranks" % n in p.stderr


def test_bench_without_a_launcher_reports_a_failing_rank(tmp_path):
    p, lines = _run_launcherless(tmp_path, 2, ["--fail-rank", "1"])
    assert p.returncode != 0 and not lines
