"""The N > 1 code path of bench.py on the one GPU of the test box: torch.distributed.run with ONE rank and the RCCL
(`nccl`) backend forced on -- init_process_group on a device, the engine on torch's current stream, the gather of the
32-byte rows through RCCL, the barrier / all_reduce bracketing of the timed region.  The child is a fresh process (never a
re-exec of the test runner).  Multi-rank plumbing itself (shard bounds, uneven gathers) is covered on CPU with gloo in
tests/test_distributed_cpu.py; the 8-GPU curve is the driver's to measure."""
import json
import os
import socket
import sys

import pytest

from helpers.proc import run_child

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("config,extra", [("c3", ["--sets", "30000"]), ("c5", ["--sets", "40", "--seg-len", "3000"])])
def test_bench_one_rank_over_rccl(config, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", config, "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--force-dist"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = run_child(cmd, timeout=240, env=env, cwd=ROOT, what="bench.py, one rank over RCCL")
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert "RCCL gather" in out["config"]["parallelism"]
    assert out["dist"]["backend"] == "nccl" and out["dist"]["world_size"] == 1 and out["dist"]["rccl_version"]
    d = out["dist"]  # per-rank timing and the cost of one gather: what makes the first real multi-GPU line diagnosable
    assert len(d["per_rank_ms"]) == 1 and 0 < d["per_rank_ms"][0] <= d["per_rank_exchange_done_ms"][0] <= d["per_rank_barrier_ms"][0] * 1.001
    assert d["gather_ms"] > 0 and d["gather_bytes"] == 32 * sum(out["config"]["shard_sets"])
    assert out["roofline"]["achieved"] > 0 and out["roofline"]["launches"] > 0
    # the three synchronous legs behind the timed region: plain <= launch timing only <= timing + the kernels' counters (within noise),
    # and the roofline quoted on the launch durations of the timing-only leg
    assert 0 < out["sync_unprofiled_ms_per_step"] and 0 < out["sync_timing_only_ms_per_step"] and 0 < out["sync_ms_per_step"]
    assert "counters OFF" in out["roofline"]["timing"]
    assert 0.5 < out["align_ms_per_step"] / out["align_ms_with_counters_per_step"] < 1.5   # (tiny workloads here: the two passes differ by noise)
    assert abs(out["roofline"]["avg_launch_ms"] * out["roofline"]["launches"] / 2 - out["align_ms_per_step"]) < 1e-6 * out["align_ms_per_step"] + 1e-9


RANKS = 4  # the GPU box allows six processes on its card: the test runner + four ranks (eight ranks: tests/test_distributed_cpu.py)


@pytest.mark.parametrize("config,extra", [("c3", ["--sets", "8000"]), ("c5", ["--sets", "16", "--seg-len", "2000"])])
def test_bench_four_ranks_share_the_gpu(config, extra):
    """Four ranks, four processes, four engines on the one GPU of the test box (gloo gathers the rows through host memory:
    RCCL refuses several ranks on one device), through bench.py's DEFAULT multi-GPU step function: steps submitted to the
    engines' job slots, three in flight, every step's rows sent to rank 0 by an asynchronous gather through the ring of
    receive buffers.  What this executes that nothing else does before the driver's 8-GPU run: per-rank seeds (c3), ONE
    workload cut by summed alignment cells into shards of different sizes and the uneven gather (c5), barrier / all_reduce
    bracketing with several ranks, several engines sharing a device."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(RANKS), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(RANKS), "--config", config, "--steps", "4",
           "--warmup", "1", "--no-cpu-baseline", "--dist-backend", "gloo", "--one-device"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = run_child(cmd, timeout=240, env=env, cwd=ROOT, what="bench.py, four ranks on one GPU")
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == RANKS and out["value"] > 0
    assert out["dist"]["world_size"] == RANKS and out["dist"]["backend"] == "gloo" and out["dist"]["gather"].startswith("asynchronous")
    assert "steps in flight" in out["submission"]
    d = out["dist"]
    assert len(d["per_rank_ms"]) == RANKS and all(x > 0 for x in d["per_rank_ms"]) and len(d["per_rank_exchange_done_ms"]) == RANKS
    assert max(d["per_rank_barrier_ms"]) <= out["ms_per_step"] * 1.001 and max(d["per_rank_ms"]) <= out["ms_per_step"] * 1.001  # (the line's time is the slowest rank's)
    assert d["gather_ms"] > 0 and d["gather_bytes"] == 32 * sum(out["config"]["shard_sets"])
    shards = out["config"]["shard_sets"]
    assert len(shards) == RANKS and sum(shards) == RANKS * int(extra[1])
    if config == "c5":
        assert "cut by summed alignment cells" in out["config"]["parallelism"]

