"""Stand-in for bench.py's ranks (RPA_BENCH_RANK_SCRIPT): what `python bench.py --gpus N` starts when no launcher is around
it, without a GPU.  Every rank checks the environment torch.distributed.run gave it, joins a gloo group, leaves a file; rank 0
prints the one JSON line the parent has to relay.  `--fail-rank R` makes rank R exit with status 7 (the parent must fail too)."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, required=True)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--mark-dir", required=True)
ap.add_argument("--fail-rank", type=int, default=-1)
args = ap.parse_args()
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
assert world == args.gpus, (world, args.gpus)
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and "LOCAL_RANK" in os.environ
assert not torch.cuda.is_initialized()
if rank == args.fail_rank:
    sys.exit(7)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t)
open(os.path.join(args.mark_dir, "rank%d.pid%d" % (rank, os.getpid())), "w").write(str(os.getppid()))
dist.barrier()
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": world, "sum_of_ranks": int(t.item()), "steps": args.steps, "dist": {"world_size": dist.get_world_size()}}))
dist.destroy_process_group()
