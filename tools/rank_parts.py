#!/usr/bin/env python3
"""tools/rank_parts.py -- config 2's batch through k_rank_records ONLY (RPA_DEBUG_RANK_ONLY=1), five times; run it under
`rocprofv3 --kernel-trace --stats` with RPA_LIB_PATH pointing at a timing build (tools/rank_parts.sh) to see what a part of the kernel costs."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RPA_DEBUG_RANK_ONLY"] = "1"
import torch
import __graft_entry__ as graft
pkg = graft.load_package()
synth = importlib.import_module("taxator_tk_amd.synth")
ds = synth.make_dataset(n_sets=1_000_000, n_hits=20, seg_len=200, branching=4, genome_len=4000, seed=2, device="cuda", level_rate=0.03, query_rate=0.05)
model = pkg.RPAPredictionModel(ds.tax, alphabet=0)
model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in ds.batch.items()}
rows = torch.zeros((ds.n_sets, 8), dtype=torch.int32, device="cuda"); stats = torch.zeros((ds.n_sets, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(5):
    try:
        model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
    except pkg.RpaError as e:
        assert "RPA_DEBUG_RANK_ONLY" in str(e), e
print("done")
