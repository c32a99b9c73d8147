"""Feasibility probe: do two engines on two HIP streams overlap (one in its VALU-bound alignment kernel while the other
runs its latency-bound placement kernels)?  Two host threads, 500 k record sets each, vs one engine on 1 M sets."""
import importlib, sys, os, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
pkg = graft.load_package()
synth = importlib.import_module("taxator_tk_amd.synth")

def setup(n_sets, seed):
    ds = synth.make_dataset(n_sets=n_sets, n_hits=20, seg_len=200, branching=4, genome_len=4000, seed=seed, device="cuda")
    m = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    m.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    m.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in ds.batch.items()}
    rows = torch.zeros((n_sets, 8), dtype=torch.int32, device="cuda"); st = torch.zeros((n_sets, 4), dtype=torch.int32, device="cuda")
    return m, dev, rows, st

def run(m, dev, rows, st, n):
    for _ in range(n):
        m.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=st)

big = setup(1_000_000, 2)
a = setup(500_000, 3); b = setup(500_000, 4)
for x in (big, a, b): run(*x, 2)
torch.cuda.synchronize()
t = time.perf_counter(); run(*big, 5); torch.cuda.synchronize(); t_big = (time.perf_counter() - t) / 5
t = time.perf_counter(); run(*a, 5); run(*b, 5); torch.cuda.synchronize(); t_seq = (time.perf_counter() - t) / 5
t = time.perf_counter()
th = [threading.Thread(target=run, args=(*x, 5)) for x in (a, b)]
[x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
t_par = (time.perf_counter() - t) / 5
print("1M sets, one engine: %.2f ms | 2 x 500k sequential: %.2f ms | 2 x 500k concurrent (2 streams, 2 threads): %.2f ms" % (t_big * 1e3, t_seq * 1e3, t_par * 1e3))
