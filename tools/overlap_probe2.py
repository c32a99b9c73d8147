"""Feasibility probe 2: two engines (500 k record sets each) on two streams, the second one started with a phase offset so
that its latency-bound placement kernels fall under the first one's long alignment kernel and vice versa."""
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

def run(m, dev, rows, st, n, delay=0.0):
    if delay: time.sleep(delay)
    for _ in range(n):
        m.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=st)

a = setup(500_000, 3); b = setup(500_000, 4)
for x in (a, b): run(*x, 2)
torch.cuda.synchronize()
N = 8
t = time.perf_counter(); run(*a, N); run(*b, N); torch.cuda.synchronize(); t_seq = (time.perf_counter() - t) / N
print("2 x 500k sequential: %.2f ms per pair of calls" % (t_seq * 1e3))
for off_ms in (0.0, 1.5, 3.0, 4.5, 6.0):
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(*a, N, 0.0)), threading.Thread(target=run, args=(*b, N, off_ms / 1e3))]
    [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("offset %.1f ms: %.2f ms per pair of calls (wall %.1f ms for %d pairs)" % (off_ms, (dt - off_ms / 1e3) / N * 1e3, dt * 1e3, N))
