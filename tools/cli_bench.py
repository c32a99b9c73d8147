"""tools/cli_bench.py -- end-to-end throughput of the host `taxator` CLI (TSV in -> GFF3 out) on a synthetic dataset.

    python tools/cli_bench.py [n_sets] [n_hits] [--repeat-input N] [--indexed] [extra taxator args...]

Writes the dataset in the reference's file formats under $TMPDIR, runs taxator-tk_amd/bin/taxator --timing on it and
prints the phase times.  --repeat-input N: the alignments file N times back to back (the same queries come round again: a
query identifier that re-appears later starts a new group, as in the reference) -- 10 M record sets from a 1 M-set data set
without formatting 200 M lines in Python.  Measurement tool only.
"""
import hashlib, importlib, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("taxator-tk_amd.synth")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

args = sys.argv[1:]
n_sets = int(args.pop(0)) if args and args[0].isdigit() else 100000
n_hits = int(args.pop(0)) if args and args[0].isdigit() else 20
repeat = 1
if "--repeat-input" in args:
    k = args.index("--repeat-input"); repeat = int(args[k + 1]); del args[k:k + 2]
indexed = "--indexed" in args  # -i ref.fna.fai -v query.fna.fai (the query index does not exist: rebuilt by a scan)
extra = [x for x in args if x != "--indexed"]
t0 = time.time()
try:  # (the generator runs on the GPU when there is one: 1 M record sets in seconds instead of minutes)
    import torch
    dev = "cuda" if torch.cuda.is_available() else None
except ImportError:
    dev = None
ds = synth.make_dataset(n_sets=n_sets, n_hits=n_hits, seg_len=200, branching=4, genome_len=4000, seed=2, **({"device": dev} if dev else {}))
folder = tempfile.mkdtemp(prefix="clibench_")
synth.write_files(ds, folder)
aln = folder + "/aln.tsv"
if repeat > 1:
    with open(folder + "/aln_rep.tsv", "wb") as out:
        for _ in range(repeat):
            with open(aln, "rb") as f:
                shutil.copyfileobj(f, out, 1 << 24)
    aln = folder + "/aln_rep.tsv"
print("dataset: %d sets x %d hits written in %.1f s (alignments file %.1f MB, %d record sets)" % (n_sets, n_hits, time.time() - t0, os.path.getsize(aln) / 1e6, n_sets * repeat), flush=True)
env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
cmd = [os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator"), "-g", folder + "/mapping.tax", "-q", folder + "/query.fna", "-f", folder + "/ref.fna", "--timing"] + extra
if indexed:
    cmd += ["-i", folder + "/ref.fna.fai", "-v", folder + "/query.fna.fai"]
for rep in range(2):
    t0 = time.time()
    r = subprocess.run(cmd, stdin=open(aln), stdout=open(folder + "/out.gff3", "w"), stderr=subprocess.PIPE, text=True, env=env)
    dt = time.time() - t0
    lines = r.stderr.strip().splitlines()
    print("run %d: rc=%d wall %.2f s  %.0f record sets/s  %s" % (rep, r.returncode, dt, n_sets * repeat / dt, " | ".join(l for l in lines if l.startswith(("TIMING", "LOAD", "READERS")))), flush=True)
    if r.returncode:
        print(r.stderr[-2000:])
h = hashlib.sha1()
n_lines = 0
with open(folder + "/out.gff3", "rb") as f:
    for chunk in iter(lambda: f.read(1 << 24), b""):
        h.update(chunk); n_lines += chunk.count(b"\n")
print("gff3 sha1", h.hexdigest(), "lines", n_lines)
shutil.rmtree(folder, ignore_errors=True)
