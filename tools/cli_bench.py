"""tools/cli_bench.py -- end-to-end throughput of the host `taxator` CLI (TSV in -> GFF3 out) on a synthetic dataset.

    python tools/cli_bench.py [n_sets] [n_hits] [extra taxator args...]

Writes the dataset in the reference's file formats under $TMPDIR, runs taxator-tk_amd/bin/taxator --timing on it and
prints the phase times.  Measurement tool only.
"""
import importlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("taxator-tk_amd.synth")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_hits = int(sys.argv[2]) if len(sys.argv) > 2 else 20
extra = sys.argv[3:]
indexed = "--indexed" in extra  # -i ref.fna.fai -v query.fna.fai (the query index does not exist: rebuilt by a scan)
extra = [x for x in extra if x != "--indexed"]
t0 = time.time()
ds = synth.make_dataset(n_sets=n_sets, n_hits=n_hits, seg_len=200, branching=4, genome_len=4000, seed=2)
folder = tempfile.mkdtemp(prefix="clibench_")
synth.write_files(ds, folder)
print("dataset: %d sets x %d hits written in %.1f s (aln.tsv %.1f MB)" % (n_sets, n_hits, time.time() - t0, os.path.getsize(folder + "/aln.tsv") / 1e6), flush=True)
env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
cmd = [os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator"), "-g", folder + "/mapping.tax", "-q", folder + "/query.fna", "-f", folder + "/ref.fna", "--timing"] + extra
if indexed:
    cmd += ["-i", folder + "/ref.fna.fai", "-v", folder + "/query.fna.fai"]
for rep in range(2):
    t0 = time.time()
    r = subprocess.run(cmd, stdin=open(folder + "/aln.tsv"), stdout=open(folder + "/out.gff3", "w"), stderr=subprocess.PIPE, text=True, env=env)
    dt = time.time() - t0
    lines = r.stderr.strip().splitlines()
    print("run %d: rc=%d wall %.2f s  %.0f record sets/s  %s" % (rep, r.returncode, dt, n_sets / dt, " | ".join(l for l in lines if l.startswith(("TIMING", "STORE")))), flush=True)
import hashlib
print("gff3 sha1", hashlib.sha1(open(folder + "/out.gff3", "rb").read()).hexdigest(), "lines", sum(1 for _ in open(folder + "/out.gff3")))
