#!/usr/bin/env python3
"""tools/cli_bigstore.py [--gbases G] [--sets N] [--hits H] [--dir D] [--runs a,b,...] -- N1 (the `.fai` region fetch, "the
real end-to-end bottleneck") against a reference collection that is not a toy.

Builds tools/bigstore_gen.cpp, writes a G-Gbase reference FASTA (sequences of 10 kbp ... 10 Mbp) with its .fai, N DISTINCT record
sets of H hits each at random loci of random sequences, query FASTA + .fai, mapping and taxonomy dump, then runs the product's own
driver on them, `taxator -i ref.fna.fai -v query.fna.fai --timing < aln.tsv > out.gff3`, once per run named in --runs:
    default      indexed stores, default tile (65 536 residues), no budget: regions become resident as the stream touches them
    tile2k       --store-tile 2048: what is fetched per hit is a few kbp, not 64 kbp
    budget       --store-tile 2048 --store-gb B (B = a fifth of what `tile2k` made resident): the store is emptied and refilled again and again
    inmemory     no index files: the whole FASTA is read and sent to HBM up front (the reference's in-memory store)
Prints per run: wall time, record sets/s over the whole process and over the pipeline alone, the TIMING / LOAD / STORE / RESIDENT
lines (stage split; residues fetched, windows appended, fetch and append MB/s, times the store was emptied), and the GFF3's SHA-1
(the same for every run).  Measurement tool; test infrastructure only in that it reuses synth's mock taxonomy."""
import argparse
import hashlib
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sha1(path):
    h = hashlib.sha1()
    n = 0
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
            n += chunk.count(b"\n")
    return h.hexdigest(), n


def parse_kv(line):
    out = {}
    for tok in line.split("\t")[1:]:
        k, _, v = tok.partition("=")
        try:
            out[k] = float(v)
        except ValueError:
            out[k] = v if v else True
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbases", type=float, default=32.0)
    ap.add_argument("--sets", type=int, default=1_000_000)
    ap.add_argument("--hits", type=int, default=20)
    ap.add_argument("--dir", default=None, help="where the files go (default: a fresh directory under /dev/shm when it exists, else $TMPDIR)")
    ap.add_argument("--runs", default="default,tile2k,budget,inmemory")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--json", default=None, help="also write the results as JSON here")
    args = ap.parse_args()
    return measure(args)


def measure(args, quiet=False):
    """args: an object with gbases, sets, hits, dir, runs, keep, json (bench.py calls this for secondary.e2e_indexed)"""
    say = (lambda *a, **k: None) if quiet else print
    import __graft_entry__ as graft
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    base = args.dir or tempfile.mkdtemp(prefix="bigstore_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.makedirs(base, exist_ok=True)
    gen = os.path.join(tempfile.mkdtemp(prefix="bigstore_gen_"), "bigstore_gen")  # (not under /dev/shm: mounted noexec on the GPU boxes)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", gen, os.path.join(ROOT, "tools", "bigstore_gen.cpp")])
    tax = synth.MockTaxonomy(4)
    tax.write_ncbi_dump(os.path.join(base, "tax"))
    with open(os.path.join(base, "species.txt"), "w") as f:
        f.write("\n".join(tax.taxid[int(i)] for i in tax.species) + "\n")
    t0 = time.time()
    subprocess.check_call([gen, base, str(args.gbases), str(args.sets), str(args.hits), os.path.join(base, "species.txt")],
                          stderr=subprocess.DEVNULL if quiet else None)
    sizes = {n: os.path.getsize(os.path.join(base, n)) for n in ("ref.fna", "query.fna", "aln.tsv")}
    n_seqs = sum(1 for _ in open(os.path.join(base, "ref.fna.fai")))
    say("files: reference %.1f GB in %d sequences, queries %.2f GB, alignments %.2f GB (%d record sets x %d hits), written to %s in %.0f s"
          % (sizes["ref.fna"] / 1e9, n_seqs, sizes["query.fna"] / 1e9, sizes["aln.tsv"] / 1e9, args.sets, args.hits, base, time.time() - t0), flush=True)
    exe = os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator")
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(base, "tax"))
    common = [exe, "-g", base + "/mapping.tax", "-q", base + "/query.fna", "-f", base + "/ref.fna", "--timing"]
    idx = ["-i", base + "/ref.fna.fai", "-v", base + "/query.fna.fai"]
    results = {"files": dict(sizes, sequences=n_seqs, record_sets=args.sets, hits=args.hits), "runs": {}}
    resident_bytes = None
    first_sha = None
    for run in args.runs.split(","):
        if run == "default":
            extra = idx
        elif run == "tile2k":
            extra = idx + ["--store-tile", "2048"]
        elif run == "budget":
            gb = max(0.05, (resident_bytes or 4e9) / 5 / 1073741824.0)
            extra = idx + ["--store-tile", "2048", "--store-gb", "%.3f" % gb]
        elif run.startswith("turnover"):
            # a budget far below what the stream touches: every block of the stream (16 MB of TSV, ~15 k record sets) brings a few kbp
            # per hit into an HBM budget of turnover:G GB (default 2) and the store is emptied again and again
            gb = float(run.split(":")[1]) if ":" in run else 2.0
            extra = idx + ["--store-tile", "1024", "--store-gb", "%.3f" % gb, "--block-mb", "16"]
        elif run == "inmemory":
            extra = []
        else:
            extra = idx + run.split()  # (anything else: literal extra arguments behind the index options)
        t1 = time.time()
        with open(base + "/aln.tsv") as fin, open(base + "/out.gff3", "wb") as fout:
            r = subprocess.run(common + extra, stdin=fin, stdout=fout, stderr=subprocess.PIPE, text=True, env=env)
        wall = time.time() - t1
        lines = [l for l in r.stderr.splitlines() if l.startswith(("TIMING", "LOAD", "STORE", "RESIDENT", "READERS"))]
        digest, n_lines = sha1(base + "/out.gff3")
        tm = next((parse_kv(l) for l in lines if l.startswith("TIMING")), {})
        res = {"args": " ".join(extra[len(idx):] if extra[:len(idx)] == idx else ["(no index files)"]), "rc": r.returncode, "wall_s": wall,
               "record_sets_per_s": args.sets / wall, "pipeline_s": tm.get("wall_s"),
               "pipeline_record_sets_per_s": (args.sets / tm["wall_s"]) if tm.get("wall_s") else None, "load_s": tm.get("load_s"),
               "stages_s": {k: tm.get(k) for k in ("parse_s", "predict_s", "write_s")}, "gff3_sha1": digest, "gff3_lines": n_lines, "resident": {}, "stores": {}}
        for l in lines:
            if l.startswith("RESIDENT"):
                res["resident"][l.split("\t")[1]] = parse_kv("x\t" + "\t".join(l.split("\t")[2:]))
            if l.startswith("STORE"):
                res["stores"][l.split("\t")[1]] = parse_kv("x\t" + "\t".join(l.split("\t")[3:]))
        if run == "tile2k" and "ref" in res["stores"]:
            resident_bytes = res["stores"]["ref"].get("device_bytes")
        first_sha = first_sha or digest
        res["same_gff3_as_first_run"] = digest == first_sha
        results["runs"][run] = res
        say("run %-9s rc=%d wall %.2f s = %.0f record sets/s (pipeline %.2f s = %.0f/s)  sha1 %s %s" % (
            run, r.returncode, wall, args.sets / wall, tm.get("wall_s") or 0, (args.sets / tm["wall_s"]) if tm.get("wall_s") else 0, digest[:12],
            "" if digest == first_sha else "DIFFERENT OUTPUT"), flush=True)
        for l in lines:
            say("    " + l, flush=True)
        if r.returncode:
            say(r.stderr[-1500:], flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(results, f, indent=1)
    if not args.keep and not args.dir:
        shutil.rmtree(base, ignore_errors=True)
    return results


if __name__ == "__main__":
    main()
