#!/usr/bin/env python3
"""Hunt for the `taxator` child that never exits (round 2's driver run of `pytest -m gpu` sat in
tests/test_gpu_cli.py::test_cli_errors_like_the_reference[--record-objects] until the box's limit).

Runs the three error-path commands of that test through both input paths (columnar pipeline / --record-objects), plus one
successful run of each path, ITERS times each, PARALLEL children at a time (<= 4: a GPU box allows few processes on the
card), every child under a watchdog.  A child that overruns is examined the way tests/helpers/proc.py does it -- per-thread
/proc state, then SIGUSR1 for the driver's own "stage / engine" report -- and killed; the hunt goes on.  Every so often a
large batch goes through a fresh engine in this process (--dirty-every) so that device allocations the children get are
not factory-fresh.  RPA_POISON=1 in the environment is inherited by the children (every work buffer 0xA5 per batch).
Arms (environment): TAXATOR_RUN_EXIT_HANDLERS=1 = plain return from main (the runtime's exit handlers run);
RPA_CU_MASK_QUEUES=1 = round 2's CU-masked engine streams (the combination of the two is what hung).

Writes a progress line per 50 children and a JSON summary at the end to --out (default gpurun_out/hang_hunt.json).
Exit code 0 whatever was found: the summary says it.
"""
import argparse
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers.proc import thread_report  # noqa: E402

BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def variants(work):
    """(name, argv, stdin file, expected exit code, text expected on stderr)"""
    edge = os.path.join(work, "edge")
    shutil.copytree(os.path.join(GOLDEN, "survey_edge"), edge)
    bad = {
        "mapping": ("e0\t1\t200\t200\tnope\t101\t300\t400\t0\t200\t200\n", "bad taxon mapping for alignment reference sequence"),
        "reverse": ("e0\t200\t1\t200\tref3\t101\t300\t400\t0\t200\t200\n", "reverse query positions not allowed"),
        "range": ("e0\t1\t200\t200\tref3\t99999101\t99999300\t400\t0\t150\t200\ne0\t1\t200\t200\tref2\t101\t300\t300\t0\t180\t200\n", "SequenceRangeError"),
    }
    base = [BIN, "-g", os.path.join(edge, "mapping.tax"), "-q", os.path.join(edge, "query_edge.fna"), "-f", os.path.join(edge, "ref.fna"),
            "-i", os.path.join(edge, "ref.fna.fai"), "-p", "0"]
    out = []
    for key, (text, want) in bad.items():
        f = os.path.join(edge, "aln_%s.tsv" % key)
        open(f, "w").write(text)
        for pname, extra in (("columnar", []), ("objects", ["--record-objects"])):
            out.append(("%s/%s" % (key, pname), base + extra, f, 1, want))
    ok = [BIN, "-g", os.path.join(edge, "mapping.tax"), "-q", os.path.join(edge, "query.fna"), "-f", os.path.join(edge, "ref.fna"),
          "-i", os.path.join(edge, "ref.fna.fai"), "-p", "0"]
    for pname, extra in (("columnar", []), ("objects", ["--record-objects"])):
        out.append(("ok/%s" % pname, ok + extra, os.path.join(edge, "aln.tsv"), 0, ""))
    return edge, out


def one(v, env, limit):
    name, cmd, stdin_file, want_rc, want_text = v
    t0 = time.time()
    with open(stdin_file, "rb") as fin:
        p = subprocess.Popen(cmd, stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        out, err = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        rep = thread_report(p.pid)
        try:
            p.send_signal(signal.SIGUSR1)
        except OSError:
            pass
        time.sleep(0.5)
        p.terminate()
        try:
            out, err = p.communicate(timeout=5)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        return dict(variant=name, hung=True, seconds=time.time() - t0, threads=rep, stderr=(err or "")[-2000:])
    ok = p.returncode == want_rc and want_text in err
    r = dict(variant=name, hung=False, ok=ok, seconds=time.time() - t0)
    if not ok:
        r.update(rc=p.returncode, stderr=(err or "")[-1500:])
    return r


def dirty_device(n_sets):
    """One large batch through a fresh engine in this process: leaves the allocator's pools and the GPU's memory used."""
    import importlib
    import __graft_entry__ as graft
    pkg = graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=n_sets, n_hits=20, seg_len=200, branching=4, genome_len=4000, seed=int(time.time()) & 1023)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA, device=0)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    model.predict(ds.batch)
    del model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500, help="children per variant")
    ap.add_argument("--parallel", type=int, default=3)
    ap.add_argument("--limit", type=float, default=30.0, help="watchdog per child (seconds)")
    ap.add_argument("--dirty-every", type=int, default=400, help="children between two large in-process batches (0: never)")
    ap.add_argument("--dirty-sets", type=int, default=200000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hang_hunt.json"))
    ap.add_argument("--budget-s", type=float, default=900.0, help="stop starting children after this many seconds")
    ap.add_argument("--only", default="", help="comma-separated variant names (e.g. mapping/objects,reverse/objects); default: all")
    a = ap.parse_args()
    a.parallel = max(1, min(4, a.parallel))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    sys.path.insert(0, ROOT)
    work = tempfile.mkdtemp(prefix="hang_hunt_")
    edge, vs = variants(work)
    if a.only:
        vs = [v for v in vs if v[0] in a.only.split(",")]
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(edge, "tax"))
    jobs = [v for _ in range(a.iters) for v in vs]
    t_start = time.time()
    hung, wrong, slow = [], [], []
    per_variant = {v[0]: [0, 0.0, 0.0] for v in vs}  # runs, total seconds, max seconds
    done = 0
    progress = a.out.replace(".json", "_progress.txt")
    with ThreadPoolExecutor(a.parallel) as ex:
        pos = 0
        while pos < len(jobs) and time.time() - t_start < a.budget_s:
            chunk = jobs[pos:pos + 50]
            pos += len(chunk)
            for r in ex.map(lambda v: one(v, env, a.limit), chunk):
                done += 1
                pv = per_variant[r["variant"]]
                pv[0] += 1; pv[1] += r["seconds"]; pv[2] = max(pv[2], r["seconds"])
                if r["hung"]:
                    hung.append(r)
                elif not r["ok"]:
                    wrong.append(r)
                elif r["seconds"] > 10:
                    slow.append(r)
            with open(progress, "a") as f:
                f.write("%d children in %.0f s: %d hung, %d wrong, %d slow (>10 s)\n" % (done, time.time() - t_start, len(hung), len(wrong), len(slow)))
            if a.dirty_every and done % a.dirty_every < 50 and done >= a.dirty_every:
                try:
                    dirty_device(a.dirty_sets)
                except Exception as e:  # the hunt is about the children
                    with open(progress, "a") as f:
                        f.write("dirty_device failed: %r\n" % (e,))
    summary = dict(children=done, seconds=time.time() - t_start, parallel=a.parallel, limit_s=a.limit, poison=os.environ.get("RPA_POISON", ""),
                   env={k: v for k, v in os.environ.items() if k.startswith(("RPA_", "TAXATOR_"))},
                   hung=hung, wrong=wrong[:20], n_wrong=len(wrong), slow=slow[:20],
                   per_variant={k: dict(runs=v[0], mean_s=v[1] / max(1, v[0]), max_s=v[2]) for k, v in per_variant.items()})
    json.dump(summary, open(a.out, "w"), indent=1)
    print(json.dumps({k: summary[k] for k in ("children", "seconds", "n_wrong")} | {"n_hung": len(hung)}))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
