"""End-to-end through the host C++ mirror: alignments TSV on stdin -> `taxator` -> C ABI -> HIP -> GFF3 on stdout."""
import gzip
import importlib
import os
import shutil

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import refio as R
from helpers.proc import run_child

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator")
CHILD_LIMIT_S = 120  # a taxator run of these tests takes about a second (a cold GPU box: up to a minute for the first one)


def _run(folder, ext, extra=(), aln="aln.tsv", query=None, log=None):
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
    cmd = [BIN, "-g", os.path.join(folder, "mapping.tax"), "-q", os.path.join(folder, query or ("query" + ext)), "-f",
           os.path.join(folder, "ref" + ext), "-i", os.path.join(folder, "ref" + ext + ".fai"), "-p", "0"] + list(extra)
    if log:
        cmd += ["-l", log]
    return run_child(cmd, timeout=CHILD_LIMIT_S, stdin_path=os.path.join(folder, aln), env=env)


def _unpack(golden_dir, name, ext, tmp_path):
    work = tmp_path / name
    shutil.copytree(os.path.join(golden_dir, name), work)
    with gzip.open(work / ("ref" + ext + ".gz"), "rb") as f, open(work / ("ref" + ext), "wb") as g:
        g.write(f.read())
    return str(work)


@pytest.mark.parametrize("name,ext,extra", [("survey_dna", ".fna", []), ("survey_protein", ".faa", ["-b", "protein"])])
@pytest.mark.parametrize("path", [[], ["--record-objects"], ["--block-mb", "0.004", "-p", "3"], ["--block-mb", "0.004", "--readers", "1"],
                                  ["--block-mb", "0.004", "--readers", "4"]])
def test_cli_reproduces_reference_gff3(golden_dir, tmp_path, name, ext, extra, path):
    """Default = columnar pipeline (reader teams over the mapped input / GPU / writer threads); --record-objects = the
    per-record object loop; tiny blocks = many batches, with ONE reader (-p 3, --readers 1: groups straddle block boundaries
    and are carried) and with four teams (slices cut where the query identifier changes).  Same bytes as the reference printed."""
    folder = _unpack(golden_dir, name, ext, tmp_path)
    r = _run(folder, ext, extra + path + ["--batch", "97"])  # several GPU calls: the stale-ival state must carry across them
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(folder, "out.gff3")).read()


@pytest.mark.parametrize("name,ext,extra,alphabet,nsets", [("survey_dna", ".fna", [], 0, 150), ("survey_protein", ".faa", ["-b", "protein"], 1, 80)])
def test_cli_log_holds_the_references_alignments(oracle, golden_dir, tmp_path, name, ext, extra, alphabet, nsets):
    """taxator -l: one "+ALN i <=> j" line per alignment the control flow computes, with the distance and similarity the
    reference's log prints there (taxonpredictionmodelsequence.hh:533,636,782,802; std::fixed, two decimals) -- held
    against the reference's own log of the same input (tests/golden/survey_*/expected.json: per record set the parsed
    +ALN lines and the STATS counters).  Records that tie completely are ordered by heap address in the reference
    (core/src/alignmentrecord.hh:480) and by input order here, and every query of the survey data has such ties; so the
    input is first rewritten with each set's tied records in the order the reference happened to use (found with the
    oracle, as tests/test_oracle_placement.py does), and then the log must reproduce the reference's, set by set."""
    import itertools
    import json
    folder = _unpack(golden_dir, name, ext, tmp_path)
    exp = json.load(open(os.path.join(folder, "expected.json")))
    fds = R.Dataset(folder, alphabet)
    b = fds.batch()
    tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
    q = oracle.StoreArrays(fds.query.residues, fds.query.offsets)
    db = oracle.StoreArrays(fds.ref.residues, fds.ref.offsets)
    lines, found = [], []
    for s_, group in enumerate(fds.sets[:nsets]):
        lo, hi = b["offsets"][s_], b["offsets"][s_ + 1]
        r = b["recs"][lo:hi]
        ref = [tuple(x) for x in exp[s_]["aln"]]
        ties = {}
        for k, x in enumerate(r):
            ties.setdefault((float(x["score"]), int(x["identities"])), []).append(k)
        glist = list(ties.values())
        total = 1
        for g in glist:
            total *= int(np.prod(np.arange(1, len(g) + 1)))
        best = None
        if total <= 3000:
            for perm in itertools.product(*[itertools.permutations(g) for g in glist]):
                order = np.arange(len(r))
                for g, pg in zip(glist, perm):
                    order[list(g)] = pg
                rc, out, st, tr = oracle.predict(tax, q, db, alphabet, b["qseq"][s_], b["qlen"][s_], r[order], want_trace=True)
                if [(p_, i, j, round(d, 2), round(sm, 2)) for p_, i, j, d, sm in tr] == ref and [int(st["pass0"]), int(st["pass1"]), int(st["pass2"])] == exp[s_]["stats"]:
                    best = order
                    break
        found.append(best is not None)
        for k in (best if best is not None else range(len(group))):
            x = group[int(k)]
            lines.append("%s%s\t%d\t%d\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d" % ("*" if x.filtered else "", x.qid, x.qstart, x.qstop, x.qlen, x.rid, x.rstart,
                                                                           x.rstop, repr(float(x.score)), repr(x.evalue), x.identities, x.alnlen))
    assert sum(found) > nsets // 2
    open(os.path.join(folder, "aln_ordered.tsv"), "w").write("\n".join(lines) + "\n")
    log = str(tmp_path / "aln.log")
    r_ = _run(folder, ext, extra + ["--batch", "37"], aln="aln_ordered.tsv", log=log)
    assert r_.returncode == 0, r_.stderr
    got, cur = [], None
    text = open(log).read()
    assert text.count("\nSTATS\t") == text.count("\n\nID\t") + 1 and text.endswith("\n\n")  # a blank line behind every STATS line (:837)
    for line in open(log):
        if line.startswith("ID\t"):
            cur = dict(id=line.rstrip("\n").split("\t")[1], aln=[], stats=None); got.append(cur)
        elif line.startswith("    +ALN "):
            head, fields = line[len("    +ALN "):].rstrip("\n").split("\t")
            a, b_ = head.split(" <=> ")
            kv = dict(x.split("=") for x in fields.split("; "))
            cur["aln"].append([int(kv["pass"]), -1 if a == "query" else int(a), -1 if b_ == "query" else int(b_), float(kv["dist"]), float(kv["sim"])])
        elif line.startswith("STATS\t"):
            c = line.rstrip("\n").split("\t")
            cur["stats"] = [int(c[3]), int(c[4]), int(c[5])]
    assert len(got) == nsets
    for s_, (g, e) in enumerate(zip(got, exp)):
        assert len(g["aln"]) == sum(g["stats"])
        if not found[s_]:
            continue
        assert g["id"] == e["id"]  # "<qrstart>:<qrstop>@<query id>", taxonpredictionmodelsequence.hh:344,407-408
        assert g["stats"] == e["stats"], (g["id"], g["stats"], e["stats"])
        assert g["aln"] == [[x[0], x[1], x[2], round(x[3], 2), round(x[4], 2)] for x in e["aln"]], (g["id"], g["aln"][:6], e["aln"][:6])


def test_cli_config1_plumbing(oracle, tmp_path):
    """BASELINE config 1 through files: 100 segments x 10 hits, b=2 taxonomy; GFF3 equals the oracle's."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=100, n_hits=10, seg_len=200, branching=2, genome_len=2000, seed=1, variety=True)
    folder = str(tmp_path / "c1")
    synth.write_files(ds, folder)
    log = str(tmp_path / "log.txt")
    r = _run(folder, ".fna", log=log)
    assert r.returncode == 0, r.stderr
    log2 = str(tmp_path / "log2.txt")
    r2 = _run(folder, ".fna", extra=["--record-objects"], log=log2)
    assert r2.returncode == 0 and r2.stdout == r.stdout and open(log2).read() == open(log).read()
    fds = R.Dataset(folder, 0)
    b = fds.batch()
    tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
    q = oracle.StoreArrays(fds.query.residues, fds.query.offsets)
    db = oracle.StoreArrays(fds.ref.residues, fds.ref.offsets)
    rc, out, st = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], b["recs"])
    assert rc == 0
    assert r.stdout == fds.gff3(out, b["qids"], b["qlen"])
    stats_lines = [l.split("\t") for l in open(log) if l.startswith("STATS")]
    assert [int(l[3]) + int(l[4]) + int(l[5]) for l in stats_lines] == [int(x) for x in st["pass0"] + st["pass1"] + st["pass2"]]


@pytest.mark.parametrize("force_bands", [False, True])
def test_cli_mixed_segment_lengths(oracle, tmp_path, force_bands, monkeypatch):
    """A stream of record sets of mixed shape through the driver -- segments of 120 ... 1500 nt, 8 ... 100 hits per set, shuffled --
    so that one batch holds every nucleotide kernel family: one-strip classes up to 12 words, the ring-window classes (385 ... 2048 nt)
    in front of the register strips, one pair per wavefront for the few.  With the banded launches forced onto launches of any size
    and with the engine's own choices; several GPU calls; GFF3 equal to the oracle's either way."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    if force_bands: monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")
    ds = synth.make_ragged_dataset([(150, 8, 150), (40, 20, 300), (30, 12, 500), (16, 10, 1000), (6, 10, 1500), (10, 100, 120)], seed=11, branching=3)
    folder = str(tmp_path / "ragged")
    synth.write_files(ds, folder)
    r = _run(folder, ".fna", ["--batch", "61"])
    assert r.returncode == 0, r.stderr
    r2 = _run(folder, ".fna", ["--record-objects"])
    assert r2.returncode == 0 and r2.stdout == r.stdout
    fds = R.Dataset(folder, 0)
    b = fds.batch()
    tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
    q = oracle.StoreArrays(fds.query.residues, fds.query.offsets)
    db = oracle.StoreArrays(fds.ref.residues, fds.ref.offsets)
    rc, out, st = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], b["recs"])
    assert rc == 0
    assert r.stdout == fds.gff3(out, b["qids"], b["qlen"])


def _multi_segment_dataset(synth, folder, seed, shuffle):
    """A synthetic data set whose queries carry SEVERAL segments (1-4 record sets of the generator joined into one query,
    some of them overlapping so that the split rule has to merge them), lines of a query position-sorted or shuffled."""
    ds = synth.make_dataset(n_sets=260, n_hits=8, seg_len=120, branching=3, genome_len=2000, seed=seed, variety=True)
    synth.write_files(ds, folder)  # taxonomy, mapping, reference FASTA + .fai; the query FASTA and the TSV are rewritten below
    rng = np.random.default_rng(seed)
    b = ds.batch
    qoff = ds.query_offsets.astype(np.int64)
    tsv, fasta = [], []
    s = k = 0
    while s < ds.n_sets:
        g = min(ds.n_sets - s, int(rng.integers(1, 5)))
        letters = np.zeros(0, np.uint8)
        group = []
        for t in range(s, s + g):
            seg = ds.query_letters[qoff[t]:qoff[t + 1]]
            pos = len(letters)
            if t > s and rng.random() < 0.35:
                pos -= int(rng.integers(1, 30))          # overlaps the segment before it: no cut there
            elif t > s and rng.random() < 0.5:
                letters = np.concatenate([letters, ds.query_letters[qoff[s]:qoff[s] + int(rng.integers(1, 40))]])  # a gap
                pos = len(letters)
            letters = np.concatenate([letters[:pos], seg])
            for r in range(int(b["set_offsets"][t]), int(b["set_offsets"][t + 1])):
                group.append(["*" if b["filtered"][r] else "", int(b["qstart"][r]) + pos, int(b["qstop"][r]) + pos, int(b["ref_seq"][r]), int(b["rstart"][r]),
                              int(b["rstop"][r]), float(b["score"][r]), int(b["identities"][r]), int(b["alnlen"][r])])
        if shuffle:
            rng.shuffle(group)
        else:
            group.sort(key=lambda x: (x[1], x[2]))
        for x in group:
            tsv.append("%sm%d\t%d\t%d\t%d\tref%d\t%d\t%d\t%g\t0\t%d\t%d\n" % (x[0], k, x[1], x[2], len(letters), x[3], x[4], x[5], x[6], x[7], x[8]))
        fasta.append(b">m%d\n" % k + letters.tobytes() + b"\n")
        s += g
        k += 1
    with open(os.path.join(folder, "aln.tsv"), "w") as f:
        f.writelines(tsv)
    with open(os.path.join(folder, "query.fna"), "wb") as f:
        f.writelines(fasta)
    return k


@pytest.mark.parametrize("order", ["position-sorted", "shuffled"])
def test_cli_alignments_sorted_split_and_cutoff_options(oracle, tmp_path, order):
    """The options the reference's pipelines really pass (extended/binning-last.bash:108-109: `-x $taxator_speedup -o 1`)
    and the ones next to them: `-o 1` (RecordSetGeneratorSorted: input order, core/src/alignmentrecord.hh:569-631), `-s 0`
    (one record set per query identifier, :543-566), `-x`, `-t` (taxator.cpp:284-287) -- on queries with several segments,
    through the columnar reader and the record-object loop: byte-identical to each other and to the GFF3 made from the
    oracle's rows over the record sets a literal Python reading of the reference's generators builds."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    folder = str(tmp_path / "ms")
    n_queries = _multi_segment_dataset(synth, folder, seed=81, shuffle=order == "shuffled")
    n_sets = {}
    for name, flags, kw in (("default", [], {}),
                            ("-o 1", ["-o", "1"], dict(presorted=True)),
                            ("-s 0", ["-s", "0"], dict(split=False)),
                            ("-o 1 -x 1 -t 0.3", ["-o", "1", "-x", "1", "-t", "0.3"], dict(presorted=True, x=1.0, t=0.3)),
                            ("-x 0 -t 0", ["-x", "0", "-t", "0"], dict(x=0.0, t=0.0)),
                            ("-s 0 -x 0.8 -t 0.1", ["-s", "0", "-x", "0.8", "-t", "0.1"], dict(split=False, x=0.8, t=0.1))):
        fds = R.Dataset(folder, 0, split=kw.get("split", True), presorted=kw.get("presorted", False))
        b = fds.batch()
        tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
        q = oracle.StoreArrays(fds.query.residues, fds.query.offsets)
        db = oracle.StoreArrays(fds.ref.residues, fds.ref.offsets)
        rc, out, st = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], b["recs"], exclude_factor=kw.get("x", 0.5),
                                           reeval_bandwidth=kw.get("t", 0.05))
        assert rc == 0
        want = fds.gff3(out, b["qids"], b["qlen"])
        n_sets[name] = len(out)
        for path in ([], ["--record-objects"], ["--block-mb", "0.004", "-p", "3"]):
            r = _run(folder, ".fna", extra=flags + path + ["--batch", "61"])
            assert r.returncode == 0, (name, path, r.stderr)
            assert r.stdout == want, (name, path)
    assert n_sets["-s 0"] == n_queries < n_sets["default"]
    if order == "shuffled":
        assert n_sets["-o 1"] < n_sets["default"]     # input order: a late segment's line first lifts the running stop over the earlier segments
    else:
        assert n_sets["-o 1"] == n_sets["default"]


@pytest.mark.parametrize("path", [[], ["--record-objects"]])
def test_cli_prunes_a_real_shaped_taxonomy(oracle, tmp_path, path):
    """What the driver does to every real NCBI dump before the engine sees a node index (core/taxator.cpp:324-340: mark the ranks
    of `-r`, default = the seven major ranks, then TaxonTree::deleteUnmarkedNodes, core/src/taxontree.cpp:55-70: an unmarked node
    goes, its children move to its parent, depths shrink :169-178).  The dump here has chains of "no rank" nodes inside its
    edges, as NCBI's has; through `taxator` with the default ranks, with `-r species genus family`, with `-d 0` (nothing deleted:
    deeper rows) -- GFF3 byte for byte what the oracle gives on a literal reading of the pruning (tests/helpers/refio.py) -- and
    with a reference sequence mapped onto a taxid that pruning deletes: the reference then dies on the first alignment line that
    names it (core/src/alignmentrecord.hh:222-227, TaxonNotFound) after printing the record sets in front."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=160, n_hits=12, seg_len=150, branching=3, genome_len=1500, seed=911, variety=True)
    synth.stretch_dataset(ds, lambda i: (i * 7 + 3) % 4)  # 0-3 "no rank" nodes in the edge above every node
    # ... and BRANCHING nodes that pruning deletes too (a chain node is never an LCA; NCBI's "no rank" clades and sub-ranks are): every
    # order becomes a "no rank" clade, every other class a "subclass" -- by default their children move up to the class / phylum
    ds.tax.rank = ["no rank" if rk == "order" else ("subclass" if rk == "class" and k % 2 else rk) for k, rk in enumerate(ds.tax.rank)]
    assert "no rank" in ds.tax.rank[1:] and "subclass" in ds.tax.rank and int(ds.tax.depth.max()) > 12
    folder = str(tmp_path / "pr")
    synth.write_files(ds, folder)
    outs = {}
    for name, extra, kw in (("default", [], {}), ("three ranks", ["-r", "species", "genus", "family"], dict(ranks=("species", "genus", "family"))),
                            ("whole tree", ["-d", "0"], dict(prune=False)),
                            ("whole tree, ranks named", ["-r", "species", "-d", "0"], dict(prune=False))):
        r = _run(folder, ".fna", extra=extra + path)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        fds = R.Dataset(folder, 0, **kw)
        b = fds.batch()
        tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
        q = oracle.StoreArrays(fds.query.residues, fds.query.offsets)
        db = oracle.StoreArrays(fds.ref.residues, fds.ref.offsets)
        rc, out, st = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], b["recs"])
        assert rc == 0
        assert r.stdout == fds.gff3(out, b["qids"], b["qlen"]), name
        outs[name] = (r.stdout, int(fds.tax.depth.max()), len(fds.tax.parent))
    assert outs["default"][1] == 6 and outs["three ranks"][1] == 3 and outs["whole tree"][1] > 12           # levels below the root (no order left)
    assert outs["three ranks"][2] < outs["default"][2] < outs["whole tree"][2]                                  # nodes kept
    assert len({v[0] for v in outs.values()}) == 3 and outs["whole tree"][0] == outs["whole tree, ranks named"][0]
    # a reference sequence on a taxid that the default pruning deletes (a "no rank" node): fine with -d 0, fatal without
    deleted = next(t for t, rk in zip(ds.tax.taxid[1:], ds.tax.rank[1:]) if rk == "no rank")
    lines = open(os.path.join(folder, "mapping.tax")).read().split("\n")
    recs = R.parse_alignments(os.path.join(folder, "aln.tsv"))
    first_seen = {}
    for k, rec in enumerate(recs):
        first_seen.setdefault(rec.rid, k)
    victim = min(first_seen, key=lambda rid: abs(first_seen[rid] - len(recs) // 2))  # a reference the stream first names half way through
    with open(os.path.join(folder, "mapping.tax"), "w") as f:
        f.write("\n".join((l.split("\t")[0] + "\t" + deleted) if l.split("\t")[0] == victim else l for l in lines))
    r = _run(folder, ".fna", extra=["-d", "0"] + path)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run(folder, ".fna", extra=path)
    assert r.returncode != 0 and "An unrecoverable error occurred" in r.stderr and ("taxid " + deleted) in r.stderr
    first_bad = next(k for k, rec in enumerate(recs) if rec.rid == victim)
    qid_bad = recs[first_bad].qid
    printed = [l.split("\t")[0] for l in r.stdout.split("\n")[1:] if l]
    before = []
    for rec in recs[:first_bad]:
        if rec.qid != qid_bad and (not before or before[-1] != rec.qid):
            before.append(rec.qid)
    # the generator closes a group by reading the line BEHIND it (core/src/alignmentrecord.hh:464-479): when the bad line is the
    # first of its own group, the exception leaves while the group in front is still open and takes it along
    if first_bad > 0 and recs[first_bad - 1].qid != qid_bad:
        before.pop()
    assert len(before) > 20
    assert qid_bad not in printed and [q_ for k, q_ in enumerate(printed) if k == 0 or printed[k - 1] != q_] == before
    assert r.stdout == "".join(l + "\n" for l in outs["default"][0].split("\n")[:1 + len(printed)])          # ... the same lines as the good run's


def test_cli_taxator_into_binner(oracle, tmp_path):
    """`taxator | binner` on the GPU box: the GFF3 that taxator prints for queries with several segments goes through the host
    `binner` (consensus per query: regex group, majority / support, core/binner.cpp:154-340, core/src/predictionranges.hh:
    117-266) and must come out as the literal Python reading of the reference's binner (tests/test_binner_cpu.py) makes it --
    the second CLI of the path end to end in the driver's GPU record, not only in the CPU suite."""
    import subprocess
    import test_binner_cpu as TB
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    folder = str(tmp_path / "ms")
    n_queries = _multi_segment_dataset(synth, folder, seed=83, shuffle=True)
    r = _run(folder, ".fna", extra=["--batch", "500"])
    assert r.returncode == 0, r.stderr
    gff3 = r.stdout
    assert gff3.count("\n") > n_queries
    tax = R.FlatTaxonomy(os.path.join(folder, "tax"))
    binner = os.path.join(ROOT, "taxator-tk_amd", "bin", "binner")
    for extra, kw in (([], {}), (["-j", "0.6", "-s", "40"], dict(majority=0.6, min_support=40)), (["-s", "150", "-j", "0.9"], dict(majority=0.9, min_support=150))):
        head, body, err = TB.run_binner(binner, os.path.join(folder, "tax"), gff3, extra, cwd=str(tmp_path))
        exp = TB.naive_binner(gff3.split("\n"), tax, **kw)
        assert sorted(body) == sorted(exp) and len(body) == n_queries
        assert len({l.split("\t")[1] for l in body}) > 5


@pytest.mark.parametrize("path", [[], ["--record-objects"]])
def test_cli_errors_like_the_reference(golden_dir, tmp_path, path):
    folder = os.path.join(golden_dir, "survey_edge")
    bad = tmp_path / "bad.tsv"
    bad.write_text("e0\t1\t200\t200\tnope\t101\t300\t400\t0\t200\t200\n")
    work = tmp_path / "w"
    shutil.copytree(folder, work)
    shutil.copy(bad, work / "aln_bad.tsv")
    r = _run(str(work), ".fna", extra=path, aln="aln_bad.tsv", query="query_edge.fna")
    assert r.returncode == 1 and "bad taxon mapping for alignment reference sequence" in r.stderr  # alignmentrecord.hh:214-219
    (work / "aln_bad.tsv").write_text("e0\t200\t1\t200\tref3\t101\t300\t400\t0\t200\t200\n")
    r = _run(str(work), ".fna", extra=path, aln="aln_bad.tsv", query="query_edge.fna")
    assert r.returncode == 1 and "reverse query positions not allowed" in r.stderr               # alignmentrecord.hh:114
    (work / "aln_bad.tsv").write_text("e0\t1\t200\t200\tref3\t99999101\t99999300\t400\t0\t150\t200\ne0\t1\t200\t200\tref2\t101\t300\t300\t0\t180\t200\n")
    r = _run(str(work), ".fna", extra=path, aln="aln_bad.tsv", query="query_edge.fna")
    assert r.returncode == 1 and "SequenceRangeError" in r.stderr


def test_cli_leaves_at_once_when_the_first_line_does_not_parse(golden_dir, tmp_path):
    """Regression for the run that never ended: `taxator --record-objects` on an input whose first line is
    refused creates its engine, destroys it and leaves within milliseconds -- and the exit handlers of the HIP / HSA runtime
    then span for ever in 2 of 676 and 15 of 850 such runs (tools/hang_hunt.py on the MI355X box: main thread running inside
    libhsa-runtime64 under libamdhip64's static destructor, only when the engine's stream had been made with a CU mask).
    The driver now flushes and leaves with _exit once every engine is destroyed (0 of 2200).  Thirty quick runs under the
    child watchdog keep an eye on it; each must report the reference's error and exit code."""
    work = tmp_path / "w"
    shutil.copytree(os.path.join(golden_dir, "survey_edge"), work)
    (work / "aln_bad.tsv").write_text("e0\t200\t1\t200\tref3\t101\t300\t400\t0\t200\t200\n")
    slowest = 0.0
    for k in range(30):
        r = _run(str(work), ".fna", extra=["--record-objects"] if k % 3 else [], aln="aln_bad.tsv", query="query_edge.fna")
        assert r.returncode == 1 and "reverse query positions not allowed" in r.stderr, (k, r.returncode, r.stderr[-500:])
        slowest = max(slowest, r.seconds)
    assert slowest < 60, slowest


@pytest.mark.parametrize("algo,extra,kw", [
    ("dummy", [], {}), ("simple-lca", [], {}), ("megan-lca", [], {}),
    ("megan-lca", ["-t", "0.3", "-m", "150", "-e", "1e-5", "-c", "2"], dict(toppercent=0.3, minscore=150.0, maxevalue=1e-5, minsupport=2)),
    ("ic-megan-lca", ["-u", "-t", "1"], dict(toppercent=1.0, ignore_unclassified=True)),
    ("n-best-lca", ["-n", "2"], dict(nbest=2))])
def test_cli_alignment_free_models(oracle, golden_dir, tmp_path, algo, extra, kw):
    """taxator -a dummy | simple-lca | megan-lca | ic-megan-lca | n-best-lca (core/taxator.cpp:346-361) through the host
    mirror and the device kernel; no FASTA files involved.  Expected GFF3 = the oracle's rows through the GFF3 rules."""
    folder = _unpack(golden_dir, "survey_dna", ".fna", tmp_path)
    # give the mock taxonomy an "unclassified" branch so that -u has something to ignore
    names = os.path.join(folder, "tax", "names.dmp")
    lines = open(names).read().split("\n")
    marked = 0
    for k, l in enumerate(lines):
        f = l.split("\t|\t")
        if len(f) >= 4 and f[0] not in ("1",) and marked < 3 and k % 7 == 3:
            f[1] = "unclassified " + f[1]; lines[k] = "\t|\t".join(f); marked += 1
    open(names, "w").write("\n".join(lines))
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
    cmd = [BIN, "-a", algo, "-g", os.path.join(folder, "mapping.tax"), "--batch", "61"] + extra
    r = run_child(cmd, timeout=CHILD_LIMIT_S, stdin_path=os.path.join(folder, "aln.tsv"), env=env)
    assert r.returncode == 0, r.stderr
    fds = R.Dataset(folder, 0)
    b = fds.batch()
    evalue = np.array([rec.evalue for group in fds.sets for rec in group], np.float64)
    # Taxon::is_unclassified on the pruned tree, from the full tree (core/src/ncbidata.cpp:119-126)
    uncl = fds.tax.unclassified_flags()
    model = {"dummy": oracle.LCA_DUMMY, "simple-lca": oracle.LCA_SIMPLE, "megan-lca": oracle.LCA_MEGAN, "ic-megan-lca": oracle.LCA_MEGAN,
             "n-best-lca": oracle.LCA_NBEST}[algo]
    tax = oracle.TaxArrays(fds.tax.parent, fds.tax.left, fds.tax.right, fds.tax.depth)
    rc, rows = oracle.lca_predict_batch(tax, model, b["offsets"], b["qlen"], b["recs"], evalue, uncl, **kw)
    assert rc == 0
    assert r.stdout == fds.gff3(rows, b["qids"], b["qlen"])
    if algo in ("megan-lca", "n-best-lca"):  # (the LCA over ALL hits of these sets is the superkingdom)
        assert len(set(rows["lower"])) > 3


def test_cli_indexed_store_loads_only_what_the_stream_touches(oracle, tmp_path):
    """with -i the reference reads only the regions it needs through the .fai.  Here: a reference FASTA with
    100x more sequences than the alignments touch -- the GFF3 must equal the in-memory run's byte for byte, and the STORE
    report must show that only the touched sequences went to HBM (--timing prints resident / known sequences)."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=60, n_hits=6, seg_len=150, branching=2, genome_len=1500, seed=4, variety=True)
    folder = str(tmp_path / "sparse")
    synth.write_files(ds, folder)
    # pad the collection: 100x more sequences nobody refers to, spread between the real ones (.fai rewritten to match)
    rng = np.random.default_rng(1)
    real = {}
    name = None
    for line in open(os.path.join(folder, "ref.fna")):
        if line.startswith(">"): name = line[1:].strip(); real[name] = []
        else: real[name].append(line.strip())
    names = list(real)
    with open(os.path.join(folder, "ref.fna"), "w") as f, open(os.path.join(folder, "ref.fna.fai"), "w") as fai:
        def put(nm, seq):
            f.write(">%s\n" % nm)
            off = f.tell()
            for i in range(0, len(seq), 70): f.write(seq[i:i + 70] + "\n")
            fai.write("%s\t%d\t%d\t70\t71\n" % (nm, len(seq), off))
        filler = 0
        for nm in names:
            for _ in range(100):
                put("filler%d" % filler, "".join(rng.choice(list("ACGT"), int(rng.integers(200, 900))))); filler += 1
            put(nm, "".join(real[nm]))
    with open(os.path.join(folder, "mapping.tax"), "a") as f:
        for k in range(filler): f.write("filler%d\t%s\n" % (k, ds.tax.taxid[int(ds.tax.species[0])]))
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
    base = [BIN, "-g", os.path.join(folder, "mapping.tax"), "-q", os.path.join(folder, "query.fna"), "-f", os.path.join(folder, "ref.fna"), "--timing"]
    mem = run_child(base, timeout=CHILD_LIMIT_S, stdin_path=os.path.join(folder, "aln.tsv"), env=env)
    idx = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "-v", os.path.join(folder, "query.fna.fai"), "--block-mb", "0.01"],
                    timeout=CHILD_LIMIT_S, stdin_path=os.path.join(folder, "aln.tsv"), env=env)  # no query .fai on disk: rebuilt by a scan
    assert mem.returncode == 0 and idx.returncode == 0, (mem.stderr, idx.stderr)
    assert idx.stdout == mem.stdout and idx.stdout.count("\n") == 61

    def store(stderr, which):
        line = [l for l in stderr.splitlines() if l.startswith("STORE\t" + which)][0].split("\t")
        return line[2], {k: int(v) for k, v in (x.split("=") for x in line[3:])}
    kind, ref_mem = store(mem.stderr, "ref")
    assert kind == "in-memory" and ref_mem["resident"] == ref_mem["sequences"] == len(names) * 101
    kind, ref_idx = store(idx.stderr, "ref")
    touched = len(set(int(x) for x in ds.batch["ref_seq"]))
    assert kind == "indexed" and ref_idx["sequences"] == len(names) * 101 and ref_idx["resident"] == touched
    assert ref_idx["residues"] * 20 < ref_mem["residues"] and ref_idx["device_bytes"] * 5 < ref_mem["device_bytes"]  # (tables of 6464 sequences: ~130 KB)
    kind, q_idx = store(idx.stderr, "query")
    assert kind == "indexed" and q_idx["resident"] <= q_idx["sequences"] == ds.n_sets


def test_cli_region_granular_reads_of_long_references(tmp_path):
    """SURVEY 8f N1 / the reference's RandomIndexedSeqstoreRO::getSequence (core/src/sequencestorage.hh:341-369,
    core/src/faidx.h:315-350) reads [start, stop] of a sequence, not the sequence.  Here: one 50 Mbp reference and 63 of
    300 kbp, each touched only where its species' ~1.5 kbp locus sits.  With -i the GFF3 must equal the in-memory run's byte
    for byte while the resident residues follow the touched tiles, not the 69 Mbp of the collection; with a budget of
    100 KB and batches of five record sets the store is emptied and refilled along the stream (--store-gb) and nothing changes."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=400, n_hits=8, seg_len=150, branching=2, genome_len=1500, seed=9, variety=True)
    folder = str(tmp_path / "long")
    synth.write_files(ds, folder)
    rng = np.random.default_rng(2)
    # every species' genome becomes a locus inside a long sequence of random filler
    genomes, name = {}, None
    for line in open(os.path.join(folder, "ref.fna")):
        if line.startswith(">"): name = line[1:].strip(); genomes[name] = []
        else: genomes[name].append(line.strip())
    names = list(genomes)
    shift = {}
    with open(os.path.join(folder, "ref.fna"), "wb") as f, open(os.path.join(folder, "ref.fna.fai"), "w") as fai:
        for k, nm in enumerate(names):
            g = "".join(genomes[nm]).encode()
            total = 50_000_000 if k == 0 else 300_000
            at = int(rng.integers(total // 3, 2 * total // 3))
            seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total)]
            seq[at:at + len(g)] = np.frombuffer(g, np.uint8)
            shift[nm] = at
            f.write(b">" + nm.encode() + b"\n")
            off = f.tell()
            full = total // 80
            body = np.full((full, 81), ord("\n"), np.uint8)
            body[:, :80] = seq[:full * 80].reshape(full, 80)
            f.write(body.tobytes())
            if total % 80:
                f.write(seq[full * 80:].tobytes() + b"\n")
            fai.write("%s\t%d\t%d\t80\t81\n" % (nm, total, off))
    # the alignment rows move with their locus
    rows = []
    for line in open(os.path.join(folder, "aln.tsv")):
        c = line.rstrip("\n").split("\t")
        at = shift[c[4]]
        c[5] = str(int(c[5]) + at); c[6] = str(int(c[6]) + at)
        rows.append("\t".join(c))
    open(os.path.join(folder, "aln.tsv"), "w").write("\n".join(rows) + "\n")
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
    base = [BIN, "-g", os.path.join(folder, "mapping.tax"), "-q", os.path.join(folder, "query.fna"), "-f", os.path.join(folder, "ref.fna"), "--timing"]
    aln = os.path.join(folder, "aln.tsv")
    mem = run_child(base, timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
    idx = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai")], timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
    fine = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "2048"], timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
    tight = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "2048", "--store-gb", "0.0001", "--block-mb", "0.02", "--record-objects", "--batch", "5"],
                      timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
    for r in (mem, idx, fine, tight):
        assert r.returncode == 0, r.stderr[-2000:]
    assert idx.stdout == mem.stdout and fine.stdout == mem.stdout and tight.stdout == mem.stdout and mem.stdout.count("\n") > 400
    # The same budget on the DEFAULT path: the columnar pipeline keeps two batches in flight per engine from two host
    # threads, so one thread's "store full: empty it" must wait for the other thread's batch that is still running on what
    # is resident (round 3 emptied the store under it: a spurious "not resident" failure of the whole run).  Tiny blocks,
    # one and two engines, a few repetitions: dozens of resets each, racing with batches in flight.
    for extra_args in (["-p", "4"], ["-p", "4", "--devices", "0,0"], ["-p", "2"]):
        for _ in range(3):
            r = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "2048", "--store-gb", "0.0001", "--block-mb", "0.004"] + extra_args,
                          timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            assert r.stdout == mem.stdout
    # quirk 6: reverse-strand hits against an IN-MEMORY reference store are reverse-complemented here (the reference's
    # in-memory store returns the whole forward sequence, core/src/sequencestorage.hh:122-130): said once, only then
    assert mem.stderr.count("its in-memory store (-f without -i) hands back the whole") == 1
    assert "hands back the whole" not in idx.stderr

    def store(stderr):
        line = [l for l in stderr.splitlines() if l.startswith("STORE\tref")][0].split("\t")
        return line[2], {k: int(v) for k, v in (x.split("=") for x in line[3:])}
    total = 50_000_000 + 63 * 300_000
    kind, s_mem = store(mem.stderr)
    assert kind == "in-memory" and s_mem["residues"] >= total
    kind, s_idx = store(idx.stderr)
    touched = len(set(int(x) for x in ds.batch["ref_seq"]))
    assert kind == "indexed" and s_idx["resident"] == touched
    assert s_idx["live_residues"] <= touched * 3 * 65536 and s_idx["live_residues"] < total // 5     # a tile or two per locus, not 69 Mbp
    kind, s_fine = store(fine.stderr)
    assert s_fine["live_residues"] <= touched * 4 * 2048 and s_fine["live_residues"] < total // 100 and s_fine["device_bytes"] < s_mem["device_bytes"] // 50
    kind, s_tight = store(tight.stderr)
    assert s_tight["residues"] * 0.5 <= 0.0001 * 2 ** 30 * 1.5 and s_tight["residues"] < s_fine["residues"]                                        # emptied and refilled along the stream
    # Without a budget: a window that swallows resident ones leaves their residues behind in the arena; when more than half of
    # the arena is such garbage the driver empties the store and fetches what the block at hand needs again (the compaction).
    # TAXATOR_STORE_GARBAGE=0: no slack -- many small blocks, many merges, same output, and the arena stays within twice what is live.
    # (one reader: with reader teams the order in which the blocks reach the store, and with it the arena's final size, depends on timing)
    garb = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "512", "--block-mb", "0.004", "-p", "4", "--readers", "1"],
                     timeout=CHILD_LIMIT_S, stdin_path=aln, env=dict(env, TAXATOR_STORE_GARBAGE="0"))
    assert garb.returncode == 0, garb.stderr[-2000:]
    assert garb.stdout == mem.stdout
    kind, s_garb = store(garb.stderr)
    plain = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "512", "--block-mb", "0.004", "-p", "4", "--readers", "1"],
                      timeout=CHILD_LIMIT_S, stdin_path=aln, env=env)
    kind, s_plain = store(plain.stderr)
    assert plain.stdout == mem.stdout and s_plain["residues"] >= s_plain["live_residues"]
    assert s_garb["residues"] <= 2 * s_garb["live_residues"] + 64 * 512 and s_garb["residues"] <= s_plain["residues"]
    teams = run_child(base + ["-i", os.path.join(folder, "ref.fna.fai"), "--store-tile", "512", "--block-mb", "0.004", "-p", "4"],   # the default: reader teams
                      timeout=CHILD_LIMIT_S, stdin_path=aln, env=dict(env, TAXATOR_STORE_GARBAGE="0"))
    assert teams.returncode == 0 and teams.stdout == mem.stdout
    kind, s_teams = store(teams.stderr)
    assert s_teams["residues"] <= 2 * s_teams["live_residues"] + 64 * 512


@pytest.mark.parametrize("name,ext,extra", [("survey_dna", ".fna", []), ("survey_protein", ".faa", ["-b", "protein"])])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_cli_several_engines_keep_input_order(golden_dir, tmp_path, name, ext, extra, devices):
    """--devices a,b,...: one engine + two host threads per listed GPU, blocks dealt to whichever is free, GFF3 in input order
    (the replacement of doPredictionsParallel, core/taxator.cpp:181-210).  The input is a regular file, so there is one READER
    TEAM per engine as well: the mapped file is cut into slices where the query identifier changes and the teams parse slices
    side by side (the slice index is the block's place in the output).  The test box has one GPU: several engines on device 0
    exercise the same pipeline.  Tiny blocks -> dozens of batches in flight across the engines."""
    folder = _unpack(golden_dir, name, ext, tmp_path)
    n_eng = devices.count(",") + 1
    r = _run(folder, ext, extra + ["--devices", devices, "--block-mb", "0.004", "-p", "4", "--timing"])
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(folder, "out.gff3")).read()
    assert "READERS\tteams=%d\tslices" % n_eng in r.stderr
    one = _run(folder, ext, extra + ["--devices", devices, "--block-mb", "0.004", "-p", "4", "--timing", "--readers", "1"])
    assert one.returncode == 0 and one.stdout == r.stdout and "READERS\tteams=1\t" in one.stderr
    # a failure in the middle of the stream: what precedes it is printed, the exit code and message are the serial loop's --
    # whichever team meets the bad line, and with the line number counted from the start of the file
    lines = open(os.path.join(folder, "aln.tsv")).read().split("\n")
    bad = lines[:len(lines) // 2] + ["q_unknown\t1\t50\t200\tnope\t1\t50\t10\t0\t50\t50"] + lines[len(lines) // 2:]
    (tmp_path / name / "aln_bad.tsv").write_text("\n".join(bad))
    good = open(os.path.join(folder, "out.gff3")).read()
    outs = set()
    for readers in ([], ["--readers", "1"], ["--readers", "3"]):
        r = _run(folder, ext, extra + ["--devices", devices, "--block-mb", "0.004"] + readers, aln="aln_bad.tsv")
        assert r.returncode == 1 and "An unrecoverable error occurred" in r.stderr
        assert "[line %d]" % (len(lines) // 2 + 1) in r.stderr, r.stderr[-500:]
        assert good.startswith(r.stdout) and 0 < len(r.stdout) < len(good)
        outs.add(r.stdout)
    assert len(outs) == 1  # the same prefix of the output however the input was read

