"""The columnar TSV reader (taxator-tk_amd/host/columnar.*) against the record-object reader that mirrors the reference's
AlignmentRecord / RecordSetGenerator classes: same record sets, same field values (score compared by bit pattern), same
error text and the same record sets delivered before a failure -- on the fixtures and on fuzzed streams, with one and
several parser threads and with blocks so small that every group straddles a block boundary.  No GPU involved."""
import os
import random
import subprocess
import tempfile

import pytest

import __graft_entry__ as graft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "parsecheck")


@pytest.fixture(scope="module")
def parsecheck():
    if not os.path.exists(BIN):
        graft.build()
    return BIN


def _run(binary, folder, mode, split, data, threads=0, block=1 << 26, query="query_edge.fna", as_file=False):
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
    cmd = [binary, os.path.join(folder, "mapping.tax"), os.path.join(folder, query), os.path.join(folder, "ref.fna"), mode, str(split),
           str(threads), str(block)]
    if as_file:  # a regular file on stdin: the columnar reader maps it instead of reading it
        with tempfile.NamedTemporaryFile() as f:
            f.write(data)
            f.flush()
            with open(f.name, "rb") as g:
                r = subprocess.run(cmd, stdin=g, capture_output=True, env=env)
    else:        # a pipe: the read-ahead thread
        r = subprocess.run(cmd, input=data, capture_output=True, env=env)
    assert r.returncode in (0, 1), r.stderr
    return r.returncode, r.stdout


def _both(binary, folder, data, split=1, **kw):
    want = _run(binary, folder, "objects", split, data, **kw)
    for threads, block, as_file in ((1, 1 << 26, False), (4, 1 << 26, False), (3, 4096, False), (1, 4096, False),
                                    (4, 1 << 26, True), (2, 4096, True), (1, 100, True)):
        got = _run(binary, folder, "columnar", split, data, threads=threads, block=block, as_file=as_file, **kw)
        assert got == want, (threads, block, as_file, want[1][-300:], got[1][-300:])
    # the reader teams' path (taxator --devices a,b,...): the mapped file cut into slices where the query identifier changes,
    # every slice parsed on its own -- same record sets, same error text with the same "[line N]"
    if data:
        for threads, block in ((1, 1 << 26), (2, 4096), (3, 100), (1, 1)):
            got = _run(binary, folder, "slices", split, data, threads=threads, block=block, as_file=True, **kw)
            assert got == want, ("slices", threads, block, want[1][-300:], got[1][-300:])
    return want


def test_fixture_streams(parsecheck, golden_dir):
    folder = os.path.join(golden_dir, "survey_edge")
    for name, query in (("aln_edge.tsv", "query_edge.fna"), ("aln.tsv", "query.fna")):
        data = open(os.path.join(folder, name), "rb").read()
        for split in (1, 0):
            rc, out = _both(parsecheck, folder, data, split=split, query=query)
            assert rc == 0 and out.count(b"SET\t") > 0


def _good_line(rng, qid, refs):
    qs = rng.randint(1, 50)
    qe = qs + rng.randint(0, 300)
    score = rng.choice(["%d" % rng.randint(0, 2000), "%.1f" % rng.uniform(0, 900), "%g" % rng.uniform(0, 1e6), "%e" % rng.uniform(1e-3, 1e9),
                        "1e3", "52.8", "0.000123", "16777217", "123456789", "4.5E+02", "+7", "-0", ".5", "5.", "0x10", "inf", "nan",
                        "3.4028235e38", "1e-45", "0.1000000015"])
    ev = rng.choice(["0", "1e-50", "3.2e-12", "10", "0.0", "1E5", "inf", ".5e-3"])
    star = "*" if rng.random() < 0.1 else ""
    tail = "\textra\tcolumns" if rng.random() < 0.1 else ""
    return "%s%s\t%d\t%d\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d%s" % (star, qid, qs, qe, rng.randint(200, 5000), rng.choice(refs), rng.randint(1, 3000),
                                                              rng.randint(1, 3000), score, ev, rng.randint(0, 300), rng.randint(1, 400), tail)


BAD_LINES = ["", "x", "*", "e0\t1\t2", "e0\t5\t1\t200\tref3\t1\t2\t3\t0\t1\t1", "e0\t1\t2\t200\tnope\t1\t2\t3\t0\t1\t1",
             "e0\t1\t2\t200\tref3\t1\t2\tabc\t0\t1\t1", "e0\t1\t2\t200\tref3\t1\t2\t3\tzz\t1\t1", "e0\t-1\t2\t200\tref3\t1\t2\t3\t0\t1\t1",
             "e0\t1\t2\t200\tref3\t1\t2\t3\t0\t1\t1\r", "e0\t1\t99999999999\t200\tref3\t1\t2\t3\t0\t1\t1", "e0\t 1\t2\t200\tref3\t1\t2\t3\t0\t1\t1",
             "e0\t1\t2\t200\tref3\t1\t2\t1e\t0\t1\t1", "e0\t1\t2\t200\tref3\t1\t2\t3\t0\t1\t", "e0\t1\t2\t200\tref3\t1\t2\t \t0\t1\t1",
             "e0\t1\t2\t200\tref3\t+1\t2\t3\t0\t1\t1", "e0\t1\t2\t200\tref3\t1\t2\t3\t0\t1.5\t1",
             # accepted oddities (embedded NUL ends a numeric field like c_str() does; leading zeros; hex float)
             "e0\t1\x00junk\t2\t200\tref3\t1\t2\t3\x00x\t0\x00y\t1\t1", "e0\t0000000000000000000001\t2\t200\tref3\t1\t2\t0x1p3\t0\t1\t1",
             "e0\t\x00\t2\t200\tref3\t1\t2\t\x00\t0\t1\t1"]


def test_fuzzed_streams(parsecheck, golden_dir):
    folder = os.path.join(golden_dir, "survey_edge")
    refs = [l.split("\t")[0] for l in open(os.path.join(folder, "mapping.tax")) if l.strip() and not l.startswith("#")][:40]
    qids = [l[1:].split()[0] for l in open(os.path.join(folder, "query_edge.fna")) if l.startswith(">")]
    rng = random.Random(11)
    n_err = 0
    for trial in range(60):
        lines = []
        for _ in range(rng.randint(1, 40)):
            qid = rng.choice(qids) if rng.random() < 0.95 else "unknown_query"
            for _ in range(rng.randint(1, 30)):
                lines.append(_good_line(rng, qid, refs))
                if rng.random() < 0.03:
                    lines.append("# a comment line")
        if trial % 3 == 0:
            lines.insert(rng.randint(0, len(lines)), rng.choice(BAD_LINES))
        data = ("\n".join(lines) + ("\n" if trial % 2 else "")).encode()
        rc, out = _both(parsecheck, folder, data, split=trial % 4 != 3)
        n_err += rc
        if rc:
            assert out.rstrip().split(b"\n")[-1].startswith(b"ERROR ")
    assert 5 < n_err < 60


def test_large_group_and_long_line(parsecheck, golden_dir):
    """A group larger than the insertion-sort cut-off (with ties that must keep input order) and a line longer than a block."""
    folder = os.path.join(golden_dir, "survey_edge")
    rng = random.Random(5)
    lines = ["e0\t%d\t%d\t200\tref3\t%d\t%d\t%d\t0\t10\t10" % (s, s + rng.randint(0, 3), k + 1, k + 50, rng.randint(1, 5))
             for k, s in enumerate(rng.choices(range(1, 40), k=500))]
    lines.append("e1\t1\t2\t200\tref3\t1\t2\t3\t0\t1\t1\t" + "x" * 20000)
    rc, out = _both(parsecheck, folder, ("\n".join(lines) + "\n").encode())
    assert rc == 0


def _model_sets(data, split, presorted):
    """The record sets the reference's generators would build (tests/helpers/refio.py: a literal Python reading of
    core/src/alignmentrecord.hh:457-504, :543-566, :569-631) as (qid, [(qstart, qstop, rstart, rstop)...]) tuples."""
    from helpers import refio as R
    recs = R.parse_alignments([l for l in data.decode().split("\n") if l != ""])
    return [(g[0].qid, [(r.qstart, r.qstop, r.rstart, r.rstop) for r in g]) for g in R.record_sets(recs, split=split, presorted=presorted)]


def _dump_sets(out):
    sets = []
    for line in out.decode().split("\n"):
        f = line.split("\t")
        if f[0] == "SET":
            sets.append((f[1], []))
        elif len(f) == 10:
            sets[-1][1].append((int(f[1]), int(f[2]), int(f[3]), int(f[4])))
    return sets


@pytest.mark.parametrize("order", ["position-sorted", "as-it-comes"])
def test_alignments_sorted_option(parsecheck, golden_dir, monkeypatch, order):
    """`taxator -o 1` (the flag the LAST pipeline runs with, extended/binning-last.bash:109) = RecordSetGeneratorSorted
    (core/src/alignmentrecord.hh:569-631): no sort per query; a new record set starts where a line's query start lies
    beyond the running maximum of the query stops -- in INPUT order.  On position-sorted input that is what the default
    generator builds; on unsorted input it is not (and that, too, must be what the reference does: other sets).
    Both readers, every block / thread geometry, against a literal Python reading of the reference's loop."""
    folder = os.path.join(golden_dir, "survey_edge")
    refs = [l.split("\t")[0] for l in open(os.path.join(folder, "mapping.tax")) if l.strip() and not l.startswith("#")][:40]
    qids = [l[1:].split()[0] for l in open(os.path.join(folder, "query_edge.fna")) if l.startswith(">")]
    rng = random.Random(29)
    differs = 0
    for trial in range(12):
        lines = []
        qid = None
        for _ in range(rng.randint(1, 25)):
            qid = rng.choice([q for q in qids if q != qid])  # (two groups of one id in a row would be ONE group)
            group = []
            pos = 1
            for _ in range(rng.randint(1, 5)):               # segments of the query, some touching, some apart
                for _ in range(rng.randint(1, 8)):
                    qs = pos + rng.randint(0, 5)
                    qe = qs + rng.randint(0, 120)
                    group.append("%s%s\t%d\t%d\t5000\t%s\t%d\t%d\t%d\t0\t%d\t%d" % ("*" if rng.random() < 0.05 else "", qid, qs, qe, rng.choice(refs), rng.randint(1, 3000),
                                                                                rng.randint(1, 3000), rng.randint(1, 400), rng.randint(0, 100), qe - qs + 1))
                pos += rng.choice([60, 126, 127, 140, 300])
            if order == "position-sorted":
                group.sort(key=lambda l: (int(l.split("\t")[1]), int(l.split("\t")[2])))
            else:
                rng.shuffle(group)
            lines += group
            if rng.random() < 0.2:
                lines.append("# comment")
        data = ("\n".join(lines) + "\n").encode()
        monkeypatch.setenv("PARSECHECK_SORTED", "1")
        rc, out = _both(parsecheck, folder, data, split=1)
        assert rc == 0
        got = _dump_sets(out)
        assert got == _model_sets(data, True, True)
        monkeypatch.delenv("PARSECHECK_SORTED")
        rc0, out0 = _both(parsecheck, folder, data, split=1)
        assert _dump_sets(out0) == _model_sets(data, True, False)
        if order == "position-sorted":
            assert [sorted(s[1]) for s in got] == [sorted(s[1]) for s in _dump_sets(out0)]  # same cuts (ties may sit in another order)
        else:
            differs += got != _dump_sets(out0)
        # -s 0: one record set per query identifier, lines in input order, whatever -o says (:543-566; the reference's
        # `-o 1 -s 0` never returns -- RecordSetGeneratorSorted<false> does not advance, :612 -- here it is `-s 0`)
        want_whole = _model_sets(data, False, False)
        for env_sorted in (False, True):
            if env_sorted: monkeypatch.setenv("PARSECHECK_SORTED", "1")
            rc1, out1 = _both(parsecheck, folder, data, split=0)
            assert rc1 == 0 and _dump_sets(out1) == want_whole
            monkeypatch.delenv("PARSECHECK_SORTED", raising=False)
    if order != "position-sorted":
        assert differs > 6


def test_bad_line_right_behind_a_group(parsecheck, golden_dir):
    """The reference's generator closes a group by reading the record behind it: a line that does not parse there takes the
    open group with it (its record set is never predicted).  The slice path cuts the file exactly at such boundaries, so
    it has to look at the line behind a slice before it lets the slice's last group go."""
    folder = os.path.join(golden_dir, "survey_edge")
    refs = [l.split("\t")[0] for l in open(os.path.join(folder, "mapping.tax")) if l.strip() and not l.startswith("#")][:10]
    qids = [l[1:].split()[0] for l in open(os.path.join(folder, "query_edge.fna")) if l.startswith(">")]
    rng = random.Random(3)
    for bad in BAD_LINES[:8]:
        for comments_between in (False, True):
            lines = []
            for k in range(60):
                qid = qids[k % len(qids)] if k % len(qids) else qids[0]
                if lines and lines[-1].split("\t")[0].lstrip("*") == qid:
                    qid = qids[(k + 1) % len(qids)]
                for _ in range(3):
                    lines.append(_good_line(rng, qid, refs))
                if k == 40:
                    if comments_between:
                        lines.append("# a comment between the group and the bad line")
                    lines.append(bad)
            data = ("\n".join(lines) + "\n").encode()
            rc, out = _both(parsecheck, folder, data)   # (objects == columnar == slices, blocks down to one line)
            assert rc == 1 and out.rstrip().split(b"\n")[-1].startswith(b"ERROR ")
            recs = [l for l in out.split(b"\n") if l and not l.startswith(b"SET\t") and not l.startswith(b"ERROR ")]
            assert len(recs) == 40 * 3                  # groups 0..39: group 40 went down with the line behind it
