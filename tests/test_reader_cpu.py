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
