"""The indexed sequence store of the host driver (taxator -i / -v: memory-mapped FASTA + .fai, host/sequencestorage.*)
against the in-memory store: same identifiers, lengths and residues -- with a samtools-style .fai, with an index rebuilt
by a scan (no .fai), with irregular lines, CRLF line ends, blanks inside lines, empty sequences and a last line without a
line break; a .fai that does not fit the file is refused.  Also -o 1 (alignments-sorted): both readers cut the record
sets in INPUT order (core/src/alignmentrecord.hh:569-631) and agree with each other.  No GPU involved."""
import os
import random
import subprocess

import pytest

import __graft_entry__ as graft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "parsecheck")


@pytest.fixture(scope="module")
def parsecheck():
    if not os.path.exists(BIN):
        graft.build()
    return BIN


def _write_regular(path, seqs, width, eol=b"\n", last_eol=True):
    """FASTA with fixed line width + the .fai samtools would write (core/src/faidx.h:438-476)."""
    fai = []
    with open(path, "wb") as f:
        for k, (name, seq) in enumerate(seqs):
            f.write(b">" + name + b" some description" + eol)
            off = f.tell()
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width])
                if i + width < len(seq) or last_eol or k + 1 < len(seqs):
                    f.write(eol)
            fai.append(b"%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + len(eol)))
    with open(path + ".fai", "wb") as f:
        f.write(b"".join(fai))


def _check(binary, fasta, fai):
    r = subprocess.run([binary, "--fasta", fasta, fai], capture_output=True, text=True)
    return r.returncode, r.stdout.strip()


def test_indexed_store_equals_in_memory_store(parsecheck, tmp_path):
    rng = random.Random(3)
    seqs = [(b"s%d" % k, bytes(rng.choice(b"ACGTNacgtnRY") for _ in range(rng.choice([0, 1, 59, 60, 61, 120, 121, 1000, 7777]))))
            for k in range(40)]
    total = sum(len(s) for _, s in seqs)
    for width, eol, last_eol in ((60, b"\n", True), (80, b"\r\n", True), (7, b"\n", False), (10000, b"\n", True)):
        path = str(tmp_path / ("w%d_%d.fna" % (width, len(eol))))
        _write_regular(path, seqs, width, eol, last_eol)
        assert _check(parsecheck, path, path + ".fai") == (0, "OK 40 %d" % total), (width, eol)
        assert _check(parsecheck, path, "-") == (0, "OK 40 %d" % total), (width, eol)          # index rebuilt by a scan
        assert _check(parsecheck, path, str(tmp_path / "missing.fai")) == (0, "OK 40 %d" % total)
    # irregular lines, blanks inside lines: only the scanned index can describe them
    path = str(tmp_path / "irregular.fna")
    with open(path, "wb") as f:
        for name, seq in seqs:
            f.write(b">" + name + b"\n")
            i = 0
            while i < len(seq):
                n = rng.randint(1, 90)
                chunk = seq[i:i + n]
                if rng.random() < 0.2 and len(chunk) > 2:
                    chunk = chunk[:1] + b" " + chunk[1:2] + b"\t" + chunk[2:]
                f.write(chunk + rng.choice([b"\n", b"\r\n", b"\n\n"]))
                i += n
    assert _check(parsecheck, path, "-") == (0, "OK 40 %d" % total)
    # an EMPTY line in the middle of a record whose lines are otherwise regular (the length so far is a multiple of the line
    # length where it sits): the scanned index must not call the record regular (a fetch by line arithmetic would copy the
    # line break); and a samtools-style .fai that ignores the empty line must still give the right residues
    path = str(tmp_path / "blank_inside.fna")
    with open(path, "wb") as f:
        for k, (name, seq) in enumerate(seqs):
            f.write(b">" + name + b"\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + b"\n")
                if k % 3 == 0 and i == 60:
                    f.write(b"\n")                      # after exactly two full lines
    assert _check(parsecheck, path, "-") == (0, "OK 40 %d" % total)
    # a .fai that points beyond the file is refused, not followed
    path = str(tmp_path / "w60_1.fna")
    open(path + ".bad.fai", "w").write("s0\t100\t99999999\t60\t61\n")
    rc, out = _check(parsecheck, path, path + ".bad.fai")
    assert rc == 1 and out.startswith("ERROR") and "does not fit" in out


def test_alignments_sorted_option(parsecheck, golden_dir):
    """-o 1: no sort before the cut.  On input that IS sorted by (qstart, qstop) within each query the record sets equal the
    default path's; on unsorted input they follow the input order, and the two readers still agree."""
    folder = os.path.join(golden_dir, "survey_edge")
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))

    def run(mode, data, sorted_flag):
        e = dict(env, PARSECHECK_SORTED="1") if sorted_flag else env
        cmd = [parsecheck, os.path.join(folder, "mapping.tax"), os.path.join(folder, "query_edge.fna"), os.path.join(folder, "ref.fna"), mode, "1", "2", "4096"]
        r = subprocess.run(cmd, input=data, capture_output=True, env=e)
        assert r.returncode == 0, r.stdout[-300:]
        return r.stdout

    rows = [(5, 40), (1, 30), (35, 60), (100, 140), (90, 99), (141, 150)]
    unsorted = "".join("e0\t%d\t%d\t200\tref3\t%d\t%d\t%d\t0\t10\t10\n" % (a, b, a, b, 100 + a) for a, b in rows).encode()
    srt = "".join("e0\t%d\t%d\t200\tref3\t%d\t%d\t%d\t0\t10\t10\n" % (a, b, a, b, 100 + a) for a, b in sorted(rows)).encode()
    assert run("objects", srt, True) == run("objects", srt, False) == run("columnar", srt, True) == run("columnar", srt, False)
    o, c = run("objects", unsorted, True), run("columnar", unsorted, True)
    assert o == c and o != run("objects", unsorted, False)
    # input order: (5,40) (1,30) (35,60) | (100,140) (90,99: 90 <= 140 joins) | (141,150)
    assert [int(l.split(b"\t")[4]) for l in o.split(b"\n") if l.startswith(b"SET")] == [3, 2, 1]


@pytest.mark.parametrize("chunk", ["1", "37", "500", "100000"])
def test_in_memory_store_loaded_by_many_threads(parsecheck, tmp_path, chunk, monkeypatch):
    """The in-memory store is read by all host threads: the mapped file cut into chunks at line starts, records and residues
    counted per chunk, then copied to their places (host/sequencestorage.cpp).  TAXATOR_FASTA_CHUNK makes the chunks a few
    bytes long, so that every kind of line sits at a chunk seam: headers, irregular lines with blanks and CRLF, empty lines,
    a record that spans dozens of chunks, chunks without any line start of their own -- and text in front of the first
    header (dropped, as by the reference's reader).  The indexed store (its index rebuilt by a scan) must hold the same."""
    monkeypatch.setenv("TAXATOR_FASTA_CHUNK", chunk)
    rng = random.Random(5)
    seqs = [(b"r%d" % k, bytes(rng.choice(b"ACGTNacgtn") for _ in range(rng.choice([0, 1, 2, 59, 60, 61, 200, 3000]))))
            for k in range(60)]
    total = sum(len(s) for _, s in seqs)
    path = str(tmp_path / "odd.fna")
    with open(path, "wb") as f:
        for name, seq in seqs:
            f.write(b">" + name + rng.choice([b"", b" description with > inside", b"\tx"]) + rng.choice([b"\n", b"\r\n"]))
            i = 0
            while i < len(seq):
                n = rng.randint(1, 120)
                piece = seq[i:i + n]
                if rng.random() < 0.2 and len(piece) > 2:
                    piece = piece[:1] + b" " + piece[1:2] + b"\t" + piece[2:]
                f.write(piece + rng.choice([b"\n", b"\r\n", b"\n\n"]))
                i += n
    assert _check(parsecheck, path, "-") == (0, "OK 60 %d" % total)
    # text in front of the first header belongs to no record
    body = open(path, "rb").read()
    lead = str(tmp_path / "lead.fna")
    with open(lead, "wb") as f:
        f.write(b"ACGTACGT\nthis is not a record\n\n" + body)
    out = subprocess.run([parsecheck, "--fasta", lead, "-"], capture_output=True, text=True)
    # the scanned index of the indexed store starts at the first header too: both stores agree on 60 sequences
    assert out.returncode == 0 and out.stdout.strip() == "OK 60 %d" % total, out.stdout
