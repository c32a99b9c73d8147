"""Measurement tools that make their own data must make CONSISTENT data (a benchmark over files whose alignments do not match
their sequences measures nothing).  tools/bigstore_gen.cpp (round 5: the 32 GB indexed reference of tools/cli_bigstore.py) at a
small size: every alignment line's identities, length and coordinates must be what the FASTA files hold at those coordinates,
forward and reverse strand, fetched through the .fai geometry the generator wrote.  CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fetch(f, ent, a, b):  # residues a .. b (1-based inclusive) of a .fai entry (length, offset, line bases, line width)
    ln, off, lb, lw = ent
    assert 1 <= a <= b <= ln
    s = off + (a - 1) // lb * lw + (a - 1) % lb
    e = off + (b - 1) // lb * lw + (b - 1) % lb
    f.seek(s)
    return f.read(e - s + 1).replace(b"\n", b"")


def test_bigstore_generator_writes_what_its_alignments_say(tmp_path):
    exe = str(tmp_path / "bigstore_gen")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tools", "bigstore_gen.cpp")])
    out = tmp_path / "data"
    out.mkdir()
    species = tmp_path / "species.txt"
    species.write_text("".join("%d\n" % (1000 + k) for k in range(50)))
    subprocess.check_call([exe, str(out), "0.25", "3000", "12", str(species), "99"], stderr=subprocess.DEVNULL)
    fai = {l.split("\t")[0]: list(map(int, l.split("\t")[1:])) for l in open(out / "ref.fna.fai")}
    qfai = {l.split("\t")[0]: list(map(int, l.split("\t")[1:])) for l in open(out / "query.fna.fai")}
    assert sum(e[0] for e in fai.values()) == 250_000_000 and min(e[0] for e in fai.values()) >= 10_000 and len(qfai) == 3000
    mapping = dict(l.rstrip("\n").split("\t") for l in open(out / "mapping.tax"))
    assert set(mapping) == set(fai) and set(mapping.values()) <= {str(1000 + k) for k in range(50)}
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n = rev = 0
    sets = set()
    with open(out / "ref.fna", "rb") as ref, open(out / "query.fna", "rb") as qf:
        for k, line in enumerate(open(out / "aln.tsv")):
            q, qs, qe, ql, r, rs, re_, sc, ev, idn, al = line.rstrip("\n").split("\t")
            sets.add(q)
            if k % 7:
                continue
            qs, qe, ql, rs, re_, idn, al = int(qs), int(qe), int(ql), int(rs), int(re_), int(idn), int(al)
            assert qs == 1 and qe == ql == qfai[q][0] == al
            qseq = _fetch(qf, qfai[q], qs, qe)
            if rs <= re_:
                rseq = _fetch(ref, fai[r], rs, re_)
            else:
                rseq = _fetch(ref, fai[r], re_, rs).translate(comp)[::-1]
                rev += 1
            assert len(rseq) == len(qseq) == al and sum(a == b for a, b in zip(qseq, rseq)) == idn and int(sc) == 2 * idn - al
            assert idn > 0.7 * al   # (a copy of the same ancestral stretch: 5 % + 2-15 % substitutions)
            n += 1
    assert n > 500 and rev > n // 30 and len(sets) == 3000   # (0.25 Gbases over 20 k ancestral blocks: one to three copies, i.e. hits, per set)
