#!/usr/bin/env python3
"""Build tests/golden/survey_* from the run the SURVEY stage left under /tmp/probe (container-only, one-off).

PROVENANCE -- read before trusting these files as an oracle:
  The outputs (out.gff3, log.txt) were produced during the survey by the reference's *unmodified*
  RPAPredictionModel::predict / parsers / GFF3 writer (included in place from /root/reference/core), but linked
  against a throw-away stand-in for the Boost headers this image lacks.  This round's
  rules forbid making such a build, so it is NOT reproduced here and these files are SECONDARY EVIDENCE only:
  record-set-level parity stays formally "unpinned" (oracle/rpa_oracle.h).  Pair-level parity is pinned
  separately against the reference's real SeqAn (oracle/_ref).
  Inputs (taxonomy, FASTA, alignments) are synthetic data from the survey's generator, not reference files.

What is stored: the input files (subset), the reference GFF3 (subset), and `expected.json` = per record set the
ID, NUMREF, final RANGE (names), STATS pass counters and every `+ALN` line (pass, i, j, dist, sim) parsed from the
reference's -l log.  No reference source text is stored.
"""
import gzip, json, os, re, shutil, sys

SRC = "/tmp/probe"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_log(path):
    blocks, cur, p = [], None, None
    for line in open(path):
        if line.startswith("ID\t"):
            cur = {"id": line.split("\t")[1].strip(), "aln": [], "range": None}
            blocks.append(cur)
            p = None
        elif line.startswith("  PASS"):
            p = int(line.split("\t")[1])
        elif line.startswith("  NUMREF"):
            cur["numref"] = int(line.split("\t")[1])
        elif line.lstrip().startswith("RANGE\t"):
            cur["range"] = line.strip().split("\t")[1:4]  # rtax, lower, upper (names); the last one wins
        elif line.startswith("STATS"):
            f = line.rstrip("\n").split("\t")
            cur["stats"] = [int(f[3]), int(f[4]), int(f[5])]
        else:
            m = re.match(r"\s+\+ALN (\S+) <=> (\S+)\tdist=([\d.]+); sim=([\d.]+)", line)
            if m:
                i, j = m.group(1), m.group(2)
                cur["aln"].append([p, -1 if i == "query" else int(i), -1 if j == "query" else int(j),
                                   float(m.group(3)), float(m.group(4))])
    return blocks


def copy_tax(src, dst):
    os.makedirs(os.path.join(dst, "tax"), exist_ok=True)
    for n in ("nodes.dmp", "names.dmp"):
        shutil.copy(os.path.join(src, "tax", n), os.path.join(dst, "tax", n))
    shutil.copy(os.path.join(src, "mapping.tax"), os.path.join(dst, "mapping.tax"))


def subset(src, dst, nsets, ext, with_gff=True):
    os.makedirs(dst, exist_ok=True)
    copy_tax(src, dst)
    with open(os.path.join(src, "ref" + ext), "rb") as f, gzip.GzipFile(os.path.join(dst, "ref" + ext + ".gz"), "wb", mtime=0) as g:
        g.write(f.read())
    shutil.copy(os.path.join(src, "ref" + ext + ".fai"), os.path.join(dst, "ref" + ext + ".fai"))
    blocks = parse_log(os.path.join(src, "log.txt"))[:nsets]
    keep_q, lines = [], []
    qids = [b["id"].split("@", 1)[1] for b in blocks]
    last_q = qids[-1]
    for line in open(os.path.join(src, "aln.tsv")):
        q = line.lstrip("*").split("\t", 1)[0]
        if q not in keep_q:
            if len(keep_q) and keep_q[-1] == last_q:
                break
            keep_q.append(q)
        lines.append(line)
    open(os.path.join(dst, "aln.tsv"), "w").writelines(lines)
    want = set(keep_q)
    with open(os.path.join(dst, "query" + ext), "w") as out:
        on = False
        for line in open(os.path.join(src, "query" + ext)):
            if line.startswith(">"):
                on = line[1:].split()[0] in want
            if on:
                out.write(line)
    if with_gff:
        g = open(os.path.join(src, "out.gff3")).read().splitlines(True)
        open(os.path.join(dst, "out.gff3"), "w").writelines(g[: 1 + len(blocks)])
    json.dump(blocks, open(os.path.join(dst, "expected.json"), "w"), separators=(",", ":"))
    print(dst, len(blocks), "sets")


if __name__ == "__main__":
    subset(os.path.join(SRC, "data2"), os.path.join(HERE, "survey_dna"), 400, ".fna")
    subset(os.path.join(SRC, "data3"), os.path.join(HERE, "survey_protein"), 300, ".faa")
    # small set + hand-made edge cases (n==0, n==1, perfect-hit shortcut, reverse strand, split query); log only
    dst = os.path.join(HERE, "survey_edge")
    src = os.path.join(SRC, "data")
    os.makedirs(dst, exist_ok=True)
    copy_tax(src, dst)
    for n in ("ref.fna", "ref.fna.fai", "query.fna", "query_edge.fna", "aln.tsv", "aln_edge.tsv"):
        shutil.copy(os.path.join(src, n), os.path.join(dst, n))
    json.dump(parse_log(os.path.join(src, "log.txt")), open(os.path.join(dst, "expected.json"), "w"), separators=(",", ":"))
    json.dump(parse_log(os.path.join(src, "log_edge.txt")), open(os.path.join(dst, "expected_edge.json"), "w"), separators=(",", ":"))
