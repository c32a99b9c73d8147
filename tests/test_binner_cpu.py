"""`binner` (host C++, taxator-tk_amd/host/binner.cpp): consensus binning of taxator's GFF3 segments.

No GPU involved.  PARITY UNPINNED: the reference has no test or fixture for binner and cannot be built here (Boost), so
the C++ program is checked against a second, deliberately literal reading of core/binner.cpp:56-346 and
core/src/predictionranges.hh:30-266 in Python below (lists of dicts, parent walks, the reference's 16-bit supports),
on the reference's own GFF3 output (tests/golden/survey_dna/out.gff3) and on generated multi-segment inputs, plus
hand-worked cases.  Group ORDER follows std::unordered_map in both the reference and the C++ program; the test compares
sorted lines.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import refio as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "binner")
U16 = 0xFFFF


@pytest.fixture(scope="module")
def binner_bin():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "taxator-tk_amd", "host")])
    return BIN


def parse_gff3(line, tax):
    f = line.split("\t")
    rec = dict(qid=f[0], begin=int(f[3]), end=int(f[4]))
    for kv in f[8].split(";"):
        if not kv:
            continue
        k, v = kv.split("=", 1)
        if k == "seqlen":
            rec["qlen"] = int(v)
        elif k == "tax":
            parts = v.split("-")
            tid, _, sup = parts[0].partition(":")
            support = int(sup) if sup else rec["end"] - rec["begin"] + 1
            last = tax.index_of[tid]
            rec["lower"] = last
            ts = []
            for part in parts[1:]:
                tid, _, sup = part.partition(":")
                node = tax.index_of[tid]
                p = last
                while p != node:
                    ts.insert(0, support)
                    p = int(tax.parent[p])
                if sup:
                    support = int(sup)
                last = node
            ts.insert(0, support)
            rec["upper"] = last
            rec["ts"] = ts
    return rec


def support_at(rec, tax, depth):
    idx = depth - int(tax.depth[rec["upper"]])
    if idx >= 0:
        return rec["ts"][idx] if idx < len(rec["ts"]) else rec["ts"][-1]
    return 0


def combine(records, tax, majority, min_support):
    """combinePredictionRanges, literal; returns (upper node, support at upper, summed length)."""
    tl = []
    summed_support, summed_length, seen = 0, 0, set()
    for r in records:
        i = int(tax.depth[r["lower"]])
        support = support_at(r, tax, i) & U16
        summed_support = (summed_support + support) & U16
        if r["qid"] not in seen:
            seen.add(r["qid"]); summed_length += r["qlen"]
        path = []
        n = r["lower"]
        while True:
            path.insert(0, n)
            if n == 0:
                break
            n = int(tax.parent[n])
        direct = [0] * (i + 1); total = [0] * (i + 1)
        direct[i] = total[i] = support
        for k in range(i - 1, -1, -1):
            direct[k] = support_at(r, tax, k) & U16
            total[k] = max(total[k + 1], direct[k])
        tl.append(dict(path=path, pos=0, direct=direct, total=total, ended=False))
    thresh = max(int(np.float32(majority) * np.float32(summed_support)) & U16, min_support & U16)

    def cur(t):
        return t["path"][t["pos"]]

    def end_state():
        for t in tl:
            t["ended"] = int(tax.depth[cur(t)]) == len(t["direct"]) - 1

    def get_support():
        d = s = 0
        for t in tl:
            k = int(tax.depth[cur(t)])
            d = (d + t["direct"][k]) & U16; s = (s + t["total"][k]) & U16
        return d, s

    def reduce_to_majority():
        if len(tl) < 2:
            return False
        sup, max_node, max_sup = {}, None, np.float32(0)
        for t in tl:
            node = cur(t)
            sup[node] = np.float32(sup.get(node, np.float32(0)) + np.float32(t["total"][int(tax.depth[node])]))
            if sup[node] > max_sup:
                max_sup, max_node = sup[node], node
        if len(sup) == 1:
            return False
        tl[:] = [t for t in tl if cur(t) == max_node]
        return True

    end_state()
    d, s = get_support()
    path, lower_idx, running = [], -1, 0
    while tl:
        node = cur(tl[0])
        if d >= thresh:
            lower_idx = running
        path.append([node, d, s, False])
        tl[:] = [t for t in tl if not t["ended"]]
        for t in tl:
            t["pos"] += 1
        running += 1
        end_state()
        path[-1][3] = reduce_to_majority()
        d, s = get_support()
    if lower_idx >= 0:
        upper_node, upper_support = path[lower_idx][0], path[lower_idx][2]
        for j in range(lower_idx, -1, -1):
            if path[j][1] >= thresh:
                upper_support, upper_node = path[j][2], path[j][0]
                if path[j][3]:
                    break
        return upper_node, upper_support, summed_length
    for i in range(len(path) - 1, -1, -1):
        if path[i][2] >= thresh:
            return path[i][0], path[i][2], summed_length
    return path[0][0], path[0][2], summed_length


def naive_binner(lines, tax, regex="(.+)", majority=0.7, min_support=50, sample_min="0", identity=None, ranks=None):
    groups, old = {}, None
    for line in lines:
        if not line or line[0] == "#":
            continue
        rec = parse_gff3(line, tax)
        m = re.fullmatch(regex, rec["qid"])
        name = m.group(1)
        if name != old:
            groups.setdefault(name, []); old = name
        groups[name].append(rec)
    # STEP 1 (core/binner.cpp:207-283): support of every taxon over the sample, pruning of weak range ends
    support, min_found = {0: 0}, 1 << 32
    for recs in groups.values():
        for r in recs:
            n = r["lower"]
            tot = support_at(r, tax, int(tax.depth[n]))
            min_found = min(min_found, tot)
            support[n] = support.get(n, 0) + tot
            if n != 0:
                n = int(tax.parent[n])
                while n != 0:
                    tot = max(tot, support_at(r, tax, int(tax.depth[n])))
                    support[n] = support.get(n, 0) + tot
                    n = int(tax.parent[n])
                tot = max(tot, support_at(r, tax, 0))
                support[0] += tot
    min_sample = int(np.float32(support[0]) * np.float32(float(sample_min))) if "." in sample_min else int(sample_min)
    if min_found < min_sample:
        for name in groups:
            kept = []
            for r in groups[name]:
                n = r["lower"]
                while n != r["upper"] and support.get(n, 0) < min_sample:
                    n = int(tax.parent[n])
                if n == r["upper"] and support.get(n, 0) < min_sample:
                    continue
                if n != r["lower"]:
                    r["ts"] = r["ts"][: int(tax.depth[n]) - int(tax.depth[r["upper"]]) + 1]
                    r["lower"] = n
                kept.append(r)
            groups[name] = kept
    out = []
    for name, recs in groups.items():
        if not recs:
            continue
        if len(recs) > 1:
            node, usup, length = combine(recs, tax, majority, min_support)
            final = dict(upper=node, ts=[usup], qlen=length)   # only the upper node's support is ever printed
        else:
            final = recs[0]
        node, length = final["upper"], final["qlen"]
        if node != 0 and identity:
            rank_pid = np.float32(support_at(final, tax, int(tax.depth[node])) / float(length))
            chain, n = [], node
            while True:
                chain.insert(0, n)
                if n == 0:
                    break
                n = int(tax.parent[n])
            min_pid, predict = np.float32(0), 0
            for n in chain[1:]:
                if ranks[n] in identity:
                    min_pid = max(min_pid, np.float32(identity[ranks[n]]))
                if rank_pid < min_pid:
                    break
                predict = n
            node = predict
        out.append("%s\t%s\t%d\t%d" % (name, tax.taxid[node], support_at(final, tax, int(tax.depth[node])), length))
    return out


def run_binner(binner_bin, tax_dir, text, extra=(), cwd=None):
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=tax_dir)
    r = subprocess.run([binner_bin, "-n", "sample1", "-l", os.path.join(cwd or "/tmp", "binning.log")] + list(extra), input=text,
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    head = [l for l in r.stdout.split("\n") if l.startswith(("#", "@")) or not l]
    body = [l for l in r.stdout.split("\n") if l and not l.startswith(("#", "@"))]
    return head, body, r.stderr


def test_header_and_single_segments(binner_bin, golden_dir, tmp_path):
    folder = os.path.join(golden_dir, "survey_dna")
    tax = R.FlatTaxonomy(os.path.join(folder, "tax"))
    text = open(os.path.join(folder, "out.gff3")).read()
    head, body, err = run_binner(binner_bin, os.path.join(folder, "tax"), text, cwd=str(tmp_path))
    joined = "\n".join(head)
    assert "@Version:0.9.1" in joined and "@SampleID:sample1" in joined and "@_TaxatorTK_Version:" in joined  # core/src/bioboxes.cpp:36-58
    assert "@@SequenceID\tTaxID\t_TaxatorTK_Support\t_TaxatorTK_Length" in joined
    assert "Analyzing sample composition" in err and "Consensus taxonomy assignment" in err
    exp = naive_binner(text.split("\n"), tax)
    assert sorted(body) == sorted(exp) and len(body) > 100


def _segments(rng, tax, n_groups):
    """GFF3 lines of sequences cut into 1..6 segments whose ranges mostly lie on the path of one species per sequence."""
    leaves = [i for i in range(len(tax.taxid)) if tax.depth[i] == tax.depth.max()]
    lines = []
    for g in range(n_groups):
        true = int(rng.choice(leaves))
        nseg = int(rng.integers(1, 7))
        qlen = int(rng.integers(300, 70000 if g % 7 == 0 else 3000))
        for sgm in range(nseg):
            node = true if rng.random() < 0.75 else int(rng.choice(leaves))
            up_steps = int(rng.integers(0, 4))
            lower = node
            for _ in range(int(rng.integers(0, 3))):
                lower = int(tax.parent[lower])
            upper = lower
            for _ in range(up_steps):
                upper = int(tax.parent[upper])
            support = int(rng.integers(20, min(qlen, 69000)))
            begin = int(rng.integers(1, 100)); end = begin + int(rng.integers(support, support + 50))
            chain, n = [], lower
            while n != upper:
                chain.append(n); n = int(tax.parent[n])
            if chain:
                s2 = int(support * rng.uniform(0.5, 1.0))
                t = "%s:%d-" % (tax.taxid[chain[0]], support) + "".join("%s-" % tax.taxid[c] for c in chain[1:-1])
                if len(chain) > 1:
                    t += "%s:%d-" % (tax.taxid[chain[-1]], s2)
                t += tax.taxid[upper]
            else:
                t = "%s:%d" % (tax.taxid[upper], support)
            qid = "contig%d_%d" % (g, sgm)
            lines.append("%s\ttaxator-tk\tsequence_feature\t%d\t%d\t0\t.\t.\tseqlen=%d;tax=%s;rtax=%s" % (qid, begin, end, qlen, t, tax.taxid[lower]))
    return lines


@pytest.mark.parametrize("extra,kw", [([], {}), (["-j", "0.5", "-s", "10"], dict(majority=0.5, min_support=10)),
                                      (["-g", "(contig[0-9]+)_.*", "-s", "200"], dict(regex="(contig[0-9]+)_.*", min_support=200)),
                                      (["-g", "(contig[0-9]+)_.*", "-j", "0.9", "-s", "70000"], dict(regex="(contig[0-9]+)_.*", majority=0.9, min_support=70000))])
def test_consensus_matches_literal_restatement(binner_bin, golden_dir, tmp_path, extra, kw):
    folder = os.path.join(golden_dir, "survey_dna")
    tax = R.FlatTaxonomy(os.path.join(folder, "tax"))
    rng = np.random.default_rng(9)
    lines = _segments(rng, tax, 400)
    head, body, _ = run_binner(binner_bin, os.path.join(folder, "tax"), "\n".join(["##gff-version 3"] + lines) + "\n", extra, cwd=str(tmp_path))
    exp = naive_binner(lines, tax, **kw)
    assert sorted(body) == sorted(exp)
    if "regex" in kw:
        assert len(body) == 400 and len({l.split("\t")[1] for l in body}) > 20
        log = open(os.path.join(str(tmp_path), "binning.log")).read()
        assert "using direct binning mode..." in log or "using fallback binning mode..." in log


def test_hand_worked_cases(binner_bin, golden_dir, tmp_path):
    """Three segments of one contig on the lineage root > A > B > {C1, C2}: supports 300 and 200 at C1, 100 at C2.
    Threshold = max(0.7 * 600, 50) = 420.  Walking down: root (direct 0), A (0), B (0) keep every path; below B the
    majority branch C1 (total 500) wins, its direct support 500 >= 420 -> direct mode, node C1, support 500."""
    folder = os.path.join(golden_dir, "survey_dna")
    tax = R.FlatTaxonomy(os.path.join(folder, "tax"))
    leaves = [i for i in range(len(tax.taxid)) if tax.depth[i] == tax.depth.max()]
    c1 = leaves[0]
    c2 = next(l for l in leaves[1:] if tax.parent[l] == tax.parent[c1])
    other = next(l for l in leaves if tax.parent[tax.parent[l]] != tax.parent[tax.parent[c1]])
    def gff(qid, node, support):
        return "%s\ttaxator-tk\tsequence_feature\t1\t%d\t0\t.\t.\tseqlen=1000;tax=%s:%d;rtax=%s" % (qid, support, tax.taxid[node], support, tax.taxid[node])
    text = "\n".join([gff("x_1", c1, 300), gff("x_2", c1, 200), gff("x_3", c2, 100),   # -> C1, 500
                      gff("y_1", c1, 100), gff("y_2", c2, 100), gff("y_3", other, 100),  # no branch reaches 70 % -> falls back above
                      gff("z_1", c2, 77)]) + "\n"                                        # single segment passes through
    head, body, _ = run_binner(binner_bin, os.path.join(folder, "tax"), text, ["-g", "([a-z])_.*"], cwd=str(tmp_path))
    got = {l.split("\t")[0]: l.split("\t") for l in body}
    assert got["x"][1] == tax.taxid[c1] and got["x"][2] == "500" and got["x"][3] == "3000"
    assert got["z"][1] == tax.taxid[c2] and got["z"][2] == "77" and got["z"][3] == "1000"
    # y: threshold max(0.7*300, 50) = 210; the three paths only share ancestors whose total support is 300 >= 210
    lca = c1
    while not (tax.left[lca] <= tax.left[other] and tax.right[lca] >= tax.right[other]):
        lca = int(tax.parent[lca])
    assert got["y"][1] == tax.taxid[lca] and got["y"][2] == "300"


def test_cli_surface(binner_bin):
    r = subprocess.run([binner_bin, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--sample-identifier", "--glob-identifier", "--sequence-min-support", "--signal-majority", "--identity-constrain",
                 "--files", "--logfile"):  # core/binner.cpp:67-80
        assert flag in r.stdout
    r = subprocess.run([binner_bin], capture_output=True, text=True)
    assert r.returncode == 1 and "--sample-identifier" in r.stderr


def _ranks(folder, tax):
    rank_of = {}
    for line in open(os.path.join(folder, "tax", "nodes.dmp")):
        f = line.rstrip("\n").split("\t|\t")
        rank_of[f[0]] = f[2]
    return [rank_of[t] for t in tax.taxid]


@pytest.mark.parametrize("extra,kw", [(["-m", "4000"], dict(sample_min="4000")), (["-m", "0.02", "-g", "(contig[0-9]+)_.*"], dict(sample_min="0.02", regex="(contig[0-9]+)_.*")),
                                      (["-i", "species:0.8", "-i", "genus:0.5", "-g", "(contig[0-9]+)_.*"],
                                       dict(identity={"species": 0.8, "genus": 0.5}, regex="(contig[0-9]+)_.*")),
                                      (["-i", "family:0.3"], dict(identity={"family": 0.3}))])
def test_noise_removal_and_identity_constraints(binner_bin, golden_dir, tmp_path, extra, kw):
    folder = os.path.join(golden_dir, "survey_dna")
    tax = R.FlatTaxonomy(os.path.join(folder, "tax"))
    rng = np.random.default_rng(10)
    lines = _segments(rng, tax, 300)
    head, body, err = run_binner(binner_bin, os.path.join(folder, "tax"), "\n".join(lines) + "\n", extra, cwd=str(tmp_path))
    exp = naive_binner(lines, tax, ranks=_ranks(folder, tax), **kw)
    assert sorted(body) == sorted(exp)
    if "sample_min" in kw:
        assert " taxa removed" in err and not "Noise removal: 0 taxa removed" in err

