"""The host loader's taxonomy pruning (taxator-tk_amd/host/taxonomy.cpp), which decides every node index the engine ever sees on
real data: `-r` marks ranks, `-d 1` (default) deletes what is not marked -- children of a deleted node move to its parent,
depths shrink, a deleted taxid can no longer be looked up (core/taxator.cpp:324-340, core/src/taxontree.cpp:55-70,169-178,
core/src/taxonomyinterface.cpp:8-12).  Held against the literal Python reading of those lines (tests/helpers/refio.FlatTaxonomy)
on dumps with "no rank" / sub-rank nodes inside the edges, as NCBI's have.  CPU only (`parsecheck --taxonomy`, no engine)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import refio as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "parsecheck")


@pytest.fixture(scope="module")
def parsecheck():
    if not os.path.exists(BIN):
        graft.build()
    return BIN


def _dump(tmp_path, seed, branching, extra):
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    tax = synth.MockTaxonomy(branching)
    st = synth.StretchedTaxonomy(tax, extra)
    folder = str(tmp_path / ("tax%d" % seed))
    st.write_ncbi_dump(folder)
    return st, folder


def _load(parsecheck, folder, delete, ranks=(), lookups=()):
    cmd = [parsecheck, "--taxonomy", "1" if delete else "0"] + list(ranks) + (["--"] + list(lookups) if lookups else [])
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TAXATORTK_TAXONOMY_NCBI=folder), timeout=120)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    nodes = [l.split("\t") for l in r.stdout.splitlines() if "\t" in l]
    found = {l.split()[1]: l.split()[0] for l in r.stdout.splitlines() if l.startswith(("NODE ", "MISSING "))}
    return nodes, found


@pytest.mark.parametrize("ranks,delete", [((), True), (("species", "genus", "family"), True), (("genus",), True), ((), False),
                                          (("species",), False), (("no rank",), True), (("species", "nonsense"), True)])
def test_pruned_taxonomy_equals_the_literal_reading(parsecheck, tmp_path, ranks, delete):
    rng = np.random.default_rng(5)
    extra_of = rng.integers(0, 4, 100000)
    st, folder = _dump(tmp_path, 1, 3, lambda i: int(extra_of[i % len(extra_of)]))
    assert st.rank.count("no rank") > len(st.rank) // 3
    deleted_taxid = next(t for t, rk in zip(st.taxid[1:], st.rank[1:]) if rk == "no rank")
    kept_taxid = next(t for t, rk in zip(st.taxid[1:], st.rank[1:]) if rk == "genus")
    nodes, found = _load(parsecheck, folder, delete, ranks, lookups=[deleted_taxid, kept_taxid, "1", "424242424"])
    ref = R.FlatTaxonomy(folder, ranks=ranks or R.DEFAULT_RANKS, prune=delete)
    # the same nodes under the same parents at the same depths (sibling order, hence the numbering, is each implementation's own)
    got = {n[0]: (n[1], int(n[2])) for n in nodes}
    exp = {t: (ref.taxid[int(ref.parent[k])], int(ref.depth[k])) for k, t in enumerate(ref.taxid)}
    assert got == exp and len(nodes) == len(ref.taxid)
    assert nodes[0][0] == "1" and nodes[0][1] == "1" and int(nodes[0][2]) == 0             # the root stays whatever its rank is (taxontree.cpp:56)
    # nested sets: a child's interval inside its parent's, siblings disjoint -- what isParentOf / getLCA read
    left = {n[0]: int(n[3]) for n in nodes}
    right = {n[0]: int(n[4]) for n in nodes}
    for t, (p_, d) in got.items():
        assert left[t] < right[t]
        if t != "1":
            assert left[p_] < left[t] and right[t] < right[p_] and d == got[p_][1] + 1
    assert sorted(list(left.values()) + list(right.values())) == list(range(1, 2 * len(nodes) + 1))
    keep = set(ranks or R.DEFAULT_RANKS)
    if delete:
        assert all(n[5] in keep for n in nodes[1:])
        assert (found[deleted_taxid] == "MISSING") == ("no rank" not in keep)                 # quirk 10: a mapping onto it aborts the run
        assert (found[kept_taxid] == "NODE") == ("genus" in keep)
    else:
        assert len(nodes) == len(st.taxid) and found[deleted_taxid] == "NODE" and found[kept_taxid] == "NODE"
    assert found["1"] == "NODE" and found["424242424"] == "MISSING"
    depth_max = max(d for _, d in got.values())
    if delete and not ranks:
        assert depth_max == 7
    if delete and ranks == ("genus",):
        assert depth_max == 1
