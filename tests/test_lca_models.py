"""The reference's alignment-free models (taxator -a dummy | simple-lca | megan-lca | n-best-lca).

CPU: the C oracle (oracle/lca_oracle.c) against a second, deliberately naive restatement in Python that follows the
reference's data structures literally (std::list of records with filterOut flags, std::set of nodes, std::multimap) --
two independent readings of core/src/taxonpredictionmodel.hh:57-253 and core/src/alignmentsfilter.hh must agree.
PARITY UNPINNED: the reference holds no test or golden output for these models and cannot be built here.
GPU: the device kernel through the C ABI (rpa_lca_predict_batch) against the oracle, bit for bit.
"""
import importlib

import numpy as np
import pytest

from helpers import oracle as O


class Tree:
    """Random rooted tree with nested-set numbers, node 0 = root, depth <= 8."""
    def __init__(self, rng, n_nodes=200, max_depth=8):
        parent = np.zeros(n_nodes, np.uint32)
        depth = np.zeros(n_nodes, np.uint8)
        for i in range(1, n_nodes):
            while True:
                # (deep trees: every other node hangs under one of the most recent nodes, so that long chains form)
                p = int(rng.integers(max(0, i - 3), i)) if (max_depth > 8 and i % 2) else int(rng.integers(0, i))
                if depth[p] < max_depth:
                    break
            parent[i], depth[i] = p, depth[p] + 1
        kids = [[] for _ in range(n_nodes)]
        for i in range(1, n_nodes):
            kids[parent[i]].append(i)
        left = np.zeros(n_nodes, np.uint32); right = np.zeros(n_nodes, np.uint32)
        ctr = 0
        stack = [(0, 0)]
        while stack:
            v, k = stack.pop()
            if k == 0:
                ctr += 1; left[v] = ctr
            if k < len(kids[v]):
                stack.append((v, k + 1)); stack.append((kids[v][k], 0))
            else:
                ctr += 1; right[v] = ctr
        self.parent, self.depth, self.left, self.right = parent, depth, left, right

    def lca(self, a, b):  # TaxonomyInterface::getLCA by walking up (independent of the nested-set formulation)
        while a != b:
            if self.depth[a] >= self.depth[b]:
                a = int(self.parent[a])
            else:
                b = int(self.parent[b])
        return a


def random_batch(rng, tree, n_sets, max_hits, score_levels=6):
    sizes = rng.integers(0, max_hits + 1, n_sets)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    H = int(offs[-1])
    recs = np.zeros(H, O.RECORD_DTYPE)
    a = rng.integers(1, 400, H); b = a + rng.integers(0, 300, H)
    recs["qstart"], recs["qstop"] = a, b
    recs["score"] = rng.integers(-1, score_levels, H).astype(np.float32) * 12.5   # ties, a negative level and zero
    recs["node"] = rng.integers(0, len(tree.parent), H)
    recs["filtered"] = rng.random(H) < 0.15
    evalue = 10.0 ** rng.integers(-30, 6, H).astype(np.float64)
    qlen = rng.integers(400, 900, n_sets).astype(np.uint32)
    return offs, qlen, recs, evalue


def naive_models(tree, model, offs, qlen, recs, evalue, uncl, toppercent, minscore, maxevalue, minsupport, nbest, ignore_unclassified):
    """Literal reading of the reference classes; returns rows as tuples (lower, upper, rtax, support, begin, end)."""
    rows = []
    f32 = np.float32
    for s in range(len(offs) - 1):
        R = [dict(qstart=int(r["qstart"]), qstop=int(r["qstop"]), score=f32(r["score"]), node=int(r["node"]), masked=bool(r["filtered"]),
                  evalue=float(evalue[offs[s] + k])) for k, r in enumerate(recs[offs[s]:offs[s + 1]])]
        unclassified = (0, 0, 0, 0, 1, int(qlen[s]))

        def simple():
            live = [r for r in R if not r["masked"]]
            if not live:
                return unclassified
            first = live[0]
            qs, qe = first["qstart"], first["qstop"]
            if qs > qe:
                qs, qe = qe, qs
            maxscore = first["score"]
            refnodes = {first["node"]}
            for r in live[1:]:
                if r["qstart"] <= r["qstop"]:
                    qs, qe = min(r["qstart"], qs), max(r["qstop"], qe)
                else:
                    qs, qe = min(r["qstop"], qs), max(r["qstart"], qe)
                if r["score"] > maxscore:
                    maxscore = r["score"]
                refnodes.add(r["node"])
            best = {r["node"] for r in R if not r["masked"] and r["score"] == maxscore}
            def fold(nodes):
                it = iter(sorted(nodes))
                acc = next(it)
                for x in it:
                    acc = tree.lca(acc, x)
                return acc
            node = fold(refnodes)
            rtax = fold(best) if len(refnodes) != len(best) else node
            return (node, node, rtax, qe - qs + 1, qs, qe)

        if model == O.LCA_DUMMY:
            rows.append(unclassified)
        elif model == O.LCA_SIMPLE:
            rows.append(simple())
        elif model == O.LCA_MEGAN:
            ms, me, tp = f32(minscore), f32(maxevalue), f32(toppercent)
            mx, support = f32(0.0), 0
            for r in R:
                if not r["masked"]:
                    if r["score"] < ms or r["evalue"] > float(me):
                        r["masked"] = True
                    elif r["score"] > mx:
                        mx = r["score"]; support += 1
            mx = f32((1.0 - float(tp)) * float(mx))
            for r in R:
                if r["score"] < mx:
                    r["masked"] = True
            if ignore_unclassified:
                for r in R:
                    if uncl[r["node"]]:
                        r["masked"] = True
            rows.append(simple() if support >= minsupport else unclassified)
        else:
            order = sorted([k for k, r in enumerate(R) if not r["masked"]], key=lambda k: (-float(R[k]["score"]), k))
            if order:
                count = nbest % (1 << 32)
                last = R[order[0]]["score"]
                j = 1
                while j < len(order):
                    if R[order[j]]["score"] != last:
                        count = (count - 1) % (1 << 32)
                        if count <= 0:
                            break
                        last = R[order[j]]["score"]
                    j += 1
                for k in order[j:]:
                    R[k]["masked"] = True
            rows.append(simple())
    return rows


CASES = [dict(model=O.LCA_DUMMY), dict(model=O.LCA_SIMPLE),
         dict(model=O.LCA_MEGAN), dict(model=O.LCA_MEGAN, toppercent=0.3, minscore=12.5, maxevalue=1e-3, minsupport=2),
         dict(model=O.LCA_MEGAN, toppercent=1.0, ignore_unclassified=True), dict(model=O.LCA_MEGAN, toppercent=0.0, minsupport=2, maxevalue=1e-10), dict(model=O.LCA_MEGAN, maxevalue=1e-40),
         dict(model=O.LCA_NBEST), dict(model=O.LCA_NBEST, nbest=2), dict(model=O.LCA_NBEST, nbest=5), dict(model=O.LCA_NBEST, nbest=0)]


def _full(case):
    d = dict(toppercent=0.05, minscore=0.0, maxevalue=1000.0, minsupport=1, nbest=1, ignore_unclassified=False)
    d.update(case)
    return d


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_oracle_matches_literal_restatement(case):
    rng = np.random.default_rng(3)
    tree = Tree(rng)
    uncl = (rng.random(len(tree.parent)) < 0.2).astype(np.uint8)
    offs, qlen, recs, evalue = random_batch(rng, tree, 600, 14)
    c = _full(case)
    tax = O.TaxArrays(tree.parent, tree.left, tree.right, tree.depth)
    rc, rows = O.lca_predict_batch(tax, c["model"], offs, qlen, recs, evalue, uncl, c["toppercent"], c["minscore"], c["maxevalue"],
                                   c["minsupport"], c["nbest"], c["ignore_unclassified"])
    assert rc == 0
    exp = naive_models(tree, c["model"], offs, qlen, recs, evalue, uncl, c["toppercent"], c["minscore"], c["maxevalue"], c["minsupport"],
                       c["nbest"], c["ignore_unclassified"])
    got = [tuple(int(rows[f][s]) for f in ("lower", "upper", "rtax", "support", "begin", "end")) for s in range(len(exp))]
    assert got == exp
    assert (rows["ival"] == -1.0).all() and (rows["flags"] == O.FLAG_IVAL_UNTOUCHED).all()
    if c["model"] != O.LCA_DUMMY:
        assert len({g[0] for g in got}) > 5 or c["maxevalue"] < 1e-30  # the cases do classify (1e-40 narrows to float 0: nothing passes)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_device_models_match_oracle(case):
    import __graft_entry__ as graft
    pkg = graft.load_package()
    rng = np.random.default_rng(5)
    tree = Tree(rng, n_nodes=3000)
    uncl = (rng.random(len(tree.parent)) < 0.2).astype(np.uint8)
    offs, qlen, recs, evalue = random_batch(rng, tree, 20000, 40)
    c = _full(case)
    tax = O.TaxArrays(tree.parent, tree.left, tree.right, tree.depth)
    rc, exp = O.lca_predict_batch(tax, c["model"], offs, qlen, recs, evalue, uncl, c["toppercent"], c["minscore"], c["maxevalue"],
                                  c["minsupport"], c["nbest"], c["ignore_unclassified"])
    assert rc == 0
    model = pkg.RPAPredictionModel(tree, alphabet=pkg.DNA)
    batch = dict(set_offsets=offs, query_len=qlen, filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "score", "node"):
        batch[f] = recs[f].copy()
    rows = model.lca_predict(batch, c["model"], evalue=evalue, node_unclassified=uncl, toppercent=c["toppercent"], minscore=c["minscore"],
                             maxevalue=c["maxevalue"], minsupport=c["minsupport"], nbest=c["nbest"], ignore_unclassified=c["ignore_unclassified"])
    assert rows.tobytes() == exp.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [dict(model=O.LCA_SIMPLE), dict(model=O.LCA_MEGAN, toppercent=0.3, minsupport=2), dict(model=O.LCA_NBEST, nbest=3)],
                         ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_device_models_on_a_deep_tree(case):
    """The alignment-free models on a tree of up to 60 levels (taxator -d 0 on the NCBI tree: ~40): the kernel climbs from
    the running LCA, any depth (the reference: core/src/taxonomyinterface.cpp:67-77)."""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    rng = np.random.default_rng(7)
    tree = Tree(rng, n_nodes=4000, max_depth=60)
    assert int(tree.depth.max()) > 15
    uncl = (rng.random(len(tree.parent)) < 0.2).astype(np.uint8)
    offs, qlen, recs, evalue = random_batch(rng, tree, 15000, 30)
    c = _full(case)
    tax = O.TaxArrays(tree.parent, tree.left, tree.right, tree.depth)
    rc, exp = O.lca_predict_batch(tax, c["model"], offs, qlen, recs, evalue, uncl, c["toppercent"], c["minscore"], c["maxevalue"],
                                  c["minsupport"], c["nbest"], c["ignore_unclassified"])
    assert rc == 0
    model = pkg.RPAPredictionModel(tree, alphabet=pkg.DNA)
    batch = dict(set_offsets=offs, query_len=qlen, filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "score", "node"):
        batch[f] = recs[f].copy()
    rows = model.lca_predict(batch, c["model"], evalue=evalue, node_unclassified=uncl, toppercent=c["toppercent"], minscore=c["minscore"],
                             maxevalue=c["maxevalue"], minsupport=c["minsupport"], nbest=c["nbest"], ignore_unclassified=c["ignore_unclassified"])
    assert rows.tobytes() == exp.tobytes()


@pytest.mark.gpu
def test_device_resident_batch():
    """rpa_lca_predict_batch with every array already in HBM (RPA_MEM_DEVICE), rows written to a device tensor."""
    import ctypes as C
    import torch
    import __graft_entry__ as graft
    pkg = graft.load_package()
    rng = np.random.default_rng(6)
    tree = Tree(rng, n_nodes=1000)
    uncl = (rng.random(len(tree.parent)) < 0.2).astype(np.uint8)
    offs, qlen, recs, evalue = random_batch(rng, tree, 5000, 30)
    tax = O.TaxArrays(tree.parent, tree.left, tree.right, tree.depth)
    rc, exp = O.lca_predict_batch(tax, O.LCA_MEGAN, offs, qlen, recs, evalue, uncl, toppercent=0.2, minsupport=2, ignore_unclassified=True)
    assert rc == 0
    model = pkg.RPAPredictionModel(tree, alphabet=pkg.DNA)
    dev = dict(set_offsets=torch.from_numpy(offs.view(np.int32)).cuda(), query_len=torch.from_numpy(qlen.view(np.int32)).cuda(),
               qstart=torch.from_numpy(recs["qstart"].copy().view(np.int32)).cuda(), qstop=torch.from_numpy(recs["qstop"].copy().view(np.int32)).cuda(),
               score=torch.from_numpy(recs["score"].copy()).cuda(), node=torch.from_numpy(recs["node"].copy().view(np.int32)).cuda(),
               filtered=torch.from_numpy(recs["filtered"].astype(np.uint8)).cuda())
    ev = torch.from_numpy(evalue).cuda()
    un = torch.from_numpy(uncl).cuda()
    out = torch.zeros((len(qlen), 8), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # the engine runs on its own stream
    L = model._L
    ptr = lambda t: C.c_void_p(t.data_ptr())
    b = pkg._Batch(len(qlen), len(recs), dev["set_offsets"].data_ptr(), None, dev["query_len"].data_ptr(), dev["qstart"].data_ptr(),
                   dev["qstop"].data_ptr(), None, None, dev["score"].data_ptr(), None, None, None, dev["node"].data_ptr(),
                   dev["filtered"].data_ptr(), pkg.MEM_DEVICE)
    p = pkg._LcaParams()
    L.rpa_lca_default_params(C.byref(p))
    p.model, p.toppercent, p.minsupport, p.ignore_unclassified = pkg.LCA_MEGAN, 0.2, 2, 1
    rc = L.rpa_lca_predict_batch(model._h, C.byref(b), ptr(ev), ptr(un), C.byref(p), ptr(out), pkg.MEM_DEVICE)
    assert rc == 0, L.rpa_last_error(model._h)
    assert out.cpu().numpy().tobytes() == exp.tobytes()
