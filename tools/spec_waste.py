#!/usr/bin/env python3
"""tools/spec_waste.py [c2|c4] [N_SETS] -- where do the speculative alignments go?  (CPU only; test infrastructure: uses oracle/.)

Replays the reference's control flow (oracle/rpa_oracle.c:514-727, the three passes of RPAPredictionModel::predict) in Python
on a sample of a BASELINE configuration, taking every distance from the oracle's own trace, checks that the replay demands
exactly the traced alignments, and then plays the engine's request policies over the same loops: how many alignments a policy
asks for (computed), how many of them the loops consume (demanded), how many request rounds a record set needs.

Policies (pass-1 chunk S1, pass-2 chunk S2; 0 = the whole loop at once):
  plain        rows [lo, hi) of the loop that can need an alignment at all (RPA_PASS1_FILTER=0)
  filtered     the same minus the rows that the thresholds known when the chunk is made already exclude (=1)
  first        filtered + the chunk's first hit is taken for the loop's next outgroup hit: its thresholds apply to the hits
               behind it; first-score / first-pid: only its score cut-off / only its identity thresholds.  first-score is
               what the engine does (=2, chunk_end_by_score: hits are in score order, the cut-off is a chunk end)
  cautious     every hit already in the chunk is taken for an outgroup hit (pass 1) / to lift unode_global (pass 2)
  pass 2 otherwise applies the thresholds of its anchor and the unode test at the time the chunk is made (mode 3);
  +p2firstK: an anchor's first pass-2 chunk holds K hits (RPA_SPEC2_FIRST); +p2groupK: at most K hits per chunk that would
  lift unode_global to the same node (or below one already asked for K times) -- both cut pass 2's waste, both cost rounds.
Extra arguments key=value override the dataset (level_rate=0.02 query_rate=0.1 variety=1 ...).
"""
import importlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle as O  # noqa: E402

CONFIGS = {
    "c2": dict(hits=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, seed=2),
    "c4": dict(hits=20, seg_len=67, alphabet=1, genome_len=1500, level_rate=0.06, query_rate=0.08, seed=2),
    "c4_200": dict(hits=20, seg_len=200, alphabet=1, genome_len=1500, level_rate=0.06, query_rate=0.08, seed=2),
}
LIFT_STATS = {}
P2_FIRST = 0  # pass 2: size of an anchor's first chunk (0: S2)
P2_GROUP = 0  # pass 2: hits per chunk and lift target (0: no limit)
P2_DEFER = 0  # pass 2: hits that an earlier hit of their chunk may take out of the loop wait for a second launch of the same round
F32 = np.float32
FLT_MAX = F32(np.finfo(np.float32).max)


class Loop:
    """One anchor's loop of pass 1 or pass 2: the rows in visit order with what happened to each."""
    def __init__(self, kind, anchor):
        self.kind, self.anchor = kind, anchor
        self.rows = []      # (i, what, state) what in {"align", "free", "skip"}; state = thresholds BEFORE the row was looked at
        self.end = None     # first row index the loop did not look at (n if it ran out of rows)
        self.extra = False  # pass 2: the anchor <-> query alignment happened inside this loop
        self.lift = {}      # pass 2: row -> True where the row's alignment lifted unode_global


def replay(tax, recs, qrlength, D, X=F32(0.5), reeval=0.05):
    """recs: the set's active records in the reference's order.  D: {(pass, i, anchor): (dist, sim)} from the oracle's trace.
    Returns (demanded keys in order, pass-0 rows, loops)."""
    n = len(recs)
    score = recs["score"].astype(F32)
    ident = recs["identities"].astype(np.int64)
    node = recs["node"].astype(np.int64)
    demanded = []

    def get(p, i, j):
        demanded.append((p, i, j))
        return D[(p, i, j)]

    bw = F32(F32(1.0 - reeval) * score[0])
    qd = np.full(n, FLT_MAX, F32)
    qs = np.zeros(n, F32)
    qgroup = np.zeros(n, bool)
    best = 0
    lca_all = int(node[0])
    pass0_rows = []
    for i in range(n):
        if recs["alnlen"][i] == qrlength and ident[i] == qrlength:
            qgroup[i] = True; d = F32(0); s = F32(ident[i])
        elif score[i] >= bw:
            qgroup[i] = True
            d, s = get(0, i, -1)
            d = F32(d); s = F32(s)  # (the traced similarity is already max(sim, identities))
            pass0_rows.append(i)
        else:
            d = FLT_MAX; s = F32(ident[i])
        qd[i] = d; qs[i] = s
        if d < qd[best]:
            best = i
        elif d == qd[best]:
            if s > qs[best]: best = i
            elif s == qs[best] and score[i] > score[best]: best = i
        lca_all = tax.lca(lca_all, int(node[i]))
    rtax = int(node[best])
    for i in range(n):
        if not qgroup[i]:
            continue
        if qd[i] != qd[best] or qs[i] != qs[best] or score[i] != score[best]:
            qgroup[i] = False
        else:
            rtax = tax.lca(rtax, int(node[i]))
    root = 0 if tax.parent[0] == 0 else int(np.nonzero(tax.parent == np.arange(len(tax.parent)))[0][0])
    loops = []
    lnode_g = unode_g = rtax
    bf_max = F32(1.0)
    outgroup = np.zeros(n, bool)
    lrd_min = 255
    qd_after0 = qd.copy()
    while True:
        a = int(np.nonzero(qgroup)[0][0])
        qgroup[a] = False
        qdist = qd[a]
        L = Loop(1, a)
        bf_d = [F32(0)]; bf_n = [int(node[a])]
        lnode = rtax
        ldist = F32(0); udist = FLT_MAX
        og = []
        qpid_upper = 0.0; cut_g = 0.0; cut_h = 0.0; scut = 0
        i = 0
        while lnode != root and i < n and score[i] >= F32(scut):
            state = (max(cut_g, cut_h), scut)
            qsearchpid = float(ident[i]) / qrlength
            qpid = float(qs[i]) / qrlength
            if qpid >= max(cut_g, cut_h):
                if i == a:
                    dist = F32(0); L.rows.append((i, "free", state))
                elif qd[i] == 0:
                    dist = qd[a]; L.rows.append((i, "free", state))
                else:
                    dist = F32(get(1, i, a)[0]); L.rows.append((i, "align", state))
                bf_d.append(dist); bf_n.append(int(node[i]))
                if dist == 0:
                    qgroup[i] = False
                elif dist <= qdist:
                    lnode = tax.lca(lnode, int(node[i]))
                    if dist > ldist: ldist = dist
                else:
                    if dist < udist:
                        udist = dist
                        if qsearchpid > qpid_upper:
                            qpid_upper = qsearchpid
                            cut_g = qsearchpid * 2.0 - 1.0
                            cut_h = qsearchpid * float(X)
                        if not scut: scut = int(F32(score[i] * X))
                    og.append((i, int(dist)))
            else:
                L.rows.append((i, "skip", state))
            i += 1
        L.end = i
        loops.append(L)
        bandfactor = F32(tax.bandfactor(np.array(bf_d, F32), np.array(bf_n, np.uint32)))
        if bandfactor > bf_max: bf_max = bandfactor
        qdist_ex = F32(qdist * bandfactor)
        mud = F32(2147483647.0)
        kept = []
        for (k, dd) in og:
            dist = F32(dd); keep = True
            if dist > qdist_ex:
                if dist > mud: keep = False
                elif dist < mud: mud = dist
            else:
                if mud > qdist_ex: mud = dist
                else: mud = mud if not (mud < dist) else dist
            if keep: kept.append((k, dd))
        unode = lnode
        for (k, dd) in kept:
            dist = F32(dd)
            if dist > mud: continue
            unode = tax.lca(int(node[k]), unode)
            lrd = int(tax.depth[tax.lca(int(node[k]), rtax)])
            if lrd > lrd_min: continue
            elif lrd < lrd_min:
                lrd_min = lrd; outgroup[:] = False
            outgroup[k] = True
        unode_g = tax.lca(unode_g, unode)
        lnode_g = tax.lca(lnode_g, lnode)
        if not (qgroup.any() and lnode_g != root):
            break
    while outgroup.any():
        a = int(np.nonzero(outgroup)[0][0])
        outgroup[a] = False
        if unode_g == lca_all:
            continue
        qpid_a = float(qs[a]) / qrlength
        thr = max(qpid_a * 2.0 - 1.0, qpid_a * float(X))
        fcut = F32(score[a] * X)
        L = Loop(2, a)
        L.thr, L.fcut, L.unode_at_start = thr, fcut, unode_g
        i = 0
        while i < n and score[i] >= fcut:
            qpid = float(qs[i]) / qrlength
            if qpid >= thr:
                if i == a:
                    dist = F32(0); L.rows.append((i, "free", unode_g))
                else:
                    cn = int(node[i])
                    if tax.is_parent_of(unode_g, cn) or cn == unode_g:
                        L.rows.append((i, "skip", unode_g)); i += 1; continue
                    st_unode = unode_g
                    dist = F32(get(2, i, a)[0]); L.rows.append((i, "align", st_unode))
                    qd[i] = dist
                if dist == 0:
                    outgroup[i] = False
                else:
                    if qd[a] == FLT_MAX:
                        d2, s2 = get(2, -1, a)
                        qd[a] = F32(d2); qs[a] = F32(s2); L.extra = True
                    qdist_ex = F32(qd[a] * bf_max)
                    lifted = bool(dist <= qdist_ex)
                    if i != a:
                        di = int(tax.depth[tax.lca(int(node[i]), rtax)]); da = int(tax.depth[tax.lca(rtax, int(node[a]))])
                        LIFT_STATS.setdefault(di - da, [0, 0])[0 if lifted else 1] += 1
                    if lifted:
                        unode_g = tax.lca(unode_g, int(node[i]))
                        L.lift[i] = True
            else:
                L.rows.append((i, "skip", unode_g))
            i += 1
        L.end = i
        loops.append(L)
    return demanded, pass0_rows, loops, qd_after0


def play(tax, recs, loops, qd0, qrlength, S1, S2, mode):
    """Requests of a policy over the recorded loops.  mode: "plain" | "filtered" | "cautious" (pass 1: filtered + every hit
    already in the chunk is taken for the loop's next outgroup hit, i.e. its thresholds are applied to the hits behind it;
    pass 2: every hit already in the chunk is taken to lift unode_global).  A loop pauses at the first alignment it needs
    that its current chunk does not hold.  "first": like cautious in pass 1 but only the chunk's first hit (the one the loop
    paused at, or the first one behind the anchor) is taken for an outgroup hit; pass 2 as filtered.  Returns (computed, demanded, request rounds after pass 0)."""
    n = len(recs)
    score = recs["score"].astype(F32)
    ident = recs["identities"].astype(np.int64)
    node = recs["node"].astype(np.int64)
    X = 0.5
    computed = demanded = rounds = 0
    cat = {"p1 first chunk": [0, 0], "p1 later chunks": [0, 0], "p2 first chunk": [0, 0], "p2 later chunks": [0, 0]}  # asked, used
    for L in loops:
        n_chunk = 0
        emitted = {}
        dropped = set()
        for (i, what, st) in L.rows:
            if what != "align":
                continue
            demanded += 1
            assert i not in dropped, "a dropped hit was consumed"
            if i not in emitted:  # pause: a new chunk from row i
                n_chunk += 1
                key = ("p1 " if L.kind == 1 else "p2 ") + ("first chunk" if n_chunk == 1 else "later chunks")
                lo = i
                S = S1 if L.kind == 1 else (S2 if (n_chunk > 1 or not P2_FIRST) else P2_FIRST)
                hi = n if S == 0 else min(n, lo + S)
                rounds += 1
                if L.kind == 1:
                    thr, scut = st if mode != "plain" else (0.0, 0)
                    for r in range(lo, hi):
                        if r == L.anchor or qd0[r] == 0: continue   # the anchor; kLiteZeroQd
                        if mode != "plain":
                            if score[r] < F32(scut): break
                            if float(recs["_qs"][r]) / qrlength < thr: continue
                        computed += 1; cat[key][0] += 1; emitted[r] = key
                        if mode == "cautious" or (mode in ("first", "first-score", "first-pid") and r == lo):
                            pid = float(ident[r]) / qrlength
                            if mode != "first-score": thr = max(thr, pid * 2.0 - 1.0, pid * X)
                            if not scut and mode != "first-pid": scut = int(F32(score[r] * F32(X)))
                else:
                    un = st
                    groups = {}
                    first_launch = []
                    for r in range(lo, hi):
                        if r == L.anchor: continue
                        if score[r] < L.fcut: break
                        if float(recs["_qs"][r]) / qrlength < L.thr: continue
                        cn = int(node[r])
                        if tax.is_parent_of(un, cn) or cn == un: continue  # unode_global when the chunk is made
                        if P2_GROUP:  # at most P2_GROUP hits per chunk that would lift unode_global to the same node
                            tgt = tax.lca(st, cn)
                            # a hit is also covered by an emitted group whose target is an ancestor-or-self of its own target
                            covered = sum(c for (t, c) in groups.items() if t == tgt or tax.is_parent_of(t, tgt))
                            if covered >= P2_GROUP: continue
                            groups[tgt] = groups.get(tgt, 0) + 1
                        if P2_DEFER:
                            # Two launches inside ONE round: a hit that an earlier hit of the same chunk would take out of the loop by
                            # lifting unode_global (depth of LCA(hit, best) not above the earlier hit's) waits for the first launch's
                            # results and is dropped when such an earlier hit of the first launch did lift
                            cb = int(tax.depth[tax.lca(cn, st)])
                            cover = [j for (j, cj) in first_launch if cj <= cb]
                            if cover:
                                play.deferred += 1
                                if any(L.lift.get(j) for j in cover):
                                    emitted[r] = key; play.dropped += 1   # never consumed: the loop steps over it (asserted below)
                                    dropped.add(r)
                                    continue
                            else:
                                first_launch.append((r, cb))
                        computed += 1; cat[key][0] += 1; emitted[r] = key
                        if mode == "cautious": un = tax.lca(un, cn)
            cat[emitted[i]][1] += 1
        if L.kind == 2 and L.extra:
            computed += 1; demanded += 1   # anchor <-> query range: asked for with the chunk that needs it, always used
    play.cat = cat
    return computed, demanded, rounds


def main():
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_sets = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    cfg = dict(CONFIGS[cfg_name])
    variety = False
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "variety": variety = bool(int(v))
        else: cfg[k] = type(cfg[k])(v)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    ds = synth.make_dataset(n_sets=n_sets, n_hits=cfg["hits"], seg_len=cfg["seg_len"], branching=4, genome_len=cfg["genome_len"], seed=cfg["seed"],
                            alphabet=cfg["alphabet"], level_rate=cfg["level_rate"], query_rate=cfg["query_rate"], variety=variety)
    tax = O.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = O.StoreArrays(synth.encode(ds.query_letters, cfg["alphabet"]), ds.query_offsets)
    db = O.StoreArrays(synth.encode(ds.ref_letters, cfg["alphabet"]), ds.ref_offsets)
    aos = synth.records_aos(ds.batch, O.RECORD_DTYPE)
    so = ds.batch["set_offsets"]
    policies = [("plain", S1, S2) for (S1, S2) in ((0, 10), (4, 10))] + \
               [("filtered", S1, S2) for (S1, S2) in ((0, 10), (4, 10))] + \
               [("cautious", S1, S2) for (S1, S2) in ((0, 10),)] + \
               [("first", S1, S2) for (S1, S2) in ((0, 10), (4, 10), (8, 10))] + [("first-score", 0, 10), ("first-pid", 0, 10)] + \
               [("first-score+p2defer", 0, S2, 1) for S2 in (10, 20, 0)] + [("first-score+p2first", 0, 10, k) for k in (3, 6)] + [("first-score+p2group", 0, S2, k) for S2 in (10, 20) for k in (1, 2, 3)]
    totals = {p: [0, 0, 0] for p in policies}
    cats = {p: {} for p in policies}
    pass0 = 0
    per_pass = [0, 0, 0]
    n_done = 0
    for s in range(n_sets):
        recs = aos[so[s]:so[s + 1]]
        rc, out, st, trace = O.predict(tax, q, db, cfg["alphabet"], int(ds.batch["query_seq"][s]), int(ds.batch["query_len"][s]), recs, want_trace=True)
        assert rc == 0
        act = recs[recs["filtered"] == 0] if "filtered" in recs.dtype.names else recs
        n = len(act)
        per_pass[0] += int(st["pass0"]); per_pass[1] += int(st["pass1"]); per_pass[2] += int(st["pass2"])
        if n < 2 or not trace and int(st["pass0"]) == 0:
            continue
        order = sorted(range(n), key=lambda k: (-float(act["score"][k]), -int(act["identities"][k])))  # stable: ties keep input order
        act = act[order]
        qrstart, qrstop = int(act["qstart"].min()), int(act["qstop"].max())
        qrlength = qrstop - qrstart + 1
        if act["alnlen"][0] == qrlength and act["identities"][0] == qrlength:
            continue  # shortcut: no alignments
        D = {(p, i, j): (d, sm) for (p, i, j, d, sm) in trace}
        demanded, pass0_rows, loops, qd0 = replay(tax, act, qrlength, D)
        assert demanded == [(p, i, j) for (p, i, j, d, sm) in trace], (s, demanded[:6], trace[:6])
        # querysimilarity after pass 0 (what the pass-1 / pass-2 thresholds look at)
        qs = act["identities"].astype(F32).copy()
        for i in pass0_rows: qs[i] = F32(D[(0, i, -1)][1])
        act2 = np.zeros(n, dtype=act.dtype.descr + [("_qs", "<f4")])
        for f in act.dtype.names: act2[f] = act[f]
        act2["_qs"] = qs
        pass0 += len(pass0_rows)
        for p in policies:
            global P2_FIRST, P2_GROUP, P2_DEFER
            P2_DEFER = 1 if "p2defer" in p[0] else 0
            play.deferred = getattr(play, "deferred", 0); play.dropped = getattr(play, "dropped", 0)
            P2_FIRST = p[3] if len(p) > 3 and "p2first" in p[0] else 0
            P2_GROUP = p[3] if len(p) > 3 and "p2group" in p[0] else 0
            c, d, r = play(tax, act2, loops, qd0, qrlength, p[1], p[2], p[0].split("+")[0])
            totals[p][0] += c; totals[p][1] += d; totals[p][2] += r
            for k, (a_, u_) in play.cat.items():
                e = cats[p].setdefault(k, [0, 0]); e[0] += a_; e[1] += u_
        n_done += 1
    print("%s: %d record sets replayed (%d with alignments); demanded per pass: %s" % (cfg_name, n_sets, n_done, per_pass))
    dem_all = sum(per_pass)
    print("pass-2 hits by depth(lca(hit, rtax)) - depth(lca(anchor, rtax)) -> [lifted unode_global, did not]:", dict(sorted(LIFT_STATS.items())))
    for p in policies:
        c, d, r = totals[p]
        print("  %-9s" % (p[0] + (str(p[3]) if len(p) > 3 else "")) + " S1=%d S2=%-2d  computed/demanded %.3f  (pass 1+2: asked %d, used %d = %.0f %%)  request rounds after pass 0 per set %.2f"
              % (p[1], p[2], (pass0 + c) / dem_all, c, d, 100.0 * d / max(1, c), r / max(1, n_done)))
        print("            " + ";  ".join("%s asked %d used %d" % (k, v[0], v[1]) for k, v in cats[p].items()))
    print("p2defer (all three settings together): %d hits waited for a second launch, %d of them were dropped" % (getattr(play, "deferred", 0), getattr(play, "dropped", 0)))


if __name__ == "__main__":
    main()
