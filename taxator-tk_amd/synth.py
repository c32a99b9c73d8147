"""Synthetic RPA workloads -- data plumbing for bench.py, smoke() and the parity tests.

Generates, with fixed seeds, what the reference's `taxator -a rpa` consumes: a mock NCBI-style taxonomy already pruned
to root + 7 ranks, reference genomes evolved down that tree (3 %/level: 80 % substitutions, 10 % insertions, 10 %
deletions), query segments cut from a random species at a random locus with 5 % extra divergence and 1 % N, and for
every query n alignment rows against the true species plus n-1 other species at the homologous locus (20 % of the
species are stored reverse-complemented, so their hits come with rstart > rstop), with
`score = 2*identities - alnlen`, identities = matches along the true (by-construction) alignment, i.e. what a gapped
aligner such as LAST/BLAST would report.  No reference file or code is involved; torch is used for the vectorised
gathers so that the 1 M x 20 configuration can be generated on the GPU in seconds.
"""
import os

import numpy as np
import torch

LETTERS = np.frombuffer(b"ACGTN", np.uint8)
RANKS = ("superkingdom", "phylum", "class", "order", "family", "genus", "species")

_DNA5 = np.full(256, 4, np.uint8)
for _c, _o in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"U", 3)):
    _DNA5[_c[0]] = _o
    _DNA5[_c.lower()[0]] = _o
_AA = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"
_AA27 = np.full(256, 25, np.uint8)
for _o, _c in enumerate(_AA):
    _AA27[ord(_c)] = _o
    _AA27[ord(_c.lower())] = _o


def encode(letters, alphabet):
    """ASCII residue bytes -> alphabet ordinals (same tables the engine applies on the device)."""
    return (_DNA5 if alphabet == 0 else _AA27)[np.asarray(letters, np.uint8)]


class MockTaxonomy:
    """root(taxid 1) -> one superkingdom (taxid 2) -> `branching` children per rank down to species; flattened."""

    def __init__(self, branching):
        self.branching = branching
        parent, depth, left, right, taxid, rank = [0], [0], [0], [0], ["1"], ["no rank"]
        counter = [0]
        next_id = [100]
        self.species = []

        def visit(idx, level):
            counter[0] += 1
            left[idx] = counter[0]
            if level == len(RANKS):
                self.species.append(idx)
            else:
                kids = 1 if level == 0 else branching
                for _ in range(kids):
                    c = len(parent)
                    parent.append(idx)
                    depth.append(level + 1)
                    left.append(0)
                    right.append(0)
                    if level == 0:
                        taxid.append("2")
                    else:
                        next_id[0] += 1
                        taxid.append(str(next_id[0]))
                    rank.append(RANKS[level])
                    visit(c, level + 1)
            counter[0] += 1
            right[idx] = counter[0]

        import sys
        sys.setrecursionlimit(10000)
        visit(0, 0)
        self.parent = np.array(parent, np.uint32)
        self.depth = np.array(depth, np.uint8)
        self.left = np.array(left, np.uint32)
        self.right = np.array(right, np.uint32)
        self.taxid = taxid
        self.rank = rank
        self.species = np.array(self.species, np.uint32)

    def write_ncbi_dump(self, folder):
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "nodes.dmp"), "w") as f:
            for i, t in enumerate(self.taxid):
                f.write("%s\t|\t%s\t|\t%s\t|\tXX\t|\n" % (t, self.taxid[self.parent[i]], self.rank[i]))
        with open(os.path.join(folder, "names.dmp"), "w") as f:
            for i, t in enumerate(self.taxid):
                f.write("%s\t|\t%s\t|\t\t|\tscientific name\t|\n" % (t, "root" if i == 0 else "T" + t))


class StretchedTaxonomy:
    """A taxonomy with the same leaves and the same branching as `tax`, but with chains of extra single-child nodes put
    into its edges, the way the full NCBI tree ("no rank" clades between the named ranks, taxator -d 0) relates to the
    one pruned to seven ranks: the edge above node i gets extra(i) nodes.  `new_index[i]` is where node i went; nodes are
    renumbered parent-before-child with fresh nested-set values.  For the deep-taxonomy tests (up to 63 levels)."""

    def __init__(self, tax, extra):
        parent = np.asarray(tax.parent)
        n = len(parent)
        order = np.argsort(np.asarray(tax.depth), kind="stable")  # parents before children
        new_parent, new_depth, taxid, rank = [0], [0], [tax.taxid[0]], [tax.rank[0]]
        self.new_index = np.zeros(n, np.uint32)
        fresh = [10_000_000]
        for i in order:
            if i == 0:
                continue
            p_ = int(self.new_index[parent[i]])
            for _ in range(int(extra(int(i)))):
                new_parent.append(p_); new_depth.append(new_depth[p_] + 1)
                fresh[0] += 1
                taxid.append(str(fresh[0])); rank.append("no rank")
                p_ = len(new_parent) - 1
            new_parent.append(p_); new_depth.append(new_depth[p_] + 1)
            taxid.append(tax.taxid[i]); rank.append(tax.rank[i])
            self.new_index[i] = len(new_parent) - 1
        m = len(new_parent)
        kids = [[] for _ in range(m)]
        for c in range(1, m):
            kids[new_parent[c]].append(c)
        left, right = np.zeros(m, np.uint32), np.zeros(m, np.uint32)
        counter, stack = 0, [(0, 0)]
        while stack:
            v, k = stack.pop()
            if k == 0:
                counter += 1; left[v] = counter
            if k < len(kids[v]):
                stack.append((v, k + 1)); stack.append((kids[v][k], 0))
            else:
                counter += 1; right[v] = counter
        self.parent = np.array(new_parent, np.uint32)
        self.depth = np.array(new_depth, np.uint8)
        self.left, self.right = left, right
        self.taxid, self.rank = taxid, rank
        self.species = self.new_index[np.asarray(tax.species)]
        self.branching = getattr(tax, "branching", None)

    write_ncbi_dump = MockTaxonomy.write_ncbi_dump


def stretch_dataset(ds, extra):
    """The same workload on a StretchedTaxonomy: ds.tax replaced, the batch's node column renumbered.  In place."""
    st = StretchedTaxonomy(ds.tax, extra)
    ds.batch["node"] = st.new_index[np.asarray(ds.batch["node"])].astype(np.uint32)
    ds.tax = st
    return ds


def _mutate(seq, anc, rate, rng, alphabet_size=4):
    n = len(seq)
    r = rng.random(n)
    sub = r < 0.8 * rate
    dele = (r >= 0.8 * rate) & (r < 0.9 * rate)
    ins = (r >= 0.9 * rate) & (r < rate)
    seq = seq.copy()
    seq[sub] = (seq[sub] + rng.integers(1, alphabet_size, int(sub.sum()))) % alphabet_size
    counts = 1 - dele.astype(np.int64) + ins.astype(np.int64)
    out = np.repeat(seq, counts)
    out_anc = np.repeat(anc, counts)
    second = np.cumsum(counts)[ins] - 1  # the duplicated copy of an inserting base becomes the inserted residue
    out[second] = rng.integers(0, alphabet_size, len(second))
    return out, out_anc


class Dataset:
    pass


def make_dataset(n_sets, n_hits, seg_len=200, branching=4, genome_len=4000, seed=1, device="cpu", alphabet=0,
                 level_rate=0.03, query_rate=0.05, n_rate=0.01, rc_fraction=0.2, variety=False, chunk=65536):
    """Build a device-agnostic synthetic workload.  Returns a Dataset with numpy arrays (host) -- move them yourself.

    variety=True additionally injects the control-flow edge cases the reference handles (masked records, sets with
    no / one active record, an exact copy of the best reference => perfect-hit shortcut, duplicate reference rows).
    """
    rng = np.random.default_rng(seed)
    asize = 4 if alphabet == 0 else 20
    tax = MockTaxonomy(branching)
    n_species = len(tax.species)
    assert n_hits <= n_species
    # --- genomes down the tree (numpy, sequential over nodes) ---
    anc0 = rng.integers(0, asize, genome_len).astype(np.uint8)
    level = [(anc0, np.arange(genome_len, dtype=np.int32))]
    for lv in range(1, len(RANKS)):  # phylum .. species
        nxt = []
        for seq, anc in level:
            for _ in range(branching):
                nxt.append(_mutate(seq, anc, level_rate, rng, asize))
        level = nxt
    g_len = np.array([len(s) for s, _ in level], np.int64)
    g_off = np.zeros(n_species + 1, np.int64)
    g_off[1:] = np.cumsum(g_len)
    genomes = np.concatenate([s for s, _ in level])
    ancs = np.concatenate([a for _, a in level])
    if alphabet == 0 and n_rate > 0:
        nmask = rng.random(len(genomes)) < n_rate * 0.2
        genomes[nmask] = 4
    is_rc = rng.random(n_species) < (rc_fraction if alphabet == 0 else 0.0)

    # --- choose species + locus per record set ---
    W = seg_len
    p = rng.integers(0, genome_len - W, n_sets)
    if n_hits > 64:  # (redrawing whole rows until they hold no duplicate never ends for hundreds of hits out of thousands of species)
        sp = np.argsort(rng.random((n_sets, n_species)), axis=1)[:, :n_hits].astype(np.int64)
    else:
        sp = rng.integers(0, n_species, (n_sets, n_hits))
    for _ in range(64):  # make the species of a set distinct
        srt = np.sort(sp, axis=1)
        dup_rows = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
        if len(dup_rows) == 0:
            break
        sp[dup_rows] = rng.integers(0, n_species, (len(dup_rows), n_hits))
    # locus [rs, re) of ancestral window [p, p+W) in every chosen species: searchsorted per species group
    flat_sp = sp.reshape(-1)
    flat_p = np.repeat(p, n_hits)
    order = np.argsort(flat_sp, kind="stable")
    rs = np.zeros(len(flat_sp), np.int64)
    re = np.zeros(len(flat_sp), np.int64)
    bounds = np.searchsorted(flat_sp[order], np.arange(n_species + 1))
    for s in range(n_species):
        idx = order[bounds[s]:bounds[s + 1]]
        if len(idx) == 0:
            continue
        a = ancs[g_off[s]:g_off[s + 1]]
        rs[idx] = np.searchsorted(a, flat_p[idx], "left")
        re[idx] = np.searchsorted(a, flat_p[idx] + W, "left")
    rs = rs.reshape(n_sets, n_hits)
    re = np.maximum(re.reshape(n_sets, n_hits), rs + 8)
    re = np.minimum(re, g_len[sp])

    dev = torch.device(device)
    t_gen = torch.from_numpy(genomes).to(dev)
    # --- queries: the true species' locus (hit 0) + extra divergence ---
    q_src_len = (re[:, 0] - rs[:, 0])
    Wmax = int(q_src_len.max())
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 7919 + 13)
    t_anc = torch.from_numpy(ancs.astype(np.int64)).to(dev)
    q_letters_parts, q_lens, q_anc_parts = [], [], []
    qchunk = max(1, min(chunk, (1 << 27) // max(1, Wmax)))
    for c0 in range(0, n_sets, qchunk):
        c1 = min(n_sets, c0 + qchunk)
        m = c1 - c0
        base = torch.from_numpy(g_off[sp[c0:c1, 0]] + rs[c0:c1, 0]).to(dev)
        ln = torch.from_numpy(q_src_len[c0:c1]).to(dev)
        col = torch.arange(Wmax, device=dev)
        valid = col[None, :] < ln[:, None]
        idx = (base[:, None] + col[None, :]).clamp_(max=len(genomes) - 1)
        src = t_gen[idx]
        r = torch.rand((m, Wmax), device=dev, generator=gen)
        sub = r < 0.8 * query_rate
        dele = (r >= 0.8 * query_rate) & (r < 0.9 * query_rate)
        ins = (r >= 0.9 * query_rate) & (r < query_rate)
        rnd = torch.randint(1, asize, (m, Wmax), device=dev, generator=gen, dtype=torch.uint8)
        src4 = torch.where(src >= asize, torch.zeros_like(src), src)
        src = torch.where(sub & (src < asize), (src4 + rnd) % asize, src)
        counts = (valid & ~dele).to(torch.int64) + (valid & ins).to(torch.int64)
        tot = counts.sum(1)
        row_off = torch.cumsum(tot, 0) - tot
        pos = row_off[:, None] + torch.cumsum(counts, 1) - counts
        out = torch.zeros(int(tot.sum()), dtype=torch.uint8, device=dev)
        out_anc = torch.full((int(tot.sum()),), -1, dtype=torch.int64, device=dev)  # ancestral coordinate of each query base
        keep = valid & ~dele
        out[pos[keep]] = src[keep]
        out_anc[pos[keep]] = t_anc[idx][keep]
        insm = valid & ins
        ins_pos = pos[insm] + keep[insm].to(torch.int64)
        out[ins_pos] = torch.randint(0, asize, (int(insm.sum()),), device=dev, generator=gen, dtype=torch.uint8)
        out_anc[ins_pos] = -1
        if alphabet == 0 and n_rate > 0:
            nm = torch.rand(out.shape, device=dev, generator=gen) < n_rate
            out[nm] = 4
        q_letters_parts.append(out.cpu().numpy())
        q_anc_parts.append(out_anc.cpu().numpy())
        q_lens.append(tot.cpu().numpy())
    q_codes = np.concatenate(q_letters_parts)
    q_anc = np.concatenate(q_anc_parts)
    q_len = np.concatenate(q_lens).astype(np.int64)
    q_off = np.zeros(n_sets + 1, np.int64)
    q_off[1:] = np.cumsum(q_len)

    # --- alignment rows ---
    qstart = rng.integers(1, 7, (n_sets, n_hits)).astype(np.int64)
    qstop = q_len[:, None] - rng.integers(0, 6, (n_sets, n_hits))
    qstop = np.maximum(qstop, qstart + 8)
    alnlen = qstop - qstart + 1
    x = rs + 1 + (qstart - 1)                       # forward reference coordinates, 1-based inclusive
    y = re - (q_len[:, None] - qstop)
    y = np.maximum(y, x)
    glen = g_len[sp]
    x = np.clip(x, 1, glen)
    y = np.clip(y, 1, glen)
    # identities = query bases inside [qstart, qstop] whose homologous base (same ancestral coordinate) exists in the
    # hit's locus and carries the same letter -- what a gapped aligner would report, known here by construction
    ident = np.zeros((n_sets, n_hits), np.int64)
    t_q = torch.from_numpy(q_codes).to(dev)
    t_qa = torch.from_numpy(q_anc).to(dev)
    Lq = int(q_len.max())
    Lr = int((re - rs).max())
    big = torch.iinfo(torch.int64).max
    step = max(1, min(chunk // 16, (1 << 26) // max(1, n_hits * max(Lq, Lr))))  # keep the (m, n_hits, L) temporaries bounded
    for c0 in range(0, n_sets, step):
        c1 = min(n_sets, c0 + step)
        qcol = torch.arange(Lq, device=dev)
        qb = torch.from_numpy(q_off[c0:c1]).to(dev)
        ql = torch.from_numpy(q_len[c0:c1]).to(dev)
        qidx = (qb[:, None] + qcol[None, :]).clamp_(max=len(q_codes) - 1)
        qv = t_q[qidx]                                    # (m, Lq)
        qa = torch.where(qcol[None, :] < ql[:, None], t_qa[qidx], torch.full_like(qidx, -1))
        rcol = torch.arange(Lr, device=dev)
        gb = torch.from_numpy(g_off[sp[c0:c1]] + rs[c0:c1]).to(dev)       # (m, n)
        rl = torch.from_numpy((re - rs)[c0:c1]).to(dev)
        ridx = (gb[:, :, None] + rcol).clamp_(max=len(genomes) - 1)
        ra = torch.where(rcol[None, None, :] < rl[:, :, None], t_anc[ridx], torch.full_like(ridx, big))
        rv = t_gen[ridx]
        qa_b = qa[:, None, :].expand(-1, n_hits, -1).contiguous()
        pos = torch.searchsorted(ra, qa_b).clamp_(max=Lr - 1)               # (m, n, Lq)
        hit = (torch.gather(ra, 2, pos) == qa_b) & (qa_b >= 0) & (torch.gather(rv, 2, pos) == qv[:, None, :])
        lo_ = torch.from_numpy(qstart[c0:c1] - 1).to(dev)
        hi_ = torch.from_numpy(qstop[c0:c1]).to(dev)
        inwin = (qcol[None, None, :] >= lo_[:, :, None]) & (qcol[None, None, :] < hi_[:, :, None])
        ident[c0:c1] = (hit & inwin).sum(2).cpu().numpy()
    score = (2 * ident - alnlen).astype(np.float32)
    rc_hit = is_rc[sp]
    rstart = np.where(rc_hit, glen - x + 1, x)
    rstop = np.where(rc_hit, glen - y + 1, y)
    filtered = np.zeros((n_sets, n_hits), np.uint8)

    # --- reference store: forward genomes, reverse-complemented for the RC species ---
    ref_codes = genomes.copy()
    for s in np.nonzero(is_rc)[0]:
        seg = ref_codes[g_off[s]:g_off[s + 1]][::-1].copy()
        lt4 = seg < 4
        seg[lt4] = 3 - seg[lt4]
        ref_codes[g_off[s]:g_off[s + 1]] = seg
    node = tax.species[sp]

    if variety:
        # masked records
        filtered[rng.random((n_sets, n_hits)) < 0.03] = 1
        k = np.arange(n_sets)
        filtered[k % 23 == 5] = 1                         # no active record at all
        one = k % 23 == 7
        filtered[one] = 1
        filtered[one, rng.integers(0, n_hits, int(one.sum()))] = 0   # exactly one active record
        # perfect best hit: the query range equals the reference region of hit 1 (shortcut, no alignment)
        perf = np.nonzero(k % 23 == 9)[0]
        for s in perf:
            h = 1 % n_hits
            qs_, qe_ = 1, int(q_len[s])
            sp_h = sp[s, h]
            x0 = int(rs[s, h]) + 1
            y0 = min(int(glen[s, h]), x0 + qe_ - 1)
            ln = y0 - x0 + 1
            if ln != qe_:
                continue
            q_codes[q_off[s]:q_off[s] + ln] = genomes[g_off[sp_h] + x0 - 1:g_off[sp_h] + y0]
            qstart[s, :] = 1
            qstop[s, :] = qe_
            alnlen[s, :] = qe_
            x[s, h], y[s, h] = x0, y0
            ident[s, h] = ln
            score[s, h] = np.float32(2 * ln - ln) + np.float32(1000.0)
            rstart[s, h] = glen[s, h] - x0 + 1 if rc_hit[s, h] else x0
            rstop[s, h] = glen[s, h] - y0 + 1 if rc_hit[s, h] else y0
        # duplicated reference rows (distance 0 between two hits, ties in score)
        dup = k % 23 == 11
        if n_hits >= 3:
            for arr in (sp, qstart, qstop, alnlen, ident, score, rstart, rstop, node):
                arr[dup, 2] = arr[dup, 1]

    # records of a set in the order the reference's record-set generator hands them to predict():
    # sorted by (qstart, qstop), input order on ties (core/src/alignmentrecord.hh:480)
    true_node = node[:, 0].copy()  # the species the query was cut from
    key = qstart * (int(qstop.max()) + 1) + qstop
    perm = np.argsort(key, axis=1, kind="stable")
    for arr_name in ("sp", "qstart", "qstop", "alnlen", "ident", "score", "rstart", "rstop", "node", "filtered"):
        arr = locals()[arr_name]
        arr[...] = np.take_along_axis(arr, perm, axis=1)

    ds = Dataset()
    ds.alphabet = alphabet
    ds.tax = tax
    ds.n_sets, ds.n_hits = n_sets, n_hits
    table = LETTERS if alphabet == 0 else np.frombuffer(_AA.encode(), np.uint8)[:20]
    ds.query_letters = table[q_codes]
    ds.query_offsets = q_off.astype(np.uint64)
    ds.ref_letters = table[ref_codes]
    ds.ref_offsets = g_off.astype(np.uint64)
    ds.species_rc = is_rc
    ds.true_node = true_node
    H = n_sets * n_hits
    ds.batch = dict(
        set_offsets=(np.arange(n_sets + 1, dtype=np.int64) * n_hits).astype(np.uint32),
        query_seq=np.arange(n_sets, dtype=np.uint32),
        query_len=q_len.astype(np.uint32),
        qstart=qstart.reshape(H).astype(np.uint32), qstop=qstop.reshape(H).astype(np.uint32),
        rstart=rstart.reshape(H).astype(np.uint32), rstop=rstop.reshape(H).astype(np.uint32),
        score=score.reshape(H).astype(np.float32), identities=ident.reshape(H).astype(np.uint32),
        alnlen=alnlen.reshape(H).astype(np.uint32), ref_seq=sp.reshape(H).astype(np.uint32),
        node=node.reshape(H).astype(np.uint32), filtered=filtered.reshape(H))
    return ds


def records_aos(batch, dtype):
    """SoA batch -> the array-of-structs the C oracle takes (tests/helpers/oracle.RECORD_DTYPE)."""
    recs = np.zeros(len(batch["qstart"]), dtype)
    for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "ref_seq", "node"):
        recs[f] = batch[f]
    recs["filtered"] = batch["filtered"] if batch.get("filtered") is not None else 0
    return recs


def write_files(ds, folder, line_width=80):
    """Dump a Dataset in the reference's on-disk formats (doc/fileformats.md): tax/, mapping.tax, ref FASTA + .fai,
    query FASTA, alignments TSV."""
    os.makedirs(folder, exist_ok=True)
    ds.tax.write_ncbi_dump(os.path.join(folder, "tax"))
    ext = ".fna" if ds.alphabet == 0 else ".faa"
    n_species = len(ds.ref_offsets) - 1
    with open(os.path.join(folder, "mapping.tax"), "w") as f:
        for s in range(n_species):
            f.write("ref%d\t%s\n" % (s, ds.tax.taxid[int(ds.tax.species[s % len(ds.tax.species)])]))  # (make_ragged_dataset: one reference collection per class, back to back)
    with open(os.path.join(folder, "ref" + ext), "wb") as f, open(os.path.join(folder, "ref" + ext + ".fai"), "w") as fai:
        for s in range(n_species):
            seq = ds.ref_letters[int(ds.ref_offsets[s]):int(ds.ref_offsets[s + 1])].tobytes()
            f.write(b">ref%d\n" % s)
            off = f.tell()
            for i in range(0, len(seq), line_width):
                f.write(seq[i:i + line_width] + b"\n")
            fai.write("ref%d\t%d\t%d\t%d\t%d\n" % (s, len(seq), off, line_width, line_width + 1))
    with open(os.path.join(folder, "query" + ext), "wb") as f:
        qo = np.asarray(ds.query_offsets).astype(np.int64).tolist()
        whole = ds.query_letters.tobytes()
        parts = []
        for s in range(len(qo) - 1):
            seq = whole[qo[s]:qo[s + 1]]
            parts.append(b">q%d\n" % s)
            parts.extend(seq[i:i + line_width] + b"\n" for i in range(0, len(seq), line_width))
            if len(parts) > 200000:
                f.write(b"".join(parts)); parts = []
        f.write(b"".join(parts))
    b = ds.batch
    # (chunks of rows formatted in one go: ~1 us per line instead of ~2.5 with a write per line -- 20 M lines of config 2 in 20 s)
    so = np.asarray(b["set_offsets"]).astype(np.int64)
    set_of = np.repeat(np.arange(ds.n_sets, dtype=np.int64), np.diff(so))
    qlen = np.asarray(b["query_len"])[set_of]
    qname = np.asarray(b["query_seq"]).astype(np.int64)[set_of] if "query_seq" in b else set_of  # (the query of set s is query_seq[s]: s itself except in make_ragged_dataset's shuffled streams)
    star = np.asarray(b["filtered"]).astype(bool) if "filtered" in b and b["filtered"] is not None else np.zeros(len(set_of), bool)
    fmt = "%sq%d\t%d\t%d\t%d\tref%d\t%d\t%d\t%g\t0\t%d\t%d\n"
    with open(os.path.join(folder, "aln.tsv"), "w") as f:
        step = 250000
        for lo in range(0, len(set_of), step):
            hi = min(len(set_of), lo + step)
            cols = [np.where(star[lo:hi], "*", "").tolist(), qname[lo:hi].tolist(), b["qstart"][lo:hi].tolist(), b["qstop"][lo:hi].tolist(),
                    qlen[lo:hi].tolist(), b["ref_seq"][lo:hi].tolist(), b["rstart"][lo:hi].tolist(), b["rstop"][lo:hi].tolist(),
                    np.asarray(b["score"][lo:hi], np.float64).tolist(), b["identities"][lo:hi].tolist(), b["alnlen"][lo:hi].tolist()]
            f.write("".join([fmt % r for r in zip(*cols)]))


def embed_in_big_tree(tax, n_total, max_depth, seed=7):
    """`tax` (nodes 0 .. n0-1 keep their indices) inside a tree of n_total nodes of NCBI shape: the other n_total - n0 nodes are
    clades no record refers to, hung under random earlier nodes whose depth is below max_depth (four of five under a node of
    the previous batch, so the tree grows level by level).  What the engine sees of a real taxonomy: most of its tables are
    about taxa the batch never names.  Returns an object with parent / depth / left / right (fresh nested-set numbers)."""
    rng = np.random.default_rng(seed)
    n0 = len(tax.parent)
    n = max(n_total, n0)
    parent = np.zeros(n, np.uint32)
    depth = np.zeros(n, np.uint8)
    parent[:n0] = np.asarray(tax.parent)
    depth[:n0] = np.asarray(tax.depth)
    lo, prev_lo = n0, 0
    while lo < n:
        hi = min(n, int(lo * 1.22) + 1)
        deep = rng.random(hi - lo) < 0.8
        cand = np.where(deep, rng.integers(prev_lo, lo, hi - lo), rng.integers(0, lo, hi - lo))
        cand[depth[cand] >= max_depth] = 0
        parent[lo:hi] = cand
        depth[lo:hi] = depth[cand] + 1
        prev_lo, lo = lo, hi
    order = np.argsort(parent[1:], kind="stable") + 1
    start = np.searchsorted(parent[order], np.arange(n + 1))
    left = np.zeros(n, np.uint32)
    right = np.zeros(n, np.uint32)
    nxt = start[:-1].copy()
    stack, ctr = [0], 1
    left[0] = 1
    while stack:
        v = stack[-1]
        if nxt[v] < start[v + 1]:
            c = int(order[nxt[v]]); nxt[v] += 1
            ctr += 1; left[c] = ctr
            stack.append(c)
        else:
            ctr += 1; right[v] = ctr
            stack.pop()

    class Tree:
        pass
    t = Tree()
    t.parent, t.depth, t.left, t.right = parent, depth, left, right
    return t


def make_ragged_dataset(classes, seed=1, device="cpu", branching=4, shuffle=True, level_rate=0.03, query_rate=0.05, alphabet=0):
    """Record sets of MIXED shape in one batch: `classes` = [(n_sets, n_hits, seg_len), ...]; every class is a make_dataset()
    workload of its own (own genomes, same mock taxonomy), the classes' reference and query collections laid out back to back
    in one store each, the record sets of all classes shuffled into one stream.  Sets of 1 ... hundreds of hits and segments
    of 50 ... 1500 nt side by side is what a real alignment stream looks like (core/USAGE.md:90-96) and what the uniform
    BASELINE configurations never show the engine: ragged tiles, several length bins per round, wide path labels."""
    rng = np.random.default_rng(seed)
    parts = []
    for k, (n_sets, n_hits, seg_len) in enumerate(classes):
        parts.append(make_dataset(n_sets=n_sets, n_hits=n_hits, seg_len=seg_len, branching=branching, genome_len=max(2000, 4 * seg_len),
                                  seed=seed * 1000 + k, device=device, level_rate=level_rate, query_rate=query_rate, alphabet=alphabet))
    ds = Dataset()
    ds.alphabet = alphabet
    ds.tax = parts[0].tax
    ds.n_sets = sum(p.n_sets for p in parts)
    ds.n_hits = None
    ds.classes = list(classes)
    ds.query_letters = np.concatenate([p.query_letters for p in parts])
    ds.ref_letters = np.concatenate([p.ref_letters for p in parts])
    q_off, r_off, q_base, r_base, qseq_base, rseq_base = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)], 0, 0, [], []
    nq = nr = 0
    for p in parts:
        q_off.append(p.query_offsets[1:] + np.uint64(q_base)); r_off.append(p.ref_offsets[1:] + np.uint64(r_base))
        q_base += int(p.query_offsets[-1]); r_base += int(p.ref_offsets[-1])
        qseq_base.append(nq); rseq_base.append(nr)
        nq += len(p.query_offsets) - 1; nr += len(p.ref_offsets) - 1
    ds.query_offsets = np.concatenate(q_off)
    ds.ref_offsets = np.concatenate(r_off)
    # per set: (class, set inside the class); shuffled
    cls = np.concatenate([np.full(p.n_sets, k, np.int64) for k, p in enumerate(parts)])
    loc = np.concatenate([np.arange(p.n_sets, dtype=np.int64) for p in parts])
    perm = rng.permutation(len(cls)) if shuffle else np.arange(len(cls))
    cls, loc = cls[perm], loc[perm]
    hits = np.array([c[1] for c in classes], np.int64)[cls]
    so = np.zeros(len(cls) + 1, np.int64)
    so[1:] = np.cumsum(hits)
    H = int(so[-1])
    # records: position of every record of the stream inside its class's arrays
    rec_cls = np.repeat(cls, hits)
    rec_in = np.repeat(loc * hits, hits) + (np.arange(H) - np.repeat(so[:-1], hits))
    batch = {"set_offsets": so.astype(np.uint32)}
    batch["query_seq"] = (np.array(qseq_base, np.int64)[cls] + loc).astype(np.uint32)
    batch["query_len"] = np.zeros(len(cls), np.uint32)
    for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "ref_seq", "node", "filtered"):
        batch[f] = np.zeros(H, parts[0].batch[f].dtype)
    for k, p in enumerate(parts):
        m = cls == k
        batch["query_len"][m] = p.batch["query_len"][loc[m]]
        mr = rec_cls == k
        for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "node", "filtered"):
            batch[f][mr] = p.batch[f][rec_in[mr]]
        batch["ref_seq"][mr] = (p.batch["ref_seq"][rec_in[mr]].astype(np.int64) + rseq_base[k]).astype(np.uint32)
    ds.batch = batch
    ds.set_class = cls
    return ds
