#!/usr/bin/env python3
"""Bit-level model of the banded strip kernel (k_myers_band, csrc/align_dna_kernels.hip) with small words, checked
against plain Levenshtein.  Design aid + CPU test of the exactness argument (tests/test_band_model_cpu.py):

  * a strip of HS rows sweeps only the column groups [g_lo, g_hi) that Ukkonen's band  min(0, delta) - h <= j - i <= max(0, delta) + h  touches;
  * a strip that starts at a column > 0 starts from VP = all ones on top of the value the strip above had there
    (an upper bound of the true column), column groups the strip above never computed hand down "+1 per column";
  * what crosses a strip boundary per column is HP and HN of the strip's last row; the add carry into the strip below
    IS the HN bit (carry out of the top bit <=> vp & D0 at the top bit), so no third plane;
  * the result D' is >= the true distance d, and D' == d whenever d <= k = 2 h + |delta|; D' > k proves d > k.
"""
import random


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[len(b)]


def banded_strips(pat, txt, h, HS=8, G=4, probe_groups=None):
    """pat = rows (shorter), txt = columns.  Returns (D', ok).  HS rows per strip, G columns per group.
    h: one half width, or one per strip (each at most HS above the one before: a strip never starts left of the one above)."""
    m, n = len(pat), len(txt)
    assert m >= 1 and n >= 1
    delta = n - m                       # may be negative: more rows than columns
    dlo, dhi = min(0, delta), max(0, delta)
    mask = (1 << HS) - 1
    nstrips = (m + HS - 1) // HS
    hs = list(h) if isinstance(h, (list, tuple)) else [h] * (nstrips + 1)
    hs += [hs[-1]] * (nstrips + 1 - len(hs))
    n_groups = (n + G - 1) // G
    planes = {}          # (strip, group) -> (P bits list, N bits list) per column of the group
    saved_next = 0
    bottom = 0
    VP = VN = 0
    for s in range(nstrips):
        R = s * HS
        first, last = s == 0, s + 1 == nstrips
        eq = {}
        for c in set(txt):
            e = 0
            for r in range(HS):
                if R + r < m and pat[R + r] == c:
                    e |= 1 << r
            eq[c] = e
        h, h_nxt, h_prv = hs[s], hs[s + 1], hs[s - 1] if s else 0
        g_lo = max(0, R + dlo - h) // G
        g_hi = min(n_groups, (R + HS + dhi + h - 1) // G + 1)
        g_next = max(0, R + HS + dlo - h_nxt) // G
        ga_hi = min(n_groups, max(0, R + dhi + h_prv - 1) // G + 1) if not first else 0   # window end of the strip above
        assert g_lo < g_hi
        VP, VN = mask, 0
        bottom = (0 if first else saved_next) + HS
        if first:
            assert g_lo == 0
        tmp = None
        for g in range(g_lo, g_hi):
            if g == g_next:
                tmp = bottom
            cols = min(G, n - g * G)
            Pout, Nout = [], []
            for k in range(cols):
                j = g * G + k
                if first or g >= ga_hi:
                    hp_in, hn_in = 1, 0
                else:
                    pin, nin = planes[(s - 1, g)]
                    hp_in, hn_in = pin[k], nin[k]
                e = eq[txt[j]]
                X = e | VN
                sm = VP + (X & VP) + hn_in           # add carry == HN of the row above
                D0 = ((sm ^ VP) | X) & mask
                HN = VP & D0
                HP = (VN | ~(VP | D0)) & mask
                Xs = ((HP << 1) | hp_in) & mask
                Hs = ((HN << 1) | hn_in) & mask
                VN = Xs & D0
                VP = (Hs | ~(Xs | D0)) & mask
                Pout.append((HP >> (HS - 1)) & 1)
                Nout.append((HN >> (HS - 1)) & 1)
                # the carry out of the add must equal HN's top bit
                assert (sm >> HS) == ((HN >> (HS - 1)) & 1)
            bottom += sum(Pout) - sum(Nout)
            if not last:
                planes[(s, g)] = (Pout, Nout)
        if not last:
            if g_next == g_hi:   # the strip below starts right behind this strip's window
                tmp = bottom
            assert tmp is not None, (s, g_lo, g_hi, g_next)
            saved_next = tmp
    # phantom rows below the pattern
    tail_rows = nstrips * HS - m
    dist = bottom
    for r in range(HS - tail_rows, HS):
        dist -= ((VP >> r) & 1) - ((VN >> r) & 1)
    k = 2 * min(hs[:nstrips]) + abs(delta)
    return dist, dist <= k


def check(trials=3000, seed=1):
    rng = random.Random(seed)
    bad = 0
    for t in range(trials):
        HS = rng.choice([4, 8]); G = rng.choice([2, 4, 8])
        m = rng.randint(1, 40)
        a = [rng.randrange(3) for _ in range(m)]
        # mutate
        b = []
        rate = rng.choice([0.0, 0.05, 0.2, 0.5])
        for c in a:
            r = rng.random()
            if r < rate / 3: continue
            if r < 2 * rate / 3: b.append(rng.randrange(3)); b.append(c); continue
            if r < rate: b.append(rng.randrange(3)); continue
            b.append(c)
        if rng.random() < 0.3:
            b += [rng.randrange(3) for _ in range(rng.randint(0, 10))]
        if not b: b = [0]
        pat, txt = (a, b) if rng.random() < 0.5 else (b, a)   # rows / columns: either may be the longer one
        d = levenshtein(pat, txt)
        for h in (0, 1, 2, 3, 5, 8, 13, 100):
            dp, ok = banded_strips(pat, txt, h, HS, G)
            k = 2 * h + abs(len(txt) - len(pat))
            assert dp >= d, (pat, txt, h, dp, d)
            if d <= k:
                assert ok and dp == d, (pat, txt, h, HS, G, dp, d)
            else:
                assert not ok, (pat, txt, h, dp, d, k)
        # a different half width per strip (the kernel refreshes its estimate after every strip)
        for _ in range(4):
            nstrips = (len(pat) + HS - 1) // HS
            hs = [rng.randint(0, 12)]
            for _s in range(nstrips):
                hs.append(rng.randint(0, hs[-1] + HS))
            dp, ok = banded_strips(pat, txt, hs, HS, G)
            k = 2 * min(hs[:nstrips]) + abs(len(txt) - len(pat))
            assert dp >= d, (pat, txt, hs, dp, d)
            if d <= k:
                assert ok and dp == d, (pat, txt, hs, HS, G, dp, d)
            else:
                assert not ok, (pat, txt, hs, dp, d, k)
    return trials


def banded_window_ring(pat, txt, WW, W=4):
    """The model of k_myers_window<WW> (round 5): a window of WW words (WW even) that slides down one word per W columns, half a
    group out of phase with the text groups -- the first W/2 columns of group g update words [g - WW/2, g + WW/2 - 1] (clipped at
    word 0), the last W/2 columns words [g - WW/2 + 1, g + WW/2] -- with ONLY the window's words alive: a word that leaves hands
    its vertical deltas to the running score T, a word that enters starts as "all +1", and nothing clips the window at the
    pattern's end (rows past the pattern match nothing), so the kernel needs no bound on the pattern's length.  Guaranteed band:
    |j - i| <= (WW - 1) W / 2, i.e. k = (WW - 1) W - |delta|  (WW = 4: banded_window_phase).  Returns (D', applicable, ok)."""
    m, n = len(pat), len(txt)
    assert WW % 2 == 0 and W % 2 == 0 and m >= 1
    ad = abs(n - m)
    reach = (WW - 1) * W // 2
    if ad >= reach:
        return None, False, False
    k = 2 * reach - ad
    mask = (1 << W) - 1
    H = WW // 2
    VP, VN = {}, {}   # word index -> bits; a word not in here is in its initial state (all +1)

    def eqmask(c, wi):
        return sum(1 << r for r in range(W) if wi * W + r < m and pat[wi * W + r] == c)

    def sweep(j0, j1, wlo, whi):
        live = set(range(wlo, whi + 1))
        for wi in list(VP):            # (the kernel holds WW registers: nothing outside the window may be alive)
            if wi not in live: del VP[wi]; del VN[wi]
        assert len(live) <= WW
        for j in range(j0, j1):
            hp_in, hn_in = 1, 0
            for wi in range(wlo, whi + 1):
                vp, vn = VP.get(wi, mask), VN.get(wi, 0)
                X = eqmask(txt[j], wi) | vn
                sm = vp + (X & vp) + hn_in
                D0 = ((sm ^ vp) | X) & mask
                HN = vp & D0
                HP = (vn | ~(vp | D0)) & mask
                Xs = ((HP << 1) | hp_in) & mask
                Hs = ((HN << 1) | hn_in) & mask
                VN[wi] = Xs & D0
                VP[wi] = (Hs | ~(Xs | D0)) & mask
                hp_in, hn_in = (HP >> (W - 1)) & 1, (HN >> (W - 1)) & 1

    T = 0
    half = W // 2
    top = 0
    for g in range((n + W - 1) // W):
        c0 = g * W
        ca = min(n, c0 + half)
        top = max(0, g - H)
        sweep(c0, ca, top, g + H - 1)
        T += ca - c0
        if ca < n:
            if g - H >= 0:  # the window moves down a word: its first word leaves (always a whole word of the pattern, see the kernel)
                assert (g - H + 1) * W <= m
                T += bin(VP[g - H]).count("1") - bin(VN[g - H]).count("1")
            top = max(0, g - H + 1)
            cb = min(n, c0 + W)
            sweep(ca, cb, top, g + H)
            T += cb - ca
    dist = T
    for wi in range(top, (m + W - 1) // W):  # the window's words and the untouched ones below it, rows of the pattern only
        for r in range(W):
            if wi * W + r < m:
                dist += ((VP.get(wi, mask) >> r) & 1) - ((VN.get(wi, 0) >> r) & 1)
    return dist, True, dist <= k


def check_window_ring(trials=4000, seed=9):
    rng = random.Random(seed)
    hit = 0
    for t in range(trials):
        W = rng.choice([4, 8]); WW = rng.choice([4, 6, 8, 10, 14])
        m = rng.randint(1, 14 * W)
        a = [rng.randrange(3) for _ in range(m)]
        b = []
        rate = rng.choice([0.0, 0.05, 0.2, 0.5, 0.9])
        for c in a:
            r = rng.random()
            if r < rate / 3: continue
            if r < 2 * rate / 3: b.append(rng.randrange(3)); b.append(c); continue
            if r < rate: b.append(rng.randrange(3)); continue
            b.append(c)
        if rng.random() < 0.3:
            b += [rng.randrange(3) for _ in range(rng.randint(0, 3 * W))]
        if not b: b = [0]
        d = levenshtein(a, b)
        dp, applicable, ok = banded_window_ring(a, b, WW, W)
        if not applicable:
            continue
        k = (WW - 1) * W - abs(len(b) - m)
        assert dp >= d, (a, b, WW, W, dp, d)
        if d <= k:
            assert ok and dp == d, (a, b, WW, W, dp, d, k)
            hit += 1
        else:
            assert not ok, (a, b, WW, W, dp, d, k)
    assert hit > trials // 5
    return trials


if __name__ == "__main__":
    print("window ring ok", check_window_ring(20000))
    print("ok", check())


# ---- K1 form: the whole pattern in registers, a STATIC window of words per column group ---------------------------
def banded_window(pat, txt, NW, HW, W=4):
    """k_myers_short_band (align_dna_kernels.hip): pattern rows in NW words of W bits, text in groups of W columns; group g
    updates only the words [g - HW, g + HW] (clipped).  Words below the window keep their initial VP = all ones until the
    window reaches them, words above it are dropped ("+1 per column" comes down from there).  The score is followed at
    the top of the window: T = D'[row above the window's first word][column], +1 per column, plus the dropped word's
    vertical deltas whenever the window moves down.  Returns (D', applicable, ok)."""
    m, n = len(pat), len(txt)
    assert 1 <= m <= NW * W
    delta = n - m
    ad = abs(delta)
    if ad >= W * HW:
        return None, False, False
    h = W * HW - ad
    k = 2 * h + ad
    mask = (1 << W) - 1
    eq = {}
    for c in set(txt):
        eq[c] = [sum(1 << r for r in range(W) if wi * W + r < m and pat[wi * W + r] == c) for wi in range(NW)]
    VP = [mask] * NW
    VN = [0] * NW
    T = 0
    n_groups = (n + W - 1) // W
    assert n_groups <= NW + HW
    for g in range(n_groups):
        wlo, whi = max(0, g - HW), min(NW - 1, g + HW)
        cols = min(W, n - g * W)
        for kcol in range(cols):
            e = eq[txt[g * W + kcol]]
            hp_in, hn_in = 1, 0
            for wi in range(wlo, whi + 1):
                vp, vn = VP[wi], VN[wi]
                X = e[wi] | vn
                sm = vp + (X & vp) + hn_in
                D0 = ((sm ^ vp) | X) & mask
                HN = vp & D0
                HP = (vn | ~(vp | D0)) & mask
                Xs = ((HP << 1) | hp_in) & mask
                Hs = ((HN << 1) | hn_in) & mask
                VN[wi] = Xs & D0
                VP[wi] = (Hs | ~(Xs | D0)) & mask
                hp_in, hn_in = (HP >> (W - 1)) & 1, (HN >> (W - 1)) & 1
        T += cols
        if g + 1 < n_groups and g - HW >= 0 and g - HW <= NW - 1:   # the window's first word moves down: fold the dropped word in
            w0 = g - HW
            T += bin(VP[w0]).count("1") - bin(VN[w0]).count("1")
    wlo_final = max(0, (n_groups - 1) - HW) if n_groups else 0
    dist = T
    for wi in range(wlo_final, NW):
        for r in range(W):
            if wi * W + r < m:
                dist += ((VP[wi] >> r) & 1) - ((VN[wi] >> r) & 1)
    return dist, True, dist <= k


def check_window(trials=3000, seed=2):
    rng = random.Random(seed)
    for t in range(trials):
        W = rng.choice([3, 4, 8]); NW = rng.randint(1, 7); HW = rng.randint(1, 3)
        m = rng.randint(1, NW * W)
        a = [rng.randrange(3) for _ in range(m)]
        b = []
        rate = rng.choice([0.0, 0.05, 0.2, 0.5, 0.9])
        for c in a:
            r = rng.random()
            if r < rate / 3: continue
            if r < 2 * rate / 3: b.append(rng.randrange(3)); b.append(c); continue
            if r < rate: b.append(rng.randrange(3)); continue
            b.append(c)
        if rng.random() < 0.3:
            b += [rng.randrange(3) for _ in range(rng.randint(0, 6))]
        if not b: b = [0]
        d = levenshtein(a, b)
        dp, applicable, ok = banded_window(a, b, NW, HW, W)
        if not applicable:
            continue
        k = 2 * (W * HW - abs(len(b) - m)) + abs(len(b) - m)
        assert dp >= d, (a, b, NW, HW, W, dp, d)
        if d <= k:
            assert ok and dp == d, (a, b, NW, HW, W, dp, d, k)
        else:
            assert not ok, (a, b, NW, HW, W, dp, d, k)
    return trials


if __name__ == "__main__":
    print("window ok", check_window())


# ---- K1, four-word window phase-shifted by half a group (k_myers_short_band<NW, 4 words>) --------------------------------
def banded_window_phase(pat, txt, NW, W=4):
    """Window of FOUR words that slides down one word per W columns, half a group out of phase with the text groups: the
    first W/2 columns of group g update words [g-2, g+1], the last W/2 columns words [g-1, g+2] (clipped).  Guaranteed
    band: |j - i| <= 1.5 W - ... => h = 1.5 W - |delta|, k = 2 h + |delta| = 3 W - |delta|  (W = 32: k = 96 - |delta|).
    Returns (D', applicable, ok)."""
    m, n = len(pat), len(txt)
    assert 1 <= m <= NW * W and W % 2 == 0
    ad = abs(n - m)
    if 2 * ad >= 3 * W:
        return None, False, False
    k = 3 * W - ad
    mask = (1 << W) - 1
    eq = {}
    for c in set(txt):
        eq[c] = [sum(1 << r for r in range(W) if wi * W + r < m and pat[wi * W + r] == c) for wi in range(NW)]
    VP = [mask] * NW
    VN = [0] * NW
    T = 0
    half = W // 2

    def sweep(j0, j1, wlo, whi):
        for j in range(j0, j1):
            e = eq[txt[j]]
            hp_in, hn_in = 1, 0
            for wi in range(wlo, whi + 1):
                vp, vn = VP[wi], VN[wi]
                X = e[wi] | vn
                sm = vp + (X & vp) + hn_in
                D0 = ((sm ^ vp) | X) & mask
                HN = vp & D0
                HP = (vn | ~(vp | D0)) & mask
                Xs = ((HP << 1) | hp_in) & mask
                Hs = ((HN << 1) | hn_in) & mask
                VN[wi] = Xs & D0
                VP[wi] = (Hs | ~(Xs | D0)) & mask
                hp_in, hn_in = (HP >> (W - 1)) & 1, (HN >> (W - 1)) & 1

    n_groups = (n + W - 1) // W
    top = 0  # first word of the window in effect
    for g in range(n_groups):
        c0 = g * W
        ca = min(n, c0 + half)
        top = max(0, g - 2)
        sweep(c0, ca, top, min(NW - 1, g + 1))
        T += ca - c0
        if ca < n:  # columns follow: the window moves down a word (if its first word was a real one)
            if g - 2 >= 0 and g - 2 <= NW - 1:
                T += bin(VP[g - 2]).count("1") - bin(VN[g - 2]).count("1")
            top = max(0, g - 1)
            cb = min(n, c0 + W)
            sweep(ca, cb, top, min(NW - 1, g + 2))
            T += cb - ca
    dist = T
    for wi in range(min(top, NW), NW):
        for r in range(W):
            if wi * W + r < m:
                dist += ((VP[wi] >> r) & 1) - ((VN[wi] >> r) & 1)
    return dist, True, dist <= k


def check_window_phase(trials=4000, seed=5):
    rng = random.Random(seed)
    for t in range(trials):
        W = rng.choice([4, 8]); NW = rng.randint(1, 8)
        m = rng.randint(1, NW * W)
        a = [rng.randrange(3) for _ in range(m)]
        b = []
        rate = rng.choice([0.0, 0.05, 0.2, 0.5, 0.9])
        for c in a:
            r = rng.random()
            if r < rate / 3: continue
            if r < 2 * rate / 3: b.append(rng.randrange(3)); b.append(c); continue
            if r < rate: b.append(rng.randrange(3)); continue
            b.append(c)
        if rng.random() < 0.3:
            b += [rng.randrange(3) for _ in range(rng.randint(0, 6))]
        if not b: b = [0]
        d = levenshtein(a, b)
        dp, applicable, ok = banded_window_phase(a, b, NW, W)
        if not applicable:
            continue
        k = 3 * W - abs(len(b) - m)
        assert dp >= d, (a, b, NW, W, dp, d)
        if d <= k:
            assert ok and dp == d, (a, b, NW, W, dp, d, k)
        else:
            assert not ok, (a, b, NW, W, dp, d, k)
    return trials


def banded_window_phase3(pat, txt, NW, W=8):
    """The window of banded_window_phase with a THREE-word middle: of the W columns of text group g the first W/4 update
    words [g-2, g+1], the middle W/2 words [g-1, g+1], the last W/4 words [g-1, g+2] (clipped) -- word g-2 leaves a
    quarter of a group earlier and word g+2 enters a quarter later than there.  Guaranteed band: |j - i| <= 1.25 W, so
    h = 1.25 W - |delta| and k = 2 h + |delta| = 2.5 W - |delta| (W = 32: k = 80 - |delta|; 3.5 words per column instead
    of 4).  Returns (D', applicable, ok)."""
    m, n = len(pat), len(txt)
    assert 1 <= m <= NW * W and W % 4 == 0
    ad = abs(n - m)
    if 4 * ad >= 5 * W:
        return None, False, False
    k = (5 * W) // 2 - ad
    mask = (1 << W) - 1
    eq = {}
    for c in set(txt):
        eq[c] = [sum(1 << r for r in range(W) if wi * W + r < m and pat[wi * W + r] == c) for wi in range(NW)]
    VP = [mask] * NW
    VN = [0] * NW
    T = 0
    q = W // 4

    def sweep(j0, j1, wlo, whi):
        for j in range(j0, j1):
            e = eq[txt[j]]
            hp_in, hn_in = 1, 0
            for wi in range(wlo, whi + 1):
                vp, vn = VP[wi], VN[wi]
                X = e[wi] | vn
                sm = vp + (X & vp) + hn_in
                D0 = ((sm ^ vp) | X) & mask
                HN = vp & D0
                HP = (vn | ~(vp | D0)) & mask
                Xs = ((HP << 1) | hp_in) & mask
                Hs = ((HN << 1) | hn_in) & mask
                VN[wi] = Xs & D0
                VP[wi] = (Hs | ~(Xs | D0)) & mask
                hp_in, hn_in = (HP >> (W - 1)) & 1, (HN >> (W - 1)) & 1

    n_groups = (n + W - 1) // W
    top = 0  # first word of the window in effect
    for g in range(n_groups):
        c0 = g * W
        ca = min(n, c0 + q)
        top = max(0, g - 2)
        sweep(c0, ca, top, min(NW - 1, g + 1))
        T += ca - c0
        if ca < n:  # columns follow: word g-2 leaves the window (if it was a real one)
            if 0 <= g - 2 <= NW - 1:
                T += bin(VP[g - 2]).count("1") - bin(VN[g - 2]).count("1")
            top = max(0, g - 1)
            cb = min(n, c0 + 3 * q)
            sweep(ca, cb, top, min(NW - 1, g + 1))
            T += cb - ca
            if cb < n:  # word g+2 enters
                cc = min(n, c0 + W)
                sweep(cb, cc, top, min(NW - 1, g + 2))
                T += cc - cb
    dist = T
    for wi in range(min(top, NW), NW):
        for r in range(W):
            if wi * W + r < m:
                dist += ((VP[wi] >> r) & 1) - ((VN[wi] >> r) & 1)
    return dist, True, dist <= k


def check_window_phase3(trials=4000, seed=7):
    rng = random.Random(seed)
    hit = 0
    for t in range(trials):
        W = rng.choice([4, 8, 12]); NW = rng.randint(1, 8)
        m = rng.randint(1, NW * W)
        a = [rng.randrange(3) for _ in range(m)]
        b = []
        rate = rng.choice([0.0, 0.05, 0.2, 0.5, 0.9])
        for c in a:
            r = rng.random()
            if r < rate / 3: continue
            if r < 2 * rate / 3: b.append(rng.randrange(3)); b.append(c); continue
            if r < rate: b.append(rng.randrange(3)); continue
            b.append(c)
        if rng.random() < 0.3:
            b += [rng.randrange(3) for _ in range(rng.randint(0, 6))]
        if not b: b = [0]
        d = levenshtein(a, b)
        dp, applicable, ok = banded_window_phase3(a, b, NW, W)
        if not applicable:
            continue
        k = (5 * W) // 2 - abs(len(b) - m)
        assert dp >= d, (a, b, NW, W, dp, d)
        if d <= k:
            assert ok and dp == d, (a, b, NW, W, dp, d, k)
            hit += 1
        else:
            assert not ok, (a, b, NW, W, dp, d, k)
    assert hit > trials // 5
    return trials


if __name__ == "__main__":
    print("window phase ok", check_window_phase())
    print("window phase3 ok", check_window_phase3(20000))
