#!/usr/bin/env python3
"""Bit-level model of the banded strip kernel (k_myers_band, csrc/align_dna_kernels.hip) with small words, checked
against plain Levenshtein.  Design aid + CPU test of the exactness argument (tests/test_band_model_cpu.py):

  * a strip of HS rows sweeps only the column groups [g_lo, g_hi) that Ukkonen's band  -h <= j - i <= delta + h  touches;
  * a strip that starts at a column > 0 starts from VP = all ones on top of the value the strip above had there
    (an upper bound of the true column), column groups the strip above never computed hand down "+1 per column";
  * what crosses a strip boundary per column is HP and HN of the strip's last row; the add carry into the strip below
    IS the HN bit (carry out of the top bit <=> vp & D0 at the top bit), so no third plane;
  * the result D' is >= the true distance d, and D' == d whenever d <= k = 2 h + delta; D' > k proves d > k.
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
    """pat = rows (shorter), txt = columns.  Returns (D', ok, probe_min).  HS rows per strip, G columns per group."""
    m, n = len(pat), len(txt)
    assert n >= m >= 1
    delta = n - m
    mask = (1 << HS) - 1
    nstrips = (m + HS - 1) // HS
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
        g_lo = (R - h) // G if R > h else 0
        g_hi = min(n_groups, (R + HS + delta + h - 1) // G + 1)
        g_next = (R + HS - h) // G if R + HS > h else 0
        ga_hi = min(n_groups, (R + delta + h - 1) // G + 1) if not first else 0   # window end of the strip above
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
    k = 2 * h + delta
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
        pat, txt = (a, b) if len(a) <= len(b) else (b, a)
        d = levenshtein(pat, txt)
        for h in (0, 1, 2, 3, 5, 8, 13, 100):
            dp, ok = banded_strips(pat, txt, h, HS, G)
            k = 2 * h + len(txt) - len(pat)
            assert dp >= d, (pat, txt, h, dp, d)
            if d <= k:
                assert ok and dp == d, (pat, txt, h, HS, G, dp, d)
            else:
                assert not ok, (pat, txt, h, dp, d, k)
    return trials


if __name__ == "__main__":
    print("ok", check())
