"""queue_walk_k2 (blance_amd/csrc/k_queue_walk.h), the hand-written gfx950 assembly of the queue kernel's lean walk, executed on
the CPU from its preprocessed text (tests/gcn_vector.py) and compared, batch by batch, with a Python restatement of the C++
twin of that loop (k_pass_queue.h, "the lean walk"): same steps taken, same nodes chosen, same window, THETA, stale lanes and
LDS tables afterwards, the same step and the same code where it leaves.  Random batches cover: k = 1 and k = 2, rows with no,
few and many bits, windows that are full / short / drained, partitions without own nodes, the folded row, promotions, keys
at and beyond THETA, counters at the edge of the tables, slow and stale lanes, higher priority nodes inside the window.
The SIMT emulator runs the twin, the device runs both (tests/test_hip_parity.py); this is the check of the text itself
that needs no GPU."""
import math
import os
import random
import shutil
import struct

import numpy as np
import pytest

import gcn_vector as GV

M32, M64 = (1 << 32) - 1, (1 << 64) - 1
INT_MAX = 0x7fffffff
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def sortable(x):
    if x == 0.0:
        x = 0.0                                   # (-0 -> +0: the text's v_cmp_eq_f64 / v_cndmask pair)
    b = f64_bits(x)
    return (b ^ M64) if b >> 63 else (b | (1 << 63))


def key_of(cnt, nt_term, ff, sh):
    r = float(cnt)
    if nt_term is not None:
        r = r + nt_term
    r = r + ff
    r = math.ldexp(r, -sh)
    return sortable(r)


def lt(ka, na, kb, nb):
    return ka < kb or (ka == kb and na < nb)


class Scenario:
    """One batch: the node tables, the window, the 64 steps' data, the LDS image and the registers of the text."""

    def __init__(self, seed):
        rng = random.Random(seed)
        self.rng = rng
        self.k = rng.choice([1, 2])
        self.fold = rng.random() < 0.25
        self.B = rng.choice([64, 64, 64, 37, 5])
        N = self.N = rng.choice([96, 200, 512, 1024])
        NP = rng.choice([1000, 65536, 1048576])
        self.lp = [i / NP for i in range(512)]
        self.ff = [(0.001 * i) / NP for i in range(2048)]
        base = rng.choice([0, 3, 40, 200])
        spread = rng.choice([0, 1, 2, 6, 30])
        hi_tot = rng.random() < 0.1                                   # counters near the end of the table of (0.001 t) / NP
        ties = rng.random() < 0.35                                    # many nodes with ONE key: the node id decides everywhere
        self.cnt = [base + rng.randint(0, spread) for _ in range(N)]
        self.tot = [min(c * 3 + (0 if ties else rng.randint(0, 5)) + (2030 if hi_tot else 0), 2047) for c in self.cnt]
        self.sh = [0 if ties else rng.choice([0, 0, 0, 1, 2]) for _ in range(N)]
        self.nt = [rng.choice([0, 0, 1, 5, 60, 509, 511]) if self.fold else 0 for _ in range(N)]
        self.gB = [key_of(self.cnt[n], self.lp[self.nt[n]] if self.fold else None, self.ff[self.tot[n]], self.sh[n]) for n in range(N)]
        order = sorted(range(N), key=lambda n: (self.gB[n], n))
        wcnt = self.wcnt = rng.choice([64, 64, 64, 50, 31, 8, 1, 0])
        self.wk = [self.gB[n] for n in order[:wcnt]] + [M64] * (64 - wcnt)
        self.wn = order[:wcnt] + [INT_MAX] * (64 - wcnt)
        if wcnt < N:
            self.thK, self.thN = self.gB[order[wcnt]], order[wcnt]
        else:
            self.thK, self.thN = M64, INT_MAX
        # the steps
        k = self.k
        self.own_a, self.own_b, self.ka, self.kb, self.h0, self.w, self.ov = [], [], [], [], [], [], []
        self.lastK, self.lastN = [], []
        self.always = self.slow = self.stale = 0
        self.act = (1 << self.B) - 1
        dens = rng.choice([0.0, 0.0, 0.02, 0.3, 0.9, 0.97, 1.0])
        forced = rng.random() < 0.3                                   # lanes that hold all their nodes but failed validation
        self.bits = []
        for lane in range(64):
            nown = 0 if self.fold else rng.choice([k, k, k, k - 1, 0])
            near = order[:80] if rng.random() < 0.5 else list(range(N))
            own = rng.sample(near, nown)
            keys = []
            for n in own:                                             # exact scores: at or above the node's g
                bump = rng.choice([0, 0, 0, 1, 3, 40])
                keys.append(key_of(self.cnt[n], self.lp[bump], self.ff[self.tot[n]], self.sh[n]) if bump else self.gB[n])
            pairs = sorted(zip(keys, own))
            oa, ob = (pairs[0][1] if nown > 0 else -1), (pairs[1][1] if nown > 1 else -1)
            ka, kb = (pairs[0][0] if nown > 0 else M64), (pairs[1][0] if nown > 1 else M64)
            self.own_a.append(oa); self.own_b.append(ob); self.ka.append(ka); self.kb.append(kb)
            if nown == k and k > 0:
                lk, ln = pairs[k - 1]
                if forced and rng.random() < 0.5:
                    self.always |= 1 << lane
            else:
                lk, ln = M64, INT_MAX
                self.always |= 1 << lane
            self.lastK.append(lk); self.lastN.append(ln)
            self.h0.append(rng.choice(order[:70]) if rng.random() < 0.3 else -1)
            self.w.append(rng.choice([1, 1, 2, 5, 8]))
            lows = [rng.choice(order[:70]) if rng.random() < 0.15 else 0xffff for _ in range(2)]
            self.ov.append((lows[0] & 0xffff) | ((lows[1] if lows[1] != 0xffff else -1) << 16))
            if rng.random() < 0.03:
                self.slow |= 1 << lane
            if rng.random() < 0.02:
                self.stale |= 1 << lane
            row = 0
            for n in range(N):
                if rng.random() < dens:
                    row |= 1 << n
            self.bits.append(row)
        if rng.random() < 0.08:                                       # a batch in which every step keeps its nodes
            self.always = self.slow = self.stale = 0
            self.lastK = [0] * 64
        self.cur = rng.choice([0, 0, 0, 3, 20])
        self.o1 = [rng.randint(0, N) for _ in range(64)]
        self.o2 = [rng.randint(0, N) for _ in range(64)]

    # ---- the C++ twin of the loop, restated
    def model(self):
        k, B, fold = self.k, self.B, self.fold
        wk, wn, wcnt = list(self.wk), list(self.wn), self.wcnt
        thK, thN = self.thK, self.thN
        cnt, tot, nt, gB = list(self.cnt), list(self.tot), list(self.nt), list(self.gB)
        stale, moved, cur = self.stale, 0, self.cur
        o1, o2 = list(self.o1), list(self.o2)
        code = None
        while True:
            if cur >= B:
                code = 0
                break
            fK, fN = (wk[0], wn[0]) if wcnt > 0 else (thK, thN)
            notstay = 0
            for l in range(64):
                if self.lastK[l] > fK or (self.lastK[l] == fK and self.lastN[l] >= fN):
                    notstay |= 1 << l
            notstay = (notstay | self.always | stale) & self.act & (M64 << cur) & M64
            if not notstay:
                cur, code = B, 0
                break
            f = (notstay & -notstay).bit_length() - 1
            if ((self.slow | stale) >> f) & 1:
                cur, code = f, 1
                break
            w, oa, ob, hh = self.w[f], self.own_a[f], self.own_b[f], self.h0[f]
            ka, kb = self.ka[f], self.kb[f]
            elig = [l < wcnt and wn[l] != oa and wn[l] != ob and wn[l] != hh for l in range(64)]
            dirty = [elig[l] and not fold and ((self.bits[f] >> (wn[l] & 0xfff)) & 1) == 1 for l in range(64)]
            clean = [l for l in range(64) if elig[l] and not dirty[l]]
            if len(clean) < k:
                cur, code = f, (2 if wcnt < 32 else 1)
                break
            c1, c2 = clean[0], clean[k - 1]
            dl = [l for l in range(64) if dirty[l] and l <= c2]
            if dl:
                if any(tot[wn[l]] >= 0x800 for l in dl):
                    cur, code = f, 2
                    break
                ckey = wk[c2]
                if any(ckey >= key_of(cnt[wn[l]], self.lp[1], self.ff[tot[wn[l]]], self.sh[wn[l]]) for l in dl):
                    cur, code = f, 1
                    break
            t1, t2 = (wk[c1], wn[c1]), (wk[c2], wn[c2])
            na, nb = (INT_MAX if oa < 0 else oa), (INT_MAX if ob < 0 else ob)
            lv1 = lv2 = en1 = en2 = -1
            if k == 2:
                if lt(ka, na, *t1):
                    if lt(kb, nb, *t1):
                        cur = f + 1
                        continue
                    r1, r2, en1, lv1, last = oa, t1[1], t1[1], ob, t1
                elif lt(ka, na, *t2):
                    r1, r2, en1, lv1, last = t1[1], oa, t1[1], ob, (ka, na)
                else:
                    r1, r2, en1, en2, lv1, lv2, last = t1[1], t2[1], t1[1], t2[1], oa, ob, t2
            else:
                if lt(ka, na, *t1):
                    cur = f + 1
                    continue
                r1, r2, en1, lv1, last = t1[1], -1, t1[1], oa, t1
            if not lt(last[0], last[1], thK, thN):
                cur, code = f, 2
                break
            l0 = self.ov[f] & 0xffff
            l0 = l0 - 0x10000 if l0 >> 15 else l0
            l1 = self.ov[f] >> 16                                      # (arithmetic: -1 = none)
            if en1 in (l0, l1) or (en2 >= 0 and en2 in (l0, l1)):
                cur, code = f, 2
                break
            chg = [lv1, lv2, en1, en2]
            newc, newt, newn, newk = {}, {}, {}, {}
            bail = False
            for j, x in enumerate(chg):
                if x < 0:
                    continue
                d = -w if j < 2 else w
                newc[j], newt[j] = cnt[x] + d, tot[x] + d
                if (newt[j] & M32) >= 0x800:
                    bail = True
            if not bail and fold:
                for j, x in enumerate(chg):
                    if x >= 0:
                        newn[j] = nt[x] + (1 if j >= 2 else 0)
                        if newn[j] >= 0x200:
                            bail = True
            if bail:
                cur, code = f, 2
                break
            for j, x in enumerate(chg):
                if x < 0:
                    continue
                newk[j] = key_of(newc[j], self.lp[newn[j]] if fold else None, self.ff[newt[j]], self.sh[x])
            for j, x in enumerate(chg):
                if x < 0:
                    continue
                cnt[x], tot[x], gB[x] = newc[j], newt[j], newk[j]
                if fold:
                    nt[x] = newn[j]
            for j, x in enumerate(chg):
                if x < 0:
                    continue
                if x in wn[:64]:
                    p = wn.index(x)
                    wk[p:] = wk[p + 1:] + [M64]
                    wn[p:] = wn[p + 1:] + [INT_MAX]
                    wcnt -= 1
                if lt(newk[j], x, thK, thN):
                    less = sum(1 for l in range(64) if lt(wk[l], wn[l], newk[j], x))
                    if less >= 64:
                        thK, thN = newk[j], x
                    else:
                        if wcnt >= 64:
                            thK, thN = wk[63], wn[63]
                            wcnt -= 1
                        wk[less + 1:] = wk[less:63]
                        wn[less + 1:] = wn[less:63]
                        wk[less], wn[less] = newk[j], x
                        wcnt += 1
                for l in range(64):
                    if self.own_a[l] == x or self.own_b[l] == x:
                        stale |= 1 << l
            o1[f], o2[f] = r1, r2
            moved |= 1 << f
            cur = f + 1
        return dict(code=code, cur=cur, wcnt=wcnt, wk=wk, wn=wn, thK=thK, thN=thN, stale=stale, moved=moved, o1=o1, o2=o2,
                    cnt=[c & M32 for c in cnt], tot=[t & M32 for t in tot], nt=nt, gB=gB)      # (as the 32-bit words of the LDS image)

    # ---- the text, interpreted
    def run_text(self, prog):
        wv = GV.Wave()
        N = self.N
        NXp = (N + 63) & ~63
        BW = ((NXp >> 5) + 3) & ~3
        at = 0

        def place(nbytes, align=16):
            nonlocal at
            at = (at + align - 1) & ~(align - 1)
            a = at
            at += nbytes
            return a
        gB_o, cnt_o, tot_o = place(8 * NXp), place(4 * NXp), place(4 * NXp)
        lp_o, ff_o = place(8 * 512), place(8 * 2048)
        bits_o = place(4 * 64 * BW)
        sh_o, nt_o = place(NXp), place(2 * NXp)
        for n in range(N):
            wv.lds_wr(gB_o + 8 * n, self.gB[n], 8)
            wv.lds_wr(cnt_o + 4 * n, self.cnt[n], 4)
            wv.lds_wr(tot_o + 4 * n, self.tot[n], 4)
            wv.lds_wr(sh_o + n, self.sh[n], 1)
            wv.lds_wr(nt_o + 2 * n, self.nt[n], 2)
        for i in range(512):
            wv.lds_wr(lp_o + 8 * i, f64_bits(self.lp[i]), 8)
        for i in range(2048):
            wv.lds_wr(ff_o + 8 * i, f64_bits(self.ff[i]), 8)
        for f in range(64):
            for wd in range(BW):
                wv.lds_wr(bits_o + 4 * (f * BW + wd), (self.bits[f] >> (32 * wd)) & M32, 4)

        def setv(r, vals):
            wv.v[r] = np.array([v & M32 for v in vals], dtype=np.uint32)

        def setv64(r, vals):
            setv(r, [v & M32 for v in vals])
            setv(r + 1, [(v >> 32) & M32 for v in vals])

        def sets64(r, v):
            wv.s[r], wv.s[r + 1] = v & M32, (v >> 32) & M32
        setv64(200, self.wk); setv(202, self.wn); setv(203, self.o1); setv(204, self.o2)
        setv(205, self.lastN); setv64(206, self.lastK); setv(208, self.own_a); setv(209, self.own_b)
        setv64(210, self.ka); setv64(212, self.kb); setv(214, self.h0); setv(215, self.w); setv(216, self.ov)
        setv(217, list(range(64)))
        wv.s[40], wv.s[41] = self.cur, self.wcnt
        sets64(42, self.thK); wv.s[44] = self.thN & M32
        sets64(46, self.stale); sets64(48, 0); sets64(50, self.always); sets64(52, self.slow); sets64(54, self.act)
        wv.s[56] = (cnt_o >> 2) | ((tot_o >> 2) << 16)
        wv.s[57] = (sh_o >> 2) | ((ff_o >> 2) << 16)
        wv.s[58] = (bits_o >> 2) | ((BW * 4) << 16)
        wv.s[59] = (gB_o >> 2) | (self.B << 16) | ((1 if self.k == 2 else 0) << 24) | ((1 if self.fold else 0) << 25)
        wv.s[45] = (nt_o >> 2) | ((lp_o >> 2) << 16)
        sets64(38, f64_bits(self.lp[1]))
        wv.s[60] = 0xdead
        wv.run(prog)
        g64 = lambda r: [int(wv.v[r][i]) | (int(wv.v[r + 1][i]) << 32) for i in range(64)]
        s32 = lambda x: x - (1 << 32) if x >> 31 else x
        return dict(code=wv.s[60], cur=wv.s[40], wcnt=wv.s[41], wk=g64(200), wn=[s32(int(x)) for x in wv.v[202]],
                    thK=wv.s[42] | (wv.s[43] << 32), thN=s32(wv.s[44]), stale=wv.s[46] | (wv.s[47] << 32),
                    moved=wv.s[48] | (wv.s[49] << 32), o1=[s32(int(x)) for x in wv.v[203]], o2=[s32(int(x)) for x in wv.v[204]],
                    cnt=[wv.lds_rd(cnt_o + 4 * n, 4) for n in range(N)], tot=[wv.lds_rd(tot_o + 4 * n, 4) for n in range(N)],
                    nt=[wv.lds_rd(nt_o + 2 * n, 2) for n in range(N)], gB=[wv.lds_rd(gB_o + 8 * n, 8) for n in range(N)],
                    executed=wv.executed)


@pytest.fixture(scope="module")
def program():
    if not (shutil.which(HIPCC) or os.path.exists(HIPCC)):
        pytest.skip("needs hipcc to preprocess the translation unit")
    return GV.queue_walk_program()


def _compare(sc, got, want):
    wc = want["wcnt"]
    for key in ("code", "cur", "wcnt", "thK", "thN", "stale", "moved", "cnt", "tot", "gB"):
        assert got[key] == want[key], (key, got[key] if not isinstance(got[key], list) else "...", want[key] if not isinstance(want[key], list) else "...")
    assert got["wk"][:wc] == want["wk"][:wc] and got["wn"][:wc] == want["wn"][:wc]
    if sc.fold:
        assert got["nt"] == want["nt"]
    for l in range(64):
        if (want["moved"] >> l) & 1:
            assert got["o1"][l] == want["o1"][l], l
            if sc.k == 2:
                assert got["o2"][l] == want["o2"][l], l
        else:                                           # lanes that did not move keep what the registers held
            assert got["o1"][l] == sc.o1[l] and got["o2"][l] == sc.o2[l], l


@pytest.mark.parametrize("block", range(4))
def test_text_equals_twin_on_random_batches(program, block):
    codes, steps = {}, 0
    for seed in range(block * 100, block * 100 + 100):
        sc = Scenario(seed)
        want = sc.model()
        got = sc.run_text(program)
        try:
            _compare(sc, got, want)
        except AssertionError as e:
            raise AssertionError("seed %d (k %d fold %s B %d wcnt %d): %s" % (seed, sc.k, sc.fold, sc.B, sc.wcnt, e))
        codes[want["code"]] = codes.get(want["code"], 0) + 1
        steps += bin(want["moved"]).count("1")
    assert steps > 100 and len(codes) >= 2, (codes, steps)
