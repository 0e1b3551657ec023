"""Parity (MI355X): the device lextree search (s3a_lexsearch_*, cmusphinx_amd/csrc/s3a_lextree.hip)
against the oracle and the recorded reference-grade trace, operation by operation: root entry,
HMM evaluation, ORDER-PRESERVING phone-level propagation (active-list order, tie-breaks,
clear-before/after-entry), word exits, active-senone marking, list swap -- and the fused
single-synchronisation frame (s3a_lexsearch_frame_search) against the step-by-step calls."""
import os

import numpy as np
import pytest

import lextree_trace
import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def load():
    return lextree_trace.from_npz(np.load(os.path.join(GOLDEN, "lextree_trace_tidigits.npz")))


def make_gpu(gpu_lib, tr, stream=None):
    ne = tr["n_emit"]
    tm = gpu_lib.Tmat.init_logs3(np.asarray(tr["tp"], np.int32).reshape(tr["n_tmat"], ne, ne + 1))
    return gpu_lib.LexSearch(tr["trees"], tm, tr["sseq"], tr["comsseq"], tr["comstate_off"], tr["comstate"],
                             tr["n_sen"], stream=stream)


def test_gpu_replays_the_recorded_trace(gpu_lib):
    tr = load()
    ls = make_gpu(gpu_lib, tr)
    n_frames, n_exits = lextree_trace.Replayer(tr, ls).run()
    assert n_frames == 60 and n_exits > 20


class Lockstep:
    """Feed identical inputs to the oracle and the GPU and compare everything after each op."""

    def __init__(self, a, b, T):
        self.a, self.b, self.T = a, b, T

    def same(self, label, lists=(0, 1)):
        for t in range(self.T):
            for which in lists:
                assert np.array_equal(self.a.active(t, which), self.b.active(t, which)), (label, t, which)
            assert np.array_equal(self.a.state(t), self.b.state(t)), (label, t)


def synth_forest(rng, n_tree=4, n_node=700, n_sseq=300, n_sen=900, n_lc=9):
    """Random lextrees with the reference's shape: several left-context root lists sharing
    root nodes, first-level nodes with SEVERAL parents, leaves carrying word ids."""
    trees = []
    for _ in range(n_tree):
        n_root = 24
        parent_of = [[] for _ in range(n_node)]
        children = [[] for _ in range(n_node)]
        lvl1 = list(range(n_root, n_root + 60))
        for c in lvl1:                                   # multi-parent first level
            for p in rng.choice(n_root, size=rng.integers(1, 5), replace=False):
                children[int(p)].append(c)
        for c in range(n_root + 60, n_node):             # single-parent below
            p = int(rng.integers(n_root, c))
            children[p].append(c)
        wid = np.full(n_node, -1, np.int32)
        leaf = [v for v in range(n_node) if not children[v]]
        wid[leaf] = rng.integers(0, 500, len(leaf))
        prob = -rng.integers(0, 40000, n_node).astype(np.int32)
        off = np.zeros(n_node + 1, np.int32)
        flat = []
        for v in range(n_node):
            off[v] = len(flat); flat += children[v]
        off[n_node] = len(flat)
        lc = np.arange(n_lc, dtype=np.int16)
        lro = [0]; lr = []
        for _ in range(n_lc):
            lst = rng.permutation(n_root)[: rng.integers(3, n_root)]
            lr += [int(v) for v in lst]; lro.append(len(lr))
        trees.append(dict(n_node=n_node, n_lc=n_lc, n_root=n_root, type=0,
                          ssid=rng.integers(0, n_sseq, n_node).astype(np.int32),
                          tmatid=rng.integers(0, 5, n_node).astype(np.int32),
                          composite=(rng.random(n_node) < 0.2).astype(np.uint8), wid=wid, prob=prob,
                          child_off=off, child=np.array(flat, np.int32), lc=lc,
                          lcroot_off=np.array(lro, np.int32), lcroot=np.array(lr, np.int32),
                          root=np.arange(n_root, dtype=np.int32)))
    tp = np.full((5, 3, 4), O.LOGPROB_ZERO, np.int32)
    for i in range(3):
        tp[:, i, i] = -rng.integers(300, 6000, 5); tp[:, i, i + 1] = -rng.integers(300, 6000, 5)
    n_comstate = 40
    coff = np.concatenate([[0], np.cumsum(rng.integers(1, 6, n_comstate))]).astype(np.int32)
    return dict(n_tree=n_tree, n_emit=3, n_tmat=5, n_sseq=n_sseq, n_comsseq=n_sseq, n_comstate=n_comstate,
                n_sen=n_sen, tp=tp.ravel(), sseq=rng.integers(0, n_sen, n_sseq * 3).astype(np.int16),
                comsseq=rng.integers(0, n_comstate, n_sseq * 3).astype(np.int16), comstate_off=coff,
                comstate=rng.integers(0, n_sen, coff[-1]).astype(np.int16), trees=trees, events=[])


@pytest.mark.parametrize("seed,coarse", [(1, 1), (2, 64), (3, 4096)])
def test_random_forest_lockstep_with_oracle(gpu_lib, seed, coarse):
    """Bigger trees, multi-parent nodes, many exact score TIES (scores quantised by `coarse`),
    several root-entry calls per frame sharing root nodes, narrow and wide beams."""
    rng = np.random.default_rng(seed)
    tr = synth_forest(rng)
    a, b = O.OracleLexSearch(tr), make_gpu(gpu_lib, tr)
    ls = Lockstep(a, b, tr["n_tree"])
    q = lambda v: (np.asarray(v) // coarse) * coarse
    for x in (a, b):
        x.enter(0, [0, 3], [0, 0], [7, 8], -1, -10**8)
        x.enter(2, [1], [0], [9], -1, -10**8)
        x.swap()
    ls.same("begin")
    for frm in range(40):
        senscr = q(-rng.integers(0, 30000, tr["n_sen"])).astype(np.int32)
        comsen = q(-rng.integers(0, 30000, tr["n_comstate"])).astype(np.int32)
        assert np.array_equal(a.sen_active(), b.sen_active()), frm
        ra, rb = a.hmm_eval(senscr, comsen, frm), b.hmm_eval(senscr, comsen, frm)
        assert all(np.array_equal(u, v) for u, v in zip(ra, rb)), frm
        ls.same(("eval", frm))
        best = int(max(ra[0].max(), -2**31 + 1))
        beam = int(rng.choice([-30000, -80000, -200000]))
        th, pth, wth = best + beam, best + beam // 2, int(max(ra[1].max(), -2**31 + 1)) + beam // 2
        a.propagate(frm, th, pth, wth); b.propagate(frm, th, pth, wth)
        ls.same(("propagate", frm))
        ea, eb = a.leaves(wth), b.leaves(wth)
        for t in range(tr["n_tree"]):
            assert all(np.array_equal(u, v) for u, v in zip(ea[t], eb[t])), (frm, t)
        # word transitions: several calls into one tree, shared roots, equal scores on purpose
        k = frm % tr["n_tree"]
        n = int(rng.integers(1, 6))
        lcs = rng.choice(9, n, replace=False)
        scr = q(best - rng.integers(0, 60000, n)).astype(np.int32)
        if n > 2:
            scr[1] = scr[0]
        hist = rng.integers(0, 10**6, n).astype(np.int32)
        for x in (a, b):
            x.enter(k, lcs, scr, hist, frm, best + beam)
        ls.same(("enter", frm))
        a.swap(); b.swap()
        ls.same(("swap", frm))
    assert sum(len(a.active(t, 0)) for t in range(tr["n_tree"])) > 50


def test_fused_frame_search_equals_stepwise(gpu_lib):
    rng = np.random.default_rng(11)
    tr = synth_forest(rng)
    a, b = make_gpu(gpu_lib, tr), make_gpu(gpu_lib, tr)
    ls = Lockstep(a, b, tr["n_tree"])
    for x in (a, b):
        x.enter(1, [2, 5], [0, -100], [1, 2], -1, -10**8); x.swap()
    hmmbeam, pbeam, wbeam = -150000, -120000, -90000
    for frm in range(25):
        senscr = -rng.integers(0, 30000, tr["n_sen"]).astype(np.int32)
        comsen = -rng.integers(0, 30000, tr["n_comstate"]).astype(np.int32)
        best, wbest, nact = a.hmm_eval(senscr, comsen, frm)
        bh, bw = int(best.max()), int(wbest.max())
        w32 = lambda v: ((v + 2**31) % 2**32) - 2**31       # int32 wrap-around, as in the reference's C
        a.propagate(frm, w32(bh + hmmbeam), w32(bh + pbeam), w32(bw + wbeam))
        ea = a.leaves(w32(bw + wbeam))
        res, eb = b.frame_search(senscr, comsen, frm, hmmbeam, pbeam, wbeam)
        assert (res.best_hmm, res.best_word, res.n_hmm) == (bh, bw, int(nact.sum()))
        assert (res.thres, res.phone_thres, res.word_thres) == (w32(bh + hmmbeam), w32(bh + pbeam), w32(bw + wbeam))
        for t in range(tr["n_tree"]):
            assert all(np.array_equal(u, v) for u, v in zip(ea[t], eb[t])), (frm, t)
        ls.same(("fused", frm))
        for x in (a, b):
            x.enter(frm % 4, [frm % 9], [bh - 5000], [frm], frm, bh + hmmbeam); x.swap()


def test_histbin_lockstep_with_oracle(gpu_lib):
    """lextree_hmm_histbin: same bins and the same REORDERED active lists (bins ascending, reverse
    list order inside a bin), on lists long enough to span several 1024-element passes."""
    rng = np.random.default_rng(21)
    tr = synth_forest(rng, n_tree=3, n_node=5000)
    a, b = O.OracleLexSearch(tr), make_gpu(gpu_lib, tr)
    ls = Lockstep(a, b, tr["n_tree"])
    for x in (a, b):
        for t in range(3):
            x.enter(t, list(range(9)), [0] * 9, list(range(9)), -1, -10**9)
        x.swap()
    for frm in range(16):
        senscr = (-rng.integers(0, 3000, tr["n_sen"]) * 16).astype(np.int32)      # many equal scores
        comsen = (-rng.integers(0, 3000, tr["n_comstate"]) * 16).astype(np.int32)
        ra, rb = a.hmm_eval(senscr, comsen, frm), b.hmm_eval(senscr, comsen, frm)
        best = int(ra[0].max())
        for nbin, bw in ((1000, 97), (1000, 5000), (37, 1500)):
            ba, bb = np.zeros(nbin, np.int32), np.zeros(nbin, np.int32)
            for t in range(3):
                a.histbin(t, best, ba, bw); b.histbin(t, best, bb, bw)
            assert np.array_equal(ba, bb) and ba.sum() == ra[2].sum(), (frm, nbin, bw)
            ls.same(("histbin", frm, nbin, bw))
        a.propagate(frm, best - 10**7, best - 10**7, best - 10**7); b.propagate(frm, best - 10**7, best - 10**7, best - 10**7)
        ls.same(("propagate", frm))
        a.swap(); b.swap()
    assert max(len(a.active(t, 0)) for t in range(3)) > 1500     # more than one 1024-element pass


class OracleFrame:
    """The reference's frame (srch_utt_decode_blk's body for mode 4) on the CPU oracle: active
    senones -> CI + gated CD scoring -> composite senones -> HMM evaluation -> beams (with
    histogram pruning, srch_time_switch_tree.c:849-905) -> propagation -> word exits."""

    def __init__(self, tr, om, cd2cisen, n_ci, ci_pbeam, comwt):
        self.tr, self.lex = tr, O.OracleLexSearch(tr)
        self.fs = O.OracleFrameScorer(om, cd2cisen, n_ci, ci_pbeam)
        self.comwt = comwt

    def frame(self, feat, frm, hmmbeam, pbeam, wbeam, maxhmmpf, phone_uses_wbeam=0):
        tr, lex = self.tr, self.lex
        sa = lex.sen_active()
        best, ns, ng, cin, cig, cib = self.fs.step(feat, frm, sa)
        comsen = O.comsenscr(tr["comstate_off"], tr["comstate"], self.comwt, self.fs.senscr)
        b, w, n = lex.hmm_eval(self.fs.senscr, comsen, frm)
        bh, bw_, nh = int(b.max()), int(w.max()), int(n.sum())
        hb, pb, wb, hist = hmmbeam, pbeam, wbeam, False
        if nh > maxhmmpf + (maxhmmpf >> 1):
            hist = True
            width = -hmmbeam // 1000
            bins = np.zeros(1000, np.int32)
            for t in range(lex.T):
                lex.histbin(t, bh, bins, width)
            i = j = 0
            while i < 1000 and j < maxhmmpf:
                i += 1
                if i < 1000:
                    j += int(bins[i])
            hb = -(i * width); pb = max(hb, pbeam); wb = max(hb, wbeam)
        w32 = lambda v: ((v + 2**31) % 2**32) - 2**31       # int32 wrap-around, as in the reference's C
        th, pth, wth = w32(bh + hb), w32(bh + pb), w32(bw_ + wb)
        if phone_uses_wbeam:            # srch_time_switch_tree.c:975-1003: -ptranskip frames
            pth = wth
        lex.propagate(frm, th, pth, wth)
        return dict(best=best, counts=(ns, ng, cin, cig, cib), bh=bh, bw=bw_, n=nh, th=th, pth=pth, wth=wth,
                    hist=hist, exits=lex.leaves(wth))


@pytest.mark.parametrize("seed,maxhmmpf,ci_pbeam,pbeam,ptranskip",
                         [(5, 20000, 1e-80, -2000000, 0), (6, 150, 1e-80, -2000000, 0), (7, 400, 1e-12, -2000000, 0),
                          (8, 20000, 1e-80, -3400000, 0),       # phone beam WIDER than the HMM beam
                          (9, 20000, 1e-80, -2000000, 2),       # every 2nd frame: word threshold for phones
                          (10, 300, 1e-80, -3400000, 3),
                          (12, 20000, 1e-80, -2000000, 0),      # 10 000 nodes: lists of several 1024-position chunks
                          (13, 2500, 1e-80, -2000000, 0)])      # the same with histogram pruning (reordered lists)
def test_fused_decoder_frame_lockstep_with_oracle(gpu_lib, seed, maxhmmpf, ci_pbeam, pbeam, ptranskip, variants):
    """The product path of a mode-4 frame -- s3a_decoder_score / _search / _transition -- against
    the oracle's step-by-step frame on a synthetic forest WITH a synthetic acoustic model: raw
    scores normalised inside the search kernels, inline composite senones, histogram pruning,
    two-tree transitions, next-frame senone marks consumed by the gated scorer.
    (seed 7 also takes the transition's copy path: the calls through device memory instead of the kernel
    arguments, which a frame with more than 96 lextree_enter calls would use.)"""
    force = {}
    if seed == 7:
        force["calls_by_copy"] = 1
    big = seed >= 12
    if big:     # k_dec_scan's chained multi-workgroup path (used from 16 k list positions on) on lists of 2-3 chunks
        force["scan_chained"] = 1
    variants(**force)
    from cmusphinx_amd import synth
    rng = np.random.default_rng(seed)
    tr = synth_forest(rng, n_tree=4, n_node=10000 if big else 900, n_sen=600)
    n_ci = 30
    m = synth.make_model(600, n_ci, 4, 39, 5, 3, seed=seed + 100)
    feats = synth.make_features(m, 45, seed=seed + 200)
    comwt = -rng.integers(0, 3000, tr["n_comstate"]).astype(np.int32)
    olm = O.OracleLogMath(1.0003)
    om = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    of = OracleFrame(tr, om, m["cd2cisen"], n_ci, olm.logs3(ci_pbeam), comwt)
    glm = gpu_lib.LogMath(1.0003)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], glm)
    sc = gpu_lib.Scorer(gm, m["cd2cisen"], n_ci, ci_pbeam=ci_pbeam)
    cs = gpu_lib.ComSen(tr["comstate_off"], tr["comstate"], comwt)
    with pytest.raises(gpu_lib.S3AError, match="must share one stream"):
        make_gpu(gpu_lib, tr).decoder_utt_begin(sc)
    ls = make_gpu(gpu_lib, tr, stream=gm.stream())
    lock = Lockstep(of.lex, ls, tr["n_tree"])
    hmmbeam, wbeam = (-6000000 if big else -2600000), -1500000
    longest = 0

    ls.decoder_utt_begin(sc)
    first = (0, [0, 3, 4], [0, 0, -50], [7, 8, 9]), (2, [1], [0], [9])
    for g in first:
        of.lex.enter(g[0], g[1], g[2], g[3], -1, -10**9)
    of.lex.swap()
    ls.decoder_transition(sc, cs, -1, -10**9, *first)
    # (the fused transition leaves the emptied next-list counter to be overwritten by the coming
    # frame's search instead of zeroing it: compare the active lists and every HMM)
    lock.same("begin", lists=(0,))
    n_hist = 0
    for frm in range(len(feats)):
        uw = 1 if (ptranskip and frm % ptranskip == 0) else 0
        o = of.frame(feats[frm], frm, hmmbeam, pbeam, wbeam, maxhmmpf, uw)
        ls.decoder_score(sc, feats[frm], frm)
        res, exits = ls.decoder_search(sc, cs, frm, hmmbeam, pbeam, wbeam, uw, maxhmmpf)
        assert (res.best_hmm, res.best_word, res.n_hmm) == (o["bh"], o["bw"], o["n"]), frm
        assert (res.thres, res.phone_thres, res.word_thres) == (o["th"], o["pth"], o["wth"]), frm
        assert bool(res.need_histprune) == o["hist"], frm
        ns, ng, cin, cig, cib = o["counts"]
        assert tuple(res.extra[1:7]) == (ns, ng, cin, cig, cib, o["best"]), (frm, list(res.extra), o["counts"], o["best"])
        for t in range(tr["n_tree"]):
            assert all(np.array_equal(u, v) for u, v in zip(o["exits"][t], exits[t])), (frm, t)
        n_hist += o["hist"]
        longest = max(longest, res.n_hmm)
        # word transitions: a unigram-tree batch (sometimes empty) and a filler-tree call
        k = frm % 2
        n = int(rng.integers(0, 5))
        ga = (k, rng.choice(9, n, replace=False), (o["bh"] - rng.integers(0, 900000, n)).astype(np.int32),
              rng.integers(0, 10**6, n).astype(np.int32)) if n else None
        gb = (2 + k, [int(rng.integers(0, 9))], [o["bh"] - 1000], [frm]) if frm % 3 else None
        for g in (ga, gb):
            if g is not None:
                of.lex.enter(g[0], g[1], g[2], g[3], frm, o["bh"] + hmmbeam)
        of.lex.swap()
        ls.decoder_transition(sc, cs, frm, o["bh"] + hmmbeam, ga, gb)
        lock.same(("frame", frm), lists=(0,))
    assert o["n"] > 50
    if seed == 13:
        assert n_hist >= 3          # (1.5 x 2500 HMMs is reached in the busiest frames only)
    else:
        assert (n_hist > 10) == (maxhmmpf < 1000)
    if seed == 12:
        assert longest > 4 * 1024 + 500, longest       # four trees: one of them held more than one chunk
