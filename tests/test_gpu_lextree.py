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


def make_gpu(gpu_lib, tr):
    ne = tr["n_emit"]
    tm = gpu_lib.Tmat.init_logs3(np.asarray(tr["tp"], np.int32).reshape(tr["n_tmat"], ne, ne + 1))
    return gpu_lib.LexSearch(tr["trees"], tm, tr["sseq"], tr["comsseq"], tr["comstate_off"], tr["comstate"], tr["n_sen"])


def test_gpu_replays_the_recorded_trace(gpu_lib):
    tr = load()
    ls = make_gpu(gpu_lib, tr)
    n_frames, n_exits = lextree_trace.Replayer(tr, ls).run()
    assert n_frames == 60 and n_exits > 20


class Lockstep:
    """Feed identical inputs to the oracle and the GPU and compare everything after each op."""

    def __init__(self, a, b, T):
        self.a, self.b, self.T = a, b, T

    def same(self, label):
        for t in range(self.T):
            for which in (0, 1):
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
