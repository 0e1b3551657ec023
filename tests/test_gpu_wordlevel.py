"""The WORD LEVEL on the MI355X (s3a_wordlevel.h through s3a_wltest_*) against the oracle and the recorded
reference-grade traces, frame by frame: trigram look-ups, the history entries a frame leaves (ids, order, every
field), the frame's best exit and the lextree_enter calls of srch_utt_word_trans.  Bit-exact.

  * replay of the tidigits / RM1 traces (real dictionary, real trigram read from the DMP file);
  * random frames on a coarse score grid -- tied scores everywhere, so the heap's pop order decides what
    survives -- under tight -maxwpf / -maxhistpf, -bghist and a word-end beam.
"""
import os

import numpy as np
import pytest

import oracle_wordlevel as OW
import wordlevel_trace as WT
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FIELDS = ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")
HMMBEAM = -1000000


def run_lockstep(gpu_lib, t, frames, tree_type, **kw):
    """frames: iterable of (trees, prune_beam, thresh) -> drives the device and the oracle side by side"""
    ow = OW.OracleWordLevel(t, max_frames=4096, wordend=kw.get("wordend"))
    wl = gpu_lib.WlTest(t, tree_type, **kw)
    n_calls = 0
    for frm, (trees, prune_beam, thresh) in enumerate(frames):
        if callable(trees):
            trees = trees(ow, frm)
        r = ow.frame(frm, trees, prune_beam, maxwpf=kw.get("maxwpf"), maxhist=kw.get("maxhist"))
        calls, th, n_ent = wl.frame(trees, thresh - HMMBEAM, 0, prune_beam)
        assert th == thresh
        if r["calls"] is None:
            assert len(calls) == 0, frm
        else:
            lc, cs, ch = r["calls"]
            assert len(calls) == len(lc) + 1, frm
            assert np.array_equal(calls[:-1, 0], cs) and np.array_equal(calls[:-1, 1], ch), frm
            assert (calls[-1, 0], calls[-1, 1]) == r["fill"], frm
            assert n_ent == 3 * len(calls)
            n_calls += len(lc)
    a, b = ow.table(), wl.table()
    assert len(a["score"]) == len(b["score"])
    for k in FIELDS + ("frame_start", "bestscore", "bestvh"):
        assert np.array_equal(a[k], b[k]), k
    return a, b, n_calls


@pytest.mark.parametrize("name", ["tidigits", "rm1"])
def test_device_word_level_replays_the_recorded_trace(gpu_lib, name):
    tr = WT.from_npz(np.load(os.path.join(GOLDEN, f"wordlevel_{name}.npz")))
    fr = [([(t["type"], t["wid"], t["scr"], t["hist"]) for t in f["trees"]], f["prune_beam"], f["thresh"]) for f in tr["frames"]]
    a, b, n_calls = run_lockstep(gpu_lib, tr, fr, tr["tree_type"], cap=1 << 18, cand_cap=1 << 17)
    # ... and the recorded results themselves (the reference's, not just the oracle's)
    rec = {k: np.concatenate([f["res"][k] for f in tr["frames"]]) for k in ("wid", "score", "pred", "lw0", "lw1", "ascr", "lscr", "sf", "type")}
    for k, v in rec.items():
        assert np.array_equal(b[k][1:], v), k
    assert n_calls > 50 and len(b["score"]) > 100


def test_device_trigram_matches_the_oracle_on_the_rm1_lm(gpu_lib):
    tr = WT.from_npz(np.load(os.path.join(GOLDEN, "wordlevel_rm1.npz")))
    ow = OW.OracleWordLevel(tr)
    lm = gpu_lib.Lm3g(tr)               # (the host copy; the device copy is what the replay above exercises)
    rng = np.random.default_rng(3)
    for _ in range(20000):
        l1, l2, l3 = (int(x) for x in rng.integers(0, tr["n_ug"], 3))
        if rng.random() < 0.05:
            l1 = -1
        assert lm.tg_score(l1, l2, l3) == ow.tg_score(l1, l2, l3)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(maxwpf=2, maxhist=5)), (3, dict(maxwpf=3, maxhist=40, wordend=-2500)),
                                     (4, dict(maxwpf=50, maxhist=3)), (5, dict(maxwpf=6, maxhist=9))])
def test_random_frames_with_tied_scores_lockstep(gpu_lib, seed, kw):
    rng = np.random.default_rng(seed)
    t = OW.random_task(rng)
    if seed == 5:
        t["bghist"] = 1
    tree_type = [0, 0, 0, -1, -1, -1]
    fr = [((lambda ow, frm, r=rng: OW.random_frame(r, ow, frm)), int(rng.integers(-6, -1)) * 1000, -123456 - 7 * i) for i in range(160)]
    a, b, n_calls = run_lockstep(gpu_lib, t, fr, tree_type, **kw)
    assert b["n_tie_frames"] > 20 and len(b["score"]) > 150 and n_calls > 100


@pytest.mark.parametrize("seed,grid,kw", [(11, 1, dict(maxwpf=5, maxhist=12)), (12, 1, dict(maxwpf=40, maxhist=30)),
                                          (13, 10, dict(maxwpf=7, maxhist=100)), (14, 1, dict(maxwpf=6, maxhist=9))])
def test_large_frames_prune_by_selection_lockstep(gpu_lib, seed, grid, kw):
    """hundreds of entries above the pruning threshold: the device prunes by selection (best filler, the maxwpf best
    words, the maxhist best of their entries) and replays the heap only when a cut falls on a tie (grid 10: often)."""
    rng = np.random.default_rng(seed)
    t = OW.random_task(rng, n_word=500, n_ci=12, density=0.04, grid=grid)
    t["wbeam"] = -60000
    if seed == 14:
        t["bghist"] = 1
    tree_type = [0, 0, 0, -1, -1, -1]
    fr = [((lambda ow, frm, r=rng: OW.random_frame(r, ow, frm, max_exits=260, grid=grid)), -50000, -123456 - 7 * i) for i in range(40)]
    a, b, n_calls = run_lockstep(gpu_lib, t, fr, tree_type, cap=1 << 18, cand_cap=1 << 19, **kw)
    assert len(b["score"]) > 150
    if grid == 10:
        assert b["n_tie_frames"] > 0


@pytest.mark.parametrize("seed,kw", [(21, dict(maxwpf=400, maxhist=400)), (22, dict(maxwpf=9, maxhist=30))])
def test_frames_of_257_to_384_entries_are_ranked_a_thread_per_entry(gpu_lib, seed, kw):
    """what a 256-thread build of the word level got wrong on RM1 utt_33 (profiles/r4_wl_threads_experiment.txt): a frame with more than
    256 and at most WL_RANK_MAX = 384 entries above the pruning threshold is ranked all against all with a THREAD PER ENTRY -- the
    workgroup must be at least that wide (static_assert in s3a_wordlevel.h; 512 threads since round 4, the frame's own workgroup inside
    ku_frames).  Frames of exactly 260 .. 380 distinct word exits, a wide word beam so that all of them stay above the threshold, scores
    on a coarse grid (ties inside the ranking)."""
    rng = np.random.default_rng(seed)
    t = OW.random_task(rng, n_word=500, n_ci=12, density=0.04, grid=10)
    t["wbeam"] = -900000
    tree_type = [0, 0, 0, -1, -1, -1]
    n_word, n_filler = int(t["n_word"]), int(np.sum(t["is_filler"]))

    def frame(ow, frm, total):
        n_hist = ow.vh.contents.n_entry
        wid = rng.permutation(n_word - n_filler)[:total]
        cut = sorted(int(x) for x in rng.integers(0, total, 2))
        trees = []
        for k, (lo, hi) in enumerate(((0, cut[0]), (cut[0], cut[1]), (cut[1], total))):
            w = wid[lo:hi].astype(np.int32)
            scr = (rng.integers(-9000, -4000, len(w)) // 10 * 10 - 3000 * frm).astype(np.int32)
            trees.append((0, w, scr, rng.integers(0, n_hist, len(w)).astype(np.int32)))
        for k in range(3):
            trees.append((-1, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)))
        return trees
    totals = [260, 300, 340, 380, 257, 384, 383, 258] * 3
    fr = [((lambda ow, frm, n=n: frame(ow, frm, n)), -800000, -123456 - 7 * i) for i, n in enumerate(totals)]
    a, b, n_calls = run_lockstep(gpu_lib, t, fr, tree_type, cap=1 << 18, cand_cap=1 << 19, **kw)
    assert len(b["score"]) > 24 * (9 if "maxwpf" in kw and kw["maxwpf"] < 100 else 200)
