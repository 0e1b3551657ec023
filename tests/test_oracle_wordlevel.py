"""Oracle pinning, part 6: the word level (oracle/s3o_wordlevel.c: trigram look-up, vithist_rescore / _enter /
_prune with the reference heap's pop order, word transitions).

(a) End to end (needs oracle/_ref): the reference decoder with its WHOLE word level served by the oracle
    (oracle/_ref/ref_s3owl_decode = integration/sphinx3/s3amd_tst.c -DLT_ORACLE -DWL_ORACLE) reproduces the
    unmodified reference's -hyp and -hypseg byte for byte: tidigits (committed goldens) and RM1 (real 997-word
    trigram, DMP read from disk; live reference).
(b) The reference's own heap (sphinxbase util/heap.c driven by oracle/_ref/ref_dump): pop orders of value lists
    with many ties, committed as tests/golden/heap_orders.npz -- vithist_prune's order among equal scores.
(c) Replay of the committed traces (tests/golden/wordlevel_{tidigits,rm1}.npz, recorded during (a)): every frame's
    surviving entries and lextree_enter calls.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_wordlevel as OW
import wordlevel_trace as WT
from conftest import GOLDEN, ROOT

D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
OWL = os.path.join(ROOT, "oracle", "_ref", "ref_s3owl_decode")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
RM = os.path.join(ROOT, "tests", "_local_data", "rm1")
HAVE_REF = os.path.isdir("/root/reference")


def need(path):
    """oracle/_ref and the local data exist wherever /root/reference does (the build container): missing there = failure."""
    if os.path.exists(path):
        return
    if HAVE_REF or os.environ.get("S3A_ON_GPU_BOX"):
        pytest.fail(f"{path} is missing (make -C oracle ref; tools/fetch_local_data.sh)")
    pytest.skip(f"{path} not present (no /root/reference here)")


@pytest.mark.parametrize("name,extra", [("mode4_trigram", []), ("mode4_cibeam_ds2", ["-ci_pbeam", "1e-5", "-ds", "2"])])
def test_reference_decoder_with_oracle_word_level_is_byte_identical(name, extra, tmp_path):
    need(OWL)
    hyp, seg = str(tmp_path / "h.match"), str(tmp_path / "h.matchseg")
    args = [OWL, "-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5",
            "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4", "-lm", f"{D}/tidigits.DMP",
            "-hyp", hyp, "-hypseg", seg] + extra
    p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0
    assert open(hyp).read() == open(f"{D}/ref_{name}.match").read()
    assert open(seg).read() == open(f"{D}/ref_{name}.matchseg").read()


@pytest.mark.parametrize("extra", [[], ["-maxwpf", "3", "-maxhistpf", "8", "-bghist", "1"]])
def test_rm1_with_oracle_word_level_is_byte_identical(extra, tmp_path):
    need(OWL); need(REFDEC); need(RM)
    args = ["-mdef", f"{RM}/mdef", "-fdict", f"{RM}/fillerdict", "-dict", f"{RM}/RM.dictionary", "-mean", f"{RM}/means",
            "-var", f"{RM}/variances", "-mixw", f"{RM}/mixture_weights", "-tmat", f"{RM}/transition_matrices",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-epl", "4", "-fillprob", "0.02", "-maxwpf", "10",
            "-wip", "0.2", "-lm", f"{RM}/RM.2845.trigram.arpa.DMP", "-lw", "14", "-beam", "1e-140", "-wbeam", "1e-100",
            "-cepdir", f"{RM}/feat", "-cepext", ".mfc", "-ctl", f"{RM}/rm.ctl", "-ctlcount", "6", "-op_mode", "4"]
    if extra:
        args = [a for a in args]
        i = args.index("-maxwpf"); del args[i:i + 2]
    out = {}
    for tag, exe in (("ref", REFDEC), ("owl", OWL)):
        hyp, seg = str(tmp_path / f"{tag}.match"), str(tmp_path / f"{tag}.seg")
        p = subprocess.run([exe] + args + extra + ["-hyp", hyp, "-hypseg", seg], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=900)
        assert p.returncode == 0
        out[tag] = (open(hyp).read(), open(seg).read())
    assert out["owl"] == out["ref"] and out["ref"][0].count("\n") == 6


def test_heap_pop_order_matches_the_reference_heap():
    z = np.load(os.path.join(GOLDEN, "heap_orders.npz"))
    L = OW.lib()
    import ctypes as C
    n_case = int(z["n_case"])
    assert n_case >= 40
    for c in range(n_case):
        vals, order = z[f"v{c}"], z[f"o{c}"]
        n = len(vals)
        # drive the oracle's prune on a one-frame table whose scores are -vals: everything above the threshold,
        # no word / history limits, one word per entry -> the pop order is the heap's
        t = dict(n_ug=2, n_bg=0, n_tg=0, n_word=n + 2, n_ci=2, startwid=n, finishwid=n + 1, silwid=n + 1, start_lwid=0,
                 finish_lwid=1, wbeam=-10 ** 9, bghist=0, maxwpf=n + 5, maxhistpf=n + 5, n_lextree=1, epl=1,
                 ug_prob=np.zeros(2, np.int32), ug_bowt=np.zeros(2, np.int32), ug_firstbg=np.zeros(3, np.int32),
                 lwid=np.zeros(n + 2, np.int32), is_filler=np.zeros(n + 2, np.uint8), fillpen=np.zeros(n + 2, np.int32),
                 last_ci=np.zeros(n + 2, np.int32))
        ow = OW.OracleWordLevel(t, cap=n + 8, max_frames=4)
        vh = ow.vh.contents
        for i in range(n):                       # entries 1..n of frame 0, distinct LM states
            for k, v in (("score", -int(vals[i])), ("wid", i), ("lw0", i), ("lw1", -1), ("pred", 0), ("ef", 0), ("sf", 0)):
                getattr(vh, k)[1 + i] = v
            vh.valid[1 + i] = 1
        vh.n_entry = n + 1
        vh.bestscore[0] = int(-vals.min())
        got = np.full(n, -1, np.int32)
        L.s3o_vithist_prune(ow.vh, C.byref(ow.d), 0, n + 5, n + 5, -2 ** 30, got.ctypes.data_as(OW.I32P))
        assert list(got - 1) == list(order), c


@pytest.mark.parametrize("name,n_frames", [("tidigits", 137), ("rm1", 465)])
def test_oracle_replays_the_recorded_trace(name, n_frames):
    tr = WT.from_npz(np.load(os.path.join(GOLDEN, f"wordlevel_{name}.npz")))
    assert len(tr["frames"]) == n_frames
    ow = OW.OracleWordLevel(tr, max_frames=n_frames + 2)
    n_ent = n_calls = 0
    for f in tr["frames"]:
        r = ow.frame(f["frm"], [(t["type"], t["wid"], t["scr"], t["hist"]) for t in f["trees"]], f["prune_beam"])
        e = f["res"]
        assert r["n_entry"] == e["n_entry"], f["frm"]
        for k in ("wid", "score", "pred", "lw0", "lw1", "ascr", "lscr", "sf", "type"):
            assert np.array_equal(r["entries"][k], e[k]), (f["frm"], k)
        assert (r["bestscore"], r["bestvh"]) == (e["bestscore"], e["bestvh"])
        if e["n_calls"] < 0:
            assert r["calls"] is None
        else:
            assert all(np.array_equal(a, b) for a, b in zip(r["calls"], (e["lc"], e["cscr"], e["chist"])))
            n_calls += e["n_calls"]
        n_ent += e["n_entry"]
    assert n_ent > 100 and n_calls > 50


def test_trigram_backoff_chain_on_a_random_lm():
    """lm_tg_score's three levels against a brute-force reading of the same arrays."""
    rng = np.random.default_rng(7)
    t = OW.random_task(rng)
    ow = OW.OracleWordLevel(t)
    n_ug = t["n_ug"]

    def find(v, lo, hi, w):
        for i in range(lo, hi):
            if v[i] == w:
                return i
        return -1

    def bg(l1, l2):
        if l1 < 0:
            return int(t["ug_prob"][l2])
        b = find(t["bg_wid"], t["ug_firstbg"][l1], t["ug_firstbg"][l1 + 1], l2)
        return int(t["bg_prob"][b]) if b >= 0 else int(t["ug_bowt"][l1]) + int(t["ug_prob"][l2])

    hits = [0, 0, 0]
    for _ in range(3000):
        l1, l2, l3 = (int(x) for x in rng.integers(-1 if rng.random() < 0.1 else 0, n_ug, 3))
        l2, l3 = max(l2, 0), max(l3, 0)
        if l1 < 0:
            exp = bg(l2, l3)
        else:
            b = find(t["bg_wid"], t["ug_firstbg"][l1], t["ug_firstbg"][l1 + 1], l2)
            k = find(t["tg_wid"], t["bg_firsttg"][b], t["bg_firsttg"][b + 1], l3) if b >= 0 else -1
            if k >= 0:
                exp = int(t["tg_prob"][k]); hits[0] += 1
            else:
                exp = (int(t["bg_bowt"][b]) if b >= 0 else 0) + bg(l2, l3); hits[1 + (b < 0)] += 1
        assert ow.tg_score(l1, l2, l3) == exp
    assert min(hits) > 50
