"""Oracle pinning, part 2: continuous-density GMM loading + senone scoring.

oracle/s3o_mgau.c against outputs of the unmodified reference's mgau_init /
mgau_eval (tests/golden/*.npz, generator tests/golden/make_golden.py):
the tidigits model of the reference's own regression tests, seeded synthetic
models with the loader's edge cases, and the hub4-shaped bench model.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io, synth
from conftest import golden

SYN = ["deg_c5", "deg_c8", "c32", "c1_d13", "c3_d51"]


def bits(a):
    return a.view(np.int32) if a.dtype == np.float32 else a


@pytest.fixture(scope="module")
def tid(tidigits_dir, olm):
    return O.OracleMgau(s3io.read_gau(os.path.join(tidigits_dir, "means")),
                        s3io.read_gau(os.path.join(tidigits_dir, "variances")),
                        s3io.read_mixw(os.path.join(tidigits_dir, "mixture_weights")), olm)


def test_tidigits_precompute(tid):
    g = golden("tidigits_mgau.npz")
    assert np.array_equal(tid.n_comp, g["n_comp"])
    assert np.array_equal(bits(tid.lrd), bits(g["lrd"]))
    assert np.array_equal(tid.mixw, g["mixw"])
    assert np.array_equal(bits(tid.prec[::16]), bits(g["prec_every16"]))
    assert tid.distfloor == g["distfloor"][0]


def test_tidigits_scores_and_best_component(tid):
    g = golden("tidigits_mgau.npz")
    x = g["feat"]
    T, S = g["score"].shape
    sc = tid.score_all(x)
    assert np.array_equal(sc, g["score"])
    # bstidx / bstscr after mgau_eval(..., update_best_id = 1)
    tid.reset_state()
    for t in (0, T - 1):
        for s in range(0, S, 7):
            tid.eval(s, x[t], t, 1)
            assert tid.arr("bstidx", (S,), copy=False)[s] == g["bstidx"][t, s]
            assert tid.arr("bstscr", (S,), copy=False)[s] == g["bstscr"][t, s]


def make_syn(name):
    g = golden("synth_models.npz")
    kw = dict(zip(("n_sen", "n_ci_sen", "n_comp", "veclen", "n_tmat", "n_emit", "seed", "degenerate"),
                  (int(v) for v in g[name + "_kw"])))
    kw["degenerate"] = bool(kw["degenerate"])
    m = synth.make_model(**kw)
    fx = synth.make_features(m, 21, seed=kw["seed"] + 1)
    assert synth.array_crc(m["mean"], m["var"], m["mixw"], fx) == int(g[name + "_crc"][0]), \
        "synthetic generator drifted from the one the golden file was made with"
    return g, kw, m, fx


@pytest.mark.parametrize("name", SYN)
def test_synthetic_models(name, olm):
    g, kw, m, fx = make_syn(name)
    og = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    nc = g[name + "_n_comp"]
    assert np.array_equal(og.n_comp, nc)
    for s in range(kw["n_sen"]):
        assert np.array_equal(bits(og.lrd[s, :nc[s]]), bits(g[name + "_lrd"][s, :nc[s]]))
        assert np.array_equal(og.mixw[s, :nc[s]], g[name + "_mixw"][s, :nc[s]])
    assert np.array_equal(og.score_all(fx), g[name + "_score"])
    if kw["degenerate"]:
        assert (nc < kw["n_comp"]).any()            # components were removed
        assert (g[name + "_score"] == O.LOGPROB_ZERO).any()


def test_active_list_and_update_rules(tid):
    """mgau_eval with an active list (cont_mgau.c:1125-1167) and the bstidx quirk."""
    g = golden("tidigits_mgau.npz")
    x = g["feat"][3]
    s = 200
    tid.reset_state()
    full = tid.eval(s, x, 5, 1)
    b = int(tid.arr("bstidx", (tid.S,))[s])
    assert b == g["bstidx"][3, s]
    one = tid.eval(s, x, 6, 0, active=[b])
    assert one == tid.arr("bstscr", (tid.S,))[s]    # single best Gaussian == its gauscr
    assert one <= full
    assert tid.arr("updatetime", (tid.S,))[s] == 5  # update_best_id = 0 leaves updatetime
    allc = tid.eval(s, x, 7, 1, active=list(range(8)))
    assert allc == full


def test_hub4_shaped_reference_frames(olm):
    g = golden("hub4_synth.npz")
    m = synth.make_model(**synth.HUB4)
    fx = synth.make_features(m, 1000, seed=7)
    assert synth.array_crc(m["mean"], m["var"], m["mixw"], fx) == int(g["crc"][0])
    og = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    assert np.array_equal(bits(og.lrd[:64]), bits(g["lrd_head"]))
    assert np.array_equal(og.mixw[:64], g["mixw_head"])
    assert np.array_equal(og.score_all(fx[g["frames"]]), g["score"])
