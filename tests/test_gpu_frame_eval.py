"""Parity (MI355X): the gated per-frame driver, composite senones and the
batched hmm_vit_eval through the C ABI, against reference-derived goldens."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io, synth
from conftest import golden

pytestmark = pytest.mark.gpu
CASES = ["default", "masked", "cibeam", "cibeam_all", "ds2", "ds3_tight", "maxcd"]
HSETS = ["3st_tidigits", "3st_skip", "5st", "5st_noskip", "4st_any"]


@pytest.fixture(scope="module")
def tid(gpu_lib, tidigits_dir):
    lm = gpu_lib.LogMath(1.0003)
    return gpu_lib.MgauModel.init(os.path.join(tidigits_dir, "means"),
                                  os.path.join(tidigits_dir, "variances"),
                                  os.path.join(tidigits_dir, "mixture_weights"), lm)


@pytest.mark.parametrize("case", CASES)
def test_frame_eval_sequence_matches_reference(gpu_lib, tid, case):
    g = golden("tidigits_frame_eval.npz")
    cipbeam, ds, tighten, maxcd, masked = g[case + "_params"]
    sc = gpu_lib.Scorer(tid, g["cd2cisen"], 102, ds_ratio=int(ds), ci_pbeam=float(cipbeam),
                        tighten_factor=float(tighten), max_cd=int(maxcd))
    r = sc.frame_eval_seq(g["feat"], active=g["active"] if masked else None)
    assert np.array_equal(r["ci_best"], g[case + "_ci_best"])
    assert np.array_equal(r["best"], g[case + "_best"])
    assert np.array_equal(r["sen_active_out"], g[case + "_sen_active_out"])
    act = r["sen_active_out"].astype(bool)
    assert np.array_equal(r["senscr"][act], g[case + "_senscr"][act])
    assert np.array_equal(r["bstidx"], g[case + "_bstidx"])
    assert np.array_equal(r["updatetime"], g[case + "_updatetime"])
    assert np.array_equal(r["counts"], g[case + "_counts"][:, :2])


def test_ci_eval_scores(gpu_lib, tid):
    g = golden("tidigits_frame_eval.npz")
    sc = gpu_lib.Scorer(tid, g["cd2cisen"], 102)
    sc.utt_begin()
    for t in (0, 17, 47):
        ci, best = sc.ci_eval(g["feat"][t], t)
        assert np.array_equal(ci, g["default_ci_senscr"][t])
        assert best == g["default_ci_best"][t]


@pytest.mark.parametrize("name", ["deg_c5", "c32", "c1_d13"])
def test_frame_eval_on_synthetic_shapes(gpu_lib, name):
    g = golden("synth_models.npz")
    kw = dict(zip(("n_sen", "n_ci_sen", "n_comp", "veclen", "n_tmat", "n_emit", "seed", "degenerate"),
                  (int(v) for v in g[name + "_kw"])))
    kw["degenerate"] = bool(kw["degenerate"])
    m = synth.make_model(**kw)
    fx = synth.make_features(m, 21, seed=kw["seed"] + 1)
    lm = gpu_lib.LogMath(1.0003)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], lm)
    sc = gpu_lib.Scorer(gm, m["cd2cisen"], kw["n_ci_sen"], ci_pbeam=1e-2)
    r = sc.frame_eval_seq(fx)
    assert np.array_equal(r["best"], g[name + "_fe_best"])
    assert np.array_equal(r["senscr"], g[name + "_fe_senscr"])
    assert np.array_equal(r["counts"], g[name + "_fe_counts"][:, :2])


def test_scorer_rejects_interleaved_ci_senones(gpu_lib, tid):
    g = golden("tidigits_frame_eval.npz")
    bad = g["cd2cisen"].copy()
    bad[300] = 300
    with pytest.raises(gpu_lib.S3AError, match="CI senones must be exactly the first"):
        gpu_lib.Scorer(tid, bad, 102)


def test_comsenscr_matches_oracle(gpu_lib):
    rng = np.random.default_rng(5)
    n_sen, n_com = 1935, 1265                   # RM1 sizes
    lens = rng.integers(1, 40, n_com)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    lst = rng.integers(0, n_sen, off[-1]).astype(np.int16)
    wt = -rng.integers(0, 12000, n_com).astype(np.int32)
    senscr = -rng.integers(0, 900000, n_sen).astype(np.int32)
    senscr[rng.integers(0, n_sen, 20)] = O.LOGPROB_ZERO
    cs = gpu_lib.ComSen(off, lst, wt)
    assert np.array_equal(cs.comsenscr(senscr), O.comsenscr(off, lst, wt, senscr))
    with pytest.raises(gpu_lib.S3AError):
        gpu_lib.ComSen(np.array([0, 0, 2], np.int32), np.array([1, 2], np.int16), np.array([0, 0], np.int32))


@pytest.mark.parametrize("name", HSETS)
def test_hmm_vit_eval_random_protocol(gpu_lib, name):
    g = golden("hmm.npz")
    ne = int(g[name + "_ne"][0])
    tp, sseq, senscr = g[name + "_tp"], g[name + "_sseq"], g[name + "_senscr"]
    spec, enter = g[name + "_spec"], g[name + "_enter"]
    T, nsen = senscr.shape
    nh = spec.shape[0]
    tm = gpu_lib.Tmat.init_logs3(tp)
    hb = gpu_lib.HmmBatch(nh, tm, sseq, nsen)
    hb.setup(spec[:, 0], spec[:, 1], spec[:, 2])
    for t in range(T):
        who = np.nonzero(enter[t, :, 0] != -2147483648)[0]
        hb.enter(who, enter[t, who, 0], enter[t, who, 1], t)
        ret = hb.vit_eval(senscr[t])
        assert np.array_equal(ret, g[name + "_ret"][t]), t
        st = hb.get()
        exp, eh = g[name + "_state"][t], g[name + "_hist"][t]
        assert np.array_equal(st["score"][:, :ne], exp[:, :ne]), t
        assert np.array_equal(st["out_score"], exp[:, 5]), t
        assert np.array_equal(st["bestscore"], exp[:, 6]), t
        assert np.array_equal(st["hist"][:, :ne], eh[:, :ne]), t
        assert np.array_equal(st["out_hist"], eh[:, 5]), t
        mp = spec[:, 0].astype(bool)
        assert np.array_equal(st["mpx_ssid"][mp][:, :ne], exp[mp][:, 7:7 + ne]), t


def test_reference_unit_test_testhmm_on_gpu(gpu_lib, tidigits_dir):
    """sphinx3/src/tests/unit_tests/test_hmm: -4044 / -11008 / -22688."""
    lm = gpu_lib.LogMath(1.0001)
    md = s3io.read_mdef(os.path.join(tidigits_dir, "mdef"))
    tm = gpu_lib.Tmat.init(os.path.join(tidigits_dir, "transition_matrices"), lm, 1e-5)
    hb = gpu_lib.HmmBatch(2, tm, md["sseq"], md["n_sen"])
    hb.setup([0, 1], [0, 0], [0, 0])
    st = hb.get()
    assert (st["score"][:, :3] == O.LOGPROB_ZERO).all() and (st["hist"][:, :3] == -1).all()
    hb.enter([0, 1], [0, 0], [42, 69], 0)
    zeros = np.zeros(md["n_sen"], np.int32)
    hb.vit_eval(zeros)
    st = hb.get()
    assert list(st["score"][0, :3]) == [-4044, -11008, O.LOGPROB_ZERO]
    assert list(st["hist"][0, :3]) == [42, 42, -1]
    assert list(st["score"][1, :3]) == [-4044, -11008, O.LOGPROB_ZERO]
    assert list(st["mpx_ssid"][1, :3]) == [0, 0, -1]
    hb.enter([1], [0], [69], 0)
    hb.vit_eval(zeros)
    st = hb.get()
    assert list(st["score"][1, :3]) == [-4044, -11008, -22688]
    assert list(st["hist"][1, :3]) == [69, 69, 69]
