"""Parity tests proper (MI355X): HIP senone scoring through the C ABI against
the CPU oracle and the reference-derived golden fixtures.  Bit-exact (int32)
in the default S3A_GMM_EXACT mode; the S3A_GMM_FAST mode is held to its stated
tolerance of +-2 logs3 units per Gaussian score and must give the same argmax
ordering of the frame-best senone."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io, synth
from conftest import golden

pytestmark = pytest.mark.gpu
SYN = ["deg_c5", "deg_c8", "c32", "c1_d13", "c3_d51"]
FAST_TOL = 2        # logs3 units; see DESIGN.md "precision modes"


@pytest.fixture(scope="module")
def tid(gpu_lib, tidigits_dir):
    lm = gpu_lib.LogMath(1.0003)
    return gpu_lib.MgauModel.init(os.path.join(tidigits_dir, "means"),
                                  os.path.join(tidigits_dir, "variances"),
                                  os.path.join(tidigits_dir, "mixture_weights"), lm)


def test_tidigits_scores_match_reference_golden(tid):
    g = golden("tidigits_mgau.npz")
    sc, best = tid.score_frames(g["feat"])
    assert np.array_equal(sc, g["score"])
    assert np.array_equal(best, g["score"].max(1))


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64])
def test_ragged_frame_counts(tid, n):
    """Empty input, fewer frames than one 8-frame group, and ragged tails."""
    g = golden("tidigits_mgau.npz")
    feat = g["feat"][:n]
    if n == 0:
        out = tid.score_frames(np.zeros((0, 39), np.float32), want_best=False)
        assert out.shape == (0, 602)
        return
    sc = tid.score_frames(feat, want_best=False)
    assert np.array_equal(sc, g["score"][:n])


def test_mgau_eval_single_senone_dropin(tid):
    """s3a_mgau_eval == mgau_eval incl. bstidx/bstscr/updatetime and active lists."""
    g = golden("tidigits_mgau.npz")
    x = g["feat"]
    tid.reset_state()
    for t in (0, 5, 63):
        for s in (0, 101, 102, 333, 601):
            assert tid.eval(s, x[t], t, 1) == g["score"][t, s]
            bi, bs, ut = tid.state()
            assert (bi[s], bs[s], ut[s]) == (g["bstidx"][t, s], g["bstscr"][t, s], t)
    s, t = 200, 3
    full = tid.eval(s, x[t], 5, 1)
    b = int(tid.state()[0][s])
    one = tid.eval(s, x[t], 6, 0, active=[b])
    bi, bs, ut = tid.state()
    assert one == bs[s] and ut[s] == 5 and bi[s] == b       # update_best_id=0 leaves the state
    assert tid.eval(s, x[t], 7, 1, active=list(range(8))) == full
    assert tid.state()[2][s] == 7


@pytest.mark.parametrize("name", SYN)
def test_synthetic_models_match_reference_golden(gpu_lib, name):
    """Loader edge cases (removed components, floors, zero weights) and other
    shapes: 5/8/32/1/3 components, 39/13/51 dimensions (generic-veclen kernel)."""
    g = golden("synth_models.npz")
    kw = dict(zip(("n_sen", "n_ci_sen", "n_comp", "veclen", "n_tmat", "n_emit", "seed", "degenerate"),
                  (int(v) for v in g[name + "_kw"])))
    kw["degenerate"] = bool(kw["degenerate"])
    m = synth.make_model(**kw)
    fx = synth.make_features(m, 21, seed=kw["seed"] + 1)
    assert synth.array_crc(m["mean"], m["var"], m["mixw"], fx) == int(g[name + "_crc"][0])
    lm = gpu_lib.LogMath(1.0003)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], lm)
    assert np.array_equal(gm.params()["n_comp"], g[name + "_n_comp"])
    sc, best = gm.score_frames(fx)
    assert np.array_equal(sc, g[name + "_score"])
    assert np.array_equal(best, g[name + "_score"].max(1))
    gm.reset_state()
    for s in range(0, kw["n_sen"], 13):
        assert gm.eval(s, fx[4], 4, 1) == g[name + "_score"][4, s]
        bi, bs, _ = gm.state()
        assert bi[s] == g[name + "_bstidx"][4, s] and bs[s] == g[name + "_bstscr"][4, s]


def test_hub4_shaped_model_vs_reference_and_oracle(gpu_lib, olm):
    """The bench workload: 6144 senones x 8 x 39, 1000 frames."""
    g = golden("hub4_synth.npz")
    m = synth.make_model(**synth.HUB4)
    fx = synth.make_features(m, 1000, seed=7)
    assert synth.array_crc(m["mean"], m["var"], m["mixw"], fx) == int(g["crc"][0])
    lm = gpu_lib.LogMath(1.0003)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], lm)
    sc, best = gm.score_frames(fx)
    assert np.array_equal(sc[g["frames"]], g["score"])          # unmodified reference
    og = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    pick = np.r_[0:16, 500:508, 992:1000]
    assert np.array_equal(sc[pick], og.score_all(fx[pick]))     # oracle
    assert np.array_equal(best, sc.max(1))
    # size-independent properties at full size
    assert sc.min() >= O.LOGPROB_ZERO                           # clamp, cont_mgau.c:1200-1203
    # frame-synchronous launches (1 frame / 8 frames per launch) give the same integers
    fd = gpu_lib.DevBuf(fx.nbytes).upload(fx)
    sd = gpu_lib.DevBuf(sc.nbytes)
    for fpl in (1, 8, 0):
        gm.bench(fd, 1000, sd, None, fpl, 1)
        assert np.array_equal(sd.download(np.int32, sc.shape), sc)
    # permuting the frames permutes the output rows (no cross-frame state in score_frames)
    perm = np.random.default_rng(0).permutation(1000)
    assert np.array_equal(gm.score_frames(fx[perm], want_best=False), sc[perm])


def test_wsj_stress_shape_32_components(gpu_lib, olm):
    """config 5 shape: 32 Gaussians per senone (CP = 32 lanes per senone)."""
    kw = dict(synth.WSJ_STRESS)
    kw["n_sen"], kw["n_ci_sen"] = 1200, 150
    m = synth.make_model(**kw)
    fx = synth.make_features(m, 40, seed=3)
    lm = gpu_lib.LogMath(1.0003)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], lm)
    og = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    sc = gm.score_frames(fx, want_best=False)
    pick = [0, 7, 8, 39]
    assert np.array_equal(sc[pick], og.score_all(fx[pick]))
    # one frame per call: the frame-synchronous kernel (split log-add table: LDS head + comparisons for the tail)
    for k in range(40):
        assert np.array_equal(gm.score_frames(fx[[k]], want_best=False)[0], sc[k]), k


def test_wsj_stress_shape_at_full_size(gpu_lib, olm):
    """configs[4] at its full size: 8000 senones x 32 Gaussians x 39 (82 MB of parameters): ten frames against the oracle,
    through the many-frames kernel and one frame per launch (k_score_frame_sync)."""
    m = synth.make_model(**synth.WSJ_STRESS)
    assert m["mean"].shape[0] == 8000 and m["mean"].shape[1] == 32
    fx = synth.make_features(m, 10, seed=5)
    gm = gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], gpu_lib.LogMath(1.0003))
    og = O.OracleMgau(m["mean"], m["var"], m["mixw"], olm)
    exp = og.score_all(fx)
    sc = gm.score_frames(fx, want_best=False)
    assert sc.shape == (10, 8000) and np.array_equal(sc, exp)
    for k in (0, 4, 9):
        assert np.array_equal(gm.score_frames(fx[[k]], want_best=False)[0], exp[k]), k


def test_fast_mode_within_stated_tolerance(gpu_lib, tid):
    g = golden("tidigits_mgau.npz")
    tid.set_precision(gpu_lib.GMM_FAST)
    try:
        sc = tid.score_frames(g["feat"], want_best=False)
    finally:
        tid.set_precision(gpu_lib.GMM_EXACT)
    d = np.abs(sc.astype(np.int64) - g["score"])
    assert d.max() <= FAST_TOL, d.max()
    assert np.array_equal(sc.argmax(1), g["score"].argmax(1))


def test_other_logbase_falls_back_to_global_table(gpu_lib, tidigits_dir, olm):
    """base 1.0001: the 99042-entry table (198 KB) does not fit LDS."""
    lm = gpu_lib.LogMath(1.0001)
    gm = gpu_lib.MgauModel.init(os.path.join(tidigits_dir, "means"),
                                os.path.join(tidigits_dir, "variances"),
                                os.path.join(tidigits_dir, "mixture_weights"), lm)
    olm1 = O.OracleLogMath(1.0001)
    og = O.OracleMgau(s3io.read_gau(os.path.join(tidigits_dir, "means")),
                      s3io.read_gau(os.path.join(tidigits_dir, "variances")),
                      s3io.read_mixw(os.path.join(tidigits_dir, "mixture_weights")), olm1)
    x = golden("tidigits_mgau.npz")["feat"][:24]
    assert np.array_equal(gm.score_frames(x, want_best=False), og.score_all(x))
