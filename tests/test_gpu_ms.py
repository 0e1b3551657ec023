"""Parity (MI355X): the multi-stream senone scorer behind -senmgau .s3cont. / .semi.
(s3a_ms_mgau_*, cmusphinx_amd/csrc/s3a_ms.hip) against the reference's own outputs
(tests/golden/ms_mgau.npz) and, on hub4- and semi-continuous-sized synthetic models, against
the oracle -- bit for bit: scores, frame best, and the ordered top-N codeword lists."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import synth
from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,topn,masked", [("tid_top8", 8, False), ("tid_top4_masked", 4, True), ("tid_top1", 1, True)])
def test_s3cont_files_match_reference(gpu_lib, case, topn, masked):
    g = golden("ms_mgau.npz")
    d = os.path.join(GOLDEN, "tidigits")
    ms = gpu_lib.MsMgau.init(os.path.join(d, "means"), os.path.join(d, "variances"), os.path.join(d, "mixture_weights"),
                             gpu_lib.LogMath(1.0003), senmgau=".s3cont.", topn=topn)
    ms.n_mgau, ms.n_feat = ms.n_sen, 1
    scr = np.zeros(ms.n_sen, np.int32)
    for t in range(len(g["feat"])):
        sa = g["active"][t] if masked else np.ones(ms.n_sen, np.uint8)
        best = ms.frame_eval(sa, scr, g["feat"][t], t)
        a = sa.astype(bool)
        assert best == g[case + "_best"][t], t
        assert np.array_equal(scr[a], g[case + "_senscr"][t][a]), t
        if case + "_dist" in g and t < len(g[case + "_dist"]):
            dd, di = ms.last_dist()
            assert np.array_equal(dd[a], g[case + "_dist"][t][a]) and np.array_equal(di[a], g[case + "_dist_id"][t][a])


@pytest.mark.parametrize("case,topn,masked", [("semi_top4", 4, True), ("semi_top64", 64, False)])
def test_semi_arrays_match_reference(gpu_lib, case, topn, masked):
    g = golden("ms_mgau.npz")
    ms = gpu_lib.MsMgau.init_arrays(g["semi_mean"], g["semi_var"], g["semi_mixw"], 1, 64, g["semi_featlen"],
                                    gpu_lib.LogMath(1.0003), topn, sen2mgau=np.zeros(200, np.int32))
    scr = np.zeros(200, np.int32)
    for t in range(len(g["semi_feat"])):
        sa = g["semi_active"][t] if masked else np.ones(200, np.uint8)
        best = ms.frame_eval(sa, scr, g["semi_feat"][t], t)
        a = sa.astype(bool)
        assert best == g[case + "_best"][t], t
        assert np.array_equal(scr[a], g[case + "_senscr"][t][a]), t
        if topn == 4:
            dd, di = ms.last_dist()
            assert np.array_equal(dd, g[case + "_dist"][t]) and np.array_equal(di, g[case + "_dist_id"][t])


@pytest.mark.parametrize("shape", ["hub4_top4", "semi256_top4", "odd_c5_top3"])
def test_synthetic_shapes_match_oracle(gpu_lib, shape):
    rng = np.random.default_rng(3)
    if shape == "hub4_top4":
        m = synth.make_model(**synth.HUB4)
        mean, var, mixw = m["mean"], m["var"], m["mixw"]
        M, nd, fl, topn, s2m = 6144, 8, [39], 4, None
        feats = synth.make_features(m, 4, seed=9)
    elif shape == "semi256_top4":
        fl, nd, M, topn = [12, 24, 3, 12], 256, 1, 4
        D = sum(fl)
        mean = rng.standard_normal(nd * D).astype(np.float32)
        var = np.exp(rng.uniform(np.log(0.05), np.log(2.0), nd * D)).astype(np.float32)
        mixw = (rng.dirichlet(np.ones(nd) * 0.2, (3000, 4)) * 500).astype(np.float32)
        s2m = np.zeros(3000, np.int32)
        feats = (rng.standard_normal((4, D)) * 1.1).astype(np.float32)
    else:       # 5 densities (padded to 8 lanes), two streams, senones sharing 40 codebooks, tied distances
        fl, nd, M, topn = [7, 5], 5, 40, 3
        D = sum(fl)
        mean = np.round(rng.standard_normal(M * nd * D) * 2).astype(np.float32)
        var = np.full(M * nd * D, 0.5, np.float32)
        mean.reshape(M, -1)[:, :] = mean.reshape(M, -1)         # (layout [m][f][d][len])
        mean.reshape(M, nd * D)[::3, 7:14] = mean.reshape(M, nd * D)[::3, 0:7]     # duplicate densities: exact ties
        mixw = (rng.dirichlet(np.ones(nd), (300, 2)) * 100).astype(np.float32)
        s2m = rng.integers(0, M, 300).astype(np.int32)
        feats = np.round(rng.standard_normal((6, D)) * 2).astype(np.float32)
    S = mixw.size // (len(fl) * nd)
    om = O.OracleMs(mean, var, mixw, M, nd, fl, O.OracleLogMath(1.0003), topn, sen2mgau=s2m)
    gm = gpu_lib.MsMgau.init_arrays(mean, var, mixw, M, nd, fl, gpu_lib.LogMath(1.0003), topn, sen2mgau=s2m)
    scr = np.zeros(S, np.int32)
    for t in range(len(feats)):
        sa = (rng.random(S) < 0.8).astype(np.uint8)
        ob, oscr = om.frame_eval(sa, feats[t])
        gb = gm.frame_eval(sa, scr, feats[t], t)
        a = sa.astype(bool)
        assert gb == ob and np.array_equal(scr[a], oscr[a]), (shape, t)
        od, oi = om.last_dist()
        gd, gi = gm.last_dist()
        cb = np.zeros(M, bool); cb[(s2m if s2m is not None else np.arange(S))[a]] = True
        assert np.array_equal(gd[cb], od[cb]) and np.array_equal(gi[cb], oi[cb]), (shape, t)


def test_unsupported_configurations_fail_loudly(gpu_lib):
    d = os.path.join(GOLDEN, "tidigits")
    f = lambda n: os.path.join(d, n)
    lm = gpu_lib.LogMath(1.0003)
    with pytest.raises(gpu_lib.S3AError, match="interpolation"):
        gpu_lib.MsMgau.init(f("means"), f("variances"), f("mixture_weights"), lm, lambdafile="/nonexistent")
    with pytest.raises(gpu_lib.S3AError, match="mapping FILES"):
        gpu_lib.MsMgau.init(f("means"), f("variances"), f("mixture_weights"), lm, senmgau="/some/map")
