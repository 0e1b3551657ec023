"""Pin the oracle's multi-stream scorer (oracle/s3o_ms.c) against the reference's own
ms_mgau_init + ms_cont_mgau_frame_eval (tests/golden/ms_mgau.npz, produced by
oracle/_ref/ref_dump ms): precomputed determinants / precisions / -logs3 weights, the top-N
lists (sorted and codeword-order forms) and the normalised senone scores, for the tidigits model
through `.s3cont.` and a synthetic one-codebook four-stream model through `.semi.`."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io
from conftest import GOLDEN, golden


@pytest.fixture(scope="module")
def tid():
    d = os.path.join(GOLDEN, "tidigits")
    return (s3io.read_gau(os.path.join(d, "means")), s3io.read_gau(os.path.join(d, "variances")),
            s3io.read_mixw(os.path.join(d, "mixture_weights")))


def make_tid(tid, topn):
    mean, var, mixw = tid
    S, Cn, D = mean.shape
    return O.OracleMs(mean, var, mixw, S, Cn, [D], O.OracleLogMath(1.0003), topn)


def test_precomputed_model_matches_reference(tid):
    g = golden("ms_mgau.npz")
    ms = make_tid(tid, 4)
    c = ms.p.contents
    assert np.array_equal(ms.arr("det", c.n_mgau * c.n_density, np.float32).view(np.uint32),
                          g["tid_det"].ravel().view(np.uint32))
    prec = ms.arr("var", c.n_mgau * c.n_density * c.veclen, np.float32)
    assert np.array_equal(prec[::16].view(np.uint32), g["tid_prec_every16"].view(np.uint32))
    assert np.array_equal(ms.arr("pdf", c.n_sen * c.n_density, np.int32), g["tid_pdf"].ravel())


@pytest.mark.parametrize("case,topn,masked", [("tid_top8", 8, False), ("tid_top4_masked", 4, True), ("tid_top1", 1, True)])
def test_s3cont_frame_eval_matches_reference(tid, case, topn, masked):
    g = golden("ms_mgau.npz")
    ms = make_tid(tid, topn)
    S = ms.n_sen
    for t in range(len(g["feat"])):
        sa = g["active"][t] if masked else np.ones(S, np.uint8)
        best, scr = ms.frame_eval(sa, g["feat"][t])
        assert best == g[case + "_best"][t], t
        a = sa.astype(bool)
        assert np.array_equal(scr[a], g[case + "_senscr"][t][a]), t
        if case + "_dist" in g and t < len(g[case + "_dist"]):
            d, di = ms.last_dist()
            assert np.array_equal(d[a], g[case + "_dist"][t][a])         # 1-to-1 codebooks: active senone = active codebook
            assert np.array_equal(di[a], g[case + "_dist_id"][t][a])


@pytest.mark.parametrize("case,topn,masked", [("semi_top4", 4, True), ("semi_top64", 64, False)])
def test_semi_frame_eval_matches_reference(case, topn, masked):
    g = golden("ms_mgau.npz")
    fl = g["semi_featlen"]
    ms = O.OracleMs(g["semi_mean"], g["semi_var"], g["semi_mixw"], 1, 64, fl, O.OracleLogMath(1.0003), topn,
                    sen2mgau=np.zeros(200, np.int32))
    if topn == 4:
        c = ms.p.contents
        assert np.array_equal(ms.arr("det", 4 * 64, np.float32).view(np.uint32), g["semi_det"].ravel().view(np.uint32))
        assert np.array_equal(ms.arr("pdf", 200 * 4 * 64, np.int32)[::7], g["semi_pdf_every7"])
    for t in range(len(g["semi_feat"])):
        sa = g["semi_active"][t] if masked else np.ones(200, np.uint8)
        best, scr = ms.frame_eval(sa, g["semi_feat"][t])
        assert best == g[case + "_best"][t], t
        a = sa.astype(bool)
        assert np.array_equal(scr[a], g[case + "_senscr"][t][a]), t
        if topn == 4:
            d, di = ms.last_dist()
            assert np.array_equal(d, g[case + "_dist"][t]) and np.array_equal(di, g[case + "_dist_id"][t])
