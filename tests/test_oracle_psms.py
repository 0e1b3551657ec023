"""Pin the oracle's restatement of POCKETSPHINX's continuous scorer (oracle/s3o_psms.c) on the outputs of
the unmodified pocketsphinx (tests/golden/ps_ms.npz via oracle/_ref/ref_ps_dump): float32 determinants,
log-domain precisions, 8-bit weights, and the int16 negated, best-normalised senone scores with all
senones or pocketsphinx's delta-encoded active lists."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io
from conftest import GOLDEN, golden

TID_CASES = ["tid_top4_masked", "tid_top8", "tid_top1_aw3", "tid_top4_b10003"]


@pytest.fixture(scope="module")
def tid():
    d = os.path.join(GOLDEN, "tidigits")
    return (s3io.read_gau(os.path.join(d, "means")), s3io.read_gau(os.path.join(d, "variances")),
            s3io.read_mixw(os.path.join(d, "mixture_weights")))


@pytest.mark.parametrize("case", TID_CASES)
def test_cont_scores_match_pocketsphinx(tid, case):
    g, f = golden("ps_ms.npz"), golden("ms_mgau.npz")
    topn, aw, base, masked = g[case + "_params"]
    mean, var, mixw = tid
    S, Cn, D = mean.shape
    ps = O.OraclePsMs(mean, var, mixw, S, Cn, [D], int(topn), int(aw), float(base))
    scr = np.zeros(S, np.int16)
    for t in range(len(f["feat"])):
        scr[:] = 0
        ps.frame_eval(scr, f["feat"][t], f["active"][t] if masked else None)
        assert np.array_equal(scr, g[case + "_senscr"][t]), t


@pytest.mark.parametrize("case,topn,masked", [("semi_top4_masked", 4, True), ("semi_top64", 64, False)])
def test_semi_scores_match_pocketsphinx(case, topn, masked):
    g, f = golden("ps_ms.npz"), golden("ms_mgau.npz")
    ps = O.OraclePsMs(f["semi_mean"], f["semi_var"], f["semi_mixw"], 1, 64, f["semi_featlen"], topn,
                      sen2mgau=np.zeros(200, np.int32))
    scr = np.zeros(200, np.int16)
    for t in range(len(f["semi_feat"])):
        scr[:] = 0
        ps.frame_eval(scr, f["semi_feat"][t], f["semi_active"][t] if masked else None)
        assert np.array_equal(scr, g[case + "_senscr"][t]), t
