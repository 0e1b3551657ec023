"""Parity (MI355X): pocketsphinx's continuous scorer on the device (s3a_ps_ms_*, cmusphinx_amd/csrc/
s3a_psms.hip -- the object a ps_mgaufuncs_t vtable forwards to) against the unmodified pocketsphinx's own
outputs and the oracle: int16 scores bit for bit, with all senones and with delta-encoded active lists."""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import synth
from conftest import GOLDEN, golden
from test_oracle_psms import TID_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", TID_CASES)
def test_cont_files_match_pocketsphinx(gpu_lib, case):
    g, f = golden("ps_ms.npz"), golden("ms_mgau.npz")
    topn, aw, base, masked = g[case + "_params"]
    d = os.path.join(GOLDEN, "tidigits")
    ps = gpu_lib.PsMsMgau.init(os.path.join(d, "means"), os.path.join(d, "variances"), os.path.join(d, "mixture_weights"),
                               senmgau=".cont.", topn=int(topn), aw=int(aw), logbase=float(base))
    scr = np.zeros(ps.n_sen, np.int16)
    for t in range(len(f["feat"])):
        scr[:] = 0
        ps.frame_eval(scr, f["feat"][t], O.delta_encode(f["active"][t]) if masked else None, t)
        assert np.array_equal(scr, g[case + "_senscr"][t]), t


@pytest.mark.parametrize("case,topn,masked", [("semi_top4_masked", 4, True), ("semi_top64", 64, False)])
def test_semi_arrays_match_pocketsphinx(gpu_lib, case, topn, masked):
    g, f = golden("ps_ms.npz"), golden("ms_mgau.npz")
    ps = gpu_lib.PsMsMgau.init_arrays(f["semi_mean"], f["semi_var"], f["semi_mixw"], 1, 64, f["semi_featlen"], topn,
                                      sen2mgau=np.zeros(200, np.int32))
    scr = np.zeros(200, np.int16)
    for t in range(len(f["semi_feat"])):
        scr[:] = 0
        ps.frame_eval(scr, f["semi_feat"][t], O.delta_encode(f["semi_active"][t]) if masked else None, t)
        assert np.array_equal(scr, g[case + "_senscr"][t]), t


def test_hub4_shape_matches_oracle_with_sparse_lists(gpu_lib):
    """6144 codebooks x 8; sparse active lists with gaps over 255 (bridged by 255-steps); exact float ties."""
    m = synth.make_model(**synth.HUB4)
    feats = synth.make_features(m, 3, seed=4)
    mean = m["mean"].copy()
    mean[::5, 3, :] = mean[::5, 2, :]                       # duplicate densities: ties inside the top-N lists
    var = m["var"].copy(); var[::5, 3, :] = var[::5, 2, :]
    om = O.OraclePsMs(mean, var, m["mixw"], 6144, 8, [39], 4, aw=2)
    gm = gpu_lib.PsMsMgau.init_arrays(mean, var, m["mixw"], 6144, 8, [39], 4, aw=2)
    rng = np.random.default_rng(8)
    a, b = np.zeros(6144, np.int16), np.zeros(6144, np.int16)
    for t, density in enumerate((0.9, 0.01, 0.0005)):
        mask = rng.random(6144) < density
        mask[6143] = True
        a[:] = 0; b[:] = 0
        om.frame_eval(a, feats[t], mask)
        gm.frame_eval(b, feats[t], O.delta_encode(mask), t)
        assert np.array_equal(a, b), t
    a[:] = 0; b[:] = 0
    om.frame_eval(a, feats[0], None); gm.frame_eval(b, feats[0], None)
    assert np.array_equal(a, b)
