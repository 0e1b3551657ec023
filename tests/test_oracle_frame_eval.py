"""Oracle pinning, part 3: the per-frame scoring driver with CI gating.

oracle/s3o_mgau.c:s3o_approx_cont_mgau_{ci,frame}_eval against the unmodified
reference's approx_cont_mgau_ci_eval + approx_cont_mgau_frame_eval
(approx_cont_mgau.c:367-616) run over a frame sequence exactly as
srch_utt_decode_blk does with -pl_window 1 (srch.c:739-822): per-senone scores
(normalised), frame best, CI scores, forced-active flags, bstidx/updatetime
state, and the frm_sen_eval / frm_gau_eval counters, for:
  default    all senones active, -ci_pbeam 1e-80 (never closes)
  masked     random bursty active masks
  cibeam     narrow CI beam 1e-3 with masks: full eval / best-Gaussian back-off /
             CI back-off all occur
  cibeam_all narrow beam, all active
  ds2, ds3_tight   frame down-sampling with beam tightening
  maxcd      -maxcdsenpf 60: dynamic CI beam (approx_compute_dyn_ci_pbeam)
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io
from conftest import golden

CASES = ["default", "masked", "cibeam", "cibeam_all", "ds2", "ds3_tight", "maxcd"]


@pytest.fixture(scope="module")
def tid(tidigits_dir, olm):
    return O.OracleMgau(s3io.read_gau(os.path.join(tidigits_dir, "means")),
                        s3io.read_gau(os.path.join(tidigits_dir, "variances")),
                        s3io.read_mixw(os.path.join(tidigits_dir, "mixture_weights")), olm)


def test_cd2cisen_from_mdef(tidigits_dir):
    g = golden("tidigits_frame_eval.npz")
    md = s3io.read_mdef(os.path.join(tidigits_dir, "mdef"))
    assert md["n_ci_sen"] == 102 and md["n_sen"] == 602
    assert np.array_equal(md["cd2cisen"], g["cd2cisen"])


@pytest.mark.parametrize("case", CASES)
def test_frame_eval_sequence(case, tid, olm):
    g = golden("tidigits_frame_eval.npz")
    cipbeam, ds, tighten, maxcd, masked = g[case + "_params"]
    r = tid.frame_eval_seq(g["feat"], g["cd2cisen"], 102,
                           active=g["active"] if masked else None,
                           ci_pbeam=olm.logs3(cipbeam), ds=int(ds), tighten=float(tighten),
                           max_cd=int(maxcd))
    assert np.array_equal(r["best"], g[case + "_best"])
    assert np.array_equal(r["ci_best"], g[case + "_ci_best"])
    assert np.array_equal(r["sen_active_out"], g[case + "_sen_active_out"])
    act = r["sen_active_out"].astype(bool)
    # senscr of inactive senones is stale memory in the reference: compare active ones
    assert np.array_equal(r["senscr"][act], g[case + "_senscr"][act])
    assert np.array_equal(r["bstidx"], g[case + "_bstidx"])
    assert np.array_equal(r["updatetime"], g[case + "_updatetime"])
    assert np.array_equal(r["counts"], g[case + "_counts"])
    assert r["beams"][0] == g[case + "_beams"][0]


def test_gating_paths_are_exercised():
    """The fixtures must actually hit all three branches of the gate."""
    g = golden("tidigits_frame_eval.npz")
    c = g["cibeam_counts"]          # [T][ns, ng, n_cis, n_cig]
    n_active_cd = g["cibeam_sen_active_out"][:, 102:].sum(1)
    assert (c[:, 0] < n_active_cd).any()            # some active CD senones were not fully evaluated
    assert (c[:, 1] > 8 * c[:, 0]).any() or (c[:, 1] != 8 * c[:, 0]).any()  # single-Gaussian back-offs counted
    assert g["maxcd_beams"][1] != g["maxcd_beams"][0]   # dynamic beam differed from the static one
