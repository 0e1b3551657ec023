"""Oracle pinning, part 4: transition matrices and hmm_vit_eval.

(a) the reference's own unit test: sphinx3/src/tests/unit_tests/test_hmm/
    testhmm.c with its golden _testhmm_tidigits.res (copied verbatim as a data
    file): tidigits mdef + tmat, logbase 1.0001, tpfloor 1e-5; non-mpx and mpx
    3-state HMMs entered with score 0 -> -4044 / -11008 / -22688.
(b) random protocol runs through the unmodified reference's hmm_vit_eval
    (tests/golden/hmm.npz): 3-state (tidigits tmats and synthetic with skip
    arcs), 5-state with and without skips, and a 4-state set that takes the
    any-topology path; multiplex and non-multiplex HMMs mixed.
(c) tmat_init's float -> logs3 conversion against the reference.
"""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import s3io
from conftest import golden, GOLDEN

SETS = ["3st_tidigits", "3st_skip", "5st", "5st_noskip", "4st_any"]


def test_tmat_logs3_matches_reference(tidigits_dir):
    g = golden("tmat.npz")
    raw = s3io.read_tmat(os.path.join(tidigits_dir, "transition_matrices"))
    assert np.array_equal(O.tmat_logs3(raw, O.OracleLogMath(1.0003), 1e-4), g["tidigits_1e-4_b1.0003"])
    assert np.array_equal(O.tmat_logs3(raw, O.OracleLogMath(1.0001), 1e-5), g["tidigits_1e-5_b1.0001"])
    assert np.array_equal(O.tmat_logs3(g["hub4_raw"], O.OracleLogMath(1.0003), 1e-4), g["hub4_1e-4_b1.0003"])


def _parse_res():
    """SCORES / HISTID / MPX lines of _testhmm_tidigits.res, in file order."""
    blocks, cur = [], {}
    for ln in open(os.path.join(GOLDEN, "_testhmm_tidigits.res")):
        m = re.match(r"(SSID|MPX|SENSCR|SCORES|HISTID)\s+(.*)", ln)
        if not m:
            continue
        key = m.group(1)
        nums = [int(v) for v in re.findall(r"-?\d+", m.group(2))]
        if key in ("SSID", "MPX"):
            cur = {"kind": key, "head": nums}
            blocks.append(cur)
        else:
            cur[key] = nums
    return blocks


def test_reference_unit_test_testhmm(tidigits_dir):
    blocks = _parse_res()
    assert len(blocks) == 5
    lm = O.OracleLogMath(1.0001, 0, 1)
    md = s3io.read_mdef(os.path.join(tidigits_dir, "mdef"))
    tp = O.tmat_logs3(s3io.read_tmat(os.path.join(tidigits_dir, "transition_matrices")), lm, 1e-5)
    senscr = np.zeros((1, md["n_sen"]), np.int32)
    NOENT = -2147483648

    def run(mpx, ssid0, hist):
        spec = np.array([[mpx, 0, 0]], np.int32)
        enter = np.array([[[0, hist]]], np.int32)
        # testhmm.c sets hmm_mpx_ssid(&h2, 0) = 1 before the third eval: emulate by spec ssid
        spec[0, 1] = ssid0
        st, hi, ret = O.hmm_run(3, tp, md["sseq"], senscr, spec, enter)
        return st[0, 0], hi[0, 0]

    st, hi = run(0, 0, 42)                      # block 3: non-mpx after enter + eval
    assert list(st[:3]) + [st[5]] == blocks[2]["SCORES"] == [-4044, -11008, -939524096, -939524096]
    assert list(hi[:3]) + [hi[5]] == blocks[2]["HISTID"]
    st, hi = run(1, 0, 69)                      # block 4: mpx
    assert list(st[:3]) + [st[5]] == blocks[3]["SCORES"]
    assert list(hi[:3]) + [hi[5]] == blocks[3]["HISTID"]
    assert list(st[7:10]) == [0, 0, -1]         # "( 0 0 -1 )" per-state ssids


@pytest.mark.parametrize("name", SETS)
def test_hmm_vit_eval_random_protocol(name):
    g = golden("hmm.npz")
    ne = int(g[name + "_ne"][0])
    st, hi, ret = O.hmm_run(ne, g[name + "_tp"], g[name + "_sseq"], g[name + "_senscr"],
                            g[name + "_spec"], g[name + "_enter"])
    assert np.array_equal(ret, g[name + "_ret"])
    assert np.array_equal(st, g[name + "_state"])
    assert np.array_equal(hi, g[name + "_hist"])
    # the fixture is not degenerate: scores above WORST_SCORE and exits occur
    assert (g[name + "_state"][..., 5] > O.LOGPROB_ZERO).any()
