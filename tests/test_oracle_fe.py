"""Pin the oracle's MFCC front end (oracle/s3o_fe.c: framing with the final partial frame, pre-emphasis,
Hamming window, the reference's real FFT, mel filter bank, log, the three cepstral transforms, liftering,
-logspec / -smoothspec) on outputs of the UNMODIFIED reference (tests/golden/fe.npz, made by
tests/golden/make_golden.py through oracle/_ref/ref_dump fe) -- bit for bit: same machine arithmetic, same libm --
and on the reference's own committed golden chan3.mfc to the tolerance of the reference's own test (0.1,
sphinxbase/test/regression/test-sphinx_fe.sh; that file was made by an older build)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from fe_cases import FE_CASES, FE_SHORT, fe_params  # noqa: E402


def case_input(g, name):
    return g["chan3_raw"] if FE_CASES[name][0] == "chan3" else g["goforward_raw"]


@pytest.mark.parametrize("name", sorted(FE_CASES))
def test_cepstra_match_reference(name):
    g = golden("fe.npz")
    got = O.OracleFe(**fe_params(FE_CASES[name][1])).process_utt(case_input(g, name))
    ref = g["cep_" + name]
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("ns", FE_SHORT)
def test_framing_edges_match_reference(ns):
    """0 samples -> no frame; fewer than a frame -> the one zero-padded frame of fe_end_utt; and around every
    point where a full frame is added."""
    g = golden("fe.npz")
    got = O.OracleFe().process_utt(g["goforward_raw"][2000:2000 + ns])
    ref = g[f"cep_short_{ns}"]
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_committed_reference_golden_within_its_own_tolerance():
    g = golden("fe.npz")
    got = O.OracleFe(**fe_params(FE_CASES["chan3"][1])).process_utt(g["chan3_raw"])
    assert np.abs(got[:600] - g["chan3_mfc_committed"]).max() < 0.1


def test_bad_options_are_rejected():
    with pytest.raises(ValueError):
        O.OracleFe(nfft=500)
    with pytest.raises(ValueError):
        O.OracleFe(nfft=256)            # smaller than the 410-sample frame
