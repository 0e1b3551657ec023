"""Parity (MI355X): the MFCC front end on the device (s3a_fe_*, cmusphinx_amd/csrc/s3a_fe.hip) against the
unmodified reference's outputs (tests/golden/fe.npz) and the oracle.

Everything except log() is the reference's float64 / float32 arithmetic in the reference's order, so the
-logspec output may differ from the CPU's by the float32 rounding of a 1-ulp float64 difference in log
(device library vs libm), and a cepstrum by the same thing through the DCT.  Stated tolerance: 2 float32
ulps of the largest magnitude of the frame's outputs; the tests also require that at least 99.9 % of all
values are bit-identical (observed: every value)."""
import numpy as np
import pytest

from conftest import golden
from test_oracle_fe import FE_CASES, FE_SHORT, case_input, fe_params
import oracle_lib as O

pytestmark = pytest.mark.gpu


def check_close(got, ref):
    assert got.shape == ref.shape
    if ref.size == 0:
        return 1.0
    tol = 2 * np.spacing(np.abs(ref).max(axis=1, keepdims=True).astype(np.float32))
    assert (np.abs(got - ref) <= tol).all(), float(np.abs(got - ref).max())
    same = float((got.view(np.uint32) == ref.view(np.uint32)).mean())
    assert same >= 0.999, same
    return same


@pytest.mark.parametrize("name", sorted(FE_CASES))
def test_device_cepstra_match_reference(gpu_lib, name):
    g = golden("fe.npz")
    fe = gpu_lib.FrontEnd(**fe_params(FE_CASES[name][1]))
    check_close(fe.process_utt(case_input(g, name)), g["cep_" + name])


@pytest.mark.parametrize("ns", FE_SHORT)
def test_device_framing_edges(gpu_lib, ns):
    g = golden("fe.npz")
    fe = gpu_lib.FrontEnd()
    assert fe.n_frames(ns) == len(g[f"cep_short_{ns}"])
    check_close(fe.process_utt(g["goforward_raw"][2000:2000 + ns]), g[f"cep_short_{ns}"])


@pytest.mark.parametrize("opts", [dict(), dict(nfft=1024, wlen=0.05), dict(nfft=2048, nfilt=64, ncep=20, transform=1),
                                  dict(samprate=8000.0, nfft=256, nfilt=31, lowerf=200.0, upperf=3500.0, remove_dc=1)])
def test_device_matches_oracle_on_synthetic_audio(gpu_lib, opts):
    """100 s of synthetic audio (10 000 frames): noise bursts, silence (the -10 floor of the log), full-scale
    square waves."""
    rng = np.random.default_rng(11)
    n = int(100 * opts.get("samprate", 16000.0))
    x = (rng.standard_normal(n) * 3000 * (np.sin(np.arange(n) / 9000.0) ** 2)).astype(np.int16)
    x[n // 4: n // 4 + 20000] = 0
    x[n // 2: n // 2 + 20000] = np.where((np.arange(20000) // 40) % 2 == 0, 32767, -32768)
    got = gpu_lib.FrontEnd(**opts).process_utt(x)
    exp = O.OracleFe(**opts).process_utt(x)
    check_close(got, exp)


def test_device_front_end_feeds_the_feature_stage(gpu_lib):
    """raw audio -> cepstra -> 1s_c_d_dd features, both stages on the device, against the oracle's chain"""
    from test_oracle_feat import oracle_feat
    g = golden("fe.npz")
    cep = gpu_lib.FrontEnd().process_utt(g["goforward_raw"])
    got = gpu_lib.feat_1s_c_d_dd(cep, cmn="current", varnorm=False, agc="none")
    exp = oracle_feat(O.OracleFe().process_utt(g["goforward_raw"]), "current", 0, "none")
    assert np.abs(got - exp).max() <= 1e-5


def test_rejected_options(gpu_lib):
    with pytest.raises(RuntimeError):
        gpu_lib.FrontEnd(nfft=500)
    with pytest.raises(RuntimeError):
        gpu_lib.FrontEnd(nfft=256)
