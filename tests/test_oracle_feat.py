"""Pin the oracle's feature computation (oracle/s3o_feat.c: padding, CMN / variance normalisation /
AGC-max over the padded utterance, the 1s_c_d_dd difference streams) on the reference's own
feat_s2mfc2feat outputs (tests/golden/feat_variants.npz): bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from conftest import golden

VARIANTS = [("current", 0, "none"), ("current", 1, "max"), ("none", 0, "max"), ("current", 0, "max")]


def oracle_feat(cep, cmn, vn, agc):
    cep = np.ascontiguousarray(cep, np.float32)
    out = np.zeros((len(cep), 39), np.float32)
    O.lib().s3o_feat_1s_c_d_dd(cep.ctypes.data_as(C.c_void_p), len(cep), 13, int(cmn == "current"), int(vn),
                               int(agc == "max"), out.ctypes.data_as(C.c_void_p))
    return out


@pytest.mark.parametrize("u", [0, 1])
@pytest.mark.parametrize("cmn,vn,agc", VARIANTS)
def test_feature_streams_match_reference(u, cmn, vn, agc):
    g = golden("feat_variants.npz")
    ref = g[f"feat{u}_{cmn}_{vn}_{agc}"]
    got = oracle_feat(g[f"cep{u}"], cmn, vn, agc)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_default_variant_is_the_decoders_feature_stream():
    """-cmn current -agc none -varnorm no is what the tidigits decoding fixtures were scored on."""
    g, f = golden("feat_variants.npz"), golden("tidigits_feat.npz")
    assert np.array_equal(g["feat0_current_0_none"], f["man.ah.111a"])
