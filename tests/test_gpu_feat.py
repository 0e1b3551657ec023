"""Parity (MI355X): feature computation on the device (s3a_feat_1s_c_d_dd, cmusphinx_amd/csrc/s3a_feat.hip)
against the reference's own feat_s2mfc2feat outputs and the oracle: bit-exact float32 under every
normalisation option, on real cepstra and on long / tiny synthetic utterances."""
import numpy as np
import pytest

from conftest import golden
from test_oracle_feat import VARIANTS, oracle_feat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("u", [0, 1])
@pytest.mark.parametrize("cmn,vn,agc", VARIANTS)
def test_device_features_match_reference(gpu_lib, u, cmn, vn, agc):
    g = golden("feat_variants.npz")
    got = gpu_lib.feat_1s_c_d_dd(g[f"cep{u}"], cmn=cmn, varnorm=bool(vn), agc=agc)
    assert np.array_equal(got.view(np.uint32), g[f"feat{u}_{cmn}_{vn}_{agc}"].view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 7, 64, 65, 3000])
def test_device_features_match_oracle_on_synthetic_lengths(gpu_lib, n):
    rng = np.random.default_rng(n)
    cep = (rng.standard_normal((n, 13)) * np.linspace(8, 0.5, 13)).astype(np.float32)
    for cmn, vn, agc in VARIANTS:
        got = gpu_lib.feat_1s_c_d_dd(cep, cmn=cmn, varnorm=bool(vn), agc=agc)
        exp = oracle_feat(cep, cmn, vn, agc)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (n, cmn, vn, agc)
