"""Host C logic of libcmusphinx_amd through its C ABI (no GPU needed):
log-add tables, the S3 file envelope + checksum, mgau_init's precomputation
and tmat_init's conversion -- against the reference-derived goldens and the
oracle, bit for bit."""
import os
import shutil

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import lib, s3io, synth
from conftest import golden

CASES = [(1.0003, 0), (1.0001, 0), (1.0001, 1), (1.0001, 8), (1.002, 0)]


def bits(a):
    return a.view(np.int32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("base,shift", CASES)
def test_logmath_table_and_known_values(base, shift):
    g = golden("logmath.npz")
    key = f"b{base}_s{shift}"
    lm = lib.LogMath(base, shift, 1)
    size, width, sh = lm.table_shape()
    k = g[key + "_known"]
    assert (size, width, sh, lm.zero) == tuple(k[:4])
    assert np.array_equal(lm.table.astype(np.int64), g[key + "_table"].astype(np.int64))
    got = [lm.log(1e-150), lm.log(42.0), lm.log(1e-48),
           lm.add(lm.log(1e-48), lm.log(5e-48)), lm.add(lm.log(1e-48), lm.log(42.0)),
           lm.log10_to_log(-7.0), lm.ln_to_log(-123.456), lm.logs3(1e-80), lm.logs3(0.5),
           lm.add(lib.LOGPROB_ZERO, -12345), lm.add(-12345, lib.LOGPROB_ZERO),
           lm.add(-100, -100 - size)]
    assert got == list(k[4:16])
    kf = g[key + "_knownf"]
    assert lm.log_to_ln(lib.LOGPROB_ZERO) == kf[0]
    assert lm.exp(-5000) == kf[1]
    assert lm.base == kf[3]


def test_reference_unit_test_known_answers():
    lm = lib.LogMath(1.0001)
    assert lm.log(1e-150) == -3454050 and lm.log(42) == 37378     # sphinxbase test_log_int16.c
    lm3 = lib.LogMath(1.0003)
    assert int(lm3.log10_to_log(0.8202) * 10.5 - lm3.logs3(0.02)) == 79150   # sphinx3 test_logs3
    with pytest.raises(lib.S3AError):
        lib.LogMath(1.0)


def test_logmath_add_random_vs_oracle():
    lm, olm = lib.LogMath(1.0003), O.OracleLogMath(1.0003)
    rng = np.random.default_rng(3)
    xs = rng.integers(-1_200_000_000, 100, 4000)
    ys = xs + rng.integers(-40000, 40000, 4000)
    for x, y in zip(xs.tolist(), ys.tolist()):
        y = max(y, -2_000_000_000)
        assert lm.add(x, y) == olm.add(x, y)


def test_mgau_loader_tidigits_matches_reference(tidigits_dir):
    g = golden("tidigits_mgau.npz")
    lm = lib.LogMath(1.0003)
    m = lib.MgauModel.load_host(os.path.join(tidigits_dir, "means"),
                                os.path.join(tidigits_dir, "variances"),
                                os.path.join(tidigits_dir, "mixture_weights"), lm)
    assert (m.S, m.C, m.D) == (602, 8, 39)
    p = m.params()
    assert np.array_equal(p["n_comp"], g["n_comp"])
    assert np.array_equal(bits(p["lrd"]), bits(g["lrd"]))
    assert np.array_equal(p["mixw"], g["mixw"])
    assert np.array_equal(bits(p["prec"][::16]), bits(g["prec_every16"]))
    assert p["distfloor"] == g["distfloor"][0]
    # a host-only handle must refuse to score
    with pytest.raises(lib.S3AError, match="needs a GPU"):
        m.score_frames(g["feat"][:2])


@pytest.mark.parametrize("name", ["deg_c5", "deg_c8", "c32", "c1_d13", "c3_d51"])
def test_mgau_loader_synthetic_edge_cases(name, tmp_path, olm):
    g = golden("synth_models.npz")
    kw = dict(zip(("n_sen", "n_ci_sen", "n_comp", "veclen", "n_tmat", "n_emit", "seed", "degenerate"),
                  (int(v) for v in g[name + "_kw"])))
    kw["degenerate"] = bool(kw["degenerate"])
    mdl = synth.make_model(**kw)
    d = synth.write_model(str(tmp_path / name), mdl, chksum=True)
    lm = lib.LogMath(1.0003)
    m = lib.MgauModel.load_host(os.path.join(d, "means"), os.path.join(d, "variances"),
                                os.path.join(d, "mixture_weights"), lm)
    p = m.params()
    nc = g[name + "_n_comp"]
    assert np.array_equal(p["n_comp"], nc)
    og = O.OracleMgau(mdl["mean"], mdl["var"], mdl["mixw"], olm)
    for s in range(kw["n_sen"]):
        k = nc[s]
        assert np.array_equal(bits(p["lrd"][s, :k]), bits(g[name + "_lrd"][s, :k]))
        assert np.array_equal(p["mixw"][s, :k], g[name + "_mixw"][s, :k])
        assert np.array_equal(bits(p["mean"][s, :k]), bits(og.mean[s, :k]))
        assert np.array_equal(bits(p["prec"][s, :k]), bits(og.prec[s, :k]))


def test_envelope_errors_and_byteswap(tidigits_dir, tmp_path):
    lm = lib.LogMath(1.0003)
    src = {k: os.path.join(tidigits_dir, k) for k in ("means", "variances", "mixture_weights")}
    # corrupt one payload byte -> checksum error (bio.c:491-504 would E_FATAL; we fail the call)
    bad = tmp_path / "means_bad"
    b = bytearray(open(src["means"], "rb").read())
    b[5000] ^= 0x40
    bad.write_bytes(bytes(b))
    with pytest.raises(lib.S3AError, match="checksum error"):
        lib.MgauModel.load_host(str(bad), src["variances"], src["mixture_weights"], lm)
    # truncated file
    trunc = tmp_path / "means_trunc"
    trunc.write_bytes(bytes(b[:40000]))
    with pytest.raises(lib.S3AError):
        lib.MgauModel.load_host(str(trunc), src["variances"], src["mixture_weights"], lm)
    with pytest.raises(lib.S3AError, match="cannot open"):
        lib.MgauModel.load_host(str(tmp_path / "nope"), src["variances"], src["mixture_weights"], lm)
    # mismatched dimensions between means and mixture weights
    w = s3io.read_mixw(src["mixture_weights"])[:100]
    s3io.write_mixw(str(tmp_path / "mixw100"), w)
    with pytest.raises(lib.S3AError, match="don't match"):
        lib.MgauModel.load_host(src["means"], src["variances"], str(tmp_path / "mixw100"), lm)
    # big-endian copy of the whole model loads to the same parameters (bio.c swap path)
    def swapped(path, out):
        raw = open(path, "rb").read()
        i = raw.index(b"endhdr\n") + 7
        body = np.frombuffer(raw[i:], dtype="<u4").byteswap().tobytes()
        open(out, "wb").write(raw[:i] + body)
    for k in src:
        swapped(src[k], str(tmp_path / (k + "_be")))
    a = lib.MgauModel.load_host(src["means"], src["variances"], src["mixture_weights"], lm).params()
    c = lib.MgauModel.load_host(str(tmp_path / "means_be"), str(tmp_path / "variances_be"),
                                str(tmp_path / "mixture_weights_be"), lm).params()
    for k in ("mean", "prec", "lrd", "mixw", "n_comp"):
        assert np.array_equal(bits(a[k]), bits(c[k]))


def test_tmat_init_matches_reference(tidigits_dir):
    g = golden("tmat.npz")
    path = os.path.join(tidigits_dir, "transition_matrices")
    assert np.array_equal(lib.Tmat.init(path, lib.LogMath(1.0003), 1e-4).tp, g["tidigits_1e-4_b1.0003"])
    assert np.array_equal(lib.Tmat.init(path, lib.LogMath(1.0001), 1e-5).tp, g["tidigits_1e-5_b1.0001"])
    t = lib.Tmat.init_arrays(g["hub4_raw"], lib.LogMath(1.0003), 1e-4)
    assert (t.n_tmat, t.n_state) == (48, 3)
    assert np.array_equal(t.tp, g["hub4_1e-4_b1.0003"])
    # lower-triangular entry -> rejected like tmat_chk_uppertri (tmat.c:126-140)
    bad = g["hub4_raw"].copy()
    bad[3, 2, 0] = 5.0
    with pytest.raises(lib.S3AError, match="upper triangular"):
        lib.Tmat.init_arrays(bad, lib.LogMath(1.0003), 1e-4)


def test_s3io_roundtrip_is_byte_identical(tidigits_dir, tmp_path):
    for name, rd, wr in (("means", s3io.read_gau, s3io.write_gau),
                         ("mixture_weights", s3io.read_mixw, s3io.write_mixw),
                         ("transition_matrices", s3io.read_tmat, s3io.write_tmat)):
        src = os.path.join(tidigits_dir, name)
        out = str(tmp_path / name)
        wr(out, rd(src))
        assert open(out, "rb").read() == open(src, "rb").read()
