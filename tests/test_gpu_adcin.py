"""-adcin through the whole-utterance engine (MI355X): raw 16-bit audio -> MFCC (s3a_fe) -> 1s_c_d_dd features (s3a_feat)
-> decode, everything after the file read on the device (s3a_audio_to_feat_dev: the cepstra and the features never leave
HBM) -- what utt_decode does with -adcin on the host (libAPI/utt.c:208-233: fe_process_utt WITHOUT fe_end_utt, then
feat_s2mfc2feat_live over the whole utterance).  The unmodified reference with the same command line is the judge.
Audio: the reference's own test files (pocketsphinx goforward.raw, sphinxbase chan3.raw: tests/golden/fe.npz; the
tidigits utterance dhd.2934z.raw of pocketsphinx/test/data/tidigits)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_gpu_uttdec import AM, D, TST

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")


@pytest.fixture(scope="module")
def audio_task(tmp_path_factory):
    for b in (REF, TST):
        if not os.path.exists(b):
            pytest.fail(f"{b} is missing on the GPU box (make -C oracle ref)")
    d = tmp_path_factory.mktemp("adcin")
    g = np.load(os.path.join(GOLDEN, "fe.npz"))
    g["goforward_raw"].astype("<i2").tofile(d / "goforward.raw")
    g["chan3_raw"].astype("<i2").tofile(d / "chan3.raw")
    g["goforward_raw"][3000:3000 + 9000].astype("<i2").tofile(d / "short.raw")
    raw = open(os.path.join(D, "raw", "dhd.2934z.raw"), "rb").read()
    (d / "dhd.raw").write_bytes(raw)
    (d / "hdr.raw").write_bytes(b"H" * 44 + raw)             # the same behind a 44-byte header: its own run with -adchdr 44
    (d / "ctl").write_text("dhd\ngoforward\nchan3\nshort\n")
    (d / "ctlh").write_text("hdr\n")
    args = ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", str(d), "-cepext", ".raw", "-adcin", "yes",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-op_mode", "4", "-lm", f"{D}/tidigits.DMP"]
    return d, args


def run(exe, args, d, tag, env=None, ctl="ctl", extra=()):
    hyp, seg = str(d / f"{tag}.match"), str(d / f"{tag}.matchseg")
    p = subprocess.run([exe] + args + ["-ctl", str(d / ctl), "-hyp", hyp, "-hypseg", seg] + list(extra), capture_output=True, text=True,
                       errors="ignore", timeout=900, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2500:]
    return open(hyp).read(), open(seg).read()


@pytest.mark.parametrize("env", [{"S3A_UTT": "1"}, {"S3A_UTT": "4"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "4"},
                                 {"S3A_UTT": "4", "S3A_UTT_ENGINES": "2"}])
def test_raw_audio_decodes_as_the_reference(audio_task, env):
    d, args = audio_task
    ref = run(REF, args, d, "ref")
    assert ref[0].count("\n") == 4 and "(dhd)" in ref[0]
    got = run(TST, args, d, "utt" + "_".join(env.values()), env)
    assert got[0] == ref[0]                 # the words
    assert got[1] == ref[1]                 # ... and every score: the device front end's cepstra are the host's, bit for bit


def test_header_skip_and_other_front_end_options(audio_task):
    d, args = audio_task
    ref = run(REF, args, d, "refh", ctl="ctlh", extra=["-adchdr", "44"])
    got = run(TST, args, d, "utth", {"S3A_UTT": "2"}, ctl="ctlh", extra=["-adchdr", "44"])
    assert got == ref and "(hdr)" in ref[0]
    opt = ["-transform", "dct", "-remove_dc", "yes", "-lifter", "22", "-varnorm", "yes", "-agc", "max"]
    a2 = [a for a in args]
    for k in ("-varnorm", "-agc"):
        i = a2.index(k)
        del a2[i:i + 2]
    ref = run(REF, a2 + opt, d, "refo")
    got = run(TST, a2 + opt, d, "utto", {"S3A_UTT": "3"})
    assert got == ref


def test_dither_and_big_endian_samples(audio_task):
    """-dither yes -seed N: every sample that enters a frame gets one draw of the generator the reference's fe_init seeded, in
    sample order, the generator running on from utterance to utterance (fe_sigproc.c:606-613, 630-638) -- applied to the samples
    before they go to the device; -input_endian big: the samples swapped (fe->swap).  Every score as the reference's."""
    d, args = audio_task
    ref = run(REF, args, d, "refd", extra=["-dither", "yes", "-seed", "1234"])
    plain = run(REF, args, d, "refp")
    assert ref[1] != plain[1]                                           # (the dither is no bystander)
    for env in ({"S3A_UTT": "1"}, {"S3A_UTT": "3"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "4"}):
        got = run(TST, args, d, "uttd" + "_".join(env.values()), env, extra=["-dither", "yes", "-seed", "1234"])
        assert got == ref
    be = d / "be"
    be.mkdir(exist_ok=True)
    for n in ("dhd", "goforward", "chan3", "short"):
        np.fromfile(d / f"{n}.raw", "<i2").astype(">i2").tofile(be / f"{n}.raw")
    a2 = [a if a != str(d) else str(be) for a in args]
    refb = run(REF, a2, d, "refb", extra=["-input_endian", "big"])
    assert refb == plain
    assert run(TST, a2, d, "uttb", {"S3A_UTT": "2"}, extra=["-input_endian", "big", "-dither", "yes", "-seed", "1234"]) == ref


def test_cmn_prior(audio_task):
    """-cmn prior (the live-mode normalisation: every frame loses the mean learnt from the utterances BEFORE, cmn_prior.c): the
    subtraction and the running sums on the device, the decoder's cmn_t carried between utterances by the drop-in (window shift
    beyond 800 frames included: the control file is long enough)"""
    d, args = audio_task
    a2 = [a for a in args]
    k = a2.index("-cmn")
    a2[k + 1] = "prior"
    (d / "ctlp").write_text("dhd\ngoforward\nchan3\nshort\nchan3\ndhd\ngoforward\n")
    ref = run(REF, a2, d, "refcp", ctl="ctlp", extra=["-cmninit", "10.0"])
    cur = run(REF, args, d, "refcc", ctl="ctlp")
    assert ref[1] != cur[1] and ref[0].count("\n") == 7
    first = [l for l in ref[1].splitlines() if l.startswith("dhd ")]
    assert len(first) == 2 and first[0] != first[1]                     # (the same audio, another prior: the state moves)
    for env in ({"S3A_UTT": "1"}, {"S3A_UTT": "3"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "5"}):
        got = run(TST, a2, d, "uttcp" + "_".join(env.values()), env, ctl="ctlp", extra=["-cmninit", "10.0"])
        assert got == ref


def test_lda_feature_transform(audio_task, tmp_path):
    """-lda / -ldadim (feat_lda_transform behind the feature computation): the rows transformed on the device, float32 terms in the
    reference's order.  A full 39 x 39 rotation with the tidigits model, and a reduction to 32 dimensions with the model's
    Gaussians cut to 32 (data made here: what counts is that the reference and the device see the same numbers)."""
    from cmusphinx_amd import s3io
    d, args = audio_task
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.standard_normal((39, 39)))
    lda = (np.eye(39) * 0.9 + 0.1 * q).astype(np.float32)
    s3io.write_lda(str(tmp_path / "lda39"), lda)
    plain = run(REF, args, d, "refl0")
    ref = run(REF, args, d, "refl1", extra=["-lda", str(tmp_path / "lda39")])
    assert ref[1] != plain[1]
    assert run(TST, args, d, "uttl1", {"S3A_UTT": "3"}, extra=["-lda", str(tmp_path / "lda39")]) == ref
    # 32 dimensions: the model's means / variances cut to the first 32, the other model files as they are
    m32 = tmp_path / "m32"
    m32.mkdir()
    for f in ("mdef", "mixture_weights", "transition_matrices", "feat.params"):
        if os.path.exists(os.path.join(AM, f)):
            (m32 / f).write_bytes(open(os.path.join(AM, f), "rb").read())
    for f in ("means", "variances"):
        g = s3io.read_gau(os.path.join(AM, f))
        s3io.write_gau(str(m32 / f), np.ascontiguousarray(g[:, :, :32]))
    a2 = [str(m32) if a == AM else a for a in args]
    ex = ["-lda", str(tmp_path / "lda39"), "-ldadim", "32"]
    ref32 = run(REF, a2, d, "refl2", extra=ex)
    assert ref32[0].count("\n") == 4 and ref32[1] != ref[1]
    for env in ({"S3A_UTT": "1"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "4"}):
        assert run(TST, a2, d, "uttl2" + "_".join(env.values()), env, extra=ex) == ref32


@pytest.mark.parametrize("wtype,wpar", [("inverse_linear", "0.94"), ("affine", "1.05 12.0"), ("piecewise_linear", "0.92 6000")])
def test_frequency_warping(audio_task, wtype, wpar):
    """-warp_type / -warp_params (VTLN; fe_warp_*.c): the mel filters' corner frequencies through the warping function, float32 as
    the reference computes them; the decodes' scores equal the reference's to the last digit"""
    d, args = audio_task
    plain = run(REF, args, d, "refw0")
    ref = run(REF, args, d, "refw_" + wtype, extra=["-warp_type", wtype, "-warp_params", wpar])
    assert ref[1] != plain[1]
    assert run(TST, args, d, "uttw_" + wtype, {"S3A_UTT": "2"}, extra=["-warp_type", wtype, "-warp_params", wpar]) == ref


def test_unsupported_front_end_options_are_refused(audio_task):
    d, args = audio_task
    a2 = [a if a != "current" else "prior" for a in args]
    a2[a2.index("-varnorm") + 1] = "yes"
    p = subprocess.run([TST] + a2 + ["-ctl", str(d / "ctl")], capture_output=True, text=True, errors="ignore",
                       timeout=600, env=dict(os.environ, S3A_UTT="2"))
    assert p.returncode != 0 and "Variance normalization not implemented in live mode" in p.stderr
