"""MLLR-adapted models behind the replacement backend (MI355X): -mllr (one regression matrix for the run) and -ctl_mllr (a
matrix per utterance).  kb_setmllr (kb.c:335-365 -> adapt_set_mllr, libam/adaptor.c:106-170: reload, mllr_norm_mgau,
variance floor, mgau_precomp) stays the reference's host code; the device model takes its result (s3a_mgau_set_params)
before the next utterance is scored.  Judge: the unmodified reference with the same options, -hyp / -hypseg byte for byte
-- and the adapted runs must differ from the unadapted one (a model that silently stayed unadapted would pass otherwise)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_uttdec import AM, D, TST

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")


def write_regmat(path, seed, strength):
    """one class, one stream, 39 dimensions: A (near the identity), B, H (variance scaling)"""
    rng = np.random.default_rng(seed)
    A = np.eye(39) + strength * rng.standard_normal((39, 39)) / 39.0
    B = strength * rng.standard_normal(39)
    H = 1.0 + 0.3 * strength * rng.uniform(-1, 1, 39)
    with open(path, "w") as f:
        f.write("1\n1\n39\n")
        for row in A:
            f.write(" ".join(f"{v:.6f}" for v in row) + "\n")
        f.write(" ".join(f"{v:.6f}" for v in B) + "\n")
        f.write(" ".join(f"{v:.6f}" for v in H) + "\n")


@pytest.fixture(scope="module")
def task(tmp_path_factory):
    for b in (REF, TST):
        if not os.path.exists(b):
            pytest.fail(f"{b} is missing on the GPU box (make -C oracle ref)")
    d = tmp_path_factory.mktemp("mllr")
    write_regmat(d / "m1", 1, 0.1)           # (mild: the adapted models still recognise digits)
    write_regmat(d / "m2", 2, 0.2)
    ctl = [l.split()[0] for l in open(f"{D}/tidigits.length.arb.regression") if l.strip()][:10]
    (d / "ctl").write_text("".join(u + "\n" for u in ctl))
    # a matrix per utterance: runs of the same one, a change, a change back
    which = ["m1", "m1", "m2", "m2", "m2", "m1", "m2", "m1", "m1", "m2"]
    (d / "ctl_mllr").write_text("".join(f"{d}/{w}\n" for w in which))
    args = ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra", "-agc", "none",
            "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-op_mode", "4", "-lm", f"{D}/tidigits.DMP", "-ctl", str(d / "ctl")]
    return d, args


def run(exe, args, d, tag, env=None):
    hyp, seg = str(d / f"{tag}.match"), str(d / f"{tag}.matchseg")
    p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], capture_output=True, text=True, errors="ignore", timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2500:]
    return open(hyp).read(), open(seg).read()


@pytest.mark.parametrize("env", [{}, {"S3A_UTT": "3"}, {"S3A_UTT": "4", "S3A_UTT_ENGINES": "2"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "10"}])
def test_one_regression_matrix_for_the_run(task, env):
    """{}: the frame-synchronous slots; S3A_UTT: whole utterances on the device (every engine's model adapted)"""
    d, args = task
    plain = run(REF, args, d, "ref_plain")
    ref = run(REF, args + ["-mllr", str(d / "m2")], d, "ref_m2")
    assert ref[0].count("\n") == 10 and ref[1].count("\n") == 10 and ref[1] != plain[1]
    got = run(TST, args + ["-mllr", str(d / "m2")], d, "dev_m2_" + "_".join(env.values()), env)
    assert got == ref


@pytest.mark.parametrize("env", [{}, {"S3A_UTT": "3"}, {"S3A_UTT": "2", "S3A_UTT_ENGINES": "2", "S3A_UTT_QUEUE": "4"}])
def test_a_regression_matrix_per_utterance(task, env):
    """-ctl_mllr: the model changes between utterances; in utterance mode what is queued is decoded before the switch"""
    d, args = task
    ref = run(REF, args + ["-ctl_mllr", str(d / "ctl_mllr")], d, "ref_ctl")
    one = run(REF, args + ["-mllr", str(d / "m1")], d, "ref_m1")
    assert ref[1].count("\n") == 10 and ref[1] != one[1]
    got = run(TST, args + ["-ctl_mllr", str(d / "ctl_mllr")], d, "dev_ctl_" + "_".join(env.values()), env)
    assert got == ref
