"""LM sets through the whole-utterance engine (MI355X): -lmctlfn (several LMs, class-based ones with their -probdef classes),
-lmname, -ctl_lm (an LM per utterance; utt.c:240-241 -> srch_set_lm -> srch_TST_set_lm, srch_time_switch_tree.c:260-330: every LM
has unigram lextrees of its own).  The drop-in keeps a search space, a trigram and engines per LM and switches between utterances;
class words score with lm_t.inclass_ugscore on the device (s3a_wordlevel.h).  -hyp / -hypseg byte-identical to the unmodified
reference.  The LM files are test data of this repository (tests/golden/tidigits_clm: a class bigram over the tidigits words).

(The reference itself faults in srch_TST_uninit -> lextree_free when it shuts down with more than one LM -- after every output
file is complete: its exit status is not asserted in those runs.)"""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
C = os.path.join(GOLDEN, "tidigits_clm")
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")


def base_args():
    return ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra", "-agc", "none",
            "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4"]


def run(exe, args, tmp_path, tag, env=None, rc_ok=(0,)):
    hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT, timeout=1200, env=env,
                           cwd=str(tmp_path))
    txt = open(log, errors="ignore").read()
    assert p.returncode in rc_ok, "\n".join(l for l in txt.splitlines() if "FATAL" in l or "tst shim" in l)[-2000:]
    return open(hyp).read(), open(seg).read(), txt


@pytest.fixture()
def lmset(tmp_path):
    for b in (REFDEC, TST):
        if not os.path.exists(b):
            pytest.fail(f"{b} is missing on the GPU box (make -C oracle ref)")
    one = tmp_path / "one.lmctl"
    one.write_text(f"{{ {C}/digits.probdef }}\n{C}/digits.cls.lm digitclass {{\n[low]\n[high]\n}}\n")
    two = tmp_path / "two.lmctl"
    two.write_text(f"{{ {C}/digits.probdef }}\n{C}/digits.cls.lm digitclass {{\n[low]\n[high]\n}}\n{D}/tidigits.DMP plain\n")
    n = sum(1 for l in open(f"{D}/tidigits.length.arb.regression") if l.strip())
    ctl_lm = tmp_path / "ctl_lm"
    ctl_lm.write_text("".join(("plain\n" if k % 3 == 2 else "digitclass\n") for k in range(n)))
    return str(one), str(two), str(ctl_lm)


@pytest.mark.parametrize("env", [{"S3A_UTT": "5"}, {"S3A_UTT": "3", "S3A_UTT_QUEUE": "9"}])
def test_one_class_based_lm(lmset, env, tmp_path):
    one, _, _ = lmset
    args = base_args() + ["-lmctlfn", one, "-lmname", "digitclass"]
    ref = run(REFDEC, args, tmp_path, "ref")
    plain = open(f"{D}/ref_mode4_trigram.matchseg").read()
    assert ref[0].count("\n") == 31 and ref[1] != plain          # (the class LM is no bystander)
    gpu = run(TST, args, tmp_path, "gpu", env=dict(os.environ, **env))
    assert gpu[0] == ref[0] and gpu[1] == ref[1]


@pytest.mark.parametrize("env,extra", [({"S3A_UTT": "4"}, []), ({"S3A_UTT": "2", "S3A_UTT_QUEUE": "6"}, []),
                                       ({"S3A_UTT": "6", "S3A_UTT_ENGINES": "2"}, ["-bestpath", "1"])])
def test_an_lm_per_utterance(lmset, env, extra, tmp_path):
    """-ctl_lm: two of three utterances with the class bigram, the third with the plain trigram; what is queued is decoded with
    the LM it was queued for, then the other LM's lextrees / trigram / engines become current"""
    _, two, ctl_lm = lmset
    args = base_args() + ["-lmctlfn", two, "-ctl_lm", ctl_lm, "-lmname", "plain"] + extra
    ref = run(REFDEC, args, tmp_path, "ref", rc_ok=(0, -11))        # (faults at exit, after the files are complete: see above)
    assert ref[0].count("\n") == 31
    gpu = run(TST, args, tmp_path, "gpu", env=dict(os.environ, **env))
    assert "search space and" in gpu[2]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    # both LMs decided something: the all-plain and the all-class results differ from this one
    assert ref[1] != open(f"{D}/ref_mode4_trigram.matchseg").read()
