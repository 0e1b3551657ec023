"""The pocketsphinx first-pass cases shared by tests/test_oracle_psfwd.py (restatement vs unmodified pocketsphinx, CPU)
and tests/test_gpu_psfwd.py (device vs unmodified pocketsphinx, MI355X).  Every case is a command line of
oracle/ref_ps_fwd.c's drivers over data held under tests/golden/ (model files, dictionaries, LMs, cepstra and audio the
reference's own tests hold) or tests/_local_data/ps (tools/fetch_local_data.sh: the larger ones, git-ignored)."""
import os
import subprocess

import pytest

import psfwd_dump
from conftest import GOLDEN, ROOT

REF = os.path.join(ROOT, "oracle", "_ref")
LOCAL = os.path.join(ROOT, "tests", "_local_data", "ps")
HAVE_REF = os.path.isdir("/root/reference")
TD = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
SCD = os.path.join(GOLDEN, "ps_tidigits_sc")
TUR = os.path.join(GOLDEN, "ps_turtle")


def need(path):
    if os.path.exists(path):
        return
    if HAVE_REF or os.environ.get("S3A_ON_GPU_BOX"):
        pytest.fail(f"{path} is missing (make -C oracle ref; tools/fetch_local_data.sh)")
    pytest.skip(f"{path} not present (no /root/reference here)")


def ctl_file(tmp_path, name, lines):
    p = tmp_path / name
    p.write_text("".join(l + "\n" for l in lines))
    return str(p)


def cont_args(tmp_path):
    """the sphinx3 tidigits continuous model (8 Gaussians, text mdef) read by pocketsphinx: -senmgau .cont."""
    ctl = ctl_file(tmp_path, "tdc.ctl", [l.split()[0] for l in open(os.path.join(TD, "tidigits.length.arb.regression"))])
    return ["-mdef", f"{AM}/mdef", "-mean", f"{AM}/means", "-var", f"{AM}/variances", "-mixw", f"{AM}/mixture_weights",
            "-tmat", f"{AM}/transition_matrices", "-senmgau", ".cont.", "-topn", "4", "-dict", f"{TD}/tidigits.ps.dic",
            "-fdict", f"{TD}/fillerdict", "-lm", f"{TD}/tidigits.DMP", "-ctl", ctl, "-cepdir", f"{TD}/cepstra"]


def sc_args(tmp_path):
    """pocketsphinx's own regression test-tidigits-simple.sh: model/hmm/en/tidigits (semi-continuous) + tidigits.DMP"""
    return ["-hmm", SCD, "-lm", f"{TD}/tidigits.DMP", "-dict", f"{SCD}/tidigits.dic", "-ctl", f"{SCD}/tidigits.ctl",
            "-cepdir", f"{SCD}/cepstra"]


def turtle_args(tmp_path, utts=("goforward",)):
    need(os.path.join(LOCAL, "hub4wsj_sc_8k", "mdef"))
    ctl = ctl_file(tmp_path, "raw.ctl", utts)
    return ["-hmm", f"{LOCAL}/hub4wsj_sc_8k", "-lm", f"{TUR}/turtle.DMP", "-dict", f"{TUR}/turtle.dic", "-ctl", ctl,
            "-cepdir", f"{LOCAL}/raw", "-cepext", ".raw", "-adcin", "yes"]


def zh_args(tmp_path, utts=("goforward", "numbers", "something")):
    need(os.path.join(LOCAL, "zh_CN", "gigatdt.5000.DMP"))
    ctl = ctl_file(tmp_path, "raw.ctl", utts)
    return ["-hmm", f"{LOCAL}/tdt_sc_8k", "-lm", f"{LOCAL}/zh_CN/gigatdt.5000.DMP", "-dict", f"{LOCAL}/zh_CN/mandarin_notone.dic",
            "-ctl", ctl, "-cepdir", f"{LOCAL}/raw", "-cepext", ".raw", "-adcin", "yes"]


def class_lm_args(tmp_path, two=False):
    """the continuous tidigits model with a CLASS-based bigram through -lmctl / -lmname (tests/golden/tidigits_clm, written for
    the sphinx3 LM-set tests: its words in this dictionary's upper case); two: a set of two LMs, the class bigram current"""
    import re
    src = os.path.join(GOLDEN, "tidigits_clm")
    up = lambda t: re.sub(r"\b(one|two|three|four|five|six|seven|eight|nine|oh|zero)\b", lambda m: m.group(1).upper(), t)
    for f in ("digits.probdef", "digits.cls.lm"):
        (tmp_path / f).write_text(up(open(os.path.join(src, f)).read()))
    ctl = tmp_path / "set.lmctl"
    ctl.write_text(f"{{ {tmp_path}/digits.probdef }}\n{tmp_path}/digits.cls.lm digitclass {{\n[low]\n[high]\n}}\n"
                   + (f"{TD}/tidigits.DMP plain\n" if two else ""))
    args = cont_args(tmp_path)
    i = args.index("-lm")
    del args[i:i + 2]
    return args + ["-lmctl", str(ctl), "-lmname", "digitclass"]


FIRST_PASS_ONLY = ["-fwdflat", "no", "-bestpath", "no"]


def run(exe, args, tmp_path, tag, env=None, timeout=900):
    """-> (match text, seg text, bpdump dict, log text)"""
    need(os.path.join(REF, exe))
    m, s, b, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "seg", "bp", "log"))
    dump = [] if "-queue" in args else ["-bpdump", b]       # (the queue keeps hypotheses, not tables)
    with open(log, "w") as lf:
        p = subprocess.run([os.path.join(REF, exe)] + args + ["-hyp", m, "-hypseg", s] + dump, stdout=lf,
                           stderr=subprocess.STDOUT, timeout=timeout, env=dict(os.environ, **(env or {})))
    txt = open(log, errors="ignore").read()
    assert p.returncode == 0, f"{exe} failed:\n" + "\n".join(l for l in txt.splitlines() if "FATAL" in l or "ERROR" in l)[-2000:]
    return open(m).read(), open(s).read(), psfwd_dump.read_bpdump(b) if dump else None, txt


def assert_same(a, b, tables=True):
    assert a[0] == b[0], "hypotheses / path scores differ"
    assert a[1] == b[1], "segmentations (word, frames, acoustic and LM score) differ"
    if tables:
        d = psfwd_dump.diff_bpdumps(a[2], b[2])
        assert d == [], "\n".join(d[:10])
