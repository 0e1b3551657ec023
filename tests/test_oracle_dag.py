"""Oracle pinning, part 7: the second pass (oracle/s3o_dag.c: vithist_dag_build, dag_bypass_filler_nodes, dag_search /
dag_bestpath, dag_backtrace -- SURVEY.md 8(f).4).

oracle/_ref/ref_s3odag_decode = the unmodified reference decoder with its `bestpath_impl` slot served by the restatement
on the plain arrays of the reference's own history table.  With -bestpath 1 its -hyp / -hypseg must be byte-identical to
the unmodified reference's (live, same box); the program itself aborts when the restated lattice's node / link counts
differ from the reference's dag_t.  Cases: tidigits (31 utterances), RM1 (20 utterances, 997-word trigram: the second
pass CHANGES hypotheses there), -bestpathlw (a language-weight factor != 1: the float arithmetic of the bypass and the
LM scores), -min_endfr 1 (more nodes survive), a tight -maxlpf (the LM-operation limit makes the search fail: no line).
"""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
ODAG = os.path.join(ROOT, "oracle", "_ref", "ref_s3odag_decode")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
RM = os.path.join(ROOT, "tests", "_local_data", "rm1")
HAVE_REF = os.path.isdir("/root/reference")


def need(path):
    if os.path.exists(path):
        return
    if HAVE_REF or os.environ.get("S3A_ON_GPU_BOX"):
        pytest.fail(f"{path} is missing (make -C oracle ref; tools/fetch_local_data.sh)")
    pytest.skip(f"{path} not present (no /root/reference here)")


def tidigits_args():
    return ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra", "-agc", "none",
            "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4",
            "-lm", f"{D}/tidigits.DMP"]


def rm_args(n=20):
    return ["-mdef", f"{RM}/mdef", "-fdict", f"{RM}/fillerdict", "-dict", f"{RM}/RM.dictionary", "-mean", f"{RM}/means",
            "-var", f"{RM}/variances", "-mixw", f"{RM}/mixture_weights", "-tmat", f"{RM}/transition_matrices",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-epl", "4", "-fillprob", "0.02", "-maxwpf", "10",
            "-wip", "0.2", "-lm", f"{RM}/RM.2845.trigram.arpa.DMP", "-lw", "14", "-beam", "1e-140", "-wbeam", "1e-100",
            "-cepdir", f"{RM}/feat", "-cepext", ".mfc", "-ctl", f"{RM}/rm.ctl", "-ctlcount", str(n), "-op_mode", "4"]


def both(args, tmp_path):
    out = {}
    for tag, exe in (("ref", REFDEC), ("odag", ODAG)):
        hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
        with open(log, "w") as lf:
            p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT, timeout=900)
        txt = open(log, errors="ignore").read()
        assert p.returncode == 0, "\n".join(l for l in txt.splitlines() if "FATAL" in l or "dag oracle" in l)[-2000:]
        out[tag] = (open(hyp).read(), open(seg).read(), txt)
    assert "second pass of" in out["odag"][2]
    return out


@pytest.mark.parametrize("extra", [["-bestpath", "1"], ["-bestpath", "1", "-bestpathlw", "14", "-min_endfr", "1"]])
def test_tidigits_second_pass_from_the_oracle_is_byte_identical(extra, tmp_path):
    need(ODAG); need(REFDEC)
    out = both(tidigits_args() + extra, tmp_path)
    assert out["odag"][0] == out["ref"][0] and out["odag"][1] == out["ref"][1]
    assert out["ref"][0].count("\n") == 31


@pytest.mark.parametrize("extra", [["-bestpath", "1"], ["-bestpath", "1", "-bestpathlw", "9.5"], ["-bestpath", "1", "-min_endfr", "0", "-maxwpf", "20"]])
def test_rm1_second_pass_from_the_oracle_is_byte_identical(extra, tmp_path):
    need(ODAG); need(REFDEC); need(RM)
    args = rm_args()
    if "-maxwpf" in extra:
        i = args.index("-maxwpf"); del args[i:i + 2]
    out = both(args + extra, tmp_path)
    assert out["odag"][0] == out["ref"][0] and out["odag"][1] == out["ref"][1]
    assert out["ref"][0].count("\n") == 20
    # the second pass is not a no-op on this task: its output differs from the first pass's
    p = subprocess.run([REFDEC] + args + [x for x in extra if x not in ("-bestpath", "1")] + ["-hyp", str(tmp_path / "fp.match")],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    assert p.returncode == 0


def test_lm_operation_limit_fails_the_same_way(tmp_path):
    """-maxlpf 5: dag_bestpath gives up after 5 x nfrm LM operations ("Bestpath search failed"): the reference then writes
    NO line for the utterance; utterances whose search needs fewer operations write theirs"""
    need(ODAG); need(REFDEC); need(RM)
    out = both(rm_args(8) + ["-bestpath", "1", "-maxlpf", "5"], tmp_path)
    assert out["odag"][0] == out["ref"][0] and out["odag"][1] == out["ref"][1]
    assert 0 < out["ref"][0].count("\n") < 8
    assert "Bestpath search failed" in out["ref"][2] and "Bestpath search failed" in out["odag"][2]
