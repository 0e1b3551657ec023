"""Oracle pinning, part 8: pocketsphinx's first pass (oracle/s3o_psfwd.c: ngram_fwdtree_search and everything under it,
hmm_vit_eval in pocketsphinx's conventions, ngram_tg_score on the flat trigram -- SURVEY.md 8(f).3).

oracle/_ref/ref_ps_ofwd = the unmodified pocketsphinx decoder (libpsref.so, built from /root/reference's sources as they
lie) with ps_searchfuncs_t {start, step, finish} served by the restatement.  Against the unmodified decoder, live, same box:
hypothesis strings + path scores, the segment iterator's (word, sf, ef, ascr, lscr), and -- with the later passes off --
the WHOLE backpointer table (every bptbl_t field, the right-context score stack, the frame index, the search statistics).
Cases: pocketsphinx's own test-tidigits-simple (semi-continuous model, 31 utterances in one decoder, default passes:
fwdflat + bestpath run on the restatement's table), the same first-pass-only, a continuous model, goforward.raw with
turtle.DMP (BASELINE configs[0]'s utterance), and the 5000-word Mandarin trigram (97k dictionary words, an LM file with
UNSORTED n-gram runs, 363 roots / 6355 interior channels) with default beams, with histogram + -maxwpf pruning, with
-compallsen and a new decoder per utterance.  ref_ps_hmmcheck: hmm_vit_eval on random HMMs of every kind.
The reference's checked-in test-tidigits-simple.match agrees on every hypothesis string (its scores come from another build).
"""
import os
import subprocess

import pytest

import psfwd_cases as P


def both(args, tmp_path, env=None):
    return P.run("ref_ps_fwd", args, tmp_path, "ref"), P.run("ref_ps_ofwd", args, tmp_path, "ofwd", env=env)


def test_pocketsphinx_regression_default_passes(tmp_path):
    r, o = both(P.sc_args(tmp_path), tmp_path)
    P.assert_same(r, o, tables=False)
    assert "first pass served by oracle/s3o_psfwd.c" in o[3]
    gold = [l.rsplit("(", 1)[0].strip() for l in open(os.path.join(P.SCD, "test-tidigits-simple.match"))]
    assert [l.rsplit("(", 1)[0].strip() for l in r[0].splitlines()] == gold


def test_pocketsphinx_regression_tables(tmp_path):
    r, o = both(P.sc_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path, env={"PSO_LMCHECK": "1"})
    P.assert_same(r, o)
    assert "LM check:" in o[3] and "scores identical" in o[3]
    assert sum(v["bpidx"] for v in r[2].values()) > 5000


def test_continuous_model_tables(tmp_path):
    r, o = both(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, o)
    assert r[0].startswith("ONE ONE ONE (man/man.ah.111a")


def test_new_decoder_per_utterance(tmp_path):
    """-fresh yes: what the engine's whole-utterance lanes reproduce (results independent of the lane's history)"""
    r, o = both(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes"], tmp_path)
    P.assert_same(r, o)


@pytest.mark.parametrize("model,window", [("sc", 3), ("cont", 5), ("cont", 1)])
def test_phone_loop_lookahead(tmp_path, model, window):
    """-pl_window: the decoder's own phone loop (phone_loop_search.c, stepped in front of the search) hands its scores to the
    restatement's transitions; whole tables against the unmodified decoder, and the look-ahead must matter"""
    args = (P.sc_args if model == "sc" else P.cont_args)(tmp_path) + P.FIRST_PASS_ONLY
    r, o = both(args + ["-pl_window", str(window)], tmp_path)
    P.assert_same(r, o)
    r0 = P.run("ref_ps_fwd", args, tmp_path, "ref0")
    assert r0[0] != r[0], "the look-ahead changed no path score: the case does not test it"


@pytest.mark.parametrize("two", [False, True], ids=["one_lm", "set_of_two"])
def test_class_based_lm(tmp_path, two):
    """-lmctl with a class-based LM (ngram_ng_score's declassification, sphinxbase ngram_model.c:494-521): a class word scores as
    its class's tag word plus its in-class weight"""
    r, o = both(P.class_lm_args(tmp_path, two) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, o)
    plain = P.run("ref_ps_fwd", P.cont_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path, "plain")
    assert plain[0] != r[0], "the class LM changed no path score: the case does not test it"
    assert "Added class [low]" in r[3]


def test_goforward_raw(tmp_path):
    r, o = both(P.turtle_args(tmp_path, ("goforward", "numbers", "something")) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, o)
    assert r[0].startswith("go forward ten meters (goforward")


@pytest.mark.parametrize("extra", [[], ["-maxhmmpf", "800", "-maxwpf", "5", "-beam", "1e-60", "-wbeam", "1e-30"],
                                   ["-compallsen", "yes", "-fresh", "yes"]], ids=["default", "pruned", "allsen_fresh"])
def test_mandarin_trigram_tables(tmp_path, extra):
    r, o = both(P.zh_args(tmp_path) + P.FIRST_PASS_ONLY + extra, tmp_path, env={"PSO_LMCHECK": "3"})
    P.assert_same(r, o)
    assert "LM check:" in o[3]
    assert sum(v["bpidx"] for v in r[2].values()) > 30000


def test_mandarin_default_passes(tmp_path):
    r, o = both(P.zh_args(tmp_path, ("goforward",)), tmp_path)
    P.assert_same(r, o, tables=False)


def test_hmm_vit_eval_on_random_hmms():
    exe = os.path.join(P.REF, "ref_ps_hmmcheck")
    P.need(exe)
    p = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "200000 cases identical" in p.stdout, p.stdout[-2000:]
