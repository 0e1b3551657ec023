"""Parity (MI355X): pocketsphinx's first pass on the device (cmusphinx_amd/csrc/s3a_psfwd.hip behind
ps_searchfuncs_t, integration/pocketsphinx/ps_search_amd.c) against the unmodified pocketsphinx decoder run live on the
same box (oracle/_ref/ref_ps_fwd): hypothesis strings AND path scores, the segmentations, and -- first pass only --
the whole backpointer table, entry for entry in the reference's order, the right-context score stack, the search
statistics.  Frame-synchronous (start / step / finish with the decoder's own scorer: semi-continuous or continuous),
with the reference's fwdflat + bestpath on top of the device's table, and as whole utterances in 1 .. 31 lanes
(scoring + search on the device, hypotheses made on the device).  Bit-exact: integer work."""
import pytest

import psfwd_cases as P

pytestmark = pytest.mark.gpu


def pair(args, tmp_path, amd_extra=()):
    r = P.run("ref_ps_fwd", args, tmp_path, "ref")
    a = P.run("ref_ps_amdfwd", args + list(amd_extra), tmp_path, "amd")
    assert "first pass served by libcmusphinx_amd" in a[3] or "first pass served by" in a[3]
    return r, a


def test_pocketsphinx_regression_default_passes(tmp_path):
    """test-tidigits-simple.sh as pocketsphinx runs it: 31 utterances through one decoder, fwdflat + bestpath"""
    r, a = pair(P.sc_args(tmp_path), tmp_path)
    P.assert_same(r, a, tables=False)


def test_pocketsphinx_regression_tables(tmp_path):
    r, a = pair(P.sc_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, a)


def test_continuous_model_frame_synchronous(tmp_path):
    r, a = pair(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, a)


@pytest.mark.parametrize("lanes", [1, 8, 31])
def test_whole_utterances_on_the_device(tmp_path, lanes):
    r, a = pair(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes"], tmp_path, ["-batch", str(lanes)])
    P.assert_same(r, a)


def test_whole_utterances_all_senones(tmp_path):
    r, a = pair(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes", "-compallsen", "yes"], tmp_path, ["-batch", "16"])
    P.assert_same(r, a)


def test_whole_utterances_keep_decoder_state(tmp_path):
    """one lane, utterance after utterance WITHOUT resets = one reference decoder used the same way"""
    r, a = pair(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path, ["-batch", "1"])
    P.assert_same(r, a)


@pytest.mark.parametrize("lanes", [1, 4, 31])
def test_whole_utterances_from_a_queue(tmp_path, lanes):
    """s3a_psfwd_decode_queue: 31 utterances through 1 / 4 / 31 persistent lanes, a lane taking the next utterance when
    its own has ended (light reset in between): every result is a new decoder's, whatever lane decoded it"""
    args = P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes"]
    r = P.run("ref_ps_fwd", args, tmp_path, "ref")
    a = P.run("ref_ps_amdfwd", args + ["-batch", str(lanes), "-queue", "yes"], tmp_path, "amd")
    assert "[queue]" in a[3] or lanes == 31
    assert a[0] == r[0] and a[1] == r[1]


# ---- -pl_window: the phone loop look-ahead (phone_loop_search.c; pocketsphinx.c:242-248, :704-712, :823-826) ----
@pytest.mark.parametrize("model,window", [("sc", 3), ("cont", 5)])
def test_lookahead_frame_synchronous(tmp_path, model, window):
    """the decoder's own phone loop stays on the host and hands its scores over before every step (s3a_psfwd_set_lookahead)"""
    args = (P.sc_args if model == "sc" else P.cont_args)(tmp_path) + P.FIRST_PASS_ONLY
    r, a = pair(args + ["-pl_window", str(window)], tmp_path)
    P.assert_same(r, a)
    r0 = P.run("ref_ps_fwd", args, tmp_path, "ref0")
    assert r0[0] != r[0], "the look-ahead changed no path score: the case does not test it"


@pytest.mark.parametrize("lanes,window,extra", [(1, 3, []), (8, 5, []), (31, 1, []), (16, 4, ["-compallsen", "yes"]),
                                                (8, 2, ["-pl_beam", "1e-3", "-pl_pbeam", "1e-2"])],
                         ids=["1x3", "8x5", "31x1", "allsen", "narrow_loop_beams"])
def test_lookahead_whole_utterances(tmp_path, lanes, window, extra):
    """the phone loop ON THE DEVICE, inside the lane's launch, pl_window frames ahead of the lane's search: tables identical"""
    args = P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes", "-pl_window", str(window)] + extra
    r, a = pair(args, tmp_path, ["-batch", str(lanes)])
    P.assert_same(r, a)


def test_lookahead_whole_utterances_keep_decoder_state(tmp_path):
    """one lane, no resets: the senones the last utterance's final search step flagged are still flagged when the next
    utterance's phone loop starts (acmod_start_utt does not clear them) -- they take part in its frames' normalisation"""
    r, a = pair(P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-pl_window", "4"], tmp_path, ["-batch", "1"])
    P.assert_same(r, a)


@pytest.mark.parametrize("lanes", [4, 31])
def test_lookahead_from_a_queue(tmp_path, lanes):
    args = P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-fresh", "yes", "-pl_window", "3"]
    r = P.run("ref_ps_fwd", args, tmp_path, "ref")
    a = P.run("ref_ps_amdfwd", args + ["-batch", str(lanes), "-queue", "yes"], tmp_path, "amd")
    assert a[0] == r[0] and a[1] == r[1]


def test_lookahead_default_passes(tmp_path):
    """fwdflat + bestpath (the reference's host code) on the device's table, the look-ahead on"""
    r, a = pair(P.sc_args(tmp_path) + ["-pl_window", "2"], tmp_path)
    P.assert_same(r, a, tables=False)


# ---- class-based LMs (-lmctl with class definitions; sphinxbase ngram_model.c:494-521) ----
def test_class_based_lm_frame_synchronous(tmp_path):
    r, a = pair(P.class_lm_args(tmp_path) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, a)
    assert "Added class [low]" in r[3]


@pytest.mark.parametrize("lanes,extra", [(8, []), (31, ["-pl_window", "3"])], ids=["8", "31_lookahead"])
def test_class_based_lm_whole_utterances(tmp_path, lanes, extra):
    """a set of two LMs, the class bigram the current one; hypotheses (segment LM scores included) made on the device"""
    r, a = pair(P.class_lm_args(tmp_path, two=True) + P.FIRST_PASS_ONLY + ["-fresh", "yes"] + extra, tmp_path, ["-batch", str(lanes)])
    P.assert_same(r, a)


def test_class_based_lm_default_passes(tmp_path):
    r, a = pair(P.class_lm_args(tmp_path), tmp_path)
    P.assert_same(r, a, tables=False)


def test_goforward_raw(tmp_path):
    r, a = pair(P.turtle_args(tmp_path, ("goforward", "numbers", "something")) + P.FIRST_PASS_ONLY, tmp_path)
    P.assert_same(r, a)
    assert r[0].startswith("go forward ten meters (goforward")


@pytest.mark.parametrize("extra", [[], ["-maxhmmpf", "800", "-maxwpf", "5", "-beam", "1e-60", "-wbeam", "1e-30"]],
                         ids=["default", "pruned"])
def test_mandarin_trigram_tables(tmp_path, extra):
    r, a = pair(P.zh_args(tmp_path) + P.FIRST_PASS_ONLY + extra, tmp_path)
    P.assert_same(r, a)


def test_mandarin_default_passes(tmp_path):
    r, a = pair(P.zh_args(tmp_path, ("goforward",)), tmp_path)
    P.assert_same(r, a, tables=False)


def test_partial_results_while_the_utterance_is_open(tmp_path):
    """live use: ps_get_hyp between ps_start_utt and ps_end_utt (the cepstra in blocks of 25 frames).  The binding resets the
    decoder's own table at start and brings the device's table over before the reference's ngram_search_hyp reads it: every
    partial string and score, and the final results, are the unmodified decoder's -- also for the SECOND utterance of a
    decoder, which must not see the first one's table."""
    args = P.cont_args(tmp_path) + P.FIRST_PASS_ONLY + ["-partial", "25"]
    r, a = pair(args, tmp_path)
    P.assert_same(r, a)
    rp, ap = open(str(tmp_path / "ref.match.partial")).read(), open(str(tmp_path / "amd.match.partial")).read()
    assert rp.count("\n") > 100 and rp == ap
