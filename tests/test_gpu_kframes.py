"""ku_frames (s3a_uttdec_opts_t.persist / .cluster, round 5): the lane's frame as the phases of ONE persistent workgroup -- or a
cluster of them with a counter barrier -- instead of twelve launches per frame, every frame's senone scores computed before the
search starts, the queue's lanes taking their utterances themselves.  The library chooses it from 8 busy lanes on (below that
the launches win); here S3A_UTT_PERSIST=1 forces it for engines of 1 .. 40 lanes so that every mode runs on the small tasks:

  KF_STATIC (s3a_uttdec_decode), KF_QUEUE (s3a_uttdec_decode_queue: no schedule, lanes refill themselves, a lane whose utterance
  overflowed is scrubbed inside the kernel), KF_WINDOW (a queue with the second pass keeps window blocks), clusters of 1 / 2 / 3
  workgroups per lane, 3- and 5-state HMMs, histogram pruning (-maxhmmpf 20: the sort inside the kernel), -ptranskip with a phone
  beam wider than the HMM beam (d_dec_weak_t), the CI gate with -ds 2, the hub4-shaped task (several-parent sets, composite senones).

Expected output: the unmodified reference's -hyp / -hypseg (committed goldens or produced live), byte for byte -- the same
bar as the launch path's tests, whose bodies these reuse."""
import os
import subprocess

import numpy as np
import pytest

import test_gpu_5state as T5
import test_gpu_dropin as TD
import test_gpu_queue as TQ
import test_gpu_uttdec as TU
from cmusphinx_amd import bundle
from test_gpu_5state import task  # noqa: F401  (fixture)
from test_gpu_uttdec import tidigits_bundle  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

FORCE = {"S3A_UTT_PERSIST": "1"}


@pytest.fixture
def persist(monkeypatch):
    monkeypatch.setenv("S3A_UTT_PERSIST", "1")


@pytest.mark.parametrize("n_lanes,cluster", [(1, 1), (3, 2), (8, 3), (40, 1)])
def test_queue_lanes_take_their_utterances_themselves(gpu_lib, tidigits_bundle, monkeypatch, persist, n_lanes, cluster):
    """KF_QUEUE on the ragged tidigits set, lanes = workgroups or clusters of 2 / 3; the engine reports that it ran ku_frames"""
    monkeypatch.setenv("S3A_UTT_CLUSTER", str(cluster))
    dec = bundle.Decoder(tidigits_bundle, n_lanes)
    utts, feats = TQ.tidigits_feats(gpu_lib)
    dec.decode_queue(feats)
    m, s = TQ.queue_lines(dec, utts)
    rm, rs = TQ.ref_lines()
    assert m == rm and s == rs
    parts = dec.ud.last_parts()
    assert parts["n_frames"] >= 1 and parts["n_score"] >= 1 and parts["cluster"] == cluster and parts["frames_ms"] > 0
    ticks, frames, launches, c = dec.ud.frame_ticks(0)
    assert frames > 0 and launches >= 1 and ticks["hmm_eval"] > 0


def test_clusters_with_the_general_barrier(gpu_lib, tidigits_bundle, monkeypatch):
    """S3A_UTT_PERSIST=2: the clusters' agent-scope barrier only (an L2 write-back per workgroup and step) -- what a cluster falls
    back to when its workgroups do not share an XCD; the default is the XCD-local barrier (checked per launch through XCC_ID)"""
    monkeypatch.setenv("S3A_UTT_PERSIST", "2")
    monkeypatch.setenv("S3A_UTT_CLUSTER", "3")
    dec = bundle.Decoder(tidigits_bundle, 5)
    utts, feats = TQ.tidigits_feats(gpu_lib)
    dec.decode_queue(feats)
    m, s = TQ.queue_lines(dec, utts)
    rm, rs = TQ.ref_lines()
    assert m == rm and s == rs
    assert dec.ud.last_parts()["cluster"] == 3


def test_launches_and_ku_frames_leave_the_same_tables(gpu_lib, tidigits_bundle, monkeypatch):
    """s3a_uttdec_decode both ways: every lane's whole history table and frame statistics, word for word"""
    utts, feats = TQ.tidigits_feats(gpu_lib)
    res = {}
    for mode in ("-1", "1"):
        monkeypatch.setenv("S3A_UTT_PERSIST", mode)
        dec = bundle.Decoder(tidigits_bundle, 6)
        dec.decode(feats[:6])
        res[mode] = [dec.ud.result(z) for z in range(6)]
        assert (dec.ud.last_parts()["n_frames"] > 0) == (mode == "1")
    for a, b in zip(res["-1"], res["1"]):
        assert len(a["score"]) == len(b["score"]) > 1 and a["n_frm"] == b["n_frm"]
        for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type", "frame_start", "bestscore", "bestvh", "frame_stat"):
            assert np.array_equal(a[k], b[k]), k


def test_queue_reuse_overflow_and_static_decodes(gpu_lib, tidigits_bundle, persist):
    """the launch path's own tests of engine reuse and of a lane scrubbed after a capacity error, through ku_frames"""
    TQ.test_queue_order_and_reuse_of_an_engine(gpu_lib, tidigits_bundle)
    TQ.test_a_lane_whose_utterance_overflowed_is_scrubbed_on_the_device(gpu_lib, tidigits_bundle)
    TU.test_every_lane_that_overflowed_restarts_clean(gpu_lib, tidigits_bundle)
    TU.test_hypothesis_records_are_fixed_size_and_self_contained(gpu_lib, tidigits_bundle)


def test_second_pass_inside_a_queue_as_groups(gpu_lib, tidigits_bundle, persist, monkeypatch):
    """a queue with -bestpath 1 (round 6): groups of at most n_lanes utterances, longest first, each ONE KF_STATIC launch, then the lanes'
    hypotheses, vithist_utt_end + the second pass on the lanes' own tables, lextree_utt_end -- nothing waited for between groups; when a
    group's scores do not fit the score buffer (S3A_UTT_SCORE_ROWS): K-frame window blocks between refill events, as before"""
    TQ.test_second_pass_inside_the_queue_through_the_c_abi(gpu_lib, tidigits_bundle)
    monkeypatch.setenv("S3A_UTT_SCORE_ROWS", "90")
    TQ.test_second_pass_inside_the_queue_through_the_c_abi(gpu_lib, tidigits_bundle)


def test_the_score_buffer_in_parts(gpu_lib, tidigits_bundle, persist, monkeypatch):
    """s3a_uttdec_opts_t.score_rows_max (S3A_UTT_SCORE_ROWS in this harness) caps the rows of senone scores a call keeps on the device:
    the queue then goes through in PARTS -- consecutive utterances whose rows fit, one ku_frames launch per part, the buffer reused,
    every part's utterances taken longest first -- which an uncapped 288 GB device never does by itself.  At least three parts here;
    then the same engine uncapped-sized calls again (buffers that already exist), and the cap too small for one utterance is refused"""
    utts, feats = TQ.tidigits_feats(gpu_lib)
    rm, rs = TQ.ref_lines()
    longest, total = max(len(f) for f in feats), sum(len(f) for f in feats)
    cap = max(longest, total // 5)
    monkeypatch.setenv("S3A_UTT_SCORE_ROWS", str(cap))
    dec = bundle.Decoder(tidigits_bundle, 4)
    for _ in range(2):
        dec.decode_queue(feats)
        m, s = TQ.queue_lines(dec, utts)
        assert m == rm and s == rs
        parts = dec.ud.last_parts()
        assert parts["n_frames"] >= 3 and parts["n_score"] >= parts["n_frames"], parts          # (one ku_frames launch per part)
    dec.decode_queue(feats[:9])
    m, s = TQ.queue_lines(dec, utts[:9])
    assert m == rm[:9] and s == rs[:9]
    monkeypatch.setenv("S3A_UTT_SCORE_ROWS", str(longest - 1))
    small = bundle.Decoder(tidigits_bundle, 4)
    with pytest.raises(Exception, match="do not fit"):
        small.decode_queue(feats)


def test_static_decode_falls_back_to_window_blocks(gpu_lib, tidigits_bundle, persist, monkeypatch):
    """s3a_uttdec_decode with more frames than the score buffer may hold: kf_decode_static answers S3A_EUNSUP and the lanes' frames go
    through ku_frames in K-frame window blocks (KF_WINDOW) -- the same tables, word for word, as the uncapped call"""
    utts, feats = TQ.tidigits_feats(gpu_lib)
    res = {}
    for cap in ("0", "64"):
        monkeypatch.setenv("S3A_UTT_SCORE_ROWS", cap)
        dec = bundle.Decoder(tidigits_bundle, 6)
        dec.decode(feats[:6])
        res[cap] = [dec.ud.result(z) for z in range(6)]
        parts = dec.ud.last_parts()
        assert (parts["n_frames"] == 1 if cap == "0" else parts["n_frames"] > 4) and parts["cluster"] >= 1, parts    # (a launch per K-frame block)
        for z, (u, uid) in enumerate(utts[:6]):
            rec = dec.hyp(z, uid, z)
            assert rec.status == 0 and dec.format(rec)[0] == TQ.ref_lines()[0][z]
    for a, b in zip(res["0"], res["64"]):
        for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type", "frame_start", "bestscore", "bestvh", "frame_stat"):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name,lanes,cluster", [("mode4_trigram", "4", "1"), ("mode4_cibeam_ds2", "4", "2"), ("mode4_cibeam_ds2", "40", "1")])
def test_drop_in_program_through_ku_frames(name, lanes, cluster, tmp_path):
    env = dict(FORCE, S3A_UTT=lanes, S3A_UTT_CLUSTER=cluster)
    hyp, seg, log = (str(tmp_path / f"kf_{name}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([TD.TST] + TD.common() + TD.RUNS[name] + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT,
                           timeout=900, env=dict(os.environ, **env))
    assert p.returncode == 0, open(log, errors="ignore").read()[-2000:]
    assert open(hyp).read() == open(os.path.join(TD.D, f"ref_{name}.match")).read()
    assert open(seg).read() == open(os.path.join(TD.D, f"ref_{name}.matchseg")).read()


@pytest.mark.parametrize("what,extra", [("histogram", ["-maxhmmpf", "20"]),
                                        ("weak", ["-ptranskip", "2", "-beam", "1e-80", "-pbeam", "1e-100", "-wbeam", "1e-40"])])
def test_histogram_pruning_and_weak_hmms_inside_the_kernel(what, extra, tmp_path):
    args = TD.RUNS["mode4_trigram"] + extra
    ref_hyp, ref_seg, _ = TD.run(TD.REFDEC, args, tmp_path, "cpu_" + what)
    for lanes, cluster in (("4", "1"), ("5", "3")):
        hyp, seg, log = (str(tmp_path / f"kf_{what}_{lanes}.{e}") for e in ("match", "matchseg", "log"))
        with open(log, "w") as lf:
            p = subprocess.run([TD.TST] + TD.common() + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT, timeout=900,
                               env=dict(os.environ, **FORCE, S3A_UTT=lanes, S3A_UTT_CLUSTER=cluster))
        assert p.returncode == 0, open(log, errors="ignore").read()[-2000:]
        assert open(hyp).read() == ref_hyp and open(seg).read() == ref_seg


@pytest.mark.parametrize("env", [{"S3A_UTT": "4"}, {"S3A_UTT": "3", "S3A_UTT_CLUSTER": "2"}, {"S3A_UTT": "6", "S3A_UTT_QUEUE": "12"},
                                 {"S3A_UTT": "2", "S3A_UTT_QUEUE": "6", "S3A_UTT_SCORE_ROWS": "420"},          # the queue in >= 3 parts
                                 {"S3A_UTT": "6", "S3A_UTT_SCORE_ROWS": "100"}])                               # S3A_EUNSUP -> window blocks
def test_hub4_shaped_decode_through_ku_frames(env, tmp_path):
    """several-parent sets (46 left-context variants per root), composite senones, ~3000 active HMMs per frame"""
    args = TD.synth_task("hub4", tmp_path, 6, 200)
    ref = TD.decode_task(TD.REFDEC, args, tmp_path, "ref")
    got = TD.decode_task(TD.TST, args, tmp_path, "kf", dict(FORCE, **env))
    assert got[0] == ref[0] and got[1] == ref[1]


@pytest.mark.parametrize("env", [{"S3A_UTT": "4"}, {"S3A_UTT": "3", "S3A_UTT_CLUSTER": "2", "S3A_UTT_QUEUE": "8"}])
def test_five_state_hmms_through_ku_frames(env, task):  # noqa: F811
    T5.test_five_state_decode_matches_reference(task, "kf" + env["S3A_UTT"], dict(FORCE, **env))



def test_clusters_at_scale_on_the_hub4_shaped_task(tmp_path):
    """16 lanes x 3 workgroups (48 workgroups, two lanes' clusters per XCD) with the XCD-local barrier on the task that has several-parent
    sets and composite senones: the barrier's riskiest regime outside the bench's 128-lane projection"""
    args = TD.synth_task("hub4", tmp_path, 16, 120)
    ref = TD.decode_task(TD.REFDEC, args, tmp_path, "ref")
    got = TD.decode_task(TD.TST, args, tmp_path, "kf", dict(FORCE, S3A_UTT="16", S3A_UTT_CLUSTER="3"))
    assert got[0] == ref[0] and got[1] == ref[1] and ref[0].count("\n") == 16
    got = TD.decode_task(TD.TST, args, tmp_path, "kfq", dict(FORCE, S3A_UTT="8", S3A_UTT_CLUSTER="4", S3A_UTT_QUEUE="16"))
    assert got[0] == ref[0] and got[1] == ref[1]


def test_the_relay_hands_lanes_over_to_clusters(gpu_lib, tidigits_bundle):
    """a call ends with its slowest lane: when no utterance is left to take and few lanes are still at work they leave at a frame
    boundary and the chain's next launch continues them as clusters of 2, then 4 workgroups (s3a_variants_t.kf_relay_at brings the
    hand-overs down to 8 and 4 lanes): queue and plain decode, hypotheses and whole tables as without the relay"""
    from cmusphinx_amd import lib
    utts, feats = TQ.tidigits_feats(gpu_lib)
    rm, rs = TQ.ref_lines()
    res = {}
    try:
        for relay_at, no_relay in ((8, 0), (0, 1)):
            lib.set_variants(kf_relay_at=relay_at, kf_no_relay=no_relay)
            dec = bundle.Decoder(tidigits_bundle, 16)
            dec.decode_queue(feats)
            m, s = TQ.queue_lines(dec, utts)
            assert m == rm and s == rs
            assert dec.ud.last_relay() == (2 if relay_at else 0) and dec.ud.last_parts()["n_frames"] == 1
            dec.decode(feats[:16])
            assert dec.ud.last_relay() == (2 if relay_at else 0)
            res[relay_at] = [dec.ud.result(z) for z in range(16)]
            del dec
    finally:
        lib.set_variants()
    for a, b in zip(res[8], res[0]):
        for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type", "frame_start", "bestscore", "bestvh", "frame_stat"):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("env", [{"S3A_UTT": "12"}, {"S3A_UTT": "10", "S3A_UTT_QUEUE": "16"}])
def test_the_relay_on_the_hub4_shaped_task(env, tmp_path):
    """several-parent sets, composite senones, ~3000 active HMMs per frame across the hand-overs (12 lanes -> 6 x 2 workgroups -> 3 x 4)"""
    args = TD.synth_task("hub4", tmp_path, 16, 150)
    ref = TD.decode_task(TD.REFDEC, args, tmp_path, "ref")
    got = TD.decode_task(TD.TST, args, tmp_path, "relay", dict(env, S3A_KF_RELAY_AT="6"))
    assert got[0] == ref[0] and got[1] == ref[1] and ref[0].count("\n") == 16


@pytest.mark.parametrize("env", [{"S3A_UTT": "4", "S3A_UTT_CLUSTER": "1"}, {"S3A_UTT": "5", "S3A_UTT_CLUSTER": "3"}, {"S3A_UTT": "6", "S3A_UTT_QUEUE": "20"}])
def test_maxcdsenpf_inside_ku_frames(env, tmp_path):
    """-maxcdsenpf (approx_compute_dyn_ci_pbeam, approx_cont_mgau.c:303-357): the frame's CI beam worked out inside the kernel from the
    mask's bits in LDS and the row's CI scores (kf_dyn_ci_beam) -- RM1, 145 instead of 572 CD senones per frame survive the gate;
    one workgroup per lane, clusters (every workgroup works the beam out for itself), the queue"""
    args = TD.rm_args(["-ci_pbeam", "1e-10", "-maxcdsenpf", "150"])
    ref = TD.decode_task(TD.REFDEC, args, tmp_path, "ref")
    got = TD.decode_task(TD.TST, args, tmp_path, "kf", dict(FORCE, **env))
    assert got[0] == ref[0] and got[1] == ref[1] and ref[0].count("\n") == 20


@pytest.mark.parametrize("env", [{"S3A_UTT": "2", "S3A_UTT_CLUSTER": "4"}, {"S3A_UTT": "3", "S3A_UTT_CLUSTER": "1"}, {"S3A_UTT": "2", "S3A_UTT_QUEUE": "3", "S3A_UTT_CLUSTER": "7"}])
def test_wide_beam_word_level_inside_ku_frames(env, tmp_path):
    """configs[4] (8000 senones x 32 Gaussians, -beam 1e-120 -pbeam 1e-100 -wbeam 1e-80 -maxhmmpf 100000: > 20 000 word-level candidates
    per frame): the word level's candidate phases chunked over the lane's cluster with the cluster barrier between them (d_wl_big_p2 ..
    finish inside kf_frame) -- clusters of 4 and 7, one workgroup alone (every chunk its own), the queue"""
    args = TD.synth_task("wsj", tmp_path, 3, 50, env=dict(TASK_BEAM="1e-120", TASK_WBEAM="1e-80")) + ["-pbeam", "1e-100", "-maxhmmpf", "100000"]
    ref = TD.decode_task(TD.REFDEC, args, tmp_path, "ref")
    got = TD.decode_task(TD.TST, args, tmp_path, "kf", dict(FORCE, **env))
    assert got[0] == ref[0] and got[1] == ref[1] and ref[0].count("\n") == 3
    wl = [l for l in got[2] if "word level: at most" in l]
    assert wl and int(wl[0].split("at most")[1].split()[0]) > 20000, wl
