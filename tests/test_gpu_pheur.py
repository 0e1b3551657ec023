"""Phoneme look-ahead (-pheurtype 1..3, -pl_window, -pl_beam: lextree.c:1443-1486, fast_algo_struct.c:219-300) inside the
whole-utterance engine on the MI355X: the CI senones of every frame scored ahead, phn_heur_list of every frame, the running
heuristic threshold over the active list, the extra test on every transition.  The unmodified reference with the same
options is the judge: -hyp / -hypseg byte for byte (the options change what is recognised: the runs differ from
-pheurtype 0), on a 3-state and a 5-state synthetic task, through the drop-in program and from an exported bundle."""
import os
import subprocess

import pytest

from cmusphinx_amd import bundle, s3io, synth_task
from conftest import ROOT

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")
TASK = dict(n_sen=1500, n_ciphone=28, n_comp=4, n_words=600, seed=0x5557A7E, sep=0.3, noise=1.4)
OPTS = {
    "t1_w1": ["-pheurtype", "1", "-pl_window", "1", "-pl_beam", "1e-10"],
    "t1_w5": ["-pheurtype", "1", "-pl_window", "5", "-pl_beam", "1e-30"],
    "t2_w3": ["-pheurtype", "2", "-pl_window", "3", "-pl_beam", "1e-5"],
    "t3_w4": ["-pheurtype", "3", "-pl_window", "4", "-pl_beam", "1e-20"],
    # ... with -ptranskip: in those frames the phone threshold is bestwordscore + -wbeam, BELOW the HMM threshold -- HMMs under the
    # HMM beam propagate when an earlier parent re-entered them, and that entry must pass the look-ahead too (ku_weak_heur)
    "t1_w5_skip3": ["-pheurtype", "1", "-pl_window", "5", "-pl_beam", "1e-30", "-ptranskip", "3"],
    "t3_w4_skip2": ["-pheurtype", "3", "-pl_window", "4", "-pl_beam", "1e-20", "-ptranskip", "2"],
    "t2_w3_skip1": ["-pheurtype", "2", "-pl_window", "3", "-pl_beam", "1e-5", "-ptranskip", "1"],
    "t1_w10_hist": ["-pheurtype", "1", "-pl_window", "10", "-pl_beam", "1e-25", "-maxhmmpf", "400", "-ci_pbeam", "1e-8"],
}


def make(tmp_path_factory, name, n_emit):
    for b in (REF, TST):
        if not os.path.exists(b):
            pytest.fail(f"{b} is missing on the GPU box (make -C oracle ref)")
    d = str(tmp_path_factory.mktemp(name) / "task")
    synth_task.make_task(d, n_utt=8, n_frames=350, n_emit=n_emit, **TASK)
    args = synth_task.decoder_args(d, beam="1e-70", wbeam="1e-40") + ["-pbeam", "1e-60"]
    r = subprocess.run([REF] + args + ["-hyp", d + "/ref0.hyp", "-hypseg", d + "/ref0.hypseg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return d, args


@pytest.fixture(scope="module")
def task3(tmp_path_factory):
    return make(tmp_path_factory, "ph3", 3)


@pytest.fixture(scope="module")
def task5(tmp_path_factory):
    return make(tmp_path_factory, "ph5", 5)


def both(d, args, opt, tag, env):
    if not os.path.exists(f"{d}/ref_{opt}.hyp"):
        r = subprocess.run([REF] + args + OPTS[opt] + ["-hyp", f"{d}/ref_{opt}.hyp", "-hypseg", f"{d}/ref_{opt}.hypseg"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([TST] + args + OPTS[opt] + ["-hyp", f"{d}/{tag}.hyp", "-hypseg", f"{d}/{tag}.hypseg"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(f"{d}/{tag}.hyp").read() == open(f"{d}/ref_{opt}.hyp").read()
    assert open(f"{d}/{tag}.hypseg").read() == open(f"{d}/ref_{opt}.hypseg").read()
    return open(f"{d}/ref_{opt}.hypseg").read() != open(d + "/ref0.hypseg").read()


@pytest.mark.parametrize("opt", list(OPTS))
def test_phoneme_lookahead_matches_reference_3state(task3, opt):
    d, args = task3
    changed = both(d, args, opt, "u4_" + opt, {"S3A_UTT": "4"})
    assert changed                  # the look-ahead is no bystander: the reference's own result moves with it
    both(d, args, opt, "u3pf_" + opt, {"S3A_UTT": "3", "S3A_UTT_WIN": "0"})
    both(d, args, opt, "u1_" + opt, {"S3A_UTT": "1"})


@pytest.mark.parametrize("opt", ["t1_w5", "t2_w3", "t3_w4"])
def test_phoneme_lookahead_matches_reference_5state(task5, opt):
    d, args = task5
    both(d, args, opt, "u4_" + opt, {"S3A_UTT": "4"})


@pytest.mark.parametrize("opt", ["t1_w5", "t3_w4"])
def test_phoneme_lookahead_inside_ku_frames_5state(task5, opt):
    """... and with 5-state HMMs (ku_frames<5, *, HEUR>): one workgroup per lane, clusters of 2 as a queue"""
    d, args = task5
    both(d, args, opt, "kf4_" + opt, {"S3A_UTT_PERSIST": "1", "S3A_UTT": "4", "S3A_UTT_CLUSTER": "1"})
    both(d, args, opt, "kfq3_" + opt, {"S3A_UTT_PERSIST": "1", "S3A_UTT": "3", "S3A_UTT_QUEUE": "8", "S3A_UTT_CLUSTER": "2"})


def test_phoneme_lookahead_from_a_bundle(task3, gpu_lib):
    d, args = task3
    opt = "t1_w5"
    bp = d + "/ph.bundle"
    r = subprocess.run([TST] + args + OPTS[opt], env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(bp), r.stderr[-2000:]
    r = subprocess.run([REF] + args + OPTS[opt] + ["-hyp", f"{d}/refb.hyp", "-hypseg", f"{d}/refb.hypseg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0
    utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
    feats = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")).reshape(-1, 39) for u in utts]
    dec = bundle.Decoder(bp, 4)
    assert dec.b["pheurtype"] == 1 and dec.b["pl_window"] == 5
    hyp, seg = "", ""
    for g in (range(0, 4), range(4, 8)):
        dec.decode([feats[k] for k in g])
        for z, k in enumerate(g):
            h, s = dec.format_var(*dec.hyp_var(z, utts[k], k))
            hyp += h
            seg += s
    assert hyp == open(d + "/refb.hyp").read() and seg == open(d + "/refb.hypseg").read()
    # the same through a queue with lane refill: 8 utterances in 3 lanes (the look-ahead tables are made when a lane
    # takes its utterance)
    dec3 = bundle.Decoder(bp, 3)
    dec3.decode_queue(feats)
    out = [dec3.format_var(*dec3.queue_hyp(k, utts[k], k)) for k in range(8)]
    assert "".join(o[0] for o in out) == hyp and "".join(o[1] for o in out) == seg


KF = {"S3A_UTT_PERSIST": "1"}


@pytest.mark.parametrize("opt", ["t1_w1", "t1_w5", "t2_w3", "t3_w4", "t1_w10_hist", "t1_w5_skip3", "t3_w4_skip2", "t2_w3_skip1"])
def test_phoneme_lookahead_inside_ku_frames(task3, opt):
    """round 6: the look-ahead inside the persistent kernel (ku_frames<3, *, HEUR>): the heuristic thresholds by list position as a
    step of the frame, the extra test at every transition of the propagation's three ways (list pass, one-parent sets, the several-
    parent sets' tables in LDS) -- one workgroup per lane, clusters of 3, and a queue (groups of static launches: the look-ahead's
    tables are made when a lane begins its utterance)"""
    d, args = task3
    both(d, args, opt, "kf4_" + opt, dict(KF, S3A_UTT="4", S3A_UTT_CLUSTER="1"))
    both(d, args, opt, "kf5c3_" + opt, dict(KF, S3A_UTT="5", S3A_UTT_CLUSTER="3"))
    both(d, args, opt, "kfq3_" + opt, dict(KF, S3A_UTT="3", S3A_UTT_QUEUE="8", S3A_UTT_CLUSTER="2"))


def test_phoneme_lookahead_ku_frames_is_what_ran(task3, gpu_lib):
    """the engine reports that the calls went through ku_frames (s3a_uttdec_last_parts), static and as a queue"""
    d, args = task3
    opt = "t3_w4"
    bp = d + "/phkf.bundle"
    r = subprocess.run([TST] + args + OPTS[opt], env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(bp), r.stderr[-2000:]
    if not os.path.exists(f"{d}/ref_{opt}.hyp"):
        r = subprocess.run([REF] + args + OPTS[opt] + ["-hyp", f"{d}/ref_{opt}.hyp", "-hypseg", f"{d}/ref_{opt}.hypseg"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0
    utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
    feats = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")).reshape(-1, 39) for u in utts]
    dec = bundle.Decoder(bp, 8, opts={"persist": 1, "cluster": 2})
    dec.decode(feats)
    parts = dec.ud.last_parts()
    assert parts["n_frames"] >= 1 and parts["cluster"] == 2, parts
    out = [dec.format_var(*dec.hyp_var(z, utts[z], z)) for z in range(8)]
    assert "".join(o[0] for o in out) == open(f"{d}/ref_{opt}.hyp").read()
    assert "".join(o[1] for o in out) == open(f"{d}/ref_{opt}.hypseg").read()
    dec3 = bundle.Decoder(bp, 3, opts={"persist": 1})
    dec3.decode_queue(feats)
    parts = dec3.ud.last_parts()
    assert parts["n_frames"] == 3, parts            # 8 utterances over 3 lanes: three groups, a launch (chain) each
    out = [dec3.format_var(*dec3.queue_hyp(k, utts[k], k)) for k in range(8)]
    assert "".join(o[0] for o in out) == open(f"{d}/ref_{opt}.hyp").read()
    assert "".join(o[1] for o in out) == open(f"{d}/ref_{opt}.hypseg").read()


def test_lookahead_with_a_wide_phone_beam(task3):
    """-pbeam wider than -beam: every frame has HMMs under the HMM beam that may still propagate (refused until round 4); the launches,
    and (round 6) inside ku_frames -- d_weak_heur_t as a step of the frame, a tree per workgroup of the cluster"""
    d, args = task3
    wide = [a for a in args]
    wide[wide.index("-pbeam") + 1] = "1e-90"               # wider than -beam 1e-70
    for key, opt in (("wide_t1", "t1_w5"), ("wide_t3_skip", "t3_w4_skip2")):
        OPTS[key] = OPTS[opt]
        both(d, wide, key, "u4_" + key, {"S3A_UTT": "4"})
        both(d, wide, key, "u1_" + key, {"S3A_UTT": "1"})
        both(d, wide, key, "kf4_" + key, dict(KF, S3A_UTT="4", S3A_UTT_CLUSTER="1"))
        both(d, wide, key, "kf3c4_" + key, dict(KF, S3A_UTT="3", S3A_UTT_CLUSTER="4", S3A_UTT_QUEUE="8"))
