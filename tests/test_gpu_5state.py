"""5-state HMM topologies (hmm_vit_eval_5st_lr, libam/hmm.c:285-412: Bakis models with skip transitions) through the
whole-utterance engine on the MI355X: a synthetic 5-state task (mdef with five state ids per phone, 5 x 6 transition
matrices with i -> i + 2 skips; the reference checkout holds no 5-state continuous model) decoded by the unmodified
reference and by the device engine -- -hyp / -hypseg byte for byte -- through the drop-in program (1 ... 5 lanes, the
look-ahead scoring window and the per-frame scoring kernels) and from an exported bundle through the C ABI alone; the
frame-synchronous entry points refuse a 5-state model loudly."""
import os
import subprocess

import pytest

from cmusphinx_amd import bundle, s3io, synth_task
from conftest import ROOT

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")
TASK = dict(n_sen=1500, n_ciphone=28, n_comp=4, n_words=600, seed=0x5557A7E, sep=0.3, noise=1.4, n_emit=5)
BEAMS = dict(beam="1e-70", wbeam="1e-40")


@pytest.fixture(scope="module")
def task(tmp_path_factory):
    for b in (REF, TST):
        if not os.path.exists(b):
            pytest.fail(f"{b} is missing on the GPU box (make -C oracle ref)")
    d = str(tmp_path_factory.mktemp("five") / "task")
    synth_task.make_task(d, n_utt=10, n_frames=350, **TASK)
    head = open(os.path.join(d, "mdef")).read().split("\n", 12)
    assert len(head[10].split()) == 6 + 5 + 1                       # a CI phone line: base lft rt p attrib tmat + five state ids + N
    args = synth_task.decoder_args(d, **BEAMS) + ["-pbeam", "1e-60"]
    r = subprocess.run([REF] + args + ["-hyp", d + "/ref.hyp", "-hypseg", d + "/ref.hypseg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    hyp = open(d + "/ref.hyp").read()
    assert len(hyp.splitlines()) == 10 and sum(len(l.split()) - 1 for l in hyp.splitlines()) > 30      # words were recognised
    return d, args


def run_tst(d, args, tag, env):
    r = subprocess.run([TST] + args + ["-hyp", f"{d}/{tag}.hyp", "-hypseg", f"{d}/{tag}.hypseg"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, **env))
    return r


@pytest.mark.parametrize("tag,env", [("utt1", {"S3A_UTT": "1"}), ("utt4", {"S3A_UTT": "4"}), ("utt5_perframe", {"S3A_UTT": "5", "S3A_UTT_WIN": "0"}),
                                     ("utt4_win16", {"S3A_UTT": "4", "S3A_UTT_WIN": "16"}), ("utt4_x2", {"S3A_UTT": "4", "S3A_UTT_ENGINES": "2"})])
def test_five_state_decode_matches_reference(task, tag, env):
    d, args = task
    r = run_tst(d, args, tag, env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(f"{d}/{tag}.hyp").read() == open(d + "/ref.hyp").read()
    assert open(f"{d}/{tag}.hypseg").read() == open(d + "/ref.hypseg").read()


def test_five_state_with_histogram_pruning_and_ci_gate(task):
    d, args = task
    extra = ["-maxhmmpf", "300", "-ci_pbeam", "1e-8", "-ds", "2"]
    r = subprocess.run([REF] + args + extra + ["-hyp", d + "/refp.hyp", "-hypseg", d + "/refp.hypseg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0
    r = run_tst(d, args + extra, "p4", {"S3A_UTT": "4"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(d + "/p4.hyp").read() == open(d + "/refp.hyp").read()
    assert open(d + "/p4.hypseg").read() == open(d + "/refp.hypseg").read()


def test_five_state_from_a_bundle_through_the_c_abi(task, gpu_lib):
    d, args = task
    bp = d + "/five.bundle"
    r = subprocess.run([TST] + args, env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(bp), r.stderr[-2000:]
    utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
    feats = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")).reshape(-1, 39) for u in utts]
    dec = bundle.Decoder(bp, 5)
    hyp, seg = "", ""
    for g in (range(0, 5), range(5, 10)):
        dec.decode([feats[k] for k in g])
        for z, k in enumerate(g):
            h, s = dec.format_var(*dec.hyp_var(z, utts[k], k))
            hyp += h
            seg += s
            c = dec.ud.selfcheck(z).tolist()
            assert c[6] == 2147483647 and c[7] == 0, (k, c)             # the lane is clean for its next utterance
    assert hyp == open(d + "/ref.hyp").read() and seg == open(d + "/ref.hypseg").read()


def test_frame_synchronous_slots_refuse_five_states(task):
    d, args = task
    r = run_tst(d, args, "fs", {})
    assert r.returncode != 0 and "whole-utterance engine" in r.stderr
