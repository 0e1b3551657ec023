"""Several engines side by side on one GPU (MI355X): a host thread + stream each, their kernels truly concurrent on the
chip's hardware queues.  Every utterance must come out as it does from ONE engine alone, round after round on reused
lanes, and every lane must be CLEAN between utterances (s3a_uttdec_selfcheck: every node record an inactive HMM, the
propagation scratch consumed).

Round 3 found this the hard way: the word level cleared ctx->active at an utterance's last frame while the emission
workgroups of the SAME launch test it -- one that started late (another queue's kernel on the chip) skipped its sweep and
left the scratch set, and the lane's next utterance went wrong in 1-2 % of the cases.  Alone on the chip the launch's
workgroups all start before the word level ends, so no single-engine test could see it."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from cmusphinx_amd import bundle, lib, s3io, synth_task
from conftest import ROOT

pytestmark = pytest.mark.gpu
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")


def test_concurrent_engines_agree_with_one_engine_and_leave_their_lanes_clean(gpu_lib, tmp_path):
    if not os.path.exists(TST):
        pytest.fail(f"{TST} is missing on the GPU box (make -C oracle ref)")
    d = str(tmp_path / "task")
    U, NE, NLE = 48, 4, 6
    synth_task.make_task(d, n_utt=U, n_frames=300, **synth_task.HUB4_TASK)
    bp = str(tmp_path / "b.bundle")
    r = subprocess.run([TST] + synth_task.decoder_args(d), env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    assert r.returncode == 0
    utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
    feats = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")).reshape(-1, 39) for u in utts]
    decs = [bundle.Decoder(bp, NLE, cand_cap=1 << 16) for _ in range(NE)]
    order = sorted(range(U), key=lambda k: (-len(feats[k]), k))
    groups = [order[i:i + NLE] for i in range(0, U, NLE)]

    def decode(dec, g):
        dec.decode([feats[k] for k in g])
        out = {k: dec.format_var(*dec.hyp_var(z, utts[k], k)) for z, k in enumerate(g)}
        dirty = [(z, dec.ud.selfcheck(z).tolist()) for z in range(len(g))]
        return out, [(z, c) for z, c in dirty if c[6] != 2147483647 or c[7]]

    truth = {}
    for g in groups:
        o, dirty = decode(decs[0], g)
        assert dirty == []
        truth.update(o)
    L = gpu_lib.load()
    pool = ThreadPoolExecutor(NE)
    for rnd in range(3):
        def one(e):
            gpu_lib.check(L.s3a_set_device(0))
            bad = []
            for g in groups[e::NE]:
                o, dirty = decode(decs[e], g)
                bad += [("dirty lane", e, z, c) for z, c in dirty] + [("differs", e, k) for k in g if o[k] != truth[k]]
            return bad
        bad = [b for o in pool.map(one, range(NE)) for b in o]
        assert bad == [], (rnd, bad[:8])
