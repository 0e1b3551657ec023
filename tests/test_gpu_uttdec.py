"""Whole utterances through the C ABI alone (MI355X): cepstra -> s3a_feat_1s_c_d_dd -> s3a_uttdec_decode ->
s3a_uttdec_hyp -> s3a_hyp_format, against the unmodified reference's -hyp / -hypseg files.

The decoder is rebuilt from a BUNDLE (cmusphinx_amd/bundle.py) that the sphinx3 side of the drop-in wrote once
(S3A_EXPORT: the reference's own kb_init loads the models, the dictionary, the LM and builds the lextrees); from
there on no reference code runs: features, scoring, search, word level, history table, final </s> transition,
backtrace and output formatting are all libcmusphinx_amd's.  tidigits: the committed reference goldens.
"""
import os
import subprocess

import numpy as np
import pytest

from cmusphinx_amd import bundle, s3io
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")


@pytest.fixture(scope="module")
def tidigits_bundle(tmp_path_factory):
    if not os.path.exists(TST):
        pytest.fail(f"{TST} is missing on the GPU box (make -C oracle ref)")
    out = str(tmp_path_factory.mktemp("bundle") / "tidigits.bundle")
    args = [TST, "-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5",
            "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4", "-lm", f"{D}/tidigits.DMP"]
    p = subprocess.run(args, env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=out), stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0 and os.path.getsize(out) > 1000
    return out


def ctl_entries():
    for line in open(f"{D}/tidigits.length.arb.regression"):
        f = line.split()
        if f:
            yield f[0], (f[3] if len(f) > 3 else os.path.basename(f[0]))


@pytest.mark.parametrize("n_lanes", [1, 5])
def test_tidigits_through_the_c_abi_alone(gpu_lib, tidigits_bundle, n_lanes):
    dec = bundle.Decoder(tidigits_bundle, n_lanes)
    utts = list(ctl_entries())
    assert len(utts) == 31
    match, seg = [], []
    for k in range(0, len(utts), n_lanes):
        chunk = utts[k:k + n_lanes]
        feats = [gpu_lib.feat_1s_c_d_dd(s3io.read_mfc(f"{D}/cepstra/{u}.mfc").reshape(-1, 13), cmn="current") for u, _ in chunk]
        dec.decode(feats)
        for z, (u, uid) in enumerate(chunk):
            rec = dec.hyp(z, uid, k + z)
            assert rec.status == 0 and rec.n_frames == len(feats[z])
            m, s = dec.format(rec)
            match.append(m); seg.append(s)
    assert "".join(match) == open(f"{D}/ref_mode4_trigram.match").read()
    assert "".join(seg) == open(f"{D}/ref_mode4_trigram.matchseg").read()


def test_hypothesis_records_are_fixed_size_and_self_contained(gpu_lib, tidigits_bundle):
    """what the end-of-batch gather ships: sizeof(s3a_hyp_record_t) bytes per utterance, formatted on another 'rank'"""
    import ctypes as C
    dec = bundle.Decoder(tidigits_bundle, 2)
    utts = list(ctl_entries())[:2]
    feats = [gpu_lib.feat_1s_c_d_dd(s3io.read_mfc(f"{D}/cepstra/{u}.mfc").reshape(-1, 13), cmn="current") for u, _ in utts]
    dec.decode(feats)
    recs = [dec.hyp(z, uid, z) for z, (_, uid) in enumerate(utts)]
    assert C.sizeof(gpu_lib.HypRecord) == 96 + 8 * 4 + 250 * 24
    raw = b"".join(bytes(r) for r in recs)                    # "the wire"
    back = [gpu_lib.HypRecord.from_buffer_copy(raw[i * C.sizeof(gpu_lib.HypRecord):]) for i in range(2)]
    lines = [dec.format(r) for r in back]
    ref = open(f"{D}/ref_mode4_trigram.match").read().splitlines(keepends=True)
    assert [l[0] for l in lines] == ref[:2]
    assert back[1].utt_index == 1 and back[0].word[0].sf == 0


def test_very_short_and_ragged_utterances(gpu_lib, tidigits_bundle, tmp_path):
    """Utterances of 1, 2, 5, 12 and 40 frames in ONE batch (more lanes than utterances).  The reference cannot end an
    utterance of 1 or 2 frames (`s->funcs->utt_end failed`: no word exit reached a history entry) and writes no line
    for it; from 5 frames on it writes <sil> / a word.  Same files through S3A_UTT, and through the C ABI alone the
    hypothesis records of the two unendable utterances carry a failure status while the others format identically."""
    import struct
    REF = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
    cep = s3io.read_mfc(f"{D}/cepstra/man/man.ah.2934za.mfc").reshape(-1, 13)
    lens = [1, 2, 5, 12, 40]
    os.makedirs(tmp_path / "cep")
    for n in lens:
        with open(tmp_path / "cep" / f"t{n}.mfc", "wb") as f:
            f.write(struct.pack("<i", 13 * n))
            f.write(cep[:n].astype("<f4").tobytes())
    (tmp_path / "ctl").write_text("".join(f"t{n}\n" for n in lens))
    args = ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", str(tmp_path / "cep"),
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-ctl", str(tmp_path / "ctl"),
            "-op_mode", "4", "-lm", f"{D}/tidigits.DMP"]
    out = {}
    for tag, exe, env in (("ref", REF, None), ("utt", TST, dict(os.environ, S3A_UTT="8"))):
        hyp, seg = str(tmp_path / f"{tag}.match"), str(tmp_path / f"{tag}.matchseg")
        p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], capture_output=True, text=True, errors="ignore",
                           timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-1500:]
        out[tag] = (open(hyp).read(), open(seg).read(), p.stderr.count("utt_end failed"))
    assert out["ref"][0].count("\n") == 3 and out["ref"][2] == 2          # t1, t2: no line, two failures reported
    assert out["utt"] == out["ref"]
    dec = bundle.Decoder(tidigits_bundle, 8)
    feats = [gpu_lib.feat_1s_c_d_dd(cep[:n].copy(), cmn="current") for n in lens]
    dec.decode(feats)
    recs = [dec.hyp(z, f"t{n}", z) for z, n in enumerate(lens)]
    assert [r.status != 0 for r in recs] == [True, True, False, False, False]
    assert "".join(dec.format(r)[0] for r in recs[2:]) == out["ref"][0]
    assert "".join(dec.format(r)[1] for r in recs[2:]) == out["ref"][1]


def test_every_lane_that_overflowed_restarts_clean(gpu_lib, tidigits_bundle):
    """A history table far too small stops SEVERAL lanes of a batch in mid-frame (WL_E_TABLE is raised after the hash
    inserts).  The decode reports the first; every one of them must start its next utterance from scratch -- stale
    hash slots / lextree state in a lane that was not the first would give silently wrong history entries.  Then the
    var-length hypothesis API on the same lanes."""
    utts = list(ctl_entries())[:4]
    feats = [gpu_lib.feat_1s_c_d_dd(s3io.read_mfc(f"{D}/cepstra/{u}.mfc").reshape(-1, 13), cmn="current") for u, _ in utts]
    ref_m = open(f"{D}/ref_mode4_trigram.match").read().splitlines(keepends=True)
    ref_s = open(f"{D}/ref_mode4_trigram.matchseg").read().splitlines(keepends=True)
    small = bundle.Decoder(tidigits_bundle, 4, vh_cap=64)
    with pytest.raises(gpu_lib.S3AError, match="history table full"):
        small.decode(feats)
    errs = [small.ud.result(z)["err"] for z in range(4)]
    assert sum(1 for e in errs if e) >= 2, errs              # the point of the test: more than one lane stopped
    # the same engine, the same lanes, short utterances that fit the tiny table: frames cut so that few entries are made
    short = [f[:12].copy() for f in feats]
    small.decode(short)
    good = bundle.Decoder(tidigits_bundle, 4)
    good.decode(short)
    for z in range(4):
        a, b = small.ud.result(z), good.ud.result(z)
        assert a["err"] == 0 and all(np.array_equal(a[k], b[k]) for k in ("score", "pred", "wid", "lw0", "lw1", "sf", "ef", "ascr", "lscr"))
    # ... and full utterances after an overflow in a roomy engine whose lanes are then reused
    good.decode(feats)
    for z, (_, uid) in enumerate(utts):
        h, w = good.hyp_var(z, uid, z)
        assert h.status == 0 and len(w) == h.n_words
        assert good.format_var(h, w) == (ref_m[z], ref_s[z]) == good.format(good.hyp(z, uid, z))
