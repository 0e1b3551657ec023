"""Parity (MI355X): the device's first pass through the C ABI alone (s3a_psfwd_init / _start / _sen_active / _step /
_finish / _table / _hyp) against the pinned restatement oracle/s3o_psfwd.c on seeded synthetic search spaces
(tests/psfwd_synth.py): random lexicon trees, right-context tables and trigram LMs, 3- AND 5-state topologies (no
shipped pocketsphinx model is 5-state), with and without skip arcs, unigram / bigram / trigram LMs, -maxwpf and
-maxhmmpf pruning, a beam wide enough to force renormalize_scores.  Per frame: the active-senone flags; per
utterance: the whole backpointer table, the statistics, the device-made hypothesis.  Two utterances through the
same lane (a decoder's channels keep state across utterances -- the second result depends on it), then a reset lane."""
import numpy as np
import pytest

import psfwd_synth as S

pytestmark = pytest.mark.gpu

CASES = {
    "3state": dict(n_emit=3),
    "3state_noskip": dict(n_emit=3, skips=False),
    "5state": dict(n_emit=5),
    "5state_maxwpf": dict(n_emit=5, maxwpf=3),
    "maxwpf_maxhmmpf": dict(n_emit=3, maxwpf=2, maxhmmpf=60),
    "bigram": dict(n_emit=3, lm_order=2),
    "unigram": dict(n_emit=5, lm_order=1),
    "renormalize": dict(n_emit=3, n_real=16, beam=-268434000),
    "big": dict(n_emit=3, n_ci=30, n_real=400, n_sen=600, maxwpf=20),
    # -pl_window (phone loop look-ahead): what the decoder's phone loop adds at every transition, handed over frame by frame
    "lookahead_3state": dict(n_emit=3, pl_window=3),
    "lookahead_5state_maxwpf": dict(n_emit=5, maxwpf=3, pl_window=2),
    "lookahead_big": dict(n_emit=3, n_ci=30, n_real=400, n_sen=600, maxwpf=20, pl_window=4),
    # class-based LM: in-class weights on top of the tag words' scores (trigram and bigram)
    "class_lm": dict(n_emit=3, classes=True),
    "class_lm_bigram_lookahead": dict(n_emit=5, lm_order=2, classes=True, pl_window=2),
}


def decode(o, dev, scr, check_flags=True, pl=None):
    n = len(scr)
    o.start(); dev.start()
    for f in range(n):
        if pl is not None:
            o.set_lookahead(pl[f]); dev.set_lookahead(pl[f])
        fo = o.sen_active(f)
        if check_flags:
            fd = dev.sen_active(f)
            assert np.array_equal(fo, fd), f"frame {f}: active senones differ at {np.argwhere(fo != fd)[:5].ravel()}"
        k = int(fo.sum())
        ro, rd = o.step(scr[f], f, k), dev.step(scr[f], f, k)
        assert ro == rd, f"frame {f}: step returned {rd}, the restatement {ro}"
    o.finish(n); dev.finish(n)
    d = S.diff_tables(o.table(n), dev.table(n))
    assert d == [], "\n".join(d)
    assert o.hyp() == dev.hyp()
    return o.table(n)


@pytest.mark.parametrize("case", sorted(CASES))
def test_device_first_pass_matches_restatement(gpu_lib, case):
    kw = CASES[case]
    desc, keep = S.make_desc(11 + len(case), **kw)
    o, dev = S.Oracle(desc), S.Device(gpu_lib, desc, n_lanes=2, max_frames=128, bp_cap=1 << 18, bss_cap=1 << 22)
    n = 100 if "big" not in case else 60
    pl = S.make_lookahead(9, n, desc.n_ci) if kw.get("pl_window") else None
    t1 = decode(o, dev, S.make_senscr(5, n, desc.n_sen), pl=pl)
    assert t1["bpidx"] > 50, "the synthetic task must produce word exits"
    if case == "renormalize":
        assert t1["renorm"] == 1
    t2 = decode(o, dev, S.make_senscr(6, n, desc.n_sen), check_flags=("big" not in case), pl=pl)
    # a NEW decoder gives the first result again; a used one need not
    o.reset(); dev.reset()
    t3 = decode(o, dev, S.make_senscr(5, n, desc.n_sen), check_flags=False, pl=pl)
    assert S.diff_tables(t1, t3) == []
    del t2
    if pl is not None:      # the look-ahead decides: without it the same frames give another table
        o.reset(); dev.reset()
        t4 = decode(o, dev, S.make_senscr(5, n, desc.n_sen), check_flags=False, pl=np.zeros_like(pl))
        assert S.diff_tables(t1, t4) != []


def test_lookahead_needs_an_engine_built_for_it(gpu_lib):
    desc, keep = S.make_desc(3, n_emit=3)
    dev = S.Device(gpu_lib, desc, n_lanes=1, max_frames=32)
    with pytest.raises(gpu_lib.S3AError, match="pl_window 0"):
        dev.set_lookahead(np.zeros(desc.n_ci, np.int32))


def test_table_overflow_is_loud(gpu_lib):
    desc, keep = S.make_desc(3, n_emit=3)
    L = gpu_lib.load()
    import ctypes as C
    h = L.s3a_psfwd_init(C.byref(desc), 1, 64, 40, 400)
    assert h
    try:
        scr = S.make_senscr(5, 60, desc.n_sen)
        assert L.s3a_psfwd_start(h, 0) == 0
        rc = 1
        for f in range(60):
            rc = L.s3a_psfwd_step(h, 0, scr[f].ctypes.data_as(C.c_void_p), f, 0)
            if rc < 0:
                break
        assert rc == -3 and b"full" in L.s3a_last_error()      # S3A_ENOMEM, never a silent truncation
    finally:
        L.s3a_psfwd_free(h)
