"""Oracle pinning, part 5: the lexical-tree search operations (oracle/s3o_lextree.c).

(a) End to end (needs oracle/_ref, i.e. the build container or a box the binaries travelled
    to): the reference decoder with EVERY per-frame lextree slot of srch_funcs_t replaced by
    the oracle (oracle/_ref/ref_s3olt_decode = oracle/ref_tst_shim.c -DLT_ORACLE) must
    reproduce the unmodified reference's -hyp and -hypseg byte for byte on the tidigits
    regression set (mode 4, trigram LM; default beams and the CI-beam / -ds 2 variant).
(b) Replay of the committed operation trace (tests/golden/lextree_trace_tidigits.npz,
    recorded during such a byte-identical run): every active list, HMM state, best score
    and word exit after every operation.
"""
import os
import subprocess

import numpy as np
import pytest

import lextree_trace
import oracle_lib as O
from conftest import GOLDEN, ROOT

D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
OLT = os.path.join(ROOT, "oracle", "_ref", "ref_s3olt_decode")


@pytest.mark.skipif(not os.path.exists(OLT), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,extra", [("mode4_trigram", []), ("mode4_cibeam_ds2", ["-ci_pbeam", "1e-5", "-ds", "2"])])
def test_reference_decoder_with_oracle_lextree_is_byte_identical(name, extra, tmp_path):
    hyp, seg = str(tmp_path / "h.match"), str(tmp_path / "h.matchseg")
    args = [OLT, "-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5",
            "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4", "-lm", f"{D}/tidigits.DMP",
            "-hyp", hyp, "-hypseg", seg] + extra
    p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0
    assert open(hyp).read() == open(f"{D}/ref_{name}.match").read()
    assert open(seg).read() == open(f"{D}/ref_{name}.matchseg").read()


def test_oracle_replays_the_recorded_trace():
    tr = lextree_trace.from_npz(np.load(os.path.join(GOLDEN, "lextree_trace_tidigits.npz")))
    assert tr["n_tree"] == 6 and [t["n_node"] for t in tr["trees"]] == [112, 112, 112, 1, 1, 1]
    n_frames, n_exits = lextree_trace.Replayer(tr, O.OracleLexSearch(tr)).run()
    assert n_frames == 60 and n_exits > 20


def test_histbin_reorders_like_glist_prepend():
    """lextree_hmm_histbin (lextree.c:1314-1358): bins ascending, REVERSE insertion order inside a bin."""
    tr = lextree_trace.from_npz(np.load(os.path.join(GOLDEN, "lextree_trace_tidigits.npz")))
    ls = O.OracleLexSearch(tr)
    # run the trace until some tree has a decent active list
    ev = tr["events"]
    rep = lextree_trace.Replayer(dict(tr, events=ev[: next(i for i, (t, _) in enumerate(ev) if t == 60 and i > 400)]), ls)
    rep.run()
    t = 0
    act = ls.active(t, 0)
    assert len(act) > 5
    st = ls.state(t)
    best = int(st[act, 8].max())
    bw = 5000
    bins = np.zeros(1000, np.int32)
    O.lib().s3o_lextree_hmm_histbin(ls.lt[t], best, bins.ctypes.data_as(O.C.c_void_p), 1000, bw)
    new = ls.active(t, 0)
    k = np.minimum((best - st[act, 8].astype(np.int64)) // bw, 999)
    exp = [int(n) for b in range(1000) for n in act[k == b][::-1]]
    assert list(new) == exp and bins.sum() == len(act)
