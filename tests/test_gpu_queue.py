"""Lane refill (s3a_uttdec_decode_queue) on the MI355X: a queue of ANY number of utterances, a lane takes the next one when
its own has ended -- ctl_process (libcommon/corpus.c:538) knows no coupling between utterances either.  The ragged tidigits
set (31 utterances, 82 .. 339 frames) through engines of 1 .. 40 lanes: every utterance's -hyp / -hypseg line is the
unmodified reference's, whichever lane decoded it, whenever it started, whatever ran in that lane before (including an
utterance that stopped on a capacity error: the lane is scrubbed on the device)."""
import os

import numpy as np
import pytest

from cmusphinx_amd import s3io
from conftest import GOLDEN
from test_gpu_uttdec import D, ctl_entries, tidigits_bundle  # noqa: F401  (fixture)
from cmusphinx_amd import bundle

pytestmark = pytest.mark.gpu


def tidigits_feats(gpu_lib):
    utts = list(ctl_entries())
    feats = [gpu_lib.feat_1s_c_d_dd(s3io.read_mfc(f"{D}/cepstra/{u}.mfc").reshape(-1, 13), cmn="current") for u, _ in utts]
    return utts, feats


def ref_lines():
    return (open(f"{D}/ref_mode4_trigram.match").read().splitlines(keepends=True),
            open(f"{D}/ref_mode4_trigram.matchseg").read().splitlines(keepends=True))


def queue_lines(dec, utts, order=None):
    order = list(range(len(utts))) if order is None else order
    m, s = {}, {}
    for q, k in enumerate(order):
        hdr, words = dec.queue_hyp(q, utts[k][1], k)
        assert hdr.status == 0, (k, hdr.status)
        m[k], s[k] = dec.format_var(hdr, words)
    return [m[k] for k in range(len(utts))], [s[k] for k in range(len(utts))]


@pytest.mark.parametrize("n_lanes,window", [(1, None), (3, None), (8, None), (40, None), (3, 0), (5, 16)])
def test_ragged_queue_matches_the_reference(gpu_lib, tidigits_bundle, monkeypatch, n_lanes, window):
    """more utterances than lanes (and, at 40, fewer); look-ahead windows of 8 .. 64 frames and per-frame scoring"""
    if window is not None:
        monkeypatch.setenv("S3A_UTT_WIN", str(window))
    dec = bundle.Decoder(tidigits_bundle, n_lanes)
    utts, feats = tidigits_feats(gpu_lib)
    assert len(utts) == 31 and len({len(f) for f in feats}) > 20          # ragged
    dec.decode_queue(feats)
    m, s = queue_lines(dec, utts)
    rm, rs = ref_lines()
    assert m == rm
    assert s == rs
    assert dec.queue_hyp(0, "x", 0)[0].n_frames == len(feats[0])


def test_queue_order_and_reuse_of_an_engine(gpu_lib, tidigits_bundle):
    """longest first (what a caller does to keep the lanes busy to the end), then a plain decode, then the queue again:
    the engine goes from one mode to the other without a trace"""
    dec = bundle.Decoder(tidigits_bundle, 4)
    utts, feats = tidigits_feats(gpu_lib)
    rm, rs = ref_lines()
    order = sorted(range(len(feats)), key=lambda k: -len(feats[k]))
    dec.decode_queue([feats[k] for k in order])
    m, s = queue_lines(dec, utts, order)
    assert m == rm and s == rs
    dec.decode(feats[:4])
    for z in range(4):
        assert dec.format_var(*dec.hyp_var(z, utts[z][1], z)) == (rm[z], rs[z])
    with pytest.raises(gpu_lib.S3AError):
        dec.queue_hyp(0)                          # the last decode was no queue
    dec.decode_queue(feats[::-1])
    m, s = queue_lines(dec, utts, list(range(len(feats)))[::-1])
    assert m == rm and s == rs
    with pytest.raises(gpu_lib.S3AError):
        dec.hyp_var(0)                            # ... and now it was one: no lane results


def test_a_lane_whose_utterance_overflowed_is_scrubbed_on_the_device(gpu_lib, tidigits_bundle):
    """A history table far too small: the long utterances stop in mid-frame (WL_E_TABLE), the 12-frame ones between them
    fit.  Every lane alternates between the two; the short ones must come out as from an engine that never failed."""
    utts, feats = tidigits_feats(gpu_lib)
    short = [f[:12].copy() for f in feats[:6]]
    queue = []
    for k in range(6):
        queue += [feats[k], short[k]]
    small = bundle.Decoder(tidigits_bundle, 2, vh_cap=64)
    with pytest.raises(gpu_lib.S3AError, match="history table full"):
        small.decode_queue(queue)
    st = [small.ud.queue_status(q) for q in range(len(queue))]
    assert all(st[2 * k]["err"] != 0 for k in range(6)), st         # every long one stopped
    assert all(st[2 * k + 1]["err"] == 0 for k in range(6)), st
    good = bundle.Decoder(tidigits_bundle, 2)
    good.decode_queue(short)
    for k in range(6):
        a = small.queue_hyp(2 * k + 1, f"s{k}", k)
        b = good.queue_hyp(k, f"s{k}", k)
        assert bytes(a[0]) == bytes(b[0]) and np.array_equal(a[1], b[1])
        assert small.queue_hyp(2 * k)[0].status == -1
    # the engine's lanes were left dirty or clean as their LAST utterance left them: the next queue is decoded right
    small.decode_queue(short)
    for k in range(6):
        a, b = small.queue_hyp(k, f"s{k}", k), good.queue_hyp(k, f"s{k}", k)
        assert bytes(a[0]) == bytes(b[0]) and np.array_equal(a[1], b[1])


def test_second_pass_inside_the_queue_through_the_c_abi(gpu_lib, tidigits_bundle):
    """s3a_uttdec_enable_bestpath + s3a_uttdec_decode_queue: the second pass of a lane runs at its refill event; per utterance
    the record of a lock-step decode's s3a_uttdec_bestpath_hyp, words, scores and frame normalisers"""
    utts, feats = tidigits_feats(gpu_lib)
    lock = bundle.Decoder(tidigits_bundle, 4, bestpath=True)
    want = []
    for i in range(0, 12, 4):
        lock.decode(feats[i:i + 4])
        want += [lock.bestpath_hyp(z, utts[i + z][1], i + z) for z in range(4)]
    dec = bundle.Decoder(tidigits_bundle, 3, bestpath=True)
    dec.decode_queue(feats[:12])
    for u in range(12):
        h, w = dec.queue_bestpath_hyp(u, utts[u][1], u)
        assert h.status == 0 and h.status == want[u][0].status
        assert (h.n_frames, h.n_words, h.score, h.total_scale, h.exit_id) == (want[u][0].n_frames, want[u][0].n_words, want[u][0].score,
                                                                            want[u][0].total_scale, want[u][0].exit_id), u
        assert np.array_equal(w, want[u][1]), u
        assert dec.format_var(h, w) == lock.format_var(*want[u])


def test_queue_refuses_what_it_cannot_serve(gpu_lib, tidigits_bundle):
    utts, feats = tidigits_feats(gpu_lib)
    dec = bundle.Decoder(tidigits_bundle, 2)
    with pytest.raises(gpu_lib.S3AError):
        dec.decode_queue([feats[0], feats[1][:0]])                  # an utterance without frames


@pytest.mark.parametrize("env", [{"S3A_UTT": "4", "S3A_UTT_QUEUE": "31"}, {"S3A_UTT": "6", "S3A_UTT_ENGINES": "2", "S3A_UTT_QUEUE": "12"},
                                 {"S3A_UTT": "2", "S3A_UTT_QUEUE": "1", "S3A_UTT_WIN": "0"}])
def test_drop_in_program_with_lane_refill(tmp_path, env):
    """sphinx3_decode's command line, S3A_UTT_QUEUE: the control file in queues of 31 / 12 / 2 entries, lanes refilled,
    the -hyp / -hypseg files written in control-file order from the device's hypothesis records"""
    import subprocess
    from test_gpu_uttdec import AM, TST
    hyp, seg = str(tmp_path / "q.match"), str(tmp_path / "q.matchseg")
    args = [TST, "-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-ctl", f"{D}/tidigits.length.arb.regression",
            "-op_mode", "4", "-lm", f"{D}/tidigits.DMP", "-hyp", hyp, "-hypseg", seg]
    p = subprocess.run(args, env=dict(os.environ, **env), capture_output=True, text=True, errors="ignore", timeout=900)
    assert p.returncode == 0, p.stderr[-2500:]
    assert open(hyp).read() == open(f"{D}/ref_mode4_trigram.match").read()
    assert open(seg).read() == open(f"{D}/ref_mode4_trigram.matchseg").read()


@pytest.mark.parametrize("n_lanes,window", [(1, None), (5, None), (3, 0), (8, 16)])
def test_graph_mode_replays_a_block_of_frames(gpu_lib, tidigits_bundle, monkeypatch, n_lanes, window):
    """s3a_uttdec_opts_t.graph: the launches of a block of frames captured once as a HIP graph and replayed (the frame number
    reaches the kernels through a device counter); plain decodes and queues, the same bytes as stream mode = the reference"""
    monkeypatch.setenv("S3A_UTT_GRAPH", "1")
    if window is not None:
        monkeypatch.setenv("S3A_UTT_WIN", str(window))
    dec = bundle.Decoder(tidigits_bundle, n_lanes)
    utts, feats = tidigits_feats(gpu_lib)
    rm, rs = ref_lines()
    for k in range(0, 2 * n_lanes, n_lanes):                # two plain decodes (the second replays the first's graph)
        chunk = feats[k:k + n_lanes]
        dec.decode(chunk)
        for z in range(len(chunk)):
            assert dec.format_var(*dec.hyp_var(z, utts[k + z][1], k + z)) == (rm[k + z], rs[k + z])
    dec.decode_queue(feats)
    m, s = queue_lines(dec, utts)
    assert m == rm and s == rs
    dec.decode(feats[:1])                                   # fewer lanes: another graph
    assert dec.format_var(*dec.hyp_var(0, utts[0][1], 0)) == (rm[0], rs[0])


def test_scan_with_small_workgroups(gpu_lib, tidigits_bundle, monkeypatch):
    """the scan as 256-thread workgroups (what engines of 64 lanes and more use: four times the chunks per list) forced onto a
    3-lane engine: plain decodes and a queue, the reference's lines"""
    monkeypatch.setenv("S3A_UTT_SCAN_SMALL", "1")
    dec = bundle.Decoder(tidigits_bundle, 3)
    utts, feats = tidigits_feats(gpu_lib)
    rm, rs = ref_lines()
    dec.decode(feats[:3])
    for z in range(3):
        assert dec.format_var(*dec.hyp_var(z, utts[z][1], z)) == (rm[z], rs[z])
    dec.decode_queue(feats)
    m, s = queue_lines(dec, utts)
    assert m == rm and s == rs


@pytest.mark.parametrize("lanes", [3, 9])
def test_lattices_out_of_a_queue_through_the_c_abi(gpu_lib, tidigits_bundle, lanes):
    """s3a_uttdec_queue_keep_lattices + s3a_uttdec_queue_lattice: every utterance's lattice as a lock-step decode's s3a_uttdec_lattice hands
    it out -- nodes and links in the reference's list orders, word for word --, with 3 lanes (the second pass at refill events) and 9 (groups
    of static ku_frames launches); without the switch a queue keeps none"""
    utts, feats = tidigits_feats(gpu_lib)
    lock = bundle.Decoder(tidigits_bundle, 4, bestpath=True)
    want = []
    for i in range(0, 12, 4):
        lock.decode(feats[i:i + 4])
        want += [lock.ud.lattice(z) for z in range(4)]
    dec = bundle.Decoder(tidigits_bundle, lanes, bestpath=True)
    dec.decode_queue(feats[:12])
    with pytest.raises(Exception):
        dec.ud.queue_lattice(0)
    dec.ud.queue_keep_lattices()
    dec.decode_queue(feats[:12])
    for u in range(12):
        info, nodes, links = dec.ud.queue_lattice(u)
        wi, wn, wl = want[u]
        assert (info.n_frames, info.n_nodes, info.n_links, info.initial, info.final, info.final_ascr) == \
               (wi.n_frames, wi.n_nodes, wi.n_links, wi.initial, wi.final, wi.final_ascr), u
        assert np.array_equal(nodes, wn) and np.array_equal(links, wl), u
        h, w = dec.queue_bestpath_hyp(u, utts[u][1], u)
        assert h.status == 0
