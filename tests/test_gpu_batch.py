"""Parity (MI355X): the BATCHED fused frame (s3a_batch_*, cmusphinx_amd/csrc/s3a_batch.hip) -- several
decoders sharing every kernel launch -- against one CPU oracle per decoder, step by step: frame
results, word exits, active lists and every HMM of every decoder; decoders enter and leave
utterances at different times, use different beams / histogram caps, and are also driven from
one host thread each through the blocking rendezvous."""
import threading

import numpy as np
import pytest

import oracle_lib as O
from cmusphinx_amd import synth
from test_gpu_lextree import Lockstep, OracleFrame, make_gpu, synth_forest

pytestmark = pytest.mark.gpu
HMMBEAM, PBEAM, WBEAM = -2600000, -2000000, -1500000


class Decoder:
    """One decoder = oracle side (OracleFrame) + device side (LexSearch/Scorer/ComSen in a Batch slot)."""

    def __init__(self, gpu_lib, batch, seed, maxhmmpf, ci_pbeam, n_frames, shared=None, forest=None):
        """shared = (model dict, MgauModel): all decoders score against ONE model on the device (their
        scorers keep private Gaussian-selection state) -- the engine then runs the model-stationary
        CD kernel; None: a model of its own."""
        self.rng = np.random.default_rng(seed)
        self.tr = forest["tr"] if forest else synth_forest(self.rng, n_tree=4, n_node=700, n_sen=500)
        n_ci = 30
        m = shared[0] if shared else synth.make_model(500, n_ci, 4, 39, 5, 3, seed=seed + 100)
        self.feats = synth.make_features(m, n_frames, seed=seed + 200)
        comwt = forest["comwt"] if forest else -self.rng.integers(0, 3000, self.tr["n_comstate"]).astype(np.int32)
        olm = O.OracleLogMath(1.0003)
        self.of = OracleFrame(self.tr, O.OracleMgau(m["mean"], m["var"], m["mixw"], olm), m["cd2cisen"], n_ci,
                              olm.logs3(ci_pbeam), comwt)
        gm = shared[1] if shared else gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], gpu_lib.LogMath(1.0003))
        self.sc = gpu_lib.Scorer(gm, m["cd2cisen"], n_ci, ci_pbeam=ci_pbeam, private_state=shared is not None)
        self.cs = gpu_lib.ComSen(self.tr["comstate_off"], self.tr["comstate"], comwt)
        if forest and "proto" in forest:        # the same lextrees as another decoder: share the static arrays
            self.ls = forest["proto"].clone(stream=gm.stream())
        else:
            self.ls = make_gpu(gpu_lib, self.tr, stream=gm.stream())
            if forest is not None:
                forest["proto"] = self.ls
        self.batch, self.slot = batch, batch.attach(self.ls, self.sc, self.cs)
        self.lock = Lockstep(self.of.lex, self.ls, self.tr["n_tree"])
        self.maxhmmpf, self.frm, self.n_hist, self.most = maxhmmpf, None, 0, 0

    def begin(self):
        self.of.fs.g.reset_state()
        self.batch.utt_begin(self.slot)
        first = (0, [0, 3, 4], [0, 0, -50], [7, 8, 9]), (2, [1], [0], [9])
        for g in first:
            self.of.lex.enter(g[0], g[1], g[2], g[3], -1, -10**9)
        self.of.lex.swap()
        self.batch.transition(self.slot, -1, -10**9, *first)
        self.frm = 0

    def oracle_frame(self):
        self.o = self.of.frame(self.feats[self.frm], self.frm, HMMBEAM, PBEAM, WBEAM, self.maxhmmpf)

    def args(self):
        return (self.slot, self.feats[self.frm], self.frm, self.frm, HMMBEAM, PBEAM, WBEAM, 0, self.maxhmmpf)

    def check_and_advance(self, res, exits):
        o, frm, rng = self.o, self.frm, self.rng
        assert (res.best_hmm, res.best_word, res.n_hmm) == (o["bh"], o["bw"], o["n"]), (self.slot, frm)
        assert (res.thres, res.phone_thres, res.word_thres) == (o["th"], o["pth"], o["wth"]), (self.slot, frm)
        assert bool(res.need_histprune) == o["hist"], (self.slot, frm)
        ns, ng, cin, cig, cib = o["counts"]
        assert tuple(res.extra[1:7]) == (ns, ng, cin, cig, cib, o["best"]), (self.slot, frm)
        for t in range(self.tr["n_tree"]):
            assert all(np.array_equal(u, v) for u, v in zip(o["exits"][t], exits[t])), (self.slot, frm, t)
        self.n_hist += o["hist"]
        self.most = max(self.most, o["n"])
        k = frm % 2
        n = int(rng.integers(0, 5))
        ga = (k, rng.choice(9, n, replace=False), (o["bh"] - rng.integers(0, 900000, n)).astype(np.int32),
              rng.integers(0, 10**6, n).astype(np.int32)) if n else None
        gb = (2 + k, [int(rng.integers(0, 9))], [o["bh"] - 1000], [frm]) if frm % 3 else None
        for g in (ga, gb):
            if g is not None:
                self.of.lex.enter(g[0], g[1], g[2], g[3], frm, o["bh"] + HMMBEAM)
        self.of.lex.swap()
        self.batch.transition(self.slot, frm, o["bh"] + HMMBEAM, ga, gb)
        self.frm += 1

    def end(self):
        self.batch.utt_end(self.slot)
        self.of.lex.utt_end()
        self.lock.same(("utt_end", self.slot))       # every HMM cleared, both lists empty
        self.frm = None


@pytest.mark.parametrize("share_model", [False, True])
def test_batched_steps_match_one_oracle_per_decoder(gpu_lib, share_model):
    batch = gpu_lib.Batch(8)
    cfg = [(11, 20000, 1e-80, 30), (12, 150, 1e-80, 22), (13, 400, 1e-12, 30), (14, 20000, 1e-30, 17), (15, 90, 1e-80, 26)]
    shared = None
    if share_model:
        m = synth.make_model(500, 30, 4, 39, 5, 3, seed=999)
        shared = (m, gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], gpu_lib.LogMath(1.0003)))
    decs = [Decoder(gpu_lib, batch, *c, shared=shared) for c in cfg]
    start = [0, 0, 3, 5, 9]             # step at which each decoder begins its first utterance
    utts_done = [0] * len(decs)
    for step in range(75):
        for i, d in enumerate(decs):
            if d.frm is None and step >= start[i] and utts_done[i] < 2:
                d.begin()
        live = [d for d in decs if d.frm is not None]
        if not live:
            break
        for d in live:
            d.oracle_frame()
            batch.submit(*d.args())
        out = batch.run()
        assert set(out) == {d.slot for d in live}
        for i, d in enumerate(decs):
            if d in live:
                d.check_and_advance(*out[d.slot])
                if d.frm >= len(d.feats):
                    d.end()
                    utts_done[i] += 1
                    start[i] = step + 2 + i          # pause before the second utterance
    assert utts_done == [2] * len(decs)
    assert decs[1].n_hist > 10 and decs[4].n_hist > 10 and decs[0].n_hist == 0
    steps, frames = batch.stats()
    assert frames == 2 * sum(c[3] for c in cfg) and steps < frames / 2      # really batched


@pytest.mark.parametrize("n_dec,n_comp,n_node", [(5, 4, 700), (11, 8, 700), (19, 8, 700), (3, 8, 10000)])
def test_cloned_decoders_share_model_and_lextrees(gpu_lib, n_dec, n_comp, n_node, variants):
    """The production arrangement: ONE model and ONE set of lextrees on the device, N decoders with their
    own state (s3a_scorer_init_private + s3a_lexsearch_clone), different utterances.  With 8 Gaussians per
    senone and >= 8 decoders in a step the CD senones of all of them are one model-stationary pass
    (kb_gated_cd_multi: groups of 8 decoders, the last one partly filled); as the shorter utterances end the
    steps fall back to one launch per decoder."""
    if n_node > 1024:       # lists of several 1024-position chunks through k_dec_scan's chained multi-workgroup path
        variants(scan_chained=1)
    batch = gpu_lib.Batch(n_dec + 1)
    rng = np.random.default_rng(77)
    tr = synth_forest(rng, n_tree=4, n_node=n_node, n_sen=500)
    forest = dict(tr=tr, comwt=-rng.integers(0, 3000, tr["n_comstate"]).astype(np.int32))
    m = synth.make_model(500, 30, n_comp, 39, 5, 3, seed=998)
    shared = (m, gpu_lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], gpu_lib.LogMath(1.0003)))
    decs = [Decoder(gpu_lib, batch, 31 + i, 20000 if i % 2 else 200, 1e-80 if i % 5 < 3 else 1e-12, 20 + (3 * i) % 17,
                    shared=shared, forest=forest) for i in range(n_dec)]
    for d in decs:
        d.begin()
    while any(d.frm is not None for d in decs):
        live = [d for d in decs if d.frm is not None]
        for d in live:
            d.oracle_frame()
            batch.submit(*d.args())
        out = batch.run()
        for d in live:
            d.check_and_advance(*out[d.slot])
            if d.frm >= len(d.feats):
                d.end()
    assert sum(d.n_hist for d in decs) > (10 if n_node < 1024 else 0)
    if n_node > 1024:
        assert max(d.most for d in decs) > 2 * 1024     # (lists around / over one chunk; the launch has several per tree)


def test_blocking_rendezvous_one_thread_per_decoder(gpu_lib):
    batch = gpu_lib.Batch(4)
    decs = [Decoder(gpu_lib, batch, 21 + i, 20000 if i else 200, 1e-80, 12 + 5 * i) for i in range(3)]
    errors = []

    def worker(d):
        try:
            for _ in range(2):
                d.begin()
                while d.frm < len(d.feats):
                    d.oracle_frame()
                    d.check_and_advance(*batch.step(*d.args()))
                d.end()
        except BaseException as e:      # noqa: BLE001 -- surfaced in the main thread
            errors.append((d.slot, repr(e)))
            try:
                batch.utt_end(d.slot)   # do not leave the others waiting
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=(d,)) for d in decs]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in th)
    steps, frames = batch.stats()
    assert frames == 2 * sum(len(d.feats) for d in decs)
