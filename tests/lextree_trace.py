"""Parse / replay the lextree trace written by oracle/_ref/ref_s3olt_decode (S3O_TRACE=...).

The trace holds the flattened lextrees of a real mode-4 decode (tidigits, trigram LM) and,
per frame, every input the lextree operations consumed (senone + composite scores, beam
thresholds, root-entry calls) and every result they produced (per-tree best scores, active
lists, HMM states, word exits).  It was produced while the decoder's output was
byte-identical to the unmodified reference, so the recorded results are reference-grade.
"""
import numpy as np


def parse(path_or_bytes, max_frames=None):
    raw = np.fromfile(path_or_bytes, dtype="<i4") if isinstance(path_or_bytes, str) else \
        np.frombuffer(path_or_bytes, dtype="<i4")
    pos = 0
    recs = []
    while pos < len(raw):
        tag, n = int(raw[pos]), int(raw[pos + 1])
        recs.append((tag, raw[pos + 2: pos + 2 + n]))
        pos += 2 + n
    it = iter(recs)
    tag, hdr = next(it)
    assert tag == 1
    out = dict(n_tree=int(hdr[0]), n_emit=int(hdr[1]), n_tmat=int(hdr[2]), n_sseq=int(hdr[3]),
               n_comsseq=int(hdr[4]), n_comstate=int(hdr[5]), n_sen=int(hdr[6]))
    static = {2: "tp", 3: "sseq", 4: "comsseq", 5: "comstate_off", 6: "comstate"}
    trees, frames, cur, events = [], [], None, []
    tree = None
    for tag, d in it:
        if tag in static:
            out[static[tag]] = d.copy()
        elif tag == 10:
            tree = dict(n_node=int(d[0]), n_lc=int(d[1]), n_root=int(d[2]), type=int(d[3]),
                        lc=np.zeros(0, np.int16), lcroot_off=np.zeros(1, np.int32), lcroot=np.zeros(0, np.int32))
            trees.append(tree)
        elif 11 <= tag <= 21:
            name = {11: "ssid", 12: "tmatid", 13: "composite", 14: "wid", 15: "prob", 16: "child_off",
                    17: "child", 18: "lc", 19: "lcroot_off", 20: "lcroot", 21: "root"}[tag]
            tree[name] = d.copy()
        else:
            events.append((tag, d.copy()))
    out["trees"] = trees
    out["events"] = events
    return out


def to_npz_dict(tr, max_frames):
    """Flatten a parsed trace into arrays for np.savez (events kept as a tagged stream,
    truncated after max_frames search frames)."""
    d = {k: np.asarray(v) for k, v in tr.items() if k not in ("trees", "events")}
    for i, t in enumerate(tr["trees"]):
        for k, v in t.items():
            d[f"tree{i}_{k}"] = np.asarray(v)
    tags, lens, payload, nfr = [], [], [], 0
    for tag, data in tr["events"]:
        if tag == 40:
            nfr += 1
            if nfr > max_frames:
                break
        tags.append(tag); lens.append(len(data)); payload.append(data)
    d["ev_tag"] = np.array(tags, np.int32)
    d["ev_len"] = np.array(lens, np.int32)
    d["ev_data"] = np.concatenate(payload).astype(np.int32)
    return d


def from_npz(z):
    class _Loaded(dict):                # decompress every member once (NpzFile re-reads on each access)
        files = property(lambda self: list(self))
    z = _Loaded({k: z[k] for k in z.files})
    tr = {k: (int(z[k]) if z[k].ndim == 0 else z[k]) for k in z.files
          if not k.startswith(("tree", "ev_"))}
    trees = []
    for i in range(int(tr["n_tree"])):
        t = {k[len(f"tree{i}_"):]: z[k] for k in z.files if k.startswith(f"tree{i}_")}
        for k in ("n_node", "n_lc", "n_root", "type"):
            t[k] = int(t[k])
        trees.append(t)
    tr["trees"] = trees
    ev, off = [], 0
    for tag, n in zip(z["ev_tag"], z["ev_len"]):
        ev.append((int(tag), z["ev_data"][off:off + n]))
        off += n
    tr["events"] = ev
    return tr


class Replayer:
    """Drive a backend through the traced sequence and compare after every operation.

    backend must provide: enter(tree, lc[], scr[], hist[], cf, thresh); swap();
    hmm_eval(senscr, comsen, frm) -> (best[], wbest[], nact[]); propagate(cf, th, pth, wth);
    leaves(wth) -> list per tree of (wid[], score[], hist[]); active(tree, which) -> node ids;
    state(tree) -> [n_node][10] int32 (score0..2, hist0..2, out_score, out_hist, bestscore, frame);
    sen_active() -> uint8[n_sen] or None.
    """

    def __init__(self, tr, backend):
        self.tr, self.b = tr, backend
        self.T = tr["n_tree"]

    def _check_state(self, it, label):
        for t in range(self.T):
            _, act = next(it)
            _, nxt = next(it)
            _, st = next(it)
            assert np.array_equal(self.b.active(t, 0), act), f"{label}: active list of tree {t}"
            assert np.array_equal(self.b.active(t, 1), nxt), f"{label}: next_active list of tree {t}"
            got = self.b.state(t)
            exp = st.reshape(-1, 10)
            assert np.array_equal(got, exp), \
                f"{label}: HMM state of tree {t}, first bad node {np.nonzero((got != exp).any(1))[0][:5]}"

    def run(self):
        it = iter(self.tr["events"])
        frm = None
        n_frames = n_exits = 0
        senscr = comsen = None
        for tag, d in it:
            if tag == 30:
                t, n, cf, thresh = (int(v) for v in d)
                lc = next(it)[1]; scr = next(it)[1]; hist = next(it)[1]
                self.b.enter(t, lc, scr, hist, cf, thresh)
            elif tag == 40:
                frm = int(d[0]); n_frames += 1
                if n_frames == 1:
                    self.b.swap()       # srch_TST_begin: lextree_active_swap after the initial entries
                senscr = next(it)[1]; comsen = next(it)[1]; sen_active = next(it)[1]
                got = self.b.sen_active()
                if got is not None:
                    assert np.array_equal(got.astype(bool) | self._ci_mask(sen_active, got), sen_active.astype(bool)), \
                        f"frame {frm}: active-senone mask"
                tagr, r = next(it); assert tagr == 44
                best = next(it)[1]; wbest = next(it)[1]; nact = next(it)[1]
                gb, gw, gn = self.b.hmm_eval(senscr, comsen, frm)
                assert np.array_equal(gn, nact), f"frame {frm}: n_active"
                assert np.array_equal(gb, best) and np.array_equal(gw, wbest), f"frame {frm}: best scores"
                self._check_state(it, f"frame {frm} after hmm_eval")
            elif tag == 60:
                th, pth, wth = (int(v) for v in d)
                self.b.propagate(frm, th, pth, wth)
                self._check_state(it, f"frame {frm} after propagate_non_leaves")
                self._wth = wth
                exits = self.b.leaves(wth)
                for t in range(self.T):
                    n = int(next(it)[1][0])
                    wid = next(it)[1]; scr = next(it)[1]; hist = next(it)[1]
                    gwid, gscr, ghist = exits[t]
                    assert len(gwid) == n, f"frame {frm}: #word exits of tree {t}"
                    assert np.array_equal(gwid, wid) and np.array_equal(gscr, scr) and np.array_equal(ghist, hist), \
                        f"frame {frm}: word exits of tree {t}"
                    n_exits += n
            elif tag == 80:
                # frame_windup: swap happened in the trace just before this dump
                self.b.swap()
                # re-feed this record's remaining pieces through the checker
                first = (tag, d)
                def chain():
                    yield first
                    yield from it
                self._check_state(chain(), f"frame {frm} after frame_windup")
            elif tag == 99:
                break
        return n_frames, n_exits

    def _ci_mask(self, ref_mask, got):
        # CI senones are forced active by approx_cont_mgau_frame_eval in the reference run
        # (the trace is taken after scoring); the lextree marking alone does not set them
        m = np.zeros(len(ref_mask), bool)
        m[: self.tr.get("n_ci_sen", 0)] = True
        return m
