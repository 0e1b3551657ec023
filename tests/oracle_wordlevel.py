"""ctypes view of the oracle's word level (oracle/s3o_wordlevel.c): TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import oracle_lib as O

I32P = C.POINTER(C.c_int32)
U8P = C.POINTER(C.c_uint8)


class Lm3g(C.Structure):
    _fields_ = [("n_ug", C.c_int32), ("n_bg", C.c_int32), ("n_tg", C.c_int32)] + \
               [(k, I32P) for k in ("ug_prob", "ug_bowt", "ug_firstbg", "bg_wid", "bg_prob", "bg_bowt", "bg_firsttg",
                                    "tg_wid", "tg_prob", "inclass")]


class WDict(C.Structure):
    _fields_ = [("n_word", C.c_int32), ("n_ci", C.c_int32), ("lwid", I32P), ("is_filler", U8P), ("fillpen", I32P),
                ("last_ci", I32P)] + [(k, C.c_int32) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid")]


class Vithist(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("cap", "max_frames", "n_entry", "n_frm", "wbeam", "bghist", "overflow")] + \
               [(k, I32P) for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")] + \
               [("valid", U8P)] + [(k, I32P) for k in ("frame_start", "bestscore", "bestvh")]


def _ip(a):
    return a.ctypes.data_as(I32P)


_TYPED = False


def lib():
    global _TYPED
    L = O.lib()
    if not _TYPED:
        L.s3o_lm_tg_score.restype = C.c_int32
        L.s3o_lm_tg_score.argtypes = [C.POINTER(Lm3g)] + [C.c_int32] * 4
        L.s3o_vithist_init.restype = C.POINTER(Vithist)
        L.s3o_vithist_init.argtypes = [C.c_int32] * 4
        L.s3o_vithist_free.argtypes = [C.POINTER(Vithist)]
        L.s3o_vithist_utt_begin.argtypes = [C.POINTER(Vithist), C.c_int32, C.c_int32]
        L.s3o_vithist_rescore.restype = C.c_int32
        L.s3o_vithist_rescore.argtypes = [C.POINTER(Vithist), C.POINTER(Lm3g), C.POINTER(WDict)] + [C.c_int32] * 5
        L.s3o_vithist_prune.argtypes = [C.POINTER(Vithist), C.POINTER(WDict)] + [C.c_int32] * 4 + [I32P]
        L.s3o_vithist_frame_windup.argtypes = [C.POINTER(Vithist), C.c_int32]
        L.s3o_word_trans.restype = C.c_int32
        L.s3o_word_trans.argtypes = [C.POINTER(Vithist), C.POINTER(WDict), C.c_int32, C.c_int32] + [I32P] * 5
        L.s3o_vithist_utt_end.restype = C.c_int32
        L.s3o_vithist_utt_end.argtypes = [C.POINTER(Vithist), C.POINTER(Lm3g), C.POINTER(WDict)]
        _TYPED = True
    return L


class OracleWordLevel:
    """The word level of one decoder: LM + dictionary facts (arrays of a parsed trace / synthetic), a history
    table, and one call per frame that does what srch_TST_propagate_graph_wd_lv2 + frame_windup do."""

    def __init__(self, t, cap=1 << 20, max_frames=4096, wordend=None):
        self.L = lib()
        g = lambda k, dt=np.int32: np.ascontiguousarray(t.get(k, np.zeros(1, dt)), dtype=dt)
        self.keep = {k: g(k) for k in ("ug_prob", "ug_bowt", "ug_firstbg", "bg_wid", "bg_prob", "bg_bowt", "bg_firsttg",
                                       "tg_wid", "tg_prob", "lwid", "fillpen", "last_ci")}
        self.keep["is_filler"] = g("is_filler", np.uint8)
        k = self.keep
        self.lm = Lm3g(int(t["n_ug"]), int(t["n_bg"]), int(t["n_tg"]), _ip(k["ug_prob"]), _ip(k["ug_bowt"]),
                       _ip(k["ug_firstbg"]), _ip(k["bg_wid"]), _ip(k["bg_prob"]), _ip(k["bg_bowt"]), _ip(k["bg_firsttg"]),
                       _ip(k["tg_wid"]), _ip(k["tg_prob"]), None)
        self.d = WDict(int(t["n_word"]), int(t["n_ci"]), _ip(k["lwid"]), k["is_filler"].ctypes.data_as(U8P),
                       _ip(k["fillpen"]), _ip(k["last_ci"]), int(t["startwid"]), int(t["finishwid"]), int(t["silwid"]),
                       int(t["start_lwid"]), int(t["finish_lwid"]))
        self.t = t
        self.wordend = int(t.get("wordend", 0)) if wordend is None else wordend
        self.vh = self.L.s3o_vithist_init(cap, max_frames, int(t["wbeam"]), int(t["bghist"]))
        self.begin()

    def begin(self):
        self.L.s3o_vithist_utt_begin(self.vh, int(self.t["startwid"]), int(self.t["start_lwid"]))

    def tg_score(self, lw1, lw2, lw3, wid=0):
        return int(self.L.s3o_lm_tg_score(C.byref(self.lm), lw1, lw2, lw3, wid))

    def frame(self, frm, trees, prune_beam, maxwpf=None, maxhist=None, want_order=False):
        """trees = [(type, wid[], scr[], hist[])...]; returns (entries dict, calls (lc, scr, hist) or None, fill call)"""
        vh = self.vh.contents
        for ty, wid, scr, hist in trees:
            for w, s, h in zip(wid, scr, hist):
                assert self.L.s3o_vithist_rescore(self.vh, C.byref(self.lm), C.byref(self.d), int(w), frm, int(s), int(h), int(ty)) == 0
        n_before = vh.n_entry - vh.frame_start[frm]
        order = np.full(max(n_before, 1), -1, np.int32)
        self.L.s3o_vithist_prune(self.vh, C.byref(self.d), frm, int(self.t["maxwpf"] if maxwpf is None else maxwpf),
                                 int(self.t["maxhistpf"] if maxhist is None else maxhist), int(prune_beam), _ip(order))
        nci = int(self.t["n_ci"])
        lc, cs, ch = (np.zeros(nci + 1, np.int32) for _ in range(3))
        fs, fh = C.c_int32(0), C.c_int32(0)
        n = self.L.s3o_word_trans(self.vh, C.byref(self.d), frm, self.wordend, _ip(lc), _ip(cs), _ip(ch), C.byref(fs), C.byref(fh))
        fs0 = vh.frame_start[frm]
        ne = vh.n_entry - fs0
        ent = {k: np.ctypeslib.as_array(getattr(vh, k), (vh.n_entry,))[fs0:].copy()
               for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")} if ne else \
            {k: np.zeros(0, np.int32) for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")}
        res = dict(n_entry=ne, entries=ent, bestscore=int(vh.bestscore[frm]), bestvh=int(vh.bestvh[frm]),
                   calls=None if n < 0 else (lc[:n].copy(), cs[:n].copy(), ch[:n].copy()), fill=(fs.value, fh.value))
        if want_order:
            res["order"] = order[:n_before]
        self.L.s3o_vithist_frame_windup(self.vh, frm)
        return res

    def table(self):
        vh = self.vh.contents
        n, nf = vh.n_entry, vh.n_frm
        out = {k: np.ctypeslib.as_array(getattr(vh, k), (n,)).copy()
               for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")}
        out["frame_start"] = np.ctypeslib.as_array(vh.frame_start, (nf + 1,)).copy()
        out["bestscore"] = np.ctypeslib.as_array(vh.bestscore, (nf + 1,)).copy()
        out["bestvh"] = np.ctypeslib.as_array(vh.bestvh, (nf + 1,)).copy()
        return out


def random_task(rng, n_word=40, n_ci=8, n_filler=3, density=0.3, grid=100):
    """A small random trigram + dictionary in the trace's layout (scores on a coarse grid, so that ties are common)."""
    n_ug = n_word - n_filler + 2                    # words + <s> </s>
    ug_prob = (rng.integers(-60, -5, n_ug) * 100 + rng.integers(0, 100, n_ug) // grid * grid).astype(np.int32)
    ug_bowt = (rng.integers(-20, 0, n_ug) * 100).astype(np.int32)
    bg_w, bg_p, bg_b, firstbg, tg_w, tg_p, firsttg = [], [], [], [0], [], [], []
    for w1 in range(n_ug):
        for w2 in np.flatnonzero(rng.random(n_ug) < density):
            firsttg.append(len(tg_w))
            bg_w.append(w2); bg_p.append(int(rng.integers(-50, -2)) * 100); bg_b.append(int(rng.integers(-15, 0)) * 100)
            for w3 in np.flatnonzero(rng.random(n_ug) < density):
                tg_w.append(w3); tg_p.append(int(rng.integers(-40, -1)) * 100)
        firstbg.append(len(bg_w))
    firsttg.append(len(tg_w))
    is_filler = np.zeros(n_word, np.uint8)
    is_filler[n_word - n_filler:] = 1
    lwid = np.arange(n_word, dtype=np.int32)
    lwid[n_word - n_filler:] = -1
    fillpen = np.zeros(n_word, np.int32)
    fillpen[n_word - n_filler:] = rng.integers(-30, -10, n_filler) * 100
    last_ci = rng.integers(0, n_ci - 1, n_word).astype(np.int32)
    last_ci[n_word - n_filler:] = n_ci - 1          # fillers end in silence
    a = lambda x: np.asarray(x, np.int32)
    return dict(n_ug=n_ug, n_bg=len(bg_w), n_tg=len(tg_w), n_word=n_word, n_ci=n_ci, startwid=n_word - n_filler,
                finishwid=n_word - n_filler + 1, silwid=n_word - 1, start_lwid=n_ug - 2, finish_lwid=n_ug - 1,
                wbeam=-4000, bghist=0, maxwpf=4, maxhistpf=12, n_lextree=3, epl=3,
                ug_prob=ug_prob, ug_bowt=ug_bowt, ug_firstbg=a(firstbg), bg_wid=a(bg_w), bg_prob=a(bg_p), bg_bowt=a(bg_b),
                bg_firsttg=a(firsttg), tg_wid=a(tg_w), tg_prob=a(tg_p), lwid=lwid, is_filler=is_filler, fillpen=fillpen,
                last_ci=last_ci)


def random_frame(rng, ow, frm, n_tree=6, max_exits=14, grid=100):
    """Random word exits whose predecessors are existing history entries; scores on a coarse grid."""
    vh = ow.vh.contents
    n_hist = vh.n_entry
    t = ow.t
    n_word, n_filler = int(t["n_word"]), int(np.sum(t["is_filler"]))
    trees = []
    for k in range(n_tree):
        n = int(rng.integers(0, max_exits)) if rng.random() < 0.8 else 0
        filler_tree = k >= n_tree // 2
        if filler_tree:
            wid = rng.integers(n_word - n_filler + 2, n_word, min(n, 3)) if n_filler > 2 else np.zeros(0, np.int64)
        else:
            wid = rng.integers(0, n_word - n_filler, n)
        hist = rng.integers(0, n_hist, len(wid))
        scr = (rng.integers(-9000, -4000, len(wid)) // grid * grid - 3000 * frm).astype(np.int32)
        trees.append((-1 if filler_tree else 0, wid.astype(np.int32), scr, hist.astype(np.int32)))
    return trees
