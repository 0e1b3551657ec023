"""Seeded synthetic search descriptors (s3a_psfwd_desc_t / s3o_psfwd_desc_t: same layout) and ctypes drivers of the
oracle's restatement (oracle/libs3oracle.so) and the device engine (libcmusphinx_amd.so) -- test infrastructure.
Random lexicon trees, right-context tables, trigram LMs and 3- or 5-state topologies that no shipped model has."""
import ctypes as C
import os

import numpy as np

from conftest import ROOT

I32, I16, U16, U8 = C.c_int32, C.c_int16, C.c_uint16, C.c_uint8
P = C.c_void_p


class Desc(C.Structure):
    _fields_ = [(n, I32) for n in ("n_ci", "sil_ci", "n_emit", "n_sen", "n_sseq", "n_tmat")] + [("sseq", P), ("tp", P)] + \
        [(n, I32) for n in ("n_words", "start_wid", "finish_wid", "silence_wid")] + \
        [(n, P) for n in ("w_basewid", "w_lmwid", "w_first_ci", "w_last_ci", "w_last2_ci", "w_flags", "w_rc_off", "rc_ssid", "w_rc_row")] + \
        [("n_rc_rows", I32), ("rc_cimap", P), ("w_rc_tmat", P), ("n_root", I32), ("n_nonroot", I32)] + \
        [(n, P) for n in ("root_ci", "root_ci2", "root_tmat", "root_ssid0", "root_lc_ssid", "ch_child_off", "ch_child",
                          "ch_pen_off", "ch_pen_wid", "nr_ssid", "nr_tmat", "nr_ci")] + \
        [("n_1ph", I32), ("n_1ph_lm", I32)] + [(n, P) for n in ("sp_wid", "sp_ssid0", "sp_lc_ssid", "sp_tmat", "sp_ci")] + \
        [("n_fill", I32), ("fill_sp", P)] + [(n, I32) for n in ("lm_order", "lm_n_ug", "lm_n_bg", "lm_n_tg", "lm_zero")] + \
        [(n, P) for n in ("ug_prob", "ug_bowt", "ug_firstbg", "bg_wid", "bg_prob", "bg_bowt", "bg_firsttg", "tg_wid", "tg_prob")] + \
        [(n, I32) for n in ("beam", "pbeam", "wbeam", "lpbeam", "lponlybeam", "fillpen", "silpen", "nwpen", "pip", "maxwpf", "maxhmmpf")] + \
        [(n, I32) for n in ("pl_window", "pl_beam", "pl_pbeam", "pl_pip")] + [("ci_ssid", P), ("ci_tmat", P), ("w_lmcw", P)]


def make_desc(seed, n_emit=3, n_ci=10, n_real=40, n_sen=160, lm_order=3, maxwpf=-1, maxhmmpf=-1, skips=True, beam=-2500, pl_window=0,
              classes=False):
    """-> (Desc, dict of the numpy arrays it points into)"""
    r = np.random.default_rng(seed)
    A = {}
    d = Desc()
    n_sseq, n_tmat = 90, 6
    A["sseq"] = r.integers(0, n_sen, (n_sseq, n_emit)).astype(np.uint16)
    tp = np.full((n_tmat, n_emit, n_emit + 1), 255, np.uint8)
    for t in range(n_tmat):
        for i in range(n_emit):
            tp[t, i, i] = r.integers(0, 40); tp[t, i, i + 1] = r.integers(0, 60)
            if i + 2 <= n_emit and (n_emit == 5 or (skips and t % 2 == 0)):
                tp[t, i, i + 2] = r.integers(20, 120)
    A["tp"] = tp
    # dictionary: real words (some single-phone, some alternates), then the filler range <sil> <s> </s> ++n++
    prons, base = [], []
    for w in range(n_real):
        ln = 1 if w % 9 == 4 else int(r.integers(2, 6))
        prons.append([int(x) for x in r.integers(1, n_ci, ln)]); base.append(w)
    for w in range(3, n_real, 7):       # alternates share the base word's LM id
        if len(prons[w]) > 1 and len(prons[w - 1]) > 1:
            base[w] = w - 1
    sil_w, start_w, finish_w, noise_w = n_real, n_real + 1, n_real + 2, n_real + 3
    prons += [[0], [0], [0], [int(r.integers(1, n_ci))]]
    base += [sil_w, start_w, finish_w, noise_w]
    W = len(prons)
    flags = np.zeros(W, np.uint8)
    for w in range(W):
        flags[w] = (1 if len(prons[w]) == 1 else 0) | (2 if w in (sil_w, noise_w) else 0) | (4 if w < n_real else 0)
    A["w_basewid"] = np.array(base, np.int32)
    # LM vocabulary: base real words + <s> + </s>; one real word is NOT in the LM (and so not in the tree)
    oov = n_real - 1
    lm_words = [w for w in range(n_real) if base[w] == w and w != oov] + [start_w, finish_w]
    lmid = {w: i for i, w in enumerate(lm_words)}
    A["w_lmwid"] = np.array([lmid.get(base[w], -1) if (w < n_real or w in (start_w, finish_w)) else -1 for w in range(W)], np.int32)
    A["w_first_ci"] = np.array([p[0] for p in prons], np.int16)
    A["w_last_ci"] = np.array([p[-1] for p in prons], np.int16)
    A["w_last2_ci"] = np.array([p[-2] if len(p) > 1 else -1 for p in prons], np.int16)
    A["w_flags"] = flags
    # right contexts per (last, last2)
    rows, rc_off, rc_ssid, rc_row, cimap, rc_tmat = {}, [0], [], [], [], []
    for w in range(W):
        if len(prons[w]) == 1:
            rc_row.append(-1); rc_tmat.append(-1); rc_off.append(rc_off[-1]); continue
        key = (prons[w][-1], prons[w][-2])
        if key not in rows:
            n = int(r.integers(1, n_ci + 1))
            m = r.integers(0, n, n_ci); m[:n] = r.permutation(n)[:n] if n <= n_ci else m[:n]
            rows[key] = (len(rows), list(r.choice(n_sseq, n, replace=False)), m.astype(np.int16), int(r.integers(0, n_tmat)))
            cimap.append(rows[key][2])
        row, ss, _, tm = rows[key]
        rc_row.append(row); rc_tmat.append(tm); rc_ssid += ss; rc_off.append(rc_off[-1] + len(ss))
    A["w_rc_off"] = np.array(rc_off, np.int32); A["rc_ssid"] = np.array(rc_ssid + [0], np.uint16)
    A["w_rc_row"] = np.array(rc_row, np.int32); A["rc_cimap"] = np.array(cimap, np.int16).reshape(len(rows), n_ci)
    A["w_rc_tmat"] = np.array(rc_tmat, np.int16)
    # the tree (create_search_tree's shape: roots by first two phones, interior by path, last phone outside)
    roots, nodes = {}, []           # nodes: dict(children, pen)
    root_list = []
    for w in range(n_real):
        if len(prons[w]) == 1 or A["w_lmwid"][w] < 0:
            continue
        k = (prons[w][0], prons[w][1])
        if k not in roots:
            roots[k] = dict(children={}, order=[], pen=[]); root_list.append(k)
        node = roots[k]
        for p in range(1, len(prons[w]) - 1):
            ph = prons[w][p]
            if ph not in node["children"]:
                node["children"][ph] = dict(children={}, order=[], pen=[], ci=ph); node["order"].append(ph)
            node = node["children"][ph]
        node["pen"].append(w)
    interior = []

    def number(node):
        for ph in node["order"]:
            ch = node["children"][ph]; ch["id"] = len(interior); interior.append(ch); number(ch)
    for k in root_list:
        number(roots[k])
    n_root, n_non = len(root_list), len(interior)
    allch = [roots[k] for k in root_list] + interior
    coff, child, poff, pen = [0], [], [0], []
    for node in allch:
        child += [n_root + node["children"][ph]["id"] for ph in node["order"]]; coff.append(len(child))
        pen += node["pen"]; poff.append(len(pen))
    A["root_ci"] = np.array([k[0] for k in root_list], np.int16); A["root_ci2"] = np.array([k[1] for k in root_list], np.int16)
    A["root_tmat"] = r.integers(0, n_tmat, n_root).astype(np.int16); A["root_ssid0"] = r.integers(0, n_sseq, n_root).astype(np.uint16)
    A["root_lc_ssid"] = r.integers(0, n_sseq, (n_root, n_ci)).astype(np.uint16)
    A["ch_child_off"] = np.array(coff, np.int32); A["ch_child"] = np.array(child + [0], np.int32)
    A["ch_pen_off"] = np.array(poff, np.int32); A["ch_pen_wid"] = np.array(pen + [0], np.int32)
    A["nr_ssid"] = r.integers(0, n_sseq, n_non + 1).astype(np.uint16); A["nr_tmat"] = r.integers(0, n_tmat, n_non + 1).astype(np.int16)
    A["nr_ci"] = np.array([n["ci"] for n in interior] + [0], np.int16)
    # single-phone words: the LM's first (incl. <s>, </s>), then the fillers outside the LM
    sp = [w for w in range(W) if len(prons[w]) == 1 and A["w_lmwid"][w] >= 0 and w not in (sil_w, noise_w)]
    n_lm1 = len(sp)
    sp += [w for w in (sil_w, noise_w) if len(prons[w]) == 1]
    A["sp_wid"] = np.array(sp, np.int32); A["sp_ssid0"] = r.integers(0, n_sseq, len(sp)).astype(np.uint16)
    A["sp_lc_ssid"] = r.integers(0, n_sseq, (len(sp), n_ci)).astype(np.uint16)
    A["sp_tmat"] = r.integers(0, n_tmat, len(sp)).astype(np.int16); A["sp_ci"] = np.array([prons[w][0] for w in sp], np.int16)
    # the filler loop enters every filler-range word with a channel except <sil> and <s>: here </s> and the noise word
    A["fill_sp"] = np.array([sp.index(w) for w in (finish_w, noise_w) if w in sp], np.int32)
    # LM (values as after lw / wip: a few 10^4 per n-gram, shifted right by 10 in the search)
    V = len(lm_words)
    A["ug_prob"] = (-r.integers(20, 90, V) * 1024 - r.integers(0, 1024, V)).astype(np.int32)
    A["ug_bowt"] = (-r.integers(0, 30, V) * 1024).astype(np.int32)
    fb, bgw, bgp, bgb, ft, tgw, tgp = [0], [], [], [], [], [], []
    for u in range(V):
        n = int(r.integers(0, min(V, 25))) if lm_order > 1 else 0
        for x in sorted(r.choice(V, n, replace=False)):
            bgw.append(int(x)); bgp.append(int(-r.integers(5, 60) * 1024)); bgb.append(int(-r.integers(0, 20) * 1024)); ft.append(len(tgw))
            m = int(r.integers(0, 20)) if lm_order > 2 and r.random() < 0.5 else 0
            for y in sorted(r.choice(V, min(m, V), replace=False)):
                tgw.append(int(y)); tgp.append(int(-r.integers(2, 50) * 1024))
        fb.append(len(bgw))
    ft.append(len(tgw))
    A["ug_firstbg"] = np.array(fb, np.int32); A["bg_wid"] = np.array(bgw + [0], np.int32); A["bg_prob"] = np.array(bgp + [0], np.int32)
    A["bg_bowt"] = np.array(bgb + [0], np.int32); A["bg_firsttg"] = np.array(ft, np.int32)
    A["tg_wid"] = np.array(tgw + [0], np.int32); A["tg_prob"] = np.array(tgp + [0], np.int32)
    for k, v in A.items():
        A[k] = np.ascontiguousarray(v)
        if hasattr(d, k):
            setattr(d, k, A[k].ctypes.data)
    d.n_ci, d.sil_ci, d.n_emit, d.n_sen, d.n_sseq, d.n_tmat = n_ci, 0, n_emit, n_sen, n_sseq, n_tmat
    d.n_words, d.start_wid, d.finish_wid, d.silence_wid = W, start_w, finish_w, sil_w
    d.n_rc_rows, d.n_root, d.n_nonroot, d.n_1ph, d.n_1ph_lm, d.n_fill = len(rows), n_root, n_non, len(sp), n_lm1, len(A["fill_sp"])
    d.lm_order, d.lm_n_ug, d.lm_n_bg, d.lm_n_tg, d.lm_zero = lm_order, V, len(bgw), len(tgw), -(1 << 28)
    d.beam, d.pbeam, d.wbeam, d.lpbeam, d.lponlybeam = beam, beam, int(beam * 0.7), int(beam * 0.8), int(beam * 0.6)
    d.fillpen, d.silpen, d.nwpen, d.pip, d.maxwpf, d.maxhmmpf = -45, -20, -7, -3, maxwpf, maxhmmpf
    if classes:             # a class-based LM: in-class weights of a third of the words; one word is "not in its class" (weight 1)
        cw = np.where(r.integers(0, 3, W) == 0, -r.integers(1, 40000, W), 0).astype(np.int32)
        cw[5] = 1
        A["w_lmcw"] = np.ascontiguousarray(cw)
        d.w_lmcw = A["w_lmcw"].ctypes.data
    if pl_window > 0:       # the phone loop's HMMs (frame-synchronous use needs only pl_window != 0: the host hands the scores over)
        A["ci_ssid"] = np.ascontiguousarray(r.integers(0, n_sseq, n_ci).astype(np.uint16))
        A["ci_tmat"] = np.ascontiguousarray(r.integers(0, n_tmat, n_ci).astype(np.int16))
        d.ci_ssid, d.ci_tmat = A["ci_ssid"].ctypes.data, A["ci_tmat"].ctypes.data
        d.pl_window, d.pl_beam, d.pl_pbeam, d.pl_pip = pl_window, 4 * beam, 2 * beam, -2
    return d, A


def make_lookahead(seed, n_frames, n_ci):
    """phone_loop_search_score per frame and CI phone: 0 for the loop's best phone, negative for the others, WORST_SCORE minus
    the loop's best score for a phone the loop has pruned (hmm_clear_scores)"""
    r = np.random.default_rng(seed)
    pl = -r.integers(0, 900, (n_frames, n_ci)).astype(np.int64)
    for t in range(n_frames):
        pl[t, r.integers(0, n_ci)] = 0
        if t % 5 == 2:
            pl[t, r.integers(0, n_ci)] = -0x20000000 + int(r.integers(0, 100000))
    return np.ascontiguousarray(pl.astype(np.int32))


def make_senscr(seed, n_frames, n_sen):
    """negated scores, normalised to best = 0 per frame: a few good senones drifting over time"""
    r = np.random.default_rng(seed)
    s = r.integers(40, 420, (n_frames, n_sen)).astype(np.int32)
    for t in range(n_frames):
        good = r.integers(0, n_sen, 12)
        s[t, good] = r.integers(0, 50, 12)
        s[t, good[0]] = 0
    return s.astype(np.int16)


class Oracle:
    def __init__(self, desc):
        self.L = C.CDLL(os.path.join(ROOT, "oracle", "libs3oracle.so"))
        self.L.s3o_psfwd_init.restype = P
        self.L.s3o_psfwd_array.restype = C.POINTER(I32)
        self.L.s3o_psfwd_valid.restype = C.POINTER(U8)
        self.desc = desc
        self.h = C.c_void_p(self.L.s3o_psfwd_init(C.byref(desc)))

    def __del__(self):
        self.L.s3o_psfwd_free(self.h)

    def start(self):
        self.L.s3o_psfwd_start(self.h)

    def reset(self):
        self.L.s3o_psfwd_reset(self.h)

    def sen_active(self, f):
        fl = np.zeros(self.desc.n_sen, np.uint8)
        self.L.s3o_psfwd_sen_active(self.h, I32(f), fl.ctypes.data_as(P))
        return fl

    def step(self, senscr, f, n_active):
        return self.L.s3o_psfwd_step(self.h, senscr.ctypes.data_as(P), I32(f), I32(n_active))

    def set_lookahead(self, pl):
        self.L.s3o_psfwd_set_lookahead(self.h, pl.ctypes.data_as(P) if pl is not None else None)

    def finish(self, cf):
        self.L.s3o_psfwd_finish(self.h, I32(cf))

    def table(self, cf):
        sc = np.zeros(16, np.int32)
        self.L.s3o_psfwd_scalars(self.h, sc.ctypes.data_as(P))
        n, ns = int(sc[1]), int(sc[2])
        arr = lambda k, m: np.ctypeslib.as_array(self.L.s3o_psfwd_array(self.h, I32(k)), (max(m, 1),))[:m].copy()
        out = dict(n_frame=int(sc[0]), bpidx=n, bss_head=ns, best_score=int(sc[3]), lp_best=int(sc[4]), renorm=int(sc[5]), st=sc[6:12].copy())
        for k, name in enumerate(("frame", "wid", "bp", "score", "s_idx", "real_wid")):
            out[name] = arr(k, n)
        out["bss"] = arr(6, ns); out["idx"] = arr(7, cf + 1)
        out["valid"] = np.ctypeslib.as_array(self.L.s3o_psfwd_valid(self.h), (max(n, 1),))[:n].copy()
        return out

    def hyp(self):
        sc = I32(0)
        self.L.s3o_psfwd_find_exit.restype = I32
        bp = self.L.s3o_psfwd_find_exit(self.h, I32(-1), C.byref(sc))
        if bp < 0:
            return None, []
        a = [np.zeros(4096, np.int32) for _ in range(6)]
        n = self.L.s3o_psfwd_backtrace(self.h, I32(bp), *[x.ctypes.data_as(P) for x in a], I32(4096))
        return sc.value, [tuple(int(x[i]) for x in a) for i in range(n)]        # (wid, sf, ef, ascr, lscr, bp)


class Table(C.Structure):
    _fields_ = [(n, I32) for n in ("status", "n_frame", "n_mark", "bpidx", "bss_head", "best_score", "last_phone_best_score", "renormalized")] + \
        [("st", I32 * 8)] + [(n, C.POINTER(I32)) for n in ("frame", "wid", "bp", "score", "s_idx", "real_wid")] + \
        [("valid", C.POINTER(U8)), ("bscore_stack", C.POINTER(I32)), ("bp_table_idx", C.POINTER(I32))]


class Device:
    def __init__(self, lib, desc, n_lanes=1, max_frames=256, bp_cap=0, bss_cap=0):
        self.lib, self.L, self.desc = lib, lib.load(), desc
        self.h = self.L.s3a_psfwd_init(C.byref(desc), n_lanes, max_frames, bp_cap, bss_cap)
        if not self.h:
            raise lib.S3AError(self.L.s3a_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_psfwd_free(self.h)

    def chk(self, rc):
        if rc < 0:
            raise self.lib.S3AError(self.L.s3a_last_error().decode())
        return rc

    def start(self, lane=0):
        self.chk(self.L.s3a_psfwd_start(self.h, lane))

    def reset(self, lane=0):
        self.chk(self.L.s3a_psfwd_reset(self.h, lane))

    def sen_active(self, f, lane=0):
        fl = np.zeros(self.desc.n_sen, np.uint8)
        self.chk(self.L.s3a_psfwd_sen_active(self.h, lane, f, fl.ctypes.data_as(P)))
        return fl

    def step(self, senscr, f, n_active, lane=0):
        return self.chk(self.L.s3a_psfwd_step(self.h, lane, senscr.ctypes.data_as(P), f, n_active))

    def set_lookahead(self, pl, lane=0):
        self.chk(self.L.s3a_psfwd_set_lookahead(self.h, lane, pl.ctypes.data_as(P) if pl is not None else None))

    def finish(self, cf, lane=0):
        self.chk(self.L.s3a_psfwd_finish(self.h, lane, cf))

    def table(self, cf, lane=0):
        t = Table()
        self.chk(self.L.s3a_psfwd_table(self.h, lane, C.byref(t)))
        n, ns = t.bpidx, t.bss_head
        arr = lambda p, m, dt=np.int32: np.ctypeslib.as_array(p, (max(m, 1),))[:m].copy()
        out = dict(n_frame=t.n_frame, bpidx=n, bss_head=ns, best_score=t.best_score, lp_best=t.last_phone_best_score,
                   renorm=t.renormalized, st=np.array(list(t.st)[1:7], np.int32))
        for name in ("frame", "wid", "bp", "score", "s_idx", "real_wid"):
            out[name] = arr(getattr(t, name), n)
        out["bss"] = arr(t.bscore_stack, ns); out["idx"] = arr(t.bp_table_idx, cf + 1); out["valid"] = arr(t.valid, n)
        return out

    def hyp(self, lane=0):
        sc = I32(0)
        seg = np.zeros((4096, 6), np.int32)
        n = self.chk(self.L.s3a_psfwd_hyp(self.h, lane, C.byref(sc), seg.ctypes.data_as(P), 4096))
        if n == 0:
            return None, []
        return sc.value, [tuple(int(x) for x in seg[i]) for i in range(n)]     # (wid, sf, ef, ascr, lscr, bp)


def diff_tables(a, b):
    bad = []
    for k in ("n_frame", "bpidx", "bss_head", "best_score", "lp_best", "renorm"):
        if a[k] != b[k]:
            bad.append(f"{k}: {a[k]} != {b[k]}")
    for k in ("st", "frame", "wid", "bp", "score", "s_idx", "real_wid", "valid", "bss", "idx"):
        if a[k].shape != b[k].shape:
            bad.append(f"{k}: shape {a[k].shape} != {b[k].shape}")
        elif not np.array_equal(a[k], b[k]):
            i = int(np.argwhere(a[k] != b[k])[0][0])
            bad.append(f"{k}[{i}]: {a[k][i]} != {b[k][i]}")
    return bad
