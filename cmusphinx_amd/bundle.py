"""A decoder rebuilt from a BUNDLE through the C ABI alone.

The sphinx3 side of the drop-in (integration/sphinx3/s3amd_tst.c, `S3A_EXPORT=file`) loads the models with the
reference's own kb_init and writes everything s3a_uttdec_init takes: flattened lextrees, senone sequences,
composite senones, transition matrices, the flattened trigram, the dictionary facts, beams, pruning limits and the
acoustic model's file names.  This module reads that file and builds the decoder (LogMath, MgauModel, ComSen,
Tmat, LexSearch, Lm3g, UttDec of cmusphinx_amd.lib), so that measurements, multi-GPU drivers and tests drive
libcmusphinx_amd.so directly: the host program is needed once, for loading, never inside a timed region.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib

CFG_INTS = ["wbeam_vh", "bghist", "maxwpf", "maxhistpf", "wordend_beam", "n_lextree", "epl", "hmmbeam", "pbeam", "wbeam",
            "ptranskip", "maxhmmpf", "ds", "cond_ds", "maxcdsenpf", "hypsegscore_unscale"]
CFG_DBL = ["logbase", "varfloor", "mixwfloor", "ci_pbeam", "tighten_factor", "lw", "wip", "bestpathlw"]
CFG_DAG = ["min_endfr", "maxedge", "maxlmop", "maxlpf", "wip_logs3", "bestpath"]
TREE = {11: "ssid", 12: "tmatid", 13: "composite", 14: "wid", 15: "prob", 16: "child_off", 17: "child", 18: "lc",
        19: "lcroot_off", 20: "lcroot", 21: "root", 22: "ci"}
STATIC = {2: "tp", 3: "sseq", 4: "comsseq", 5: "comstate_off", 6: "comstate", 7: "comwt", 8: "cd2cisen",
          31: "ug_prob", 32: "ug_bowt", 33: "ug_firstbg", 34: "bg_wid", 35: "bg_prob", 36: "bg_bowt", 37: "bg_firsttg",
          38: "tg_wid", 39: "tg_prob", 41: "lwid", 42: "is_filler", 43: "fillpen", 44: "last_ci", 46: "basewid", 57: "sen2cimap"}


def _cstr(cells):
    return cells.tobytes().split(b"\0")[0].decode()


def read(path):
    raw = np.fromfile(path, dtype="<i4")
    pos, b, trees = 0, {}, []
    while pos < len(raw):
        if pos + 2 > len(raw):
            raise ValueError(f"{path}: truncated bundle (record header at word {pos})")
        tag, n = int(raw[pos]), int(raw[pos + 1])
        if n < 0 or pos + 2 + n > len(raw):
            raise ValueError(f"{path}: truncated bundle (record {tag} wants {n} words at word {pos})")
        d = raw[pos + 2: pos + 2 + n].copy()
        pos += 2 + n
        if tag == 1:
            for k, v in zip(("n_tree", "n_emit", "n_tmat", "n_sseq", "n_comsseq", "n_comstate", "n_sen", "n_ci_sen", "n_ci", "veclen"), d):
                b[k] = int(v)
        elif tag == 10:
            trees.append(dict(n_node=int(d[0]), n_lc=int(d[1]), n_root=int(d[2]), type=int(d[3]), lc=np.zeros(0, np.int32),
                              lcroot_off=np.zeros(1, np.int32), lcroot=np.zeros(0, np.int32)))
        elif tag in TREE:
            trees[-1][TREE[tag]] = d
        elif tag in STATIC:
            b[STATIC[tag]] = d
        elif tag == 30:
            b["n_ug"], b["n_bg"], b["n_tg"] = (int(x) for x in d)
        elif tag == 40:
            for k, v in zip(("n_word", "startwid", "finishwid", "silwid", "start_lwid", "finish_lwid", "sil_ci"), d):
                b[k] = int(v)
        elif tag == 45:
            b["words"] = [w.decode() for w in d.tobytes().split(b"\0")[: b["n_word"]]]
        elif tag == 50:
            for k, v in zip(CFG_INTS, d):
                b[k] = int(v)
        elif tag == 51:
            for k, v in zip(CFG_DBL, d.view("<f8")):
                b[k] = float(v)
        elif tag == 55:
            for k, v in zip(CFG_DAG, d):
                b[k] = int(v)
        elif tag == 56:
            b["pheurtype"], b["pl_beam"], b["pl_window"] = (int(x) for x in d)
        elif tag in (52, 53, 54):
            b[{52: "mean", 53: "var", 54: "mixw"}[tag]] = _cstr(d)
    b["trees"] = trees
    return b


class Decoder:
    """bundle -> s3a_uttdec_t with n_lanes lanes (and what formatting a hypothesis needs)"""

    def __init__(self, bundle, n_lanes, precision=lib.GMM_EXACT, vh_cap=0, cand_cap=0, max_frames=15000, bestpath=False,
                 keep_tables=True, link_cap=0, pair_cap=0, bestpathlw=None, opts=None):
        b = self.b = read(bundle) if isinstance(bundle, str) else bundle
        ne = b["n_emit"]
        self.logmath = lib.LogMath(b["logbase"])
        self.g = lib.MgauModel.init(b["mean"], b["var"], b["mixw"], self.logmath, varfloor=b["varfloor"], mixwfloor=b["mixwfloor"])
        if precision != lib.GMM_EXACT:
            self.g.set_precision(precision)
        self.comsen = lib.ComSen(b["comstate_off"], b["comstate"], b["comwt"])
        self.tmat = lib.Tmat.init_logs3(b["tp"].reshape(b["n_tmat"], ne, ne + 1))
        self.proto = lib.LexSearch(b["trees"], self.tmat, b["sseq"], b["comsseq"], b["comstate_off"], b["comstate"], b["n_sen"],
                                   stream=self.g.stream())
        b2 = dict(b, wbeam=b["wbeam_vh"])
        self.lm = lib.Lm3g(b2)
        self._keep = []
        cfg = self.cfg = lib.wordlevel_cfg(b2, [t["type"] for t in b["trees"]], self._keep, wordend=b["wordend_beam"])
        cfg.sil_ci = b["sil_ci"]
        cfg.hmmbeam, cfg.pbeam, cfg.wbeam, cfg.ptranskip, cfg.maxhmmpf = b["hmmbeam"], b["pbeam"], b["wbeam"], b["ptranskip"], b["maxhmmpf"]
        self.ud = lib.UttDec(self.proto, self.g, b["cd2cisen"], b["n_ci_sen"], self.comsen, self.lm, cfg, n_lanes, ds=b["ds"],
                             cond_ds=b["cond_ds"], ci_pbeam=b["ci_pbeam"], tighten_factor=b["tighten_factor"],
                             max_cd=b["maxcdsenpf"], max_frames=max_frames, vh_cap=vh_cap, cand_cap=cand_cap, opts=opts)
        self.n_lanes = n_lanes
        if b.get("pheurtype", 0) > 0:
            # -pheurtype 1..3: the phoneme look-ahead inside the engine
            n = b["n_ci_sen"] + 1
            self.ud.enable_pheur(b["pheurtype"], b["pl_beam"], b["pl_window"],
                                 [np.asarray(t["ci"], np.int32).astype(np.uint8) for t in b["trees"]],
                                 np.asarray(b["sen2cimap"], np.int32)[:n].astype(np.int16), b["n_ci"])
        self.dag_cfg = None
        if bestpath:
            # the second pass behind every decode: lattice + best path on the device
            self.dag_cfg = lib.dag_cfg(b, self._keep, bestpathlw=bestpathlw)
            self.ud.enable_bestpath(self.dag_cfg, link_cap, pair_cap, keep_tables)
        self.veclen = b["veclen"]
        ws = [w.encode() for w in b["words"]]
        self._wordstr = (C.c_char_p * len(ws))(*ws)
        self._basewid = np.ascontiguousarray(b["basewid"], np.int32)
        self._isfill = np.ascontiguousarray(b["is_filler"], np.uint8)

    def decode(self, feats):
        """feats: list (<= n_lanes) of [nfr, veclen] float32 -> device milliseconds"""
        return self.ud.decode(feats)

    def decode_queue(self, feats):
        """any number of utterances, lanes refilled as they finish -> device milliseconds"""
        return self.ud.decode_queue(feats)

    def queue_hyp(self, utt, uttid="", utt_index=0):
        return self.ud.queue_hyp(utt, uttid, utt_index)

    def queue_bestpath_hyp(self, utt, uttid="", utt_index=0):
        return self.ud.queue_bestpath_hyp(utt, uttid, utt_index)

    def hyp(self, lane, uttid="", utt_index=0):
        return self.ud.hyp(lane, uttid, utt_index)

    def hyp_var(self, lane, uttid="", utt_index=0):
        return self.ud.hyp_var(lane, uttid, utt_index)

    def bestpath_hyp(self, lane, uttid="", utt_index=0):
        """the second pass's hypothesis: (HypHeader, words)"""
        return self.ud.bestpath_hyp(lane, uttid, utt_index)

    def format_var(self, hdr, words):
        """the same from a header + words pair (no word limit)"""
        L = lib.load()
        words = np.ascontiguousarray(words, np.int32)
        cap = (1 << 16) + 64 * len(words)
        m, s = C.create_string_buffer(cap), C.create_string_buffer(cap)
        lib.check(L.s3a_hyp_format_var(C.byref(hdr), lib._p(words), self._wordstr, lib._p(self._basewid), lib._p(self._isfill),
                                       self.b["startwid"], self.b["finishwid"], np.float32(self.b["lw"]), int(self.b["wip"]),
                                       self.b["hypsegscore_unscale"], m, len(m), s, len(s)), L)
        return m.value.decode(), s.value.decode()

    def format(self, rec):
        """-> (the utterance's -hyp line, its -hypseg line)"""
        L = lib.load()
        m, s = C.create_string_buffer(1 << 16), C.create_string_buffer(1 << 16)
        lib.check(L.s3a_hyp_format(C.byref(rec), self._wordstr, lib._p(self._basewid), lib._p(self._isfill), self.b["startwid"],
                                   self.b["finishwid"], np.float32(self.b["lw"]), int(self.b["wip"]),
                                   self.b["hypsegscore_unscale"], m, len(m), s, len(s)), L)
        return m.value.decode(), s.value.decode()
