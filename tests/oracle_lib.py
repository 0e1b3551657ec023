"""ctypes binding of the CPU oracle (oracle/libs3oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

LOGPROB_ZERO = -939524096  # (int32)0xc8000000
NO_BSTIDX = -1
NOT_UPDATED = -100


class LogMath(C.Structure):
    _fields_ = [("base", C.c_double), ("log_of_base", C.c_double), ("log10_of_base", C.c_double),
                ("inv_log_of_base", C.c_double), ("inv_log10_of_base", C.c_double),
                ("shift", C.c_int), ("zero", C.c_int32), ("width", C.c_int),
                ("table_size", C.c_uint32), ("table", C.POINTER(C.c_uint32))]


class Mgau(C.Structure):
    _fields_ = [("n_mgau", C.c_int32), ("max_comp", C.c_int32), ("veclen", C.c_int32),
                ("n_comp", C.POINTER(C.c_int32)), ("mean", C.POINTER(C.c_float)),
                ("var", C.POINTER(C.c_float)), ("lrd", C.POINTER(C.c_float)),
                ("mixw", C.POINTER(C.c_int32)), ("bstidx", C.POINTER(C.c_int32)),
                ("bstscr", C.POINTER(C.c_int32)), ("updatetime", C.POINTER(C.c_int32)),
                ("distfloor", C.c_double), ("lm", C.POINTER(LogMath)),
                ("frm_sen_eval", C.c_int32), ("frm_gau_eval", C.c_int32),
                ("frm_ci_sen_eval", C.c_int32), ("frm_ci_gau_eval", C.c_int32)]


class FastGmm(C.Structure):
    _fields_ = [("ds_ratio", C.c_int32), ("cond_ds", C.c_int32), ("ci_pbeam", C.c_int32),
                ("max_cd", C.c_int32), ("tighten_factor", C.c_float),
                ("dyn_ci_pbeam", C.c_int32), ("skip_count", C.c_int32)]


class Hmm(C.Structure):
    _fields_ = [("score", C.c_int32 * 5), ("history", C.c_int64 * 5),
                ("out_score", C.c_int32), ("out_history", C.c_int64),
                ("ssid", C.c_int32), ("mpx_ssid", C.c_int32 * 5),
                ("bestscore", C.c_int32), ("tmatid", C.c_int32), ("frame", C.c_int32),
                ("mpx", C.c_uint8)]


class HmmCtx(C.Structure):
    _fields_ = [("n_emit_state", C.c_int32), ("tp", C.POINTER(C.c_int32)),
                ("senscore", C.POINTER(C.c_int32)), ("sseq", C.POINTER(C.c_int16))]


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def build():
    """(Re)build oracle/libs3oracle.so -- and oracle/_ref when /root/reference exists."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libs3oracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    L = C.CDLL(path)
    L.s3o_logmath_init.restype = C.POINTER(LogMath)
    L.s3o_logmath_init.argtypes = [C.c_double, C.c_int, C.c_int]
    L.s3o_logmath_free.argtypes = [C.POINTER(LogMath)]
    for fn in ("s3o_logmath_add",):
        getattr(L, fn).restype = C.c_int
        getattr(L, fn).argtypes = [C.POINTER(LogMath), C.c_int, C.c_int]
    L.s3o_logmath_log.restype = C.c_int
    L.s3o_logmath_log.argtypes = [C.POINTER(LogMath), C.c_double]
    L.s3o_logmath_exp.restype = C.c_double
    L.s3o_logmath_exp.argtypes = [C.POINTER(LogMath), C.c_int]
    L.s3o_logmath_log_to_ln.restype = C.c_double
    L.s3o_logmath_log_to_ln.argtypes = [C.POINTER(LogMath), C.c_int]
    L.s3o_logmath_ln_to_log.restype = C.c_int
    L.s3o_logmath_ln_to_log.argtypes = [C.POINTER(LogMath), C.c_double]
    L.s3o_logmath_log10_to_log.restype = C.c_int
    L.s3o_logmath_log10_to_log.argtypes = [C.POINTER(LogMath), C.c_double]
    L.s3o_logs3.restype = C.c_int32
    L.s3o_logs3.argtypes = [C.POINTER(LogMath), C.c_double]
    L.s3o_mgau_init.restype = C.POINTER(Mgau)
    L.s3o_mgau_init.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_int32] * 3 + \
        [C.c_double, C.c_double, C.c_int, C.POINTER(LogMath)]
    L.s3o_mgau_free.argtypes = [C.POINTER(Mgau)]
    L.s3o_mgau_eval.restype = C.c_int32
    L.s3o_mgau_eval.argtypes = [C.POINTER(Mgau), C.c_int32, C.POINTER(C.c_int32),
                                C.POINTER(C.c_float), C.c_int32, C.c_int32]
    L.s3o_mgau_reset_state.argtypes = [C.POINTER(Mgau)]
    L.s3o_approx_cont_mgau_ci_eval.argtypes = [C.POINTER(Mgau), C.POINTER(C.c_int16), C.c_int32,
                                               C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32), C.c_int32]
    L.s3o_approx_cont_mgau_frame_eval.restype = C.c_int32
    L.s3o_approx_cont_mgau_frame_eval.argtypes = [
        C.POINTER(Mgau), C.POINTER(FastGmm), C.POINTER(C.c_int16), C.c_int32,
        C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
        C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]
    L.s3o_dict2pid_comsenscr.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int16),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32)]
    L.s3o_tmat_logs3.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_double,
                                 C.POINTER(LogMath), C.POINTER(C.c_int32)]
    L.s3o_hmm_init.argtypes = [C.POINTER(HmmCtx), C.POINTER(Hmm), C.c_int, C.c_int32, C.c_int32]
    L.s3o_hmm_clear.argtypes = [C.POINTER(HmmCtx), C.POINTER(Hmm)]
    L.s3o_hmm_enter.argtypes = [C.POINTER(Hmm), C.c_int32, C.c_int64, C.c_int32]
    L.s3o_hmm_normalize.argtypes = [C.POINTER(HmmCtx), C.POINTER(Hmm), C.c_int32]
    L.s3o_hmm_vit_eval.restype = C.c_int32
    L.s3o_hmm_vit_eval.argtypes = [C.POINTER(HmmCtx), C.POINTER(Hmm)]
    _LIB = L
    return L


class OracleLogMath:
    def __init__(self, base=1.0003, shift=0, use_table=1):
        self.L = lib()
        self.p = self.L.s3o_logmath_init(base, shift, use_table)
        if not self.p:
            raise ValueError("bad base")

    def __del__(self):
        try:
            self.L.s3o_logmath_free(self.p)
        except Exception:
            pass

    @property
    def table(self):
        n = self.p.contents.table_size
        return np.ctypeslib.as_array(self.p.contents.table, shape=(n,)).copy()

    @property
    def zero(self):
        return self.p.contents.zero

    @property
    def width(self):
        return self.p.contents.width

    def add(self, x, y):
        return self.L.s3o_logmath_add(self.p, int(x), int(y))

    def log(self, p):
        return self.L.s3o_logmath_log(self.p, float(p))

    def logs3(self, p):
        return self.L.s3o_logs3(self.p, float(p))


class OracleMgau:
    """mgau_init on raw arrays + mgau_eval / approx_cont_mgau_frame_eval."""

    def __init__(self, mean, var, mixw, lm: OracleLogMath, varfloor=1e-4, mixwfloor=1e-7):
        self.L = lib()
        self.lm = lm
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        var = np.ascontiguousarray(var, dtype=np.float32)
        mixw = np.ascontiguousarray(mixw, dtype=np.float32).reshape(mean.shape[0], mean.shape[1])
        S, Cn, D = mean.shape
        self.S, self.C, self.D = S, Cn, D
        self.p = self.L.s3o_mgau_init(_ptr(mean, C.c_float), _ptr(var, C.c_float),
                                      _ptr(mixw, C.c_float), S, Cn, D,
                                      varfloor, mixwfloor, 1, lm.p)

    def __del__(self):
        try:
            self.L.s3o_mgau_free(self.p)
        except Exception:
            pass

    def arr(self, name, shape, copy=True):
        a = np.ctypeslib.as_array(getattr(self.p.contents, name), shape=shape)
        return a.copy() if copy else a

    @property
    def n_comp(self):
        return self.arr("n_comp", (self.S,))

    @property
    def mean(self):
        return self.arr("mean", (self.S, self.C, self.D))

    @property
    def prec(self):
        return self.arr("var", (self.S, self.C, self.D))

    @property
    def lrd(self):
        return self.arr("lrd", (self.S, self.C))

    @property
    def mixw(self):
        return self.arr("mixw", (self.S, self.C))

    @property
    def distfloor(self):
        return self.p.contents.distfloor

    def reset_state(self):
        self.L.s3o_mgau_reset_state(self.p)

    def eval(self, m, x, fr=0, update=1, active=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        ap = None
        if active is not None:
            act = np.ascontiguousarray(list(active) + [-1], dtype=np.int32)
            ap = _ptr(act, C.c_int32)
        return self.L.s3o_mgau_eval(self.p, int(m), ap, _ptr(x, C.c_float), int(fr), int(update))

    def score_all(self, feats):
        """mgau_eval(g, s, NULL, x, t, 1) for every frame and senone -> int32 [T][S]."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        out = np.empty((T, self.S), dtype=np.int32)
        for t in range(T):
            xp = _ptr(feats[t], C.c_float)
            for s in range(self.S):
                out[t, s] = self.L.s3o_mgau_eval(self.p, s, None, xp, t, 1)
        return out

    def frame_eval_seq(self, feats, cd2cisen, n_ci_sen, active=None, ci_pbeam=None,
                       ds=1, tighten=0.5, max_cd=100000):
        """ci_eval + frame_eval over a frame sequence (srch.c:739-822 with -pl_window 1).

        Returns dict with senscr[T][S], best[T], sen_active_out, bstidx, updatetime, counts.
        """
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        cd2cisen = np.ascontiguousarray(cd2cisen, dtype=np.int16)
        T, S = feats.shape[0], self.S
        if ci_pbeam is None:
            ci_pbeam = self.lm.logs3(1e-80)
        fg = FastGmm(ds, 0, int(ci_pbeam), int(max_cd), float(tighten), 0, 0)
        senscr = np.zeros((T, S), np.int32)
        act_out = np.zeros((T, S), np.uint8)
        bidx = np.zeros((T, S), np.int32)
        upd = np.zeros((T, S), np.int32)
        best = np.zeros(T, np.int32)
        cibest = np.zeros(T, np.int32)
        counts = np.zeros((T, 4), np.int32)
        ci = np.zeros(max(n_ci_sen, 1), np.int32)
        cur = np.zeros(S, np.int32)
        sa = np.zeros(S, np.uint8)
        rsa = np.zeros(S, np.uint8)
        b = C.c_int32(0)
        self.reset_state()
        for t in range(T):
            xp = _ptr(feats[t], C.c_float)
            self.L.s3o_approx_cont_mgau_ci_eval(self.p, _ptr(cd2cisen, C.c_int16), S, xp,
                                                _ptr(ci, C.c_int32), C.byref(b), t)
            cibest[t] = b.value
            counts[t, 2] = self.p.contents.frm_ci_sen_eval
            counts[t, 3] = self.p.contents.frm_ci_gau_eval
            sa[:] = 1 if active is None else active[t]
            best[t] = self.L.s3o_approx_cont_mgau_frame_eval(
                self.p, C.byref(fg), _ptr(cd2cisen, C.c_int16), n_ci_sen,
                _ptr(sa, C.c_uint8), _ptr(rsa, C.c_uint8), _ptr(cur, C.c_int32), xp, t,
                _ptr(ci, C.c_int32))
            counts[t, 0] = self.p.contents.frm_sen_eval
            counts[t, 1] = self.p.contents.frm_gau_eval
            senscr[t] = cur
            act_out[t] = sa
            bidx[t] = self.arr("bstidx", (S,))
            upd[t] = self.arr("updatetime", (S,))
        return dict(senscr=senscr, best=best, ci_best=cibest, sen_active_out=act_out,
                    bstidx=bidx, updatetime=upd, counts=counts,
                    beams=np.array([fg.ci_pbeam, fg.dyn_ci_pbeam], np.int32))


class OracleFrameScorer:
    """One frame at a time through s3o_approx_cont_mgau_ci_eval + _frame_eval (what the slots
    gmm_compute_lv1 / lv2 do with -pl_window 1), keeping ascr_t.senscr between frames."""

    def __init__(self, g: "OracleMgau", cd2cisen, n_ci_sen, ci_pbeam, ds=1, tighten=0.5, max_cd=100000):
        self.g = g
        self.cd2cisen = np.ascontiguousarray(cd2cisen, np.int16)
        self.n_ci = int(n_ci_sen)
        self.fg = FastGmm(ds, 0, int(ci_pbeam), int(max_cd), float(tighten), 0, 0)
        self.senscr = np.zeros(g.S, np.int32)
        self.ci = np.zeros(max(self.n_ci, 1), np.int32)
        self.rsa = np.zeros(g.S, np.uint8)
        g.reset_state()

    def step(self, feat, t, sen_active):
        """sen_active (uint8[S]) is updated in place (CI senones forced on), like the reference.
        Returns (best, n_cd_sen, n_cd_gau, n_ci_sen, n_ci_gau, ci_best)."""
        g, L = self.g, self.g.L
        x = np.ascontiguousarray(feat, np.float32)
        xp = _ptr(x, C.c_float)
        b = C.c_int32(0)
        L.s3o_approx_cont_mgau_ci_eval(g.p, _ptr(self.cd2cisen, C.c_int16), g.S, xp, _ptr(self.ci, C.c_int32),
                                       C.byref(b), t)
        cin, cig = g.p.contents.frm_ci_sen_eval, g.p.contents.frm_ci_gau_eval
        best = L.s3o_approx_cont_mgau_frame_eval(g.p, C.byref(self.fg), _ptr(self.cd2cisen, C.c_int16), self.n_ci,
                                                 _ptr(sen_active, C.c_uint8), _ptr(self.rsa, C.c_uint8),
                                                 _ptr(self.senscr, C.c_int32), xp, t, _ptr(self.ci, C.c_int32))
        return best, g.p.contents.frm_sen_eval, g.p.contents.frm_gau_eval, cin, cig, b.value


def tmat_logs3(tp, lm: OracleLogMath, tpfloor=1e-4):
    tp = np.ascontiguousarray(tp, dtype=np.float32)
    out = np.zeros(tp.shape, np.int32)
    lib().s3o_tmat_logs3(_ptr(tp, C.c_float), tp.shape[0], tp.shape[1], tpfloor, lm.p,
                         _ptr(out, C.c_int32))
    return out


def comsenscr(comstate_off, comstate, comwt, senscr):
    comstate_off = np.ascontiguousarray(comstate_off, np.int32)
    comstate = np.ascontiguousarray(comstate, np.int16)
    comwt = np.ascontiguousarray(comwt, np.int32)
    senscr = np.ascontiguousarray(senscr, np.int32)
    n = len(comwt)
    out = np.zeros(n, np.int32)
    lib().s3o_dict2pid_comsenscr(n, _ptr(comstate_off, C.c_int32), _ptr(comstate, C.c_int16),
                                 _ptr(comwt, C.c_int32), _ptr(senscr, C.c_int32),
                                 _ptr(out, C.c_int32))
    return out


def hmm_run(n_emit, tp, sseq, senscr, spec, enter):
    """Run s3o_hmm_vit_eval over T frames for NHMM HMMs (same protocol as ref_dump hmm).

    Returns (state int32 [T][N][12], hist int64 [T][N][6], ret int32 [T][N]).
    """
    L = lib()
    tp = np.ascontiguousarray(tp, np.int32)
    sseq = np.ascontiguousarray(sseq, np.int16)
    senscr = np.ascontiguousarray(senscr, np.int32)
    spec = np.ascontiguousarray(spec, np.int32)
    enter = np.ascontiguousarray(enter, np.int32)
    T, nsen = senscr.shape
    nh = spec.shape[0]
    ctx = HmmCtx(n_emit, _ptr(tp, C.c_int32), _ptr(senscr, C.c_int32), _ptr(sseq, C.c_int16))
    hs = (Hmm * nh)()
    for i in range(nh):
        L.s3o_hmm_init(C.byref(ctx), C.byref(hs[i]), int(spec[i, 0]), int(spec[i, 1]), int(spec[i, 2]))
    state = np.zeros((T, nh, 12), np.int32)
    hist = np.zeros((T, nh, 6), np.int64)
    ret = np.zeros((T, nh), np.int32)
    for t in range(T):
        ctx.senscore = _ptr(senscr[t], C.c_int32)
        for i in range(nh):
            if enter[t, i, 0] != -2147483648:
                L.s3o_hmm_enter(C.byref(hs[i]), int(enter[t, i, 0]), int(enter[t, i, 1]), t)
            ret[t, i] = L.s3o_hmm_vit_eval(C.byref(ctx), C.byref(hs[i]))
            h = hs[i]
            for j in range(5):
                state[t, i, j] = h.score[j] if j < n_emit else 0
                hist[t, i, j] = h.history[j] if j < n_emit else 0
                state[t, i, 7 + j] = h.mpx_ssid[j] if (h.mpx and j < n_emit) else -2
            state[t, i, 5] = h.out_score
            state[t, i, 6] = h.bestscore
            hist[t, i, 5] = h.out_history
    return state, hist, ret


# ---------------------------------------------------------------------------
# lextree (oracle/s3o_lextree.c)
# ---------------------------------------------------------------------------
class OracleLexSearch:
    """The trees of one decoder driven through s3o_lextree_* (sequential reference order)."""

    def __init__(self, tr):
        L = lib()
        self.L = L
        self.tr = tr
        self.keep = []
        c = lambda a, dt: np.ascontiguousarray(a, dt)
        self.tp = c(tr["tp"], np.int32); self.sseq = c(tr["sseq"], np.int16)
        self.comsseq = c(tr["comsseq"], np.int16)
        self.comstate_off = c(tr["comstate_off"], np.int32); self.comstate = c(tr["comstate"], np.int16)
        L.s3o_lextree_init.restype = C.c_void_p
        L.s3o_lextree_init.argtypes = [C.c_int32] + [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                                                      C.c_void_p, C.c_void_p]
        L.s3o_lextree_enter.argtypes = [C.c_void_p] + [C.c_int32] * 5
        L.s3o_lextree_active_swap.argtypes = [C.c_void_p]
        L.s3o_lextree_hmm_eval.restype = C.c_int32
        L.s3o_lextree_hmm_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.s3o_lextree_hmm_propagate_non_leaves.argtypes = [C.c_void_p] + [C.c_int32] * 4
        L.s3o_lextree_hmm_propagate_leaves.restype = C.c_int32
        L.s3o_lextree_hmm_propagate_leaves.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.s3o_lextree_ssid_active.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.s3o_sseq2sen_active.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.s3o_comsseq2sen_active.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.s3o_lextree_hmm_histbin.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
        L.s3o_lextree_utt_end.argtypes = [C.c_void_p]
        self.lt = []
        for t in tr["trees"]:
            arrs = dict(ssid=c(t["ssid"], np.int32), tmatid=c(t["tmatid"], np.int32),
                        composite=c(t["composite"], np.uint8), wid=c(t["wid"], np.int32),
                        prob=c(t["prob"], np.int32), child_off=c(t["child_off"], np.int32),
                        child=c(t["child"], np.int32), lc=c(t["lc"], np.int16),
                        lcroot_off=c(t["lcroot_off"], np.int32), lcroot=c(t["lcroot"], np.int32),
                        root=c(t["root"], np.int32))
            self.keep.append(arrs)
            p = lambda k: arrs[k].ctypes.data_as(C.c_void_p)
            self.lt.append(L.s3o_lextree_init(t["n_node"], p("ssid"), p("tmatid"), p("composite"), p("wid"),
                                              p("prob"), p("child_off"), p("child"), t["n_lc"], p("lc"),
                                              p("lcroot_off"), p("lcroot"), t["n_root"], p("root"),
                                              tr["n_emit"], self.tp.ctypes.data_as(C.c_void_p),
                                              self.sseq.ctypes.data_as(C.c_void_p),
                                              self.comsseq.ctypes.data_as(C.c_void_p)))
        self.T = len(self.lt)

    def _lt(self, t):
        return C.cast(self.lt[t], C.POINTER(LexTree)).contents

    def enter(self, t, lc, scr, hist, cf, thresh):
        for c in range(len(lc)):
            self.L.s3o_lextree_enter(self.lt[t], int(lc[c]), cf, int(scr[c]), int(hist[c]), thresh)

    def swap(self):
        for h in self.lt:
            self.L.s3o_lextree_active_swap(h)

    def hmm_eval(self, senscr, comsen, frm):
        s = np.ascontiguousarray(senscr, np.int32); cs = np.ascontiguousarray(comsen, np.int32)
        best, wbest, nact = [], [], []
        for t in range(self.T):
            self.L.s3o_lextree_hmm_eval(self.lt[t], _p2(s), _p2(cs), frm)
            lt = self._lt(t)
            best.append(lt.best); wbest.append(lt.wbest); nact.append(lt.n_active)
        return np.array(best, np.int32), np.array(wbest, np.int32), np.array(nact, np.int32)

    def propagate(self, cf, th, pth, wth):
        for h in self.lt:
            self.L.s3o_lextree_hmm_propagate_non_leaves(h, cf, th, pth, wth)

    def leaves(self, wth):
        out = []
        for t in range(self.T):
            n_node = self.tr["trees"][t]["n_node"]
            w = np.zeros(n_node, np.int32); s = np.zeros(n_node, np.int32); h = np.zeros(n_node, np.int32)
            n = self.L.s3o_lextree_hmm_propagate_leaves(self.lt[t], wth, _p2(w), _p2(s), _p2(h), n_node)
            assert n >= 0
            out.append((w[:n], s[:n], h[:n]))
        return out

    def active(self, t, which):
        lt = self._lt(t)
        n = lt.n_next_active if which else lt.n_active
        ptr = lt.next_active if which else lt.active
        if n == 0:
            return np.zeros(0, np.int32)
        return np.frombuffer(C.string_at(ptr, 4 * n), dtype=np.int32).copy()

    def state(self, t):
        """[n_node][10]: score0..2, hist0..2, out_score, out_hist, bestscore, frame (one bulk copy)."""
        lt = self._lt(t)
        n = lt.n_node
        raw = np.frombuffer(C.string_at(lt.hmm, n * C.sizeof(Hmm)), dtype=_HMM_DT)
        out = np.empty((n, 10), np.int32)
        out[:, 0:3] = raw["score"][:, :3]
        out[:, 3:6] = raw["history"][:, :3]
        out[:, 6] = raw["out_score"]; out[:, 7] = raw["out_history"]
        out[:, 8] = raw["bestscore"]; out[:, 9] = raw["frame"]
        return out

    def utt_end(self):
        for h in self.lt:
            self.L.s3o_lextree_utt_end(h)

    def histbin(self, t, bestscr, bins, bw):
        """lextree_hmm_histbin on tree t: bins (int32) updated in place, active list reordered."""
        assert bins.dtype == np.int32
        self.L.s3o_lextree_hmm_histbin(self.lt[t], int(bestscr), _p2(bins), len(bins), int(bw))

    def sen_active(self):
        tr = self.tr
        ssid = np.zeros(tr["n_sseq"], np.uint8); com = np.zeros(max(tr["n_comsseq"], 1), np.uint8)
        sen = np.zeros(tr["n_sen"], np.uint8)
        for h in self.lt:
            self.L.s3o_lextree_ssid_active(h, _p2(ssid), _p2(com))
        self.L.s3o_sseq2sen_active(_p2(self.sseq), tr["n_sseq"], tr["n_emit"], _p2(ssid), _p2(sen))
        self.L.s3o_comsseq2sen_active(_p2(self.comsseq), tr["n_comsseq"], tr["n_emit"], _p2(self.comstate_off),
                                      _p2(self.comstate), _p2(com), _p2(sen))
        return sen


class LexTree(C.Structure):
    _fields_ = [("n_node", C.c_int32),
                ("ssid", C.c_void_p), ("tmatid", C.c_void_p), ("wid", C.c_void_p), ("prob", C.c_void_p),
                ("composite", C.c_void_p), ("child_off", C.c_void_p), ("child", C.c_void_p),
                ("n_lc", C.c_int32), ("lc", C.c_void_p), ("lcroot_off", C.c_void_p), ("lcroot", C.c_void_p),
                ("n_root", C.c_int32), ("root", C.c_void_p),
                ("ctx", HmmCtx), ("comctx", HmmCtx),
                ("hmm", C.POINTER(Hmm)), ("active", C.POINTER(C.c_int32)), ("next_active", C.POINTER(C.c_int32)),
                ("n_active", C.c_int32), ("n_next_active", C.c_int32), ("best", C.c_int32), ("wbest", C.c_int32)]


def _p2(a):
    return a.ctypes.data_as(C.c_void_p)


# numpy view of s3o_hmm_t (must mirror the ctypes Hmm structure, incl. alignment padding)
_HMM_DT = np.dtype({"names": ["score", "history", "out_score", "out_history", "ssid", "mpx_ssid",
                              "bestscore", "tmatid", "frame", "mpx"],
                    "formats": [("<i4", 5), ("<i8", 5), "<i4", "<i8", "<i4", ("<i4", 5), "<i4", "<i4", "<i4", "u1"],
                    "offsets": [Hmm.score.offset, Hmm.history.offset, Hmm.out_score.offset, Hmm.out_history.offset,
                                Hmm.ssid.offset, Hmm.mpx_ssid.offset, Hmm.bestscore.offset, Hmm.tmatid.offset,
                                Hmm.frame.offset, Hmm.mpx.offset],
                    "itemsize": C.sizeof(Hmm)})


class Ms(C.Structure):
    _fields_ = [("n_mgau", C.c_int32), ("n_feat", C.c_int32), ("n_density", C.c_int32), ("n_sen", C.c_int32),
                ("topn", C.c_int32), ("veclen", C.c_int32), ("featlen", C.POINTER(C.c_int32)),
                ("featoff", C.POINTER(C.c_int32)), ("mean", C.POINTER(C.c_float)), ("var", C.POINTER(C.c_float)),
                ("det", C.POINTER(C.c_float)), ("pdf", C.POINTER(C.c_int32)), ("mgau", C.POINTER(C.c_int32)),
                ("min_density", C.c_double), ("dist_id", C.POINTER(C.c_int32)), ("dist", C.POINTER(C.c_int32)),
                ("mgau_active", C.POINTER(C.c_uint8)), ("lm", C.c_void_p)]


class OracleMs:
    """ms_mgau_init + ms_cont_mgau_frame_eval (-senmgau .s3cont. / .semi.) on raw arrays."""

    def __init__(self, mean, var, mixw, n_mgau, n_density, featlen, lm: OracleLogMath, topn, sen2mgau=None,
                 varfloor=1e-4, mixwfloor=1e-7):
        L = self.L = lib()
        self.lm = lm
        L.s3o_ms_init.restype = C.POINTER(Ms)
        L.s3o_ms_init.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p, C.c_int32, C.c_void_p,
                                                                       C.c_double, C.c_double, C.c_int32, C.c_void_p]
        L.s3o_ms_cont_mgau_frame_eval.restype = C.c_int32
        L.s3o_ms_cont_mgau_frame_eval.argtypes = [C.POINTER(Ms), C.c_void_p, C.c_void_p, C.c_void_p]
        L.s3o_ms_free.argtypes = [C.POINTER(Ms)]
        mean = np.ascontiguousarray(mean, np.float32).ravel(); var = np.ascontiguousarray(var, np.float32).ravel()
        fl = np.ascontiguousarray(featlen, np.int32)
        mixw = np.ascontiguousarray(mixw, np.float32)
        self.n_sen = mixw.size // (len(fl) * n_density)
        s2m = None if sen2mgau is None else np.ascontiguousarray(sen2mgau, np.int32)
        self.p = L.s3o_ms_init(_p2(mean), _p2(var), _p2(mixw), n_mgau, len(fl), n_density, _p2(fl), self.n_sen,
                               None if s2m is None else _p2(s2m), varfloor, mixwfloor, topn, lm.p)
        self.senscr = np.zeros(self.n_sen, np.int32)

    def __del__(self):
        try:
            self.L.s3o_ms_free(self.p)
        except Exception:
            pass

    def arr(self, name, n, dt):
        return np.ctypeslib.as_array(getattr(self.p.contents, name), shape=(n,)).astype(dt)

    def frame_eval(self, sen_active, feat):
        """Returns (best, senscr copy); sen_active uint8[S]."""
        sa = np.ascontiguousarray(sen_active, np.uint8)
        x = np.ascontiguousarray(feat, np.float32)
        best = self.L.s3o_ms_cont_mgau_frame_eval(self.p, _p2(sa), _p2(self.senscr), _p2(x))
        return best, self.senscr.copy()

    def last_dist(self):
        c = self.p.contents
        n = c.n_mgau * c.n_feat * c.topn
        shp = (c.n_mgau, c.n_feat, c.topn)
        return self.arr("dist", n, np.int32).reshape(shp), self.arr("dist_id", n, np.int32).reshape(shp)


def delta_encode(mask):
    """acmod_flags2list (pocketsphinx acmod.c:1220-1271): ascending senone ids as uint8 deltas, gaps over
    255 bridged by 255-steps (which the scorer treats as active senones too)."""
    lst, last = [], 0
    for s in np.nonzero(np.asarray(mask))[0]:
        d = int(s) - last
        while d > 255:
            lst.append(255); d -= 255
        lst.append(d); last = int(s)
    return np.array(lst, np.uint8)


class OraclePsMs:
    """pocketsphinx's continuous scorer (ms_mgau_init + ms_cont_mgau_frame_eval) on raw arrays."""

    def __init__(self, mean, var, mixw, n_mgau, n_density, featlen, topn, aw=1, logbase=1.0001, sen2mgau=None,
                 varfloor=1e-4, mixwfloor=1e-7):
        L = self.L = lib()
        L.s3o_psms_init.restype = C.c_void_p
        L.s3o_psms_init.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_double,
                                                                         C.c_double, C.c_int32, C.c_int32, C.c_double]
        L.s3o_psms_frame_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.s3o_psms_free.argtypes = [C.c_void_p]
        mean = np.ascontiguousarray(mean, np.float32).ravel(); var = np.ascontiguousarray(var, np.float32).ravel()
        mixw = np.ascontiguousarray(mixw, np.float32)
        fl = np.ascontiguousarray(featlen, np.int32)
        self.n_sen = mixw.size // (len(fl) * n_density)
        s2m = None if sen2mgau is None else np.ascontiguousarray(sen2mgau, np.int32)
        self.h = L.s3o_psms_init(_p2(mean), _p2(var), _p2(mixw), n_mgau, len(fl), n_density, _p2(fl), self.n_sen,
                                 None if s2m is None else _p2(s2m), varfloor, mixwfloor, topn, aw, logbase)

    def __del__(self):
        try:
            self.L.s3o_psms_free(self.h)
        except Exception:
            pass

    def frame_eval(self, senscr, feat, mask=None):
        """senscr int16[S] in place; mask None = compallsen."""
        x = np.ascontiguousarray(feat, np.float32)
        if mask is None:
            self.L.s3o_psms_frame_eval(self.h, _p2(senscr), None, 0, _p2(x), 1)
        else:
            lst = delta_encode(mask)
            self.L.s3o_psms_frame_eval(self.h, _p2(senscr), _p2(lst) if len(lst) else None, len(lst), _p2(x), 0)


class OracleFeParams(C.Structure):
    """s3o_fe_params_t"""
    _fields_ = [("samprate", C.c_float), ("frate", C.c_int32), ("wlen", C.c_float), ("alpha", C.c_float),
                ("ncep", C.c_int32), ("nfft", C.c_int32), ("nfilt", C.c_int32), ("lowerf", C.c_float),
                ("upperf", C.c_float), ("transform", C.c_int32), ("lifter", C.c_int32), ("remove_dc", C.c_int32),
                ("round_filters", C.c_int32), ("unit_area", C.c_int32), ("doublebw", C.c_int32),
                ("logspec", C.c_int32)]


FE_DEFAULTS = dict(samprate=16000.0, frate=100, wlen=0.025625, alpha=0.97, ncep=13, nfft=512, nfilt=40,
                   lowerf=133.33334, upperf=6855.4976, transform=0, lifter=0, remove_dc=0, round_filters=1,
                   unit_area=1, doublebw=0, logspec=0)


class OracleFe:
    """The MFCC front end (fe_init_auto_r + fe_process_utt + fe_end_utt), oracle/s3o_fe.c."""

    def __init__(self, **opts):
        L = self.L = lib()
        L.s3o_fe_init.restype = C.c_void_p
        L.s3o_fe_init.argtypes = [C.c_void_p]
        L.s3o_fe_free.argtypes = [C.c_void_p]
        L.s3o_fe_n_frames.argtypes = [C.c_void_p, C.c_int64]
        L.s3o_fe_process_utt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self.p = OracleFeParams(**dict(FE_DEFAULTS, **opts))
        self.h = L.s3o_fe_init(C.byref(self.p))
        if not self.h:
            raise ValueError("s3o_fe_init rejected the options")
        self.out_dim = self.p.nfilt if self.p.logspec else self.p.ncep

    def __del__(self):
        try:
            self.L.s3o_fe_free(self.h)
        except Exception:
            pass

    def n_frames(self, nsamps):
        return self.L.s3o_fe_n_frames(self.h, nsamps)

    def process_utt(self, spch):
        spch = np.ascontiguousarray(spch, np.int16)
        n = self.n_frames(len(spch))
        out = np.zeros((max(n, 1), self.out_dim), np.float32)
        self.L.s3o_fe_process_utt(self.h, _p2(spch), len(spch), _p2(out))
        return out[:n]
