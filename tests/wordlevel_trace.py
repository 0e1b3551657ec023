"""Parse / replay the word-level trace written by oracle/_ref/ref_s3owl_decode (S3O_WLTRACE=...).

The trace holds the flattened trigram + dictionary facts of a real mode-4 decode and, per frame,
the word exits the lextree search produced (the word level's input), the beam arithmetic it
needs, and what the word level left: the frame's surviving history entries and its lextree_enter
calls.  It was recorded while the decoder's -hyp / -hypseg were byte-identical to the unmodified
reference, so the recorded results are reference-grade.
"""
import numpy as np

HDR = ["n_ug", "n_bg", "n_tg", "n_word", "n_ci", "startwid", "finishwid", "silwid", "start_lwid", "finish_lwid",
       "wbeam", "bghist", "maxwpf", "maxhistpf", "n_lextree", "epl", "wordend"]
STATIC = {2: "ug_prob", 3: "ug_bowt", 4: "ug_firstbg", 5: "bg_wid", 6: "bg_prob", 7: "bg_bowt", 8: "bg_firsttg",
          9: "tg_wid", 10: "tg_prob", 11: "lwid", 12: "is_filler", 13: "fillpen", 14: "last_ci"}
ENTRY = {31: "wid", 32: "score", 33: "pred", 34: "lw0", 35: "lw1", 36: "ascr", 37: "lscr", 38: "sf", 39: "type"}


def parse(path, max_frames=None):
    raw = np.fromfile(path, dtype="<i4")
    pos, recs = 0, []
    while pos < len(raw):
        tag, n = int(raw[pos]), int(raw[pos + 1])
        recs.append((tag, raw[pos + 2: pos + 2 + n].copy()))
        pos += 2 + n
    tag, hdr = recs[0]
    assert tag == 1
    out = {k: int(v) for k, v in zip(HDR, hdr)}
    frames, cur = [], None
    for tag, d in recs[1:]:
        if tag in STATIC:
            out[STATIC[tag]] = d
        elif tag == 20:         # frame: frmno, n_exit, prune beam, enter threshold
            if max_frames is not None and len(frames) >= max_frames:
                break
            cur = dict(frm=int(d[0]), n_exit=int(d[1]), prune_beam=int(d[2]), thresh=int(d[3]), trees=[])
            frames.append(cur)
        elif tag == 21:
            cur["trees"].append(dict(type=int(d[0])))
        elif tag in (22, 23, 24):
            cur["trees"][-1][{22: "wid", 23: "scr", 24: "hist"}[tag]] = d
        elif tag == 30:         # result: cf, #entries, #calls (-1: none), bestscore, bestvh, th
            cur["res"] = dict(n_entry=int(d[1]), n_calls=int(d[2]), bestscore=int(d[3]), bestvh=int(d[4]), th=int(d[5]))
        elif tag in ENTRY:
            cur["res"][ENTRY[tag]] = d
        elif tag in (40, 41, 42):
            cur["res"][{40: "lc", 41: "cscr", 42: "chist"}[tag]] = d
    out["frames"] = [f for f in frames if "res" in f]
    return out


def to_npz_dict(tr):
    d = {k: np.asarray(v) for k, v in tr.items() if k != "frames"}
    fr = tr["frames"]
    T = len(fr[0]["trees"])
    d["n_tree"] = np.array(T)
    d["f_hdr"] = np.array([[f["frm"], f["n_exit"], f["prune_beam"], f["thresh"], f["res"]["n_entry"], f["res"]["n_calls"],
                            f["res"]["bestscore"], f["res"]["bestvh"]] for f in fr], np.int32)
    d["tree_type"] = np.array([t["type"] for t in fr[0]["trees"]], np.int32)
    d["f_nexit"] = np.array([[len(t["wid"]) for t in f["trees"]] for f in fr], np.int32)
    cat = lambda key: np.concatenate([np.concatenate([t[key] for t in f["trees"]]) for f in fr] + [np.zeros(0, np.int32)]).astype(np.int32)
    d["x_wid"], d["x_scr"], d["x_hist"] = cat("wid"), cat("scr"), cat("hist")
    for k in ENTRY.values():
        d["e_" + k] = np.concatenate([f["res"].get(k, np.zeros(0, np.int32)) for f in fr] + [np.zeros(0, np.int32)]).astype(np.int32)
    for k in ("lc", "cscr", "chist"):
        d["c_" + k] = np.concatenate([f["res"].get(k, np.zeros(0, np.int32)) for f in fr] + [np.zeros(0, np.int32)]).astype(np.int32)
    return d


def from_npz(z):
    tr = {k: (int(z[k]) if z[k].ndim == 0 else z[k]) for k in z.files if not k.startswith(("f_", "x_", "e_", "c_"))}
    T = int(z["n_tree"])
    frames, xo, eo, co = [], 0, 0, 0
    for h, ne in zip(z["f_hdr"], z["f_nexit"]):
        trees = []
        for t in range(T):
            n = int(ne[t])
            trees.append(dict(type=int(z["tree_type"][t]), wid=z["x_wid"][xo:xo + n], scr=z["x_scr"][xo:xo + n],
                              hist=z["x_hist"][xo:xo + n]))
            xo += n
        n_entry, n_calls = int(h[4]), int(h[5])
        res = dict(n_entry=n_entry, n_calls=n_calls, bestscore=int(h[6]), bestvh=int(h[7]), th=int(h[3]))
        for k in ENTRY.values():
            res[k] = z["e_" + k][eo:eo + n_entry]
        eo += n_entry
        nc = max(n_calls, 0)
        for k in ("lc", "cscr", "chist"):
            res[k] = z["c_" + k][co:co + nc]
        co += nc
        frames.append(dict(frm=int(h[0]), n_exit=int(h[1]), prune_beam=int(h[2]), thresh=int(h[3]), trees=trees, res=res))
    tr["frames"] = frames
    return tr
