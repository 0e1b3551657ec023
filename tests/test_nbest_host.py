"""N-best lists in the library (s3a_lattice_nbest, cmusphinx_amd/csrc/s3a_nbest.hip), no GPU: lattices the UNMODIFIED reference decoder
wrote for the 31 tidigits utterances (-outlatdir: dag_write's orders are the orders the device's lattice is handed out in) against the
lists it wrote for them (-nbestdir: srch_TST_nbest_impl -> dag_remove_unreachable, dag_bypass_filler_nodes, dag_compute_hscr,
dag_remove_bypass_links, nbest_search), byte for byte -- default options, -bestpathlw 14 with -beam 1e-30 and -nbest 7, -maxppath 60.
tests/golden/make_nbest_golden.py made the fixture; the GPU tests (tests/test_gpu_dag.py) do the same through the drop-in program on
tidigits, RM1 and the hub4-shaped task, the lattice coming from the device."""
import os

import numpy as np
import pytest

from cmusphinx_amd import lib

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def world():
    g = np.load(os.path.join(G, "nbest_tidigits.npz"))
    w = dict(np.load(os.path.join(G, "wordlevel_tidigits.npz"), allow_pickle=True))
    words = bytes(g["dict.words"]).decode().split("\n")
    assert len(words) == int(w["n_word"]) and words[int(w["startwid"])] == "<s>" and words[int(w["finishwid"])] == "</s>"
    base = np.array([words.index(x.split("(")[0]) for x in words], np.int32)           # dict_basewid: "word(2)" -> "word"
    lm = lib.Lm3g(w, host_only=True)
    lmath = lib.LogMath(1.0003)
    return g, w, words, base, lm, lmath


def run_case(world, name):
    g, w, words, base, lm, lmath = world
    nbest, maxppath = (int(x) for x in g[name + ".meta"])
    beam, bplw = (float(x) for x in g[name + ".fmeta"])
    keep = []
    b = dict(n_word=int(w["n_word"]), basewid=base, is_filler=w["is_filler"], lwid=w["lwid"], fillpen=w["fillpen"], lw=9.5,
             wip_logs3=lmath.logs3(0.7), **{k: int(w[k]) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid")})
    cfg = lib.dag_cfg(b, keep, bestpathlw=bplw)
    wordstr = [x.encode() for x in words]
    n, out = int(g[name + ".n"][0]), []
    for k in range(n):
        key = f"{name}.{k}"
        nw = bytes(g[key + ".words"]).decode().split("\n")
        geo = g[key + ".nodes"]
        nodes = np.zeros((len(nw), 6), np.int32)
        nodes[:, 0] = [words.index(x) for x in nw]
        nodes[:, 1:4] = geo
        lk = g[key + ".links"]
        links = np.zeros((len(lk), 5), np.int32)
        links[:, 0:3] = lk                                  # from, to, ascr (the search reads neither a link's lscr nor its end frame)
        n_frames, initial, final = (int(x) for x in g[key + ".info"])
        info = lib.LatInfo(0, n_frames, len(nodes), len(links), initial, final, 0)
        o = lib.NbestOpts(bytes(g[key + ".uttid"]), beam, lmath.logs3(beam), nbest, maxppath, lmath.logs3(0.7), 1.0003, 9.5, 0.7, 9.5)
        txt, nh, cnt, st = lib.lattice_nbest(lm, cfg, o, info, nodes, links, wordstr)
        out.append((txt, nh, cnt, st, bytes(g[key + ".text"])))
    return out


@pytest.mark.parametrize("name", ["plain", "lw14_beam1e-30_n7", "maxppath60"])
def test_lists_are_the_reference_decoders(world, name):
    res = run_case(world, name)
    assert len(res) == 31
    for k, (txt, nh, cnt, st, want) in enumerate(res):
        assert st == 0 and nh >= 1 and txt == want, (k, txt.decode()[-300:], want.decode()[-300:])
    assert sum(r[1] for r in res) > 31                      # (some utterances have more than one hypothesis)


def test_arguments_are_checked(world):
    g, w, words, base, lm, lmath = world
    keep = []
    b = dict(n_word=int(w["n_word"]), basewid=base, is_filler=w["is_filler"], lwid=w["lwid"], fillpen=w["fillpen"], lw=9.5,
             wip_logs3=lmath.logs3(0.7), **{k: int(w[k]) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid")})
    cfg = lib.dag_cfg(b, keep)
    o = lib.NbestOpts(b"x", 1e-55, lmath.logs3(1e-55), 5, 1000, lmath.logs3(0.7), 1.0003, 9.5, 0.7, 9.5)
    nodes = np.array([[13, 10, 10, 10, 0, 0], [11, 0, 0, 0, 0, 0]], np.int32)
    with pytest.raises(lib.S3AError):                        # a link that leaves the lattice
        lib.lattice_nbest(lm, cfg, o, lib.LatInfo(0, 10, 2, 1, 1, 0, 0), nodes, np.array([[1, 5, -10, 0, 9]], np.int32), [x.encode() for x in words])
    # the smallest lattice: <s> -> </s>, one hypothesis
    txt, nh, cnt, st = lib.lattice_nbest(lm, cfg, o, lib.LatInfo(0, 10, 2, 1, 1, 0, -7), nodes, np.array([[1, 0, -10, 0, 9]], np.int32),
                                         [x.encode() for x in words])
    assert st == 0 and nh == 1 and b" 0 -10 0 <s> 10 -7 " in txt and txt.endswith(b"beam %d\n" % lmath.logs3(1e-55))


RM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_local_data", "rm1")
REFDEC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "sphinx3_decode")


@pytest.mark.skipif(not (os.path.exists(REFDEC) and os.path.exists(os.path.join(RM, "rm.ctl"))),
                    reason="needs the reference build (oracle/_ref) and the RM1 task (tools/fetch_local_data.sh): the build container")
@pytest.mark.parametrize("extra,meta", [(["-nbest", "40"], (40, 1e-140, 0.0)), (["-nbest", "25", "-bestpathlw", "11"], (25, 1e-140, 11.0))])
def test_rm1_lists_live(extra, meta, tmp_path):
    """the same on RM1 (1 000 words, trigram, lists of 25 - 40 hypotheses, lattices of hundreds of links): the unmodified reference decodes
    8 utterances here and now, writes lattices and lists; the library's search on those lattices writes the same lists"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(G))
    import make_nbest_golden as MG
    import test_oracle_dag as T
    lat, nb = tmp_path / "lat", tmp_path / "nb"
    lat.mkdir(); nb.mkdir()
    args = T.rm_args(8) + extra + ["-outlatdir", str(lat), "-latext", "lat", "-nbestdir", str(nb), "-nbestext", "nbest", "-hyp", str(tmp_path / "h")]
    p = subprocess.run([REFDEC] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, errors="ignore", timeout=1500)
    assert p.returncode == 0, p.stdout[-1500:]
    w = dict(np.load(os.path.join(G, "wordlevel_rm1.npz"), allow_pickle=True))
    words = []
    for fn in ("RM.dictionary", "fillerdict"):
        for ln in open(os.path.join(RM, fn), errors="ignore"):
            ln = ln.strip()
            if ln and not ln.startswith("#") and not ln.startswith(";;"):
                words.append(ln.split()[0])
    assert len(words) == int(w["n_word"]) and words[int(w["startwid"])] == "<s>" and words[int(w["finishwid"])] == "</s>"
    base = np.array([words.index(x.split("(")[0]) for x in words], np.int32)
    lm, lmath, keep = lib.Lm3g(w, host_only=True), lib.LogMath(1.0003), []
    nbest, beam, bplw = meta
    b = dict(n_word=int(w["n_word"]), basewid=base, is_filler=w["is_filler"], lwid=w["lwid"], fillpen=w["fillpen"], lw=14.0,
             wip_logs3=lmath.logs3(0.2), **{k: int(w[k]) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid")})
    cfg = lib.dag_cfg(b, keep, bestpathlw=bplw)
    wordstr = [x.encode() for x in words]
    utts = sorted(f[:-4] for f in os.listdir(lat) if f.endswith(".lat"))
    assert len(utts) == 8
    longest = 0
    for u in utts:
        n_frames, nd, initial, final, lk = MG.parse_lattice(MG.read_text(str(lat / (u + ".lat"))))
        nodes = np.zeros((len(nd), 6), np.int32)
        nodes[:, 0] = [words.index(x[0]) for x in nd]
        nodes[:, 1:4] = [x[1:] for x in nd]
        links = np.zeros((len(lk), 5), np.int32)
        links[:, 0:3] = np.array(lk, np.int32).reshape(-1, 3)
        info = lib.LatInfo(0, n_frames, len(nodes), len(links), initial, final, 0)
        o = lib.NbestOpts(u.encode(), beam, lmath.logs3(beam), nbest, 1000000, lmath.logs3(0.2), 1.0003, 14.0, 0.2, 14.0)
        txt, nh, cnt, st = lib.lattice_nbest(lm, cfg, o, info, nodes, links, wordstr)
        want = MG.read_text(str(nb / (u + ".nbest")))
        assert st == 0 and txt == want, (u, nh, cnt)
        longest = max(longest, nh)
    assert longest == nbest
