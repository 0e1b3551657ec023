"""The second pass on the device (MI355X): lattice from the Viterbi history, filler bypass, best path
under the trigram, backtrace -- s3a_dagpass_* / s3a_uttdec_enable_bestpath.

(a) recorded history tables of the REFERENCE (tests/golden/dag_tables.npz, written by tests/golden/make_dag_golden.py
    from oracle/_ref/ref_s3odag_decode): the device's words (wid, sf, ef, ascr, lscr), lattice sizes and LM-operation
    counts against the pinned restatement's, tidigits (31 utterances, two option sets) and RM1 (real 997-word trigram;
    -min_endfr 0; a tight -maxlpf under which some searches must FAIL exactly where the reference's do);
(b) whole decodes: the drop-in with -bestpath 1 (utt mode: first pass, vithist_utt_end and second pass all on the device)
    writes the unmodified reference's -hyp / -hypseg; the same through the C ABI alone with the history tables never
    read back (keep_tables = 0).
"""
import os
import subprocess

import numpy as np
import pytest

from cmusphinx_amd import bundle, lib, s3io
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
RM = os.path.join(ROOT, "tests", "_local_data", "rm1")
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
OVERRIDES = {"tidigits": {}, "tidigits_lw14_minendfr1": {"bestpathlw": 14.0, "min_endfr": 1}, "rm1": {}, "rm1_minendfr0": {"min_endfr": 0},
             "rm1_maxlpf5": {"maxlpf": 5}}


def tidigits_args():
    return ["-dict", f"{D}/dictionary", "-fdict", f"{D}/fillerdict", "-hmm", AM, "-cepdir", f"{D}/cepstra", "-agc", "none",
            "-varnorm", "no", "-cmn", "current", "-lw", "9.5", "-ctl", f"{D}/tidigits.length.arb.regression", "-op_mode", "4",
            "-lm", f"{D}/tidigits.DMP"]


def rm_args(n=20):
    return ["-mdef", f"{RM}/mdef", "-fdict", f"{RM}/fillerdict", "-dict", f"{RM}/RM.dictionary", "-mean", f"{RM}/means",
            "-var", f"{RM}/variances", "-mixw", f"{RM}/mixture_weights", "-tmat", f"{RM}/transition_matrices",
            "-agc", "none", "-varnorm", "no", "-cmn", "current", "-epl", "4", "-fillprob", "0.02", "-maxwpf", "10",
            "-wip", "0.2", "-lm", f"{RM}/RM.2845.trigram.arpa.DMP", "-lw", "14", "-beam", "1e-140", "-wbeam", "1e-100",
            "-cepdir", f"{RM}/feat", "-cepext", ".mfc", "-ctl", f"{RM}/rm.ctl", "-ctlcount", str(n), "-op_mode", "4"]


def export(args, path):
    for p in (TST,):
        if not os.path.exists(p):
            pytest.fail(f"{p} is missing on the GPU box (make -C oracle ref)")
    r = subprocess.run([TST] + args, env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=path), stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0 and os.path.getsize(path) > 1000
    return bundle.read(path)


@pytest.fixture(scope="module")
def bundles(tmp_path_factory):
    d = tmp_path_factory.mktemp("dagb")
    out = {"tidigits": export(tidigits_args(), str(d / "t.bundle"))}
    if not os.path.isdir(RM):
        pytest.fail(f"{RM} is missing on the GPU box (tools/fetch_local_data.sh)")
    out["rm1"] = export(rm_args(), str(d / "r.bundle"))
    return out


@pytest.mark.parametrize("case", list(OVERRIDES))
def test_device_second_pass_on_recorded_reference_tables(gpu_lib, bundles, case):
    G = np.load(os.path.join(GOLDEN, "dag_tables.npz"))
    b = bundles["tidigits" if case.startswith("tidigits") else "rm1"]
    n_utt = int(G[f"{case}.n"][0])
    tabs = []
    for k in range(n_utt):
        hdr = G[f"{case}.{k}.hdr"]
        tabs.append({key: G[f"{case}.{k}.{key}"] for key in ("wid", "sf", "ef", "ascr", "lscr", "score", "hyp_wid", "hyp_sf")}
                    | {"n_frm": int(hdr[1]), "endid": int(hdr[2])})
    keep = []
    cfg = gpu_lib.dag_cfg(b, keep, **OVERRIDES[case])
    lm = gpu_lib.Lm3g(dict(b, wbeam=b["wbeam_vh"]))
    dp = gpu_lib.DagPass(lm, cfg, n_utt, max(len(t["wid"]) for t in tabs), max(t["n_frm"] for t in tabs) + 2, link_cap=1 << 17, pair_cap=1 << 14)
    res = dp.run(tabs)
    n_fail = 0
    for k, r in enumerate(res):
        hdr = G[f"{case}.{k}.hdr"]
        nw = int(hdr[4])
        assert (r.n_node, r.n_link, r.n_bypass) == (int(hdr[5]), int(hdr[6]), int(hdr[7])), (case, k, r.status)
        if nw < 0:                      # the reference's search gave up (LM operation limit): so must the device's
            assert r.status == 2 and r.n_words == 0, (case, k, r.status)
            n_fail += 1
            continue
        assert r.status == 0, (case, k, r.status)
        assert r.lmop == int(G[f"{case}.{k}.lmop"][0]), (case, k)
        exp = np.stack([G[f"{case}.{k}.o_{key}"] for key in ("wid", "sf", "ef", "ascr", "lscr")], axis=1)
        assert np.array_equal(r.words(), exp), (case, k)
    assert (n_fail > 0) == (case == "rm1_maxlpf5")


def test_a_lattice_beyond_the_link_capacity_fails_its_utterance_only(gpu_lib, bundles):
    """a dense lattice must not kill the batch.  With a link capacity smaller than some of the RM1 lattices the
    pass reports status 3 for THOSE utterances (the drop-in then writes no second-pass line for them, as after the
    reference's "Bestpath search failed") and decodes the others exactly as before; -maxedge is not enforced while the
    lattice is built (vithist_dag_build ignores dag_link's return): a tiny -maxedge only stops utterances in the bypass."""
    G = np.load(os.path.join(GOLDEN, "dag_tables.npz"))
    case, b = "rm1", bundles["rm1"]
    n_utt = int(G[f"{case}.n"][0])
    tabs, nlink = [], []
    for k in range(n_utt):
        hdr = G[f"{case}.{k}.hdr"]
        tabs.append({key: G[f"{case}.{k}.{key}"] for key in ("wid", "sf", "ef", "ascr", "lscr", "score", "hyp_wid", "hyp_sf")}
                    | {"n_frm": int(hdr[1]), "endid": int(hdr[2])})
        nlink.append(int(hdr[6]))
    cap = int(np.median(nlink))
    keep = []
    lm = gpu_lib.Lm3g(dict(b, wbeam=b["wbeam_vh"]))
    dp = gpu_lib.DagPass(lm, gpu_lib.dag_cfg(b, keep, **OVERRIDES[case]), n_utt, max(len(t["wid"]) for t in tabs),
                         max(t["n_frm"] for t in tabs) + 2, link_cap=cap, pair_cap=1 << 14)
    res = dp.run(tabs)
    st = [r.status for r in res]
    assert [s == 3 for s in st] == [n > cap for n in nlink] and 0 < sum(s == 3 for s in st) < n_utt
    for k, r in enumerate(res):
        if r.status == 0:
            exp = np.stack([G[f"{case}.{k}.o_{key}"] for key in ("wid", "sf", "ef", "ascr", "lscr")], axis=1)
            assert np.array_equal(r.words(), exp), k
    # -maxedge below every lattice's link count: the BUILD still goes through (node / link counts as recorded)
    keep2 = []
    cfg2 = gpu_lib.dag_cfg(b, keep2, **dict(OVERRIDES[case], maxedge=min(nlink) - 1))
    dp2 = gpu_lib.DagPass(lm, cfg2, n_utt, max(len(t["wid"]) for t in tabs), max(t["n_frm"] for t in tabs) + 2, link_cap=1 << 17, pair_cap=1 << 14)
    for k, r in enumerate(dp2.run(tabs)):
        assert r.n_link == nlink[k] and r.status == 3, (k, r.status)


def run(exe, args, tmp_path, tag, env=None):
    hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT, timeout=1200, env=env)
    txt = open(log, errors="ignore").read()
    assert p.returncode == 0, "\n".join(l for l in txt.splitlines() if "FATAL" in l or "tst shim" in l)[-2000:]
    return open(hyp).read(), open(seg).read(), txt


@pytest.mark.parametrize("task,lanes,extra", [("tidigits", "5", []), ("tidigits", "1", ["-bestpathlw", "14", "-min_endfr", "1"]),
                                               ("rm1", "7", []), ("rm1", "33", ["-maxlpf", "5"])])
def test_dropin_with_both_passes_on_the_device(task, lanes, extra, tmp_path):
    args = (tidigits_args() if task == "tidigits" else rm_args()) + ["-bestpath", "1"] + extra
    ref = run(REFDEC, args, tmp_path, "ref")
    gpu = run(TST, args, tmp_path, "gpu", env=dict(os.environ, S3A_UTT=lanes))
    assert "second pass (lattice + best path) of" in gpu[2] and "served by the device" in gpu[2]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    if "-maxlpf" in extra:
        assert 0 < ref[0].count("\n") < 20
    # the second pass is no bystander on RM1: its output differs from the first pass's
    if task == "rm1" and not extra:
        fp = run(REFDEC, rm_args(), tmp_path, "fp")
        assert fp[1] != ref[1]


def _files(d):
    out = {}
    for root, _, names in os.walk(d):
        for n in names:
            out[os.path.relpath(os.path.join(root, n), d)] = open(os.path.join(root, n), "rb").read()
    return out


@pytest.mark.parametrize("task,lanes,fmt,extra", [("tidigits", "6", "s3", []), ("tidigits", "31", "htk", ["-bestpath", "1"]),
                                                   ("rm1", "7", "s3", ["-bestpath", "1"]), ("rm1", "20", "htk", ["-min_endfr", "0"])])
def test_lattice_files_from_the_device_lattice(task, lanes, fmt, extra, tmp_path):
    """-outlatdir (dag_write, dag.c:731-790; dag_write_htk, :793-897): the drop-in writes the Sphinx-3 format with the
    library's formatter from the device's lattice (the dag_dump slot) and the HTK format with the reference's own writer on
    a dag_t poured from it -- and, beside it, with the library's HTK formatter: every file byte for byte the unmodified
    reference's (node order, edge order, scores, the configuration header)"""
    base = tidigits_args() if task == "tidigits" else rm_args()
    outs = {}
    for tag, exe, env in (("ref", REFDEC, None), ("gpu", TST, dict(os.environ, S3A_UTT=lanes, S3A_LAT_LIBHTK="1"))):
        d = tmp_path / f"lat_{tag}"
        d.mkdir()
        # (the same -outlatdir string in both runs: the header quotes no output path, but keep the runs alike)
        outs[tag] = run(exe, base + extra + ["-outlatdir", str(d), "-outlatfmt", fmt, "-latext", "lat"], tmp_path, tag, env=env) + (_files(str(d)),)
    ref, gpu = outs["ref"], outs["gpu"]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    lat_ref = {k: v for k, v in ref[3].items() if k.endswith(".lat")}
    lat_gpu = {k: v for k, v in gpu[3].items() if k.endswith(".lat")}
    assert len(lat_ref) == (31 if task == "tidigits" else 20) and sorted(lat_ref) == sorted(lat_gpu)
    for k in sorted(lat_ref):
        assert lat_gpu[k] == lat_ref[k], k
    assert "served by the device" in gpu[2]
    if fmt == "s3":
        assert b"Edges (FROM-NODEID TO-NODEID ASCORE)" in next(iter(lat_ref.values()))
    else:
        lib_htk = {k[:-len(".libhtk")] + ".lat": v for k, v in gpu[3].items() if k.endswith(".libhtk")}
        assert sorted(lib_htk) == sorted(lat_ref)
        for k in sorted(lat_ref):
            assert lib_htk[k] == lat_ref[k], k


@pytest.mark.parametrize("task,lanes,extra", [("tidigits", "8", []), ("rm1", "20", ["-bestpath", "1"]),
                                              ("rm1", "7", ["-bestpathlw", "11", "-nbest", "50"])])
def test_nbest_lists_on_the_device_lattice(task, lanes, extra, tmp_path):
    """-nbestdir: N-best lists by the LIBRARY's search (s3a_lattice_nbest, csrc/s3a_nbest.hip: dag_remove_unreachable,
    dag_bypass_filler_nodes, dag_compute_hscr, dag_remove_bypass_links, then A* with the reference's heap, duplicate table and
    arithmetic) on the device's lattice: the files are the unmodified reference's, byte for byte -- and so are the ones the
    reference's OWN search writes on a dag_t poured from the same lattice (S3A_REF_NBEST=1, round 4's form: the A/B).  (The
    reference's astar.c needs its own pio.h on the compiler's command line to run on a 64-bit machine at all: oracle/Makefile.)"""
    base = tidigits_args() if task == "tidigits" else rm_args()
    nb = [] if "-nbest" in extra else ["-nbest", "20"]
    outs = {}
    for tag, exe, env in (("ref", REFDEC, None), ("gpu", TST, dict(os.environ, S3A_UTT=lanes)),
                          ("gpu_refsearch", TST, dict(os.environ, S3A_UTT=lanes, S3A_REF_NBEST="1"))):
        d = tmp_path / f"nb_{tag}"
        d.mkdir()
        outs[tag] = run(exe, base + extra + nb + ["-nbestdir", str(d), "-nbestext", "nbest"], tmp_path, tag, env=env) + (_files(str(d)),)
    ref, gpu, gpu2 = outs["ref"], outs["gpu"], outs["gpu_refsearch"]
    assert "N-Best search" in gpu[2] and "in the library" in gpu[2] and "in the library" not in gpu2[2]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    assert len(ref[3]) == (31 if task == "tidigits" else 20) and sorted(ref[3]) == sorted(gpu[3]) == sorted(gpu2[3])
    assert any(v.count(b"\nT ") > 3 for v in ref[3].values())          # (lists with several hypotheses)
    for k in sorted(ref[3]):
        assert gpu[3][k] == ref[3][k], k
        assert gpu2[3][k] == ref[3][k], k


def test_nbest_lists_on_the_hub4_shaped_task(tmp_path):
    """20 k words, trigram, fillers inside the utterances: lattices of thousands of links, lists of 30 hypotheses"""
    import test_gpu_dropin as TD
    args = TD.synth_task("hub4", tmp_path, 4, 200)
    outs = {}
    for tag, exe, env in (("ref", REFDEC, None), ("gpu", TST, dict(os.environ, S3A_UTT="4"))):
        d = tmp_path / f"nb_{tag}"
        d.mkdir()
        outs[tag] = TD.decode_task(exe, args + ["-nbestdir", str(d), "-nbest", "30", "-nbestext", "nbest"], tmp_path, tag, env) + (_files(str(d)),)
    ref, gpu = outs["ref"], outs["gpu"]
    assert gpu[0] == ref[0] and gpu[1] == ref[1] and len(ref[3]) == 4 and sorted(ref[3]) == sorted(gpu[3])
    assert sum(v.count(b"\nT ") for v in ref[3].values()) > 8          # (lists of several hypotheses)
    for k in sorted(ref[3]):
        assert gpu[3][k] == ref[3][k], k


@pytest.mark.parametrize("task,lanes,queue,extra", [("tidigits", "4", "31", []), ("tidigits", "7", "16", ["-bestpathlw", "14", "-min_endfr", "1"]),
                                                     ("rm1", "6", "20", []), ("rm1", "3", "20", ["-maxlpf", "5"]),
                                                     # from 8 lanes on (round 6): groups of static ku_frames launches with the pass behind each
                                                     ("tidigits", "9", "31", ["-bestpathlw", "14", "-min_endfr", "1"]), ("rm1", "8", "20", ["-maxlpf", "5"])])
def test_second_pass_inside_a_queue_with_lane_refill(task, lanes, queue, extra, tmp_path):
    """-bestpath 1 with S3A_UTT_QUEUE: a lane that has ended runs vithist_utt_end + the second pass at its refill event,
    before its history table is reused; -hyp / -hypseg are the unmodified reference's (ragged utterances, every lane
    several utterances, searches that must fail under a tight -maxlpf included)"""
    args = (tidigits_args() if task == "tidigits" else rm_args()) + ["-bestpath", "1"] + extra
    ref = run(REFDEC, args, tmp_path, "ref")
    gpu = run(TST, args, tmp_path, "gpu", env=dict(os.environ, S3A_UTT=lanes, S3A_UTT_QUEUE=queue))
    assert "served by the device" in gpu[2] and "lane refill" in gpu[2]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    if "-maxlpf" in extra:
        assert 0 < ref[0].count("\n") < 20


@pytest.mark.parametrize("task,lanes,queue,extra", [("tidigits", "4", "31", []), ("tidigits", "9", "31", ["-bestpath", "1", "-outlatfmt", "htk"]),
                                                     ("rm1", "8", "20", ["-bestpath", "1"]), ("rm1", "5", "20", [])])
def test_lattices_and_nbest_lists_out_of_a_queue(task, lanes, queue, extra, tmp_path):
    """S3A_UTT_QUEUE with -outlatdir and -nbestdir (round 6): the lanes' lattices are read back behind every group's (below 8 lanes: every
    refill event's) second pass and kept per utterance (s3a_uttdec_queue_keep_lattices / _queue_lattice); the files -- the library's
    formatters, the library's N-best search -- and -hyp / -hypseg (the first pass's hypothesis without -bestpath, the second's with it)
    are the unmodified reference's, byte for byte"""
    base = tidigits_args() if task == "tidigits" else rm_args()
    outs = {}
    for tag, exe, env in (("ref", REFDEC, None), ("gpu", TST, dict(os.environ, S3A_UTT=lanes, S3A_UTT_QUEUE=queue))):
        dl, dn = tmp_path / f"lat_{tag}", tmp_path / f"nb_{tag}"
        dl.mkdir(); dn.mkdir()
        outs[tag] = run(exe, base + extra + ["-outlatdir", str(dl), "-latext", "lat", "-nbestdir", str(dn), "-nbest", "15", "-nbestext", "nbest"],
                        tmp_path, tag, env=env) + (_files(str(dl)), _files(str(dn)))
    ref, gpu = outs["ref"], outs["gpu"]
    assert "lane refill" in gpu[2]
    assert gpu[0] == ref[0] and gpu[1] == ref[1]
    n = 31 if task == "tidigits" else 20
    assert len(ref[3]) == n and sorted(ref[3]) == sorted(gpu[3]) and len(ref[4]) == n and sorted(ref[4]) == sorted(gpu[4])
    for k in sorted(ref[3]):
        a, b = ref[3][k], gpu[3][k]
        if "htk" in extra:          # (the HTK header prints the program's own name and date: compare from the size line on, as the lock-step test does)
            a, b = a[a.index(b"\nN="):], b[b.index(b"\nN="):]
        assert a == b, k
    for k in sorted(ref[4]):
        assert gpu[4][k] == ref[4][k], k


def test_second_pass_through_the_c_abi_alone_tables_stay_on_the_device(gpu_lib, tmp_path):
    """bundle -> s3a_uttdec_init + s3a_uttdec_enable_bestpath(keep_tables = 0): cepstra in, the second pass's hypotheses
    out; the history tables are never read back (s3a_uttdec_result refuses)"""
    args = tidigits_args() + ["-bestpath", "1"]
    ref = run(REFDEC, args, tmp_path, "ref")
    bpath = str(tmp_path / "t.bundle")
    export(args, bpath)
    dec = bundle.Decoder(bpath, 4, bestpath=True, keep_tables=False)
    utts = [l.split() for l in open(f"{D}/tidigits.length.arb.regression") if l.strip()]
    utts = [(f[0], f[3] if len(f) > 3 else os.path.basename(f[0])) for f in utts]
    match, seg = [], []
    for k in range(0, len(utts), 4):
        chunk = utts[k:k + 4]
        feats = [gpu_lib.feat_1s_c_d_dd(s3io.read_mfc(f"{D}/cepstra/{u}.mfc").reshape(-1, 13), cmn="current") for u, _ in chunk]
        dec.decode(feats)
        with pytest.raises(gpu_lib.S3AError, match="left on the device"):
            dec.ud.result(0)
        for z, (_, uid) in enumerate(chunk):
            h, w = dec.bestpath_hyp(z, uid, k + z)
            assert h.status == 0 and h.n_frames == len(feats[z])
            m, s = dec.format_var(h, w)
            match.append(m); seg.append(s)
    assert "".join(match) == ref[0] and "".join(seg) == ref[1]
