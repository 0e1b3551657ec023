"""The drop-in, end to end (MI355X): the UNMODIFIED reference decoder with the three
scoring slots of its srch_funcs_t table served by libcmusphinx_amd through the C ABI
(oracle/_ref/ref_s3amd_decode = oracle/ref_s3amd_decode.c + libs3ref.so) must
reproduce, byte for byte, the hypothesis files (-hyp) AND the per-word acoustic / LM
score segmentations (-hypseg) of the reference's own CPU scoring on the bundled
tidigits regression utterances:

  mode 2 (FSG)      vs the reference's own golden tidigits.length.arb.result
  mode 4 (fwdtree)  lextree + trigram LM, default beams
  mode 4            narrow -ci_pbeam 1e-5 and -ds 2: the CI gate, the best-Gaussian
                    back-off and frame down-sampling all fire inside a real search,
                    with the search's own active-senone masks

Identical -hypseg means identical senone scores along every surviving path and
identical frame normalisers (ascale), not just identical words.
"""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def need(*paths):
    """oracle/_ref (the reference build) and tests/_local_data travel to the GPU box with the snapshot; a gpu-marked
    test that cannot find them must FAIL, not vanish (they are git-ignored: a clean checkout has neither)."""
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:
        pytest.fail("missing on the GPU box: " + ", ".join(missing) +
                    " (make -C oracle ref; tools/fetch_local_data.sh -- both need /root/reference)")

D = os.path.join(GOLDEN, "tidigits_decode")
AM = os.path.join(GOLDEN, "tidigits")
SHIM = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_decode")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")

RUNS = {
    "mode2_fsg": ["-op_mode", "2", "-fsg", os.path.join(D, "test.digits.fsg")],
    "mode4_trigram": ["-op_mode", "4", "-lm", os.path.join(D, "tidigits.DMP")],
    "mode4_cibeam_ds2": ["-op_mode", "4", "-lm", os.path.join(D, "tidigits.DMP"),
                         "-ci_pbeam", "1e-5", "-ds", "2"],
}


def common():
    return ["-dict", os.path.join(D, "dictionary"), "-fdict", os.path.join(D, "fillerdict"),
            "-hmm", AM, "-cepdir", os.path.join(D, "cepstra"), "-agc", "none", "-varnorm", "no",
            "-cmn", "current", "-lw", "9.5", "-ctl", os.path.join(D, "tidigits.length.arb.regression")]


@pytest.fixture(autouse=True)
def _artefacts():
    need(SHIM, REFDEC, TST, PSSHIM, RM)


def run(binary, extra, tmp_path, tag):
    hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([binary] + common() + extra + ["-hyp", hyp, "-hypseg", seg],
                           stdout=lf, stderr=subprocess.STDOUT, timeout=600)
    tail = [l for l in open(log, errors="ignore").read().splitlines() if "s3amd shim" in l or "FATAL" in l]
    assert p.returncode == 0, "\n".join(tail[-10:])
    return open(hyp).read(), open(seg).read(), tail


@pytest.mark.parametrize("name", list(RUNS))
def test_reference_decoder_with_gpu_scoring_matches_reference(name, tmp_path):
    hyp, seg, tail = run(SHIM, RUNS[name], tmp_path, "gpu_" + name)
    assert any("calls served by the GPU" in l for l in tail)
    assert hyp == open(os.path.join(D, f"ref_{name}.match")).read()
    assert seg == open(os.path.join(D, f"ref_{name}.matchseg")).read()
    if name == "mode2_fsg":     # the reference's own regression golden
        assert hyp == open(os.path.join(D, "tidigits.length.arb.result")).read()


@pytest.mark.parametrize("topn", [4, 8])
def test_multistream_scorer_dropin_matches_live_reference(topn, tmp_path):
    """-senmgau .s3cont.: the reference routes gmm_compute_lv2 to ms_cont_mgau_frame_eval
    (gauden_dist top-N + senone_eval); the shim serves that slot from s3a_ms_cont_mgau_frame_eval.
    topn 4 = sorted top-N lists, topn 8 = all 8 densities in codeword order."""
    extra = RUNS["mode4_trigram"] + ["-senmgau", ".s3cont.", "-topn", str(topn)]
    ref_hyp, ref_seg, _ = run(REFDEC, extra, tmp_path, f"cpu_ms{topn}")
    hyp, seg, tail = run(SHIM, extra, tmp_path, f"gpu_ms{topn}")
    assert any("calls served by the GPU" in l for l in tail)
    assert hyp == ref_hyp and seg == ref_seg
    assert seg != open(os.path.join(D, "ref_mode4_trigram.matchseg")).read()       # really a different scorer


def test_live_cpu_reference_agrees_with_committed_golden(tmp_path):
    """The committed golden is what the reference produces on THIS box too."""
    hyp, seg, _ = run(REFDEC, RUNS["mode4_trigram"], tmp_path, "cpu_mode4")
    assert hyp == open(os.path.join(D, "ref_mode4_trigram.match")).read()
    assert seg == open(os.path.join(D, "ref_mode4_trigram.matchseg")).read()


# ---------------------------------------------------------------------------
# the whole per-frame hot path on the GPU: scoring + composite senones +
# active-senone selection + lextree HMM evaluation + phone-level propagation
# (oracle/ref_tst_shim.c); only vithist + LM stay the reference's host code
# ---------------------------------------------------------------------------
TST = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")


DRIVERS = {
    "one_decoder": {},                                          # s3a_decoder_*: one decoder, one stream
    "four_streams": {"S3A_STREAMS": "4"},                       # four decoders on four HIP streams
    "batched_4x1": {"S3A_STREAMS": "4", "S3A_BATCH": "1"},      # s3a_batch_*: four decoders share every launch
    "batched_6x2": {"S3A_STREAMS": "6", "S3A_BATCH": "2"},      # two engines of three decoders alternate
    # s3a_uttdec_*: WHOLE utterances on the device (word level, history table, trigram included), one kb_t
    "utt_1": {"S3A_UTT": "1"},
    "utt_4": {"S3A_UTT": "4"},
    "utt_3_bigwl": {"S3A_UTT": "3", "S3A_UTT_BIGWL": "1"},      # the word level's candidate phases as chip-wide launches
    "utt_8_engines2": {"S3A_UTT": "8", "S3A_UTT_ENGINES": "2"},    # two engines of four lanes, a host thread each
    "utt_40": {"S3A_UTT": "40", "S3A_UTT_MANY": "2", "S3A_UTT_SCAN_SMALL": "2"},   # the kernels / grids chosen from 32 (the scan's
                                                                # 256-thread workgroups: from 64) utterances per launch on
                                                                # (list-driven resolve, fewer workgroups that loop), forced from 2
    # the per-frame scoring kernels (ku_gated / ku_gated_cd_multi) instead of the look-ahead window + ku_select
    "utt_4_perframe": {"S3A_UTT": "4", "S3A_UTT_WIN": "0"},
    "utt_40_perframe": {"S3A_UTT": "40", "S3A_UTT_MANY": "2", "S3A_UTT_WIN": "0"},
    "utt_5_win16": {"S3A_UTT": "5", "S3A_UTT_WIN": "16"},       # another window length than the lane count picks
}


@pytest.mark.parametrize("name,driver", [("mode4_trigram", "one_decoder"), ("mode4_cibeam_ds2", "one_decoder"),
                                         ("mode4_trigram", "four_streams"), ("mode4_trigram", "batched_4x1"),
                                         ("mode4_cibeam_ds2", "batched_6x2"), ("mode4_trigram", "utt_1"),
                                         ("mode4_trigram", "utt_4"), ("mode4_cibeam_ds2", "utt_4"),
                                         ("mode4_trigram", "utt_3_bigwl"), ("mode4_trigram", "utt_40"),
                                         ("mode4_cibeam_ds2", "utt_8_engines2"), ("mode4_cibeam_ds2", "utt_4_perframe"),
                                         ("mode4_trigram", "utt_40_perframe"), ("mode4_cibeam_ds2", "utt_40_perframe"),
                                         ("mode4_cibeam_ds2", "utt_5_win16"), ("mode4_cibeam_ds2", "utt_40")])
def test_full_device_search_matches_reference(name, driver, tmp_path):
    hyp, seg, log = (str(tmp_path / f"tst_{name}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([TST] + common() + RUNS[name] + ["-hyp", hyp, "-hypseg", seg],
                           stdout=lf, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **DRIVERS[driver]))
    tail = [l for l in open(log, errors="ignore").read().splitlines() if "tst shim" in l or "FATAL" in l]
    assert p.returncode == 0, "\n".join(tail[-10:])
    assert any("frames searched by the replacement backend" in l for l in tail)
    if "S3A_BATCH" in DRIVERS[driver]:
        mb = [float(l.split("mean batch")[1].strip(" )")) for l in tail if "batched engine" in l]
        assert mb and mb[0] > 1.5, tail                         # the launches really were shared
    assert open(hyp).read() == open(os.path.join(D, f"ref_{name}.match")).read()
    assert open(seg).read() == open(os.path.join(D, f"ref_{name}.matchseg")).read()


@pytest.mark.parametrize("driver", ["one_decoder", "batched_4x1", "utt_4", "utt_40"])
def test_full_device_search_with_phone_threshold_below_hmm_threshold(driver, tmp_path):
    """-ptranskip 2 (every second frame the WORD threshold gates phone transitions) and a phone beam
    wider than the HMM beam: HMMs under the beam but over the phone threshold propagate only if a
    parent re-entered them earlier in the frame (k_dec_weak); expected output live from the reference."""
    extra = RUNS["mode4_trigram"] + ["-ptranskip", "2", "-beam", "1e-80", "-pbeam", "1e-100", "-wbeam", "1e-40"]
    ref_hyp, ref_seg, _ = run(REFDEC, extra, tmp_path, "cpu_pt")
    hyp, seg, log = (str(tmp_path / f"tst_pt.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([TST] + common() + extra + ["-hyp", hyp, "-hypseg", seg], stdout=lf,
                           stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **DRIVERS[driver]))
    tail = [l for l in open(log, errors="ignore").read().splitlines() if "tst shim" in l or "FATAL" in l]
    assert p.returncode == 0, "\n".join(tail[-10:])
    assert open(hyp).read() == ref_hyp and open(seg).read() == ref_seg
    assert ref_seg != open(os.path.join(D, "ref_mode4_trigram.matchseg")).read()


@pytest.mark.parametrize("driver", ["one_decoder", "utt_1", "utt_4", "utt_40"])
def test_full_device_search_with_histogram_pruning(driver, tmp_path):
    """-maxhmmpf 20: most frames exceed 1.5 x the cap, so lextree_hmm_histbin (bins, beam from the
    bin scan AND the reordering of the active lists) runs on the device; expected output is
    produced live by the unmodified reference."""
    extra = RUNS["mode4_trigram"] + ["-maxhmmpf", "20"]
    ref_hyp, ref_seg, _ = run(REFDEC, extra, tmp_path, "cpu_hist")
    hyp, seg, log = (str(tmp_path / f"tst_hist.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([TST] + common() + extra + ["-hyp", hyp, "-hypseg", seg],
                           stdout=lf, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **DRIVERS[driver]))
    tail = [l for l in open(log, errors="ignore").read().splitlines() if "tst shim" in l or "FATAL" in l]
    assert p.returncode == 0, "\n".join(tail[-10:])
    n = [int(l.split("applied in")[1].split()[0]) for l in tail if "histogram pruning" in l]
    assert n and n[0] > 500, tail
    assert open(hyp).read() == ref_hyp
    assert open(seg).read() == ref_seg


# ---------------------------------------------------------------------------
# RM1 (1935 senones, 6136-node lextrees with multi-parent first-level nodes, 997-word
# trigram): data is NOT committed (8 MB of third-party model files); tools/
# fetch_local_data.sh copies it from the reference checkout into tests/_local_data/,
# which travels with the gpurun snapshot.  Here the expected output is produced LIVE by
# the unmodified reference on the same box, so no golden file is needed.
# ---------------------------------------------------------------------------
RM = os.path.join(ROOT, "tests", "_local_data", "rm1")


def rm_args(extra=()):
    return list(extra) + ["-mdef", f"{RM}/mdef", "-fdict", f"{RM}/fillerdict", "-dict", f"{RM}/RM.dictionary",
            "-mean", f"{RM}/means", "-var", f"{RM}/variances", "-mixw", f"{RM}/mixture_weights",
            "-tmat", f"{RM}/transition_matrices", "-agc", "none", "-varnorm", "no", "-cmn", "current",
            "-epl", "4", "-fillprob", "0.02", "-maxwpf", "10", "-wip", "0.2",
            "-lm", f"{RM}/RM.2845.trigram.arpa.DMP", "-lw", "14", "-beam", "1e-140", "-wbeam", "1e-100",
            "-cepdir", f"{RM}/feat", "-cepext", ".mfc", "-ctl", f"{RM}/rm.ctl", "-ctlcount", "20", "-op_mode", "4"]


@pytest.mark.parametrize("binary", ["scoring_only", "full_device", "full_device_histprune", "full_device_batched",
                                    "utt_1", "utt_7_histprune", "utt_3_tight_word_limits", "utt_33", "utt_5_bestpath", "utt_4_maxcdsenpf", "utt_34_maxcdsenpf"])
def test_rm1_identical_to_live_reference(binary, tmp_path):
    exe = SHIM if binary == "scoring_only" else TST
    extra = ["-maxhmmpf", "800"] if binary in ("full_device_histprune", "full_device_batched", "utt_7_histprune") else []
    env = dict(os.environ, S3A_STREAMS="5", S3A_BATCH="2") if binary == "full_device_batched" else None
    if binary.startswith("utt_"):
        env = dict(os.environ, S3A_UTT=binary.split("_")[1], **({"S3A_UTT_MANY": "2"} if binary.split("_")[1] in ("33", "34") else {}))
    if binary.endswith("_maxcdsenpf"):          # the dynamic CI beam (approx_compute_dyn_ci_pbeam): 145 instead of 572 CD
        extra = ["-ci_pbeam", "1e-10", "-maxcdsenpf", "150"]    # senones per frame survive the gate; worked out on the device
    if binary == "utt_5_bestpath":              # the reference's SECOND pass (DAG from the history table, dag_bestpath)
        extra = ["-bestpath", "1"]              # runs unchanged on the table the device produced (srch_utt_end)
    if binary == "utt_3_tight_word_limits":     # the word level's own pruning: few words / histories per frame, bigram history
        extra = ["-maxwpf", "3", "-maxhistpf", "8", "-bghist", "1"]
    out = {}
    for tag, b in (("ref", REFDEC), ("gpu", exe)):
        hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
        with open(log, "w") as lf:
            args = rm_args(extra)
            if "-maxwpf" in extra:
                i = args.index("-maxwpf", len(extra)); del args[i:i + 2]
            p = subprocess.run([b] + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf,
                               stderr=subprocess.STDOUT, timeout=1800, env=env if tag == "gpu" else None)
        tail = [l for l in open(log, errors="ignore").read().splitlines()
                if l.startswith(("FATAL", "INFO: ")) and ("shim" in l or "SUMMARY" in l or "FATAL" in l)]
        assert p.returncode == 0, "\n".join(tail[-10:])
        out[tag] = (open(hyp).read(), open(seg).read())
        if tag == "gpu" and "-maxhmmpf" in extra:
            n = [int(l.split("applied in")[1].split()[0]) for l in open(log, errors="ignore").read().splitlines()
                 if l.startswith("INFO: ") and "tst shim" in l and "histogram pruning" in l]
            assert n and n[0] > 1000, n
        print("\n".join(t[:220] for t in tail[-3:]))
    assert out["gpu"][0] == out["ref"][0]
    assert out["gpu"][1] == out["ref"][1]
    assert out["ref"][0].count("\n") == 20


# ---------------------------------------------------------------------------
# the SECONDARY boundary: pocketsphinx's ps_mgaufuncs_t vtable (acmod.h:97-115).  oracle/ref_ps_shim.c =
# the unmodified pocketsphinx decoder (fwdtree + fwdflat + bestpath, its own feature computation and
# senone activation lists) with acmod->mgau swapped for an object that forwards frame_eval to
# s3a_ps_ms_cont_mgau_frame_eval: hypotheses AND path scores must equal the unmodified run's.
# ---------------------------------------------------------------------------
PSSHIM = os.path.join(ROOT, "oracle", "_ref", "ref_ps_shim")


def test_pocketsphinx_decoder_with_gpu_scorer_matches_pocketsphinx(tmp_path):
    ctl = tmp_path / "ps.ctl"
    ctl.write_text("".join(l.split()[0] + "\n" for l in open(os.path.join(D, "tidigits.length.arb.regression"))))
    out = {}
    for mode in ("ref", "gpu"):
        o, log = str(tmp_path / f"ps_{mode}.out"), str(tmp_path / f"ps_{mode}.log")
        with open(log, "w") as lf:
            p = subprocess.run([PSSHIM, mode, os.path.join(AM, "mdef"), os.path.join(AM, "means"),
                                os.path.join(AM, "variances"), os.path.join(AM, "mixture_weights"),
                                os.path.join(AM, "transition_matrices"), os.path.join(D, "tidigits.ps.dic"),
                                os.path.join(D, "fillerdict"), os.path.join(D, "tidigits.DMP"), str(ctl),
                                os.path.join(D, "cepstra"), o], stdout=lf, stderr=subprocess.STDOUT, timeout=600)
        tail = [l for l in open(log, errors="ignore").read().splitlines() if "ps shim" in l or "FATAL" in l]
        assert p.returncode == 0, "\n".join(tail[-5:])
        out[mode] = open(o).read()
        if mode == "gpu":
            assert any("frame_eval calls served by" in l for l in tail)
    assert out["gpu"] == out["ref"]
    assert out["ref"].count("\n") == 31 and "ONE ONE ONE (man/man.ah.111a" in out["ref"]


# ---------------------------------------------------------------------------
# BASELINE.json configs[2] / configs[4] as tests: the synthetic hub4-shaped task (6144 senones x 8, 20 000 words,
# ARPA trigram, hub4 beams) and the WSJ-shaped one (8000 x 32) decoded with a wide beam -- thousands of word exits
# and hundreds of thousands of (exit, predecessor) candidates per frame, tied scores in half of the frames --
# through the frame-synchronous drop-in and through whole utterances on the device; expected output live from the
# unmodified reference on the same files.
# ---------------------------------------------------------------------------
def synth_task(kind, tmp_path, n_utt, n_frames, env=None):
    out = subprocess.run([sys.executable, "-m", "cmusphinx_amd.synth_task", kind, str(tmp_path / "task"), f"n_utt={n_utt}",
                          f"n_frames={n_frames}"], check=True, capture_output=True, text=True, cwd=ROOT,
                         env=dict(os.environ, **(env or {}))).stdout
    return out.split(";")[1].split()


def decode_task(exe, args, tmp_path, tag, env=None):
    hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([exe] + args + ["-hyp", hyp, "-hypseg", seg], stdout=lf, stderr=subprocess.STDOUT, timeout=1800,
                           env=dict(os.environ, **(env or {})))
    tail = [l for l in open(log, errors="ignore").read().splitlines() if "tst shim" in l or "FATAL" in l or "SUMMARY" in l]
    assert p.returncode == 0, "\n".join(tail[-10:])
    return open(hyp).read(), open(seg).read(), tail


@pytest.mark.parametrize("driver", ["one_decoder", "utt_1", "utt_4", "utt_3_bigwl", "utt_40", "utt_8_engines2"])
def test_hub4_shaped_full_decode_matches_reference(driver, tmp_path):
    args = synth_task("hub4", tmp_path, 4, 250)
    ref = decode_task(REFDEC, args, tmp_path, "ref")
    got = decode_task(TST, args, tmp_path, driver, DRIVERS[driver])
    assert got[0] == ref[0] and got[1] == ref[1]
    assert ref[0].count("\n") == 4
    cd = [l for l in ref[2] if "SUMMARY" in l]
    assert cd and int(cd[0].split("cdsen/fr")[0].split()[-1]) > 3000       # most of the 6144 senones scored per frame


def test_wsj_shaped_wide_beam_decode_matches_reference(tmp_path):
    """configs[4]: 8000 senones x 32 Gaussians, -beam 1e-120 -pbeam 1e-100 -wbeam 1e-80 -maxhmmpf 100000."""
    args = synth_task("wsj", tmp_path, 2, 60, env=dict(TASK_BEAM="1e-120", TASK_WBEAM="1e-80")) + \
        ["-pbeam", "1e-100", "-maxhmmpf", "100000"]
    ref = decode_task(REFDEC, args, tmp_path, "ref")
    got = decode_task(TST, args, tmp_path, "utt_2", {"S3A_UTT": "2"})
    assert got[0] == ref[0] and got[1] == ref[1]
    wl = [l for l in got[2] if "word level: at most" in l]
    assert wl and int(wl[0].split("at most")[1].split()[0]) > 20000, wl     # a real word-level load
    one = decode_task(TST, args, tmp_path, "one", {})
    assert one[0] == ref[0] and one[1] == ref[1]


def test_live_api_s3_decode_process_with_replacement_slots(tmp_path):
    """The reference's LIVE API (libAPI/s3_decode.c): s3_decode_init, then per utterance s3_decode_begin_utt,
    s3_decode_process on blocks of 37 cepstral frames (feat_s2mfc2feat_live -> utt_decode_block), s3_decode_end_utt,
    s3_decode_hypothesis -- once with the reference's own slots, once with the replacement slots installed right after
    s3_decode_init.  Same word segments (ids, frames, scores), same strings, same -hyp / -hypseg files."""
    out = {}
    for mode in ("cpu", "gpu"):
        hyp, seg = str(tmp_path / f"live_{mode}.match"), str(tmp_path / f"live_{mode}.matchseg")
        p = subprocess.run([TST] + common() + RUNS["mode4_trigram"] + ["-hyp", hyp, "-hypseg", seg], capture_output=True,
                           text=True, errors="ignore", timeout=900, env=dict(os.environ, S3A_LIVE=mode))
        info = [l for l in p.stderr.splitlines() if "tst shim live mode" in l or "FATAL" in l]
        assert p.returncode == 0 and info, p.stderr[-2000:]
        searched = int(info[-1].split("slots,")[1].split()[0])
        assert (searched > 3000) if mode == "gpu" else (searched == 0), info[-1]
        out[mode] = ([l for l in p.stdout.splitlines() if l.startswith("LIVE ")], open(hyp).read(), open(seg).read())
    assert len(out["cpu"][0]) == 31 and out["cpu"][1].count("\n") == 31
    assert out["gpu"] == out["cpu"]


def test_one_process_per_gpu_shards_the_control_file(tmp_path):
    """RANK / WORLD_SIZE / LOCAL_RANK (as torchrun or mpirun set them): rank r decodes the r-th contiguous share of the
    control file on GPU LOCAL_RANK and writes <hyp>.part<r>; the parts in rank order are the one-process files.  Three
    ranks here, all on GPU 0 (the box has one), 31 utterances -> shares of 11, 10, 10."""
    hyp, seg = str(tmp_path / "w.match"), str(tmp_path / "w.matchseg")
    procs = []
    for r in range(3):
        env = dict(os.environ, S3A_UTT="4", WORLD_SIZE="3", RANK=str(r), LOCAL_RANK="0", S3A_NO_RCCL="1")   # (RCCL: one rank per GPU)
        procs.append(subprocess.Popen([TST] + common() + RUNS["mode4_trigram"] + ["-hyp", hyp, "-hypseg", seg],
                                      stdout=subprocess.DEVNULL, stderr=open(tmp_path / f"r{r}.log", "w"), env=env))
    for r, p in enumerate(procs):
        assert p.wait(timeout=900) == 0, open(tmp_path / f"r{r}.log", errors="ignore").read()[-2000:]
    parts = [open(f"{hyp}.part{r:03d}").read() for r in range(3)]
    assert [p.count("\n") for p in parts] == [11, 10, 10]
    assert "".join(parts) == open(os.path.join(D, "ref_mode4_trigram.match")).read()
    assert "".join(open(f"{seg}.part{r:03d}").read() for r in range(3)) == open(os.path.join(D, "ref_mode4_trigram.matchseg")).read()


def test_end_of_batch_exchange_in_c_over_rccl_one_rank(tmp_path):
    """The C side of the multi-GPU flow (s3a_gather_init / s3a_gather_hyps: RCCL loaded at run time, three all-gathers of
    (header, words) records on device buffers, rank 0 writing -hyp / -hypseg through s3a_hyp_format_var) with the one rank
    a one-GPU box allows: S3A_GATHER=1 makes the drop-in write <hyp>.gathered from the gathered records; with -bestpath 1
    the records are the SECOND pass's.  Utterances that cannot be ended get no line either way."""
    for tag, extra in (("fp", []), ("bp", ["-bestpath", "1"])):
        hyp, seg, log = (str(tmp_path / f"{tag}.{e}") for e in ("match", "matchseg", "log"))
        with open(log, "w") as lf:
            p = subprocess.run([TST] + common() + RUNS["mode4_trigram"] + extra + ["-hyp", hyp, "-hypseg", seg], stdout=lf,
                               stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, S3A_UTT="6", S3A_GATHER="1"))
        txt = open(log, errors="ignore").read()
        assert p.returncode == 0, "\n".join(l for l in txt.splitlines() if "FATAL" in l or "tst shim" in l)[-2000:]
        assert "gathered 31 utterances from 1 ranks over RCCL" in txt
        assert open(hyp + ".gathered").read() == open(hyp).read() and open(seg + ".gathered").read() == open(seg).read()
        assert open(hyp).read().count("\n") == 31


def test_an_utterance_that_overflows_its_lane_does_not_take_the_batch_down(tmp_path):
    """A history table far too small for the long utterances (S3A_UTT_VHCAP): the lanes that overflow stop loudly, the other
    utterances of the same batches are decoded, finished and written as always (their lines are the reference's, in order),
    the failed ones get no line -- as an utterance the reference itself fails on -- and the run ends with status 1."""
    hyp, seg, log = (str(tmp_path / f"o.{e}") for e in ("match", "matchseg", "log"))
    with open(log, "w") as lf:
        p = subprocess.run([TST] + common() + RUNS["mode4_trigram"] + ["-hyp", hyp, "-hypseg", seg], stdout=lf,
                           stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, S3A_UTT="4", S3A_UTT_VHCAP="300", S3A_GATHER="1"))
    txt = open(log, errors="ignore").read()
    assert p.returncode == 1, txt[-3000:]
    assert "history table full" in txt and "no hypothesis written" in txt
    ref_h = open(os.path.join(D, "ref_mode4_trigram.match")).read().splitlines()
    ref_s = open(os.path.join(D, "ref_mode4_trigram.matchseg")).read().splitlines()
    got_h, got_s = open(hyp).read().splitlines(), open(seg).read().splitlines()
    assert 0 < len(got_h) < len(ref_h) and len(got_s) == len(got_h)
    it = iter(ref_h)
    assert all(any(l == r for r in it) for l in got_h)              # a subsequence of the reference's lines, in order
    assert set(got_s) <= set(ref_s)
    # the lanes that overflowed were reset: the utterances decoded AFTER a failure on the same lanes are right (above), and
    # the exchange's records carry the failures as status -1 (rank 0 writes no line for them)
    assert open(hyp + ".gathered").read() == open(hyp).read()
