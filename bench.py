#!/usr/bin/env python3
"""bench.py -- BASELINE.json configs[3] as written: a FIXED batch of 1024 distinct synthetic 10 s utterances, hub4-shaped
CD-GMM model, full mode-4 decode, sharded over the GPUs of one node.

A "step" = one pass of the hot path over the batch: every utterance of the rank's share of the 1024-utterance control
list (contiguous -ctloffset/-ctlcount style shards: shard.shard_contiguous; --scaling weak: every rank decodes a whole
batch) is decoded from its first to its last frame on the device -- look-ahead senone scoring of all 6144 senones x 8
Gaussians x 39 (one model pass per 8 frames of all lanes), the CI gate, lextree HMM evaluation over three unigram + three
filler lextrees of a 20 000-word dictionary, histogram and beam pruning, phone-level propagation, the word level (trigram
look-ups, Viterbi history, word pruning, word transitions): s3a_uttdec_decode_dev, no host work inside an utterance --
then the hypotheses (s3a_uttdec_hyp_var: final </s> transition + backtrace) and ONE exchange of them across the ranks.
Inside a rank the share is dealt, longest first, to E engines (own stream + host thread each, their launch tails overlap);
an engine decodes its share as ONE queue with lane refill (s3a_uttdec_decode_queue_dev: a lane takes the next utterance
when its own has ended); a share of no more utterances than lanes (--refill 0: any share) runs as groups of similar
length in lock step.  Features are resident in HBM before the timed region; history tables and hypotheses come back inside it.
Arithmetic is the bit-exact mode (float32 subtract, float64 accumulate, int32 log-add).  After the timed region rank 0
formats the LAST step's gathered hypotheses and compares every -hyp / -hypseg line with the unmodified reference
decoder's (CPU, same files); a difference is an error, not a number.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The decoder is rebuilt from a bundle (cmusphinx_amd/bundle.py) through the C ABI; the sphinx3 side of the drop-in
(oracle/_ref/ref_s3amd_tst_decode = integration/sphinx3 + the unmodified reference's kb_init) runs once, UNTIMED, to load
the models / dictionary / LM, build the lextrees and write that bundle.

Prints ONE JSON line (rank 0).  Extra keys beyond the driver contract:
  roofline      the dominant kernel of a frame IN THE BENCH (HIP events around every launch of every 8th frame of every engine
                while all engines run; the look-ahead scoring launch counted per frame it serves): achieved = algorithmic
                bytes (or flops) per launch / average launch time; `alone` = the same with one engine on the chip; traffic =
                HBM bytes per launch from the committed PMC pass (profiles/)
  roofline_scoring  the engine's scoring kernel (north_star's second metric is about scoring): VALU and HBM fractions
  search        SURVEY 8(d)'s 84 B per active HMM credited ONCE per frame to the search kernels together; HMM updates/s
  kernels       every kernel class of a frame: microseconds per launch in the bench and alone, stretch, share of the frame
  wide_beam     configs[4]: the WSJ-shaped 8000 x 32 model with the wide beam in a 64-lane engine: frames/s, active HMMs and
                candidates per frame, per-kernel microseconds, the LDS / VGPR occupancy of the word-level kernels, the
                first 8 utterances compared with the unmodified reference
  ps_fwdtree    SURVEY 8(f).3: pocketsphinx's first pass on the device (s3a_psfwd_decode: scoring + search, one workgroup per
                utterance) on the same model shape, with the unmodified pocketsphinx on the host beside it
  cpu_baseline  the UNMODIFIED reference decoder (oracle/_ref/sphinx3_decode) on the same task files: 16 processes (where
                the host's aggregate peaks) over contiguous control-file shards of the batch's first 128 utterances, and
                one process alone with stat.c's sen / search / tot split
  strong_scaling_projection   what one GPU does with the 128 utterances an 8-GPU run gives it
  scoring       configs[1]: whole-utterance senone scoring and the frame-synchronous pass (1 / 2 / 8 frames per model
                pass) on the hub4 and the configs[4] (8000 x 32) model shapes, SURVEY 8(d) accounting
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VEC_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (FMA counted as 2 flops)
REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
SHIM = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # tools/gpu_round6.sh: HBM bytes per lane-frame of ku_frames (PMC passes)


def csrc_hash():
    """what ties profiles/pmc_traffic.json to the library that runs: sha256 over the kernels' sources (cmusphinx_amd/csrc/*.hip, *.h, *.c) --
    the PMC passes' script stamps the file with it, and a file whose stamp is another source's is not used"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cmusphinx_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h", ".c")):
            h.update(n.encode()); h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def run_reference(task_args, ctl, offset, count, d, tag):
    """the unmodified reference decoder on `count` utterances from `offset` (a process, not waited for)"""
    args = list(task_args)
    args[args.index("-ctl") + 1] = ctl
    hyp, seg, log = (os.path.join(d, f"{tag}.{e}") for e in ("match", "seg", "log"))
    with open(log, "w") as lf:
        p = subprocess.Popen([REFDEC] + args + ["-ctloffset", str(offset), "-ctlcount", str(count), "-hyp", hyp, "-hypseg", seg],
                             stdout=lf, stderr=subprocess.STDOUT)
    return p, hyp, seg, log


SUMMARY = re.compile(r"SUMMARY:\s+(\d+) fr;.*?([0-9.]+) xCPU\s+([0-9.]+) xClk \[Ovhrd.*?\];\s+(\d+) hmm/fr, \d+ wd/fr, ([0-9.]+) xCPU\s+([0-9.]+) xClk;"
                     r"\s+tot:\s+([0-9.]+) xCPU,\s+([0-9.]+) xClk")


def parse_summary(log):
    """stat.c:201-228 -> dict(frames, hmm, sen/search/tot xCPU and xClk)"""
    m = SUMMARY.search(open(log, errors="ignore").read())
    if not m:
        return None
    return dict(frames=int(m.group(1)), sen_xcpu=float(m.group(2)), sen_xclk=float(m.group(3)), hmm=int(m.group(4)),
                srch_xcpu=float(m.group(5)), srch_xclk=float(m.group(6)), tot_xcpu=float(m.group(7)), tot_xclk=float(m.group(8)))


def shards(n, parts):
    base, extra = divmod(n, parts)
    out, lo = [], 0
    for r in range(parts):
        c = base + (1 if r < extra else 0)
        if c:
            out.append((lo, c))
        lo += c
    return out


def cpu_batch(targs, ctl, d, n_utt, n_procs, tag):
    """the reference over utterances [0, n_utt) as n_procs concurrent processes on contiguous shards -> (aggregate
    frames/s over the decoding time of the slowest process, summaries, hyp text, hypseg text)"""
    ps = [run_reference(targs, ctl, lo, c, d, f"{tag}{i}") for i, (lo, c) in enumerate(shards(n_utt, n_procs))]
    sums, hyp, seg = [], [], []
    for q, h, s, lg in ps:
        q.wait()
        sm = parse_summary(lg)
        if q.returncode != 0 or not sm:
            return None
        sums.append(sm)
        hyp.append(open(h).read())
        seg.append(open(s).read())
    frames = sum(s["frames"] for s in sums)
    wall = max(s["frames"] * s["tot_xclk"] / 100.0 for s in sums)          # seconds of decoding of the slowest shard
    return frames / max(wall, 1e-9), sums, "".join(hyp), "".join(seg)


def cpu_baseline(targs, ctl, d, n_utt, n_procs, check_all, physical_leg):
    """sphinx3_decode on the identical files.  The reference is single-threaded, so N processes decode
    contiguous control-file shards; this host does not scale to its core count (tools/cpu_scaling.py, profiles/: 16
    processes 6.6 k frames/s, 32: 6.2 k, 64: 5.4 k, 128: 3.6 k -- every process streams the 15.7 MB model per frame), so
    `value` is the best-N leg: n_procs (default 16) processes x 8 utterances of the batch, a bounded sample (~30 s).
    single_core = one process alone.  --cpu-physical adds the N = physical-cores leg (one utterance each, ~2 min);
    --check-all lets the reference decode the REST of the batch too (unmeasured, ~3 min) so that every utterance of the
    device's output is compared."""
    one = cpu_batch(targs, ctl, d, min(2, n_utt), 1, "cpu1_")
    if not one:
        return None, None
    s1 = one[1][0]
    n_smp = min(n_utt, 8 * n_procs)
    big = cpu_batch(targs, ctl, d, n_smp, n_procs, "cpuN_")
    if not big:
        return None, None
    hyp, seg, n_chk = big[2], big[3], n_smp
    if check_all and n_smp < n_utt:
        args2 = list(targs)
        rest = [run_reference(targs, ctl, n_smp + lo, c, d, f"cpuR{i}_") for i, (lo, c) in enumerate(shards(n_utt - n_smp, n_procs))]
        for q, h, sg, _ in rest:
            q.wait()
            if q.returncode != 0:
                return None, None
            hyp += open(h).read()
            seg += open(sg).read()
        n_chk = n_utt
    out = {"value": round(big[0], 1), "unit": "frames/s", "cores": len(big[1]), "kind": "reference",
           "physical_cores_on_host": physical_cores(), "aggregate_xRT": round(big[0] / 100.0, 1),
           "per_core_in_batch": round(big[0] / len(big[1]), 1),
           "single_core": round(100.0 / max(s1["tot_xcpu"], 1e-9), 1), "single_core_xRT": round(1.0 / max(s1["tot_xcpu"], 1e-9), 2),
           "single_core_split_xCPU": {"sen": s1["sen_xcpu"], "search": s1["srch_xcpu"], "tot": s1["tot_xcpu"]},
           "single_core_split_xClk": {"sen": s1["sen_xclk"], "search": s1["srch_xclk"], "tot": s1["tot_xclk"]},
           "batch_split_xClk_mean": {k: round(float(np.mean([s[k + "_xclk"] for s in big[1]])), 3) for k in ("sen", "srch", "tot")},
           "active_hmm_per_frame": s1["hmm"], "utterances": n_smp, "frames": sum(s["frames"] for s in big[1]),
           "sample": f"unmodified oracle/_ref/sphinx3_decode (gcc -O2), full mode-4 decode of the same task files: value = the first "
                     f"{n_smp} utterances of the batch as {len(big[1])} concurrent processes over contiguous -ctloffset/-ctlcount shards, "
                     f"frames / decoding time of the slowest shard (stat.c SUMMARY xClk; model loading excluded); {len(big[1])} processes "
                     f"is where this host's aggregate peaks (profiles/r3_cpu_scaling.txt: more processes decode FEWER frames/s); "
                     f"single_core = one process alone on 2 utterances ({s1['frames']} frames), with stat.c's sen / search / tot split"}
    if physical_leg:
        pc = physical_cores()
        phys = cpu_batch(targs, ctl, d, min(n_utt, pc), pc, "cpuP_")
        if phys:
            out["procs_physical_cores"] = {"procs": len(phys[1]), "value": round(phys[0], 1), "xRT": round(phys[0] / 100.0, 1),
                                           "per_core": round(phys[0] / len(phys[1]), 1)}
    return out, (hyp, seg, n_chk)


def scoring_legs(lib, fast):
    """configs[1] (+ the configs[4] model shape): senone scoring only, features and scores resident in HBM.
    SURVEY 8(d): algorithmic bytes of a launch = the Gaussians' parameters ONCE per model pass + per frame the vector in
    and the scores out; 4 non-fusable flops per Gaussian-dimension."""
    from cmusphinx_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    out = {}
    lm = lib.LogMath(1.0003)
    for name, shape, T in (("hub4", synth.HUB4, 1000), ("wsj_shape", synth.WSJ_STRESS, 200)):
        model = synth.make_model(**shape)
        gm = lib.MgauModel.init_arrays(model["mean"], model["var"], model["mixw"], lm)
        if fast:
            gm.set_precision(lib.GMM_FAST)
        S, Cc, D = gm.S, gm.C, gm.D
        f = synth.make_features(model, T, seed=7)
        fd = lib.DevBuf(f.nbytes).upload(f)
        sd = lib.DevBuf(T * S * 4)
        bd = lib.DevBuf(T * 4)
        pick = [0, T - 1]
        got = gm.score_frames(f[pick], want_best=False)
        exp = O.OracleMgau(model["mean"], model["var"], model["mixw"], O.OracleLogMath(1.0003)).score_all(f[pick])
        assert (np.abs(got.astype(np.int64) - exp).max() <= 2) if fast else np.array_equal(got, exp), "HIP scores differ from the oracle"
        model_bytes = S * Cc * (2 * D + 2) * 4
        frame_bytes = D * 4 + S * 4
        flops_frame = 4.0 * S * Cc * D
        leg = {"shape": f"{S} senones x {Cc} Gaussians x {D}", "model_bytes_per_pass": model_bytes, "bytes_per_frame": frame_bytes}

        def entry(B, us, extra=None):
            alg = model_bytes + B * frame_bytes
            gbs = alg / (us * 1e-6) / 1e9
            tfl = B * flops_frame / (us * 1e-6) / 1e12
            e = {"frames_per_launch": B, "avg_launch_us": round(us, 3), "algorithmic_bytes_per_launch": alg,
                 "hbm": {"achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4)},
                 "valu": {"achieved_TFLOPs": round(tfl, 2), "peak_TFLOPs": FP64_VEC_PEAK_TFLOPS, "frac": round(tfl / FP64_VEC_PEAK_TFLOPS, 4)}}
            e["bound"] = "hbm" if e["hbm"]["frac"] >= e["valu"]["frac"] else "valu-f64"
            e["frac"] = max(e["hbm"]["frac"], e["valu"]["frac"])
            if extra:
                e.update(extra)
            return e
        if name == "hub4":
            for _ in range(3):
                gm.score_frames_dev(fd, T, sd, bd)
            lib.check(lib.load().s3a_dev_sync())
            gm.timer_begin()
            K = 20
            for _ in range(K):
                gm.score_frames_dev(fd, T, sd, bd)
            k_us = gm.timer_end() / K
            leg["whole_utterance"] = entry(T, k_us, {"kernel": "k_score_frames<8,exact,lds-table,512>", "frames_per_sec": round(T / (k_us * 1e-6), 1)})
        for B in (1, 2, 8):
            gm.bench(fd, T, sd, None, B, 1)
            b_us, b_kus, b_n = gm.bench(fd, T, sd, None, B, 3)
            leg["frame_sync" if B == 1 else f"frame_sync_b{B}"] = entry(B, b_kus, {"launches": b_n, "frames_per_sec": round(T / (b_us * 1e-6), 1)})
        out[name] = leg
        del gm, fd, sd, bd
    return out


def wide_beam_leg(lib, d, lanes, n_frames, fast, n_check=64):
    """BASELINE configs[4]: the WSJ-shaped model (8000 senones x 32 Gaussians, 20 k-word trigram) decoded with the wide
    beam -beam 1e-120 -pbeam 1e-100 -wbeam 1e-80 -maxhmmpf 100000 ("stress active-HMM count and LDS occupancy"):
    `lanes` utterances in ONE engine, whole utterances on the device; the first n_check utterances compared, -hyp and
    -hypseg line by line, with the unmodified reference (n_check single-utterance processes side by side)."""
    from cmusphinx_amd import bundle, s3io, synth_task
    t = os.path.join(d, "wsjtask")
    synth_task.make_task(t, n_utt=lanes, n_frames=n_frames, **synth_task.WSJ_TASK)
    targs = synth_task.decoder_args(t, beam="1e-120", wbeam="1e-80") + ["-pbeam", "1e-100", "-maxhmmpf", "100000"]
    bpath = os.path.join(t, "decoder.bundle")
    r = subprocess.run([SHIM] + targs, env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bpath), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if r.returncode != 0 or not os.path.exists(bpath):
        return {"error": "bundle export failed"}
    ctl = os.path.join(t, "ctl")
    utts = [l.split()[0] for l in open(ctl) if l.strip()]
    hfeat = [s3io.read_mfc(os.path.join(t, "feat", u + ".mfc")) for u in utts]
    dec = bundle.Decoder(bpath, lanes, precision=lib.GMM_FAST if fast else lib.GMM_EXACT, max_frames=max(len(f) for f in hfeat) // 39 + 8)
    D4x4 = 4 * ((dec.veclen + 3) // 4)
    fdev, nfr = [], []
    for f in hfeat:
        f = f.reshape(-1, dec.veclen)
        pad = np.zeros((len(f), D4x4), np.float32)
        pad[:, :dec.veclen] = f
        fdev.append(lib.DevBuf(pad.nbytes).upload(pad))
        nfr.append(len(f))
    dec.ud.decode_dev(fdev, nfr, D4x4)                      # warm-up
    ms = dec.ud.decode_dev(fdev, nfr, D4x4)
    recs = [dec.hyp_var(z, utts[z], z) for z in range(len(utts))]
    stat = [dec.ud.result(z)["frame_stat"] for z in range(len(utts))]
    res0 = [dec.ud.result(z) for z in range(min(4, len(utts)))]
    dec.ud.set_profile(4)
    dec.ud.decode_dev(fdev, nfr, D4x4)
    prof = dec.ud.profile()
    dec.ud.set_profile(0)
    # (the reference's processes only now: 64 of them beside the timed decode took the launching thread's cores -- the launch path enqueues
    # ~20 launches per frame -- and cost the leg up to a fifth of its rate from run to run)
    refs = [run_reference(targs, ctl, i, 1, t, f"wref{i}_") for i in range(min(n_check, len(utts)))]
    K = max(1, dec.ud.window())
    pf = {k: (us / n / (K if k == "ku_score_window" else 1)) for k, (us, n) in prof.items() if n > 0}
    tot = sum(pf.values())
    frames = sum(nfr)
    n_ok = n_cmp = 0
    cpu_fps = None
    cpu_frames, cpu_wall = 0, 0.0
    for i, (q, h, sg, lg) in enumerate(refs):
        q.wait()
        sm = parse_summary(lg)
        if q.returncode != 0 or not sm:
            continue
        m_, s_ = dec.format_var(*recs[i])
        n_cmp += 1
        n_ok += int(open(h).read() == m_ and open(sg).read() == s_)
        cpu_frames += sm["frames"]
        cpu_wall = max(cpu_wall, sm["frames"] * sm["tot_xclk"] / 100.0)
    if n_cmp:
        cpu_fps = cpu_frames / max(cpu_wall, 1e-9)
    resrc = json.load(open(os.path.join(ROOT, "profiles", "r4_kernel_resources.json"))) if os.path.exists(os.path.join(ROOT, "profiles",
            "r4_kernel_resources.json")) else {}
    hmm = np.concatenate([s_[:, 1] for s_ in stat])
    out = {"workload": f"configs[4]: WSJ-shaped 8000 senones x 32 Gaussians x 39, 20 k-word dictionary, ARPA trigram, -beam 1e-120 -pbeam 1e-100 "
                       f"-wbeam 1e-80 -maxhmmpf 100000; {len(utts)} utterances in one {lanes}-lane engine, whole utterances on the device",
           "lanes": lanes, "frames": int(frames), "device_ms": round(ms, 2), "frames_per_sec": round(frames / (ms * 1e-3), 1),
           "xRT": round(frames / (ms * 1e-3) / 100.0, 1),
           "per_frame": {"active_hmm_mean": round(float(hmm.mean()), 1), "active_hmm_max": int(hmm.max()),
                         "word_exits_mean": round(float(np.mean([s_[:, 7].mean() for s_ in stat])), 1),
                         "max_candidates_per_frame": int(max(r_["max_cand"] for r_ in res0)),
                                 "max_new_history_entries_per_frame": int(max(r_["max_new"] for r_ in res0)),
                         "frames_with_histogram_pruning": int(sum(int(s_[:, 6].sum()) for s_ in stat))},
           "kernels_us_per_frame": {k: round(v, 2) for k, v in sorted(pf.items(), key=lambda kv: -kv[1])}, "us_per_frame_all_lanes": round(tot, 1),
           "occupancy": {k: resrc.get(k) for k in ("ku_emit_word", "ku_hist_sort<256>", "ku_hist_sort<1024>") if k in resrc},
           "occupancy_note": "from the code objects (profiles/r4_kernel_resources.json): ku_emit_word = 512 threads x 88 VGPRs + 53 280 B LDS per workgroup "
                             "(LDS admits 3 workgroups per CU, the VGPRs 5 waves per SIMD = 20 waves = 2 workgroups of 8: TWO resident workgroups per CU; "
                             "with 1024 threads, until round 4, it was one); "
                             "ku_hist_sort<1024> 44 076 B LDS, <256> 20 028 B",
           "identical_to_reference": {"utterances_checked": n_cmp, "identical": n_ok},
           "cpu_reference": {"frames_per_sec": round(cpu_fps, 1) if cpu_fps else None, "processes": n_cmp, "frames": cpu_frames, "kind": "reference",
                             "note": "the unmodified sphinx3_decode, one single-utterance process per checked utterance side by side (stat.c xClk "
                                     "of the slowest)"}}
    assert n_cmp == 0 or n_ok == n_cmp, "wide-beam decode on the device differs from the unmodified reference"
    return out


PSREF = os.path.join(ROOT, "oracle", "_ref", "ref_ps_fwd")
PSAMD = os.path.join(ROOT, "oracle", "_ref", "ref_ps_amdfwd")


def ps_fwdtree_leg(t, lanes, n_cpu=128, n_proc=16):
    """pocketsphinx's first pass on the device, measured on configs[3]'s OWN batch: the task directory t (the
    1024 utterances the main line decodes; its phone names are in the order pocketsphinx's mdef reader insists on) decoded by
    ref_ps_amdfwd -batch <lanes> -queue yes (integration/pocketsphinx/ps_search_amd.c: features by the decoder's feat_t, then
    s3a_psfwd_decode_queue = every senone score of the batch, then ONE launch in which `lanes` persistent workgroups take
    utterance after utterance; hypotheses made on the device); the unmodified pocketsphinx (ref_ps_fwd, one host core)
    decodes the first n_cpu of them for the comparison and the CPU rate."""
    from cmusphinx_amd import synth_task
    if not (os.path.exists(PSREF) and os.path.exists(PSAMD)):
        return None
    d = t
    args = synth_task.ps_decoder_args(t)
    out = {}

    def run(exe, extra, tag, ctl=None):
        a = list(args)
        if ctl:
            a[a.index("-ctl") + 1] = ctl
        m, sg, lg = (os.path.join(d, f"ps_{tag}.{e}") for e in ("match", "seg", "log"))
        with open(lg, "w") as lf:
            r = subprocess.run([exe] + a + extra + ["-hyp", m, "-hypseg", sg], stdout=lf, stderr=subprocess.STDOUT)
        return r.returncode, open(m).read() if os.path.exists(m) else "", open(sg).read() if os.path.exists(sg) else "", open(lg, errors="ignore").read()
    # the unmodified pocketsphinx on the first n_cpu utterances: n_proc single-core processes side by side, a slice of the list each
    lines = open(os.path.join(t, "ctl")).readlines()[:n_cpu]
    n_cpu = len(lines)
    from concurrent.futures import ThreadPoolExecutor
    cuts = [(i * n_cpu // n_proc, (i + 1) * n_cpu // n_proc) for i in range(n_proc)]
    cuts = [c_ for c_ in cuts if c_[1] > c_[0]]

    def ref_slice(i):
        cf = os.path.join(d, f"ps_ctl_cpu{i}")
        with open(cf, "w") as f:
            f.writelines(lines[cuts[i][0]:cuts[i][1]])
        return run(PSREF, ["-fresh", "yes"], f"ref{i}", cf)
    with ThreadPoolExecutor(len(cuts)) as ex:
        refs_ = list(ex.map(ref_slice, range(len(cuts))))
    if any(r_[0] != 0 for r_ in refs_):
        return {"error": "the unmodified pocketsphinx failed on the task"}
    rm, rs = "".join(r_[1] for r_ in refs_), "".join(r_[2] for r_ in refs_)
    cpus = [re.search(r"decoded (\d+) frames in ([0-9.]+) s", r_[3]) for r_ in refs_]
    cpu = cpus[0]
    rc, am, asg, alog = run(PSAMD, ["-fresh", "yes", "-batch", str(lanes), "-queue", "yes"], "amd")
    if rc != 0:
        return {"error": "ref_ps_amdfwd failed: " + " | ".join(l for l in alog.splitlines() if "ERROR" in l or "FATAL" in l)[-300:]}
    dev = re.search(r"batch of (\d+) utterances, (\d+) frames: ([0-9.]+) ms on the device", alog)
    same_h = "".join(am.splitlines(keepends=True)[:n_cpu]) == rm
    same_s = "".join(asg.splitlines(keepends=True)[:n_cpu]) == rs
    tree = re.search(r"(\d+) roots, (\d+) interior channels, (\d+) single-phone words", alog)
    out = {"workload": f"configs[3]'s batch through pocketsphinx: hub4-shaped CD-GMM 6144 x 8 x 39, 20 k-word dictionary, ARPA trigram, "
                       f"pocketsphinx's default beams, first pass only; {dev.group(1)} utterances as ONE queue over {lanes} lanes "
                       f"(one persistent workgroup per lane)",
           "lanes": lanes, "utterances": int(dev.group(1)), "frames": int(dev.group(2)), "device_ms": float(dev.group(3)),
           "frames_per_sec": round(int(dev.group(2)) / (float(dev.group(3)) * 1e-3), 1),
           "xRT": round(int(dev.group(2)) / (float(dev.group(3)) * 1e-3) / 100.0, 1),
           "identical_to_pocketsphinx": {"hyp_and_score": same_h, "segmentation": same_s, "utterances_checked": n_cpu},
           "search_space": {"roots": int(tree.group(1)), "interior_channels": int(tree.group(2)), "single_phone_words": int(tree.group(3))} if tree else None,
           "cpu_pocketsphinx": {"frames": int(cpu.group(1)), "seconds": float(cpu.group(2)),
                   "frames_per_sec": round(int(cpu.group(1)) / max(float(cpu.group(2)), 1e-9), 1),
                                "cores": 1, "kind": "reference", "processes_side_by_side": len(cuts),
                                "aggregate_frames_per_sec": round(sum(int(c_.group(1)) for c_ in cpus if c_)
                                                                   / max(max(float(c_.group(2)) for c_ in cpus if c_), 1e-9), 1)} if cpu else None,
           "timed": "HIP events on the engine's stream around the batch's scoring launch + the search launch (features resident in HBM)"}
    sc = re.search(r"of which scoring ([0-9.]+) ms", alog)
    if sc and float(sc.group(1)) > 0:
        # the two kernels of the leg: k_ps_cont_tr (float32 without FMA, two frames per packed operation: the packed non-FMA rate is
        # half of MI355X_MICROARCH.md's 157.3 TFLOP/s, which counts an FMA as two) and k_psf_queue (the search: one workgroup per lane)
        fr, sms, tms = int(dev.group(2)), float(sc.group(1)), float(dev.group(3))
        flop = fr * 6144 * 8 * 39 * 4.0
        pmc = os.path.join(ROOT, "profiles", "r6_pmc_ps.json")
        traffic = None
        if os.path.exists(pmc):
            k = [v for n, v in json.load(open(pmc))["kernels"].items() if n.startswith("k_psf_queue<3")]
            traffic = k[0]["bytes_per_frame"] if k else None
        srch_ms = tms - sms
        alg = 1700 * 64 * 2 + 2 * 1500
        out["roofline_scoring"] = {"kernel": "k_ps_cont_tr<8,39>", "bound": "valu-f32 (packed, no FMA)", "avg_launch_us": round(sms * 1e3, 1),
                                   "achieved": round(flop / (sms * 1e-3) / 1e12, 2), "peak": 78.6, "unit": "TFLOP/s",
                                   "frac": round(flop / (sms * 1e-3) / 1e12 / 78.6, 4),
                                   "algorithmic_flop_per_frame": 6144 * 8 * 39 * 4}
        out["roofline_search"] = {"kernel": "k_psf_queue<3>", "bound": "hbm (64-byte channel records visited at random) / latency of a lane's phases",
                                  "avg_launch_us": round(srch_ms * 1e3, 1), "us_per_lane_frame": round(srch_ms * 1e3 * lanes / fr, 1),
                                  "algorithmic_bytes_per_frame": alg, "achieved": round(fr * alg / (srch_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                                  "unit": "GB/s", "frac": round(fr * alg / (srch_ms * 1e-3) / 1e9 / 8000.0, 4),
                                  "traffic_bytes_per_frame": traffic,
                                  "traffic_source": "profiles/r6_pmc_ps.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this very regime: 1024 utterances "
                                                    "as one queue over 512 lanes, tools/ps_pmc_regime.py)" if traffic else None,
                                  "note": "algorithmic = ~1 700 active channels per frame (pocketsphinx's own count: 1 191 - 2 270 per frame) x a 64-byte "
                                          "record read and written + the active senones' int16 scores"}
    assert same_h and same_s, "pocketsphinx first pass on the device differs from the unmodified pocketsphinx"
    return out


def exchange_run_id(dist, rank, tdev):
    """the run id of the C exchange's rendezvous (s3a_gather_init_run): the launcher's S3A_RUN_ID (launch_ranks draws one per launch);
    under a foreign launcher (torch.distributed.run: only MASTER_PORT, the same from run to run) rank 0 draws a number and
    broadcasts it over the process group that supplies the barrier; one rank alone: its pid"""
    if os.environ.get("S3A_RUN_ID", "").isdigit() and int(os.environ["S3A_RUN_ID"]) > 0:
        return int(os.environ["S3A_RUN_ID"])
    if dist is None:
        return os.getpid()
    import torch
    t = torch.tensor([(time.time_ns() ^ (os.getpid() << 40)) & ((1 << 62) - 1) or 1 if rank == 0 else 0], dtype=torch.int64, device=tdev)
    dist.broadcast(t, 0)
    return int(t.item())


def launch_ranks(n, cmd=None):
    """`python bench.py --gpus N` without a launcher around it: N ranks of this very command, one per GPU, the way the reference shards a
    control file over processes (-ctloffset / -ctlcount, main_decode.c:164-169).  The children find RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment exactly as under torch.distributed.run; rank 0's stdout (the ONE JSON line) is this
    process's.  S3A_BENCH_ONE_GPU=1: every rank on GPU 0 (the rehearsal on a one-GPU box)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    run_id = (time.time_ns() ^ (os.getpid() << 40)) & ((1 << 62) - 1) or 1     # (the exchange's rendezvous: a number of THIS launch, not the port)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), S3A_RUN_ID=str(run_id), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd if cmd is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p_.poll() is None for p_ in procs):
            time.sleep(0.2)
            bad = [p_.returncode for p_ in procs if p_.poll() not in (None, 0)]
            if bad:                 # (a rank that failed must not leave the others waiting at a barrier for ever)
                rc = bad[0]
                break
        rc = rc or next((p_.returncode for p_ in procs if p_.returncode), 0)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=1024, help="distinct synthetic utterances of the batch (configs[3]: 1024)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the fixed batch is split over the ranks (configs[3] as written); weak: every rank decodes a whole batch")
    ap.add_argument("--lanes", type=int, default=512, help="decoder lanes per GPU (utterances in flight)")
    ap.add_argument("--engines", type=int, default=1, help="decoder engines per GPU (own stream each) the lanes are split over (1 since round 5: "
                                                           "ku_frames keeps every lane on its own workgroup, so one engine fills the chip; the "
                                                           "launch path of rounds 2-4 wanted 4)")
    ap.add_argument("--min-group", type=int, default=32, help="a rank's share is cut into groups of at least this many utterances")
    ap.add_argument("--group-fixed", type=int, default=16,
            help="small shares: the per-frame cost of a group in lane equivalents (sizes the groups so that the engines finish together)")
    ap.add_argument("--refill", type=int, default=1, help="1: a share of more utterances than lanes is ONE queue per engine, a lane takes the next "
                                                          "utterance when its own has ended (s3a_uttdec_decode_queue_dev); 0: groups of similar "
                                                          "length, lanes in lock step")
    ap.add_argument("--frames", type=int, default=1000, help="nominal frames per utterance (10 s)")
    ap.add_argument("--cand-cap", type=int, default=0, help="word-level candidate capacity per lane (0: the library's default, 1 << 20)")
    ap.add_argument("--cpu-procs", type=int, default=16, help="processes of the CPU baseline's batch leg (16: where this host's aggregate peaks)")
    ap.add_argument("--no-cpu", action="store_true", help="no CPU baseline: the reference decodes (and checks) 2 utterances only")
    ap.add_argument("--check-all", action="store_true", help="the reference decodes the WHOLE batch (~3 more minutes of CPU): every utterance is compared")
    ap.add_argument("--cpu-physical", action="store_true", help="add the CPU leg with one process per physical core (~2 minutes)")
    ap.add_argument("--no-scoring", action="store_true", help="skip the scoring-only extra legs")
    ap.add_argument("--plain", action="store_true", help="the timed steps and the in-bench kernel timing only (no single-engine profile, no 8-GPU "
                                                         "projection, no extra legs): the command rocprofv3 wraps for profiles/, so that its "
                                                         "per-kernel averages are those of the bench's own regime")
    ap.add_argument("--no-ps", action="store_true", help="skip the pocketsphinx first-pass leg")
    ap.add_argument("--no-wide-beam", action="store_true", help="skip the configs[4] wide-beam leg")
    ap.add_argument("--wide-lanes", type=int, default=64, help="lanes of the wide-beam leg's engine")
    ap.add_argument("--wide-frames", type=int, default=40,
            help="nominal frames per utterance of the wide-beam leg (utterances follow LM sentences: about twice that)")
    ap.add_argument("--ps-lanes", type=int, default=512,
            help="lanes (persistent one-workgroup decoders, two per CU) the pocketsphinx leg runs the batch through as one queue")
    ap.add_argument("--only-scoring", action="store_true", help="only the scoring legs (PMC passes over the scoring kernels)")
    ap.add_argument("--fast", action="store_true", help="S3A_GMM_FAST (f32, +-2 logs3 units) instead of bit-exact")
    ap.add_argument("--variant", action="append", default=[], metavar="FIELD=INT",
            help="a field of s3a_variants_t (s3a_set_variants), e.g. hist_sort_launch=1: A/B runs of kernel variants")
    args = ap.parse_args()
    if args.plain:
        args.no_cpu = args.no_scoring = args.no_ps = args.no_wide_beam = True
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: start as many ranks as --gpus says")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # S3A_BENCH_ONE_GPU=1: every rank on GPU 0, the gather over gloo -- a rehearsal of the multi-rank flow (sharding,
    # barriers, max over ranks, the gather, rank 0's files) on a one-GPU box; RCCL cannot put two ranks on one GPU
    rehearsal = world > 1 and os.environ.get("S3A_BENCH_ONE_GPU") == "1"
    tdev = "cpu" if rehearsal else f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        if rehearsal:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cmusphinx_amd import bundle, lib, s3io, shard, synth_task
    L = lib.load()
    if args.variant:
        lib.set_variants(**{kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.variant})
    if lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libcmusphinx_amd has no CPU fallback")
    lib.check(L.s3a_set_device(local_rank))
    if args.only_scoring:
        print(json.dumps({"scoring": scoring_legs(lib, args.fast)}))
        return
    if not (os.path.exists(SHIM) and os.path.exists(REFDEC)):
        raise SystemExit("bench.py needs oracle/_ref (the reference build: kb_init loads the models); make -C oracle ref")

    # ---------------- setup (untimed): the task, the bundle, the decoder, features into HBM ----------------
    T, U, NL = args.frames, args.utts, args.lanes
    tag = os.environ.get("MASTER_PORT", str(os.getpid()))
    d = os.path.join(tempfile.gettempdir(), f"s3a_bench_{tag}")
    bpath = os.path.join(d, "decoder.bundle")
    t_setup = time.perf_counter()
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        if os.path.exists(os.path.join(d, "rccl-id")):
            os.remove(os.path.join(d, "rccl-id"))
        synth_task.make_task(d, n_utt=U, n_frames=T, sorted_names=True, **synth_task.HUB4_TASK)
        r = subprocess.run([SHIM] + synth_task.decoder_args(d), env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bpath),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert r.returncode == 0 and os.path.exists(bpath), "bundle export failed"
    if dist is not None:
        dist.barrier()
    targs = synth_task.decoder_args(d)
    ctl = os.path.join(d, "ctl")
    utts = [l.split()[0] for l in open(ctl) if l.strip()]
    hfeat = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")) for u in utts]
    t_load = time.perf_counter()
    NE = max(1, args.engines)
    assert NL % NE == 0, "--lanes must be a multiple of --engines"
    NLE = NL // NE                                  # lanes per engine
    decs = [bundle.Decoder(bpath, NLE, precision=lib.GMM_FAST if args.fast else lib.GMM_EXACT, cand_cap=args.cand_cap,
                           max_frames=max(len(f) for f in hfeat) // 39 + 8) for _ in range(NE)]
    dec = decs[0]
    t_load = time.perf_counter() - t_load
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(NE)
    D4x4 = 4 * ((dec.veclen + 3) // 4)
    fdev, nfr = [], []
    for f in hfeat:
        f = f.reshape(-1, dec.veclen)
        pad = np.zeros((len(f), D4x4), np.float32)
        pad[:, :dec.veclen] = f
        fdev.append(lib.DevBuf(pad.nbytes).upload(pad))
        nfr.append(len(f))

    # ---------------- the schedule: this rank's share, sorted by length, in groups of similar length ----------------
    def schedule(ids):
        """-> per engine the list of groups (utterance ids) it decodes one after the other.  Lanes run in lock step, so a
        group lasts as long as its longest utterance: groups hold utterances of similar length; groups are dealt to the
        engines longest first, to the engine with the least frames so far."""
        order = sorted(ids, key=lambda k: (-nfr[k], k))
        if args.refill and len(order) > NE * NLE:
            # lane refill: every engine one queue, longest first (the lanes stay busy to the end); dealt like cards so that
            # the engines' queues hold the same mix of lengths
            return [[order[e::NE]] if order[e::NE] else [] for e in range(NE)]
        if args.min_group <= len(order) <= NE * NLE and NE > 1:
            # a small share (one group per engine, e.g. 128 utterances on one of 8 GPUs): the engines run side by side, so
            # what counts is when the LAST one finishes.  A group costs its longest utterance x (lanes + a fixed part per
            # frame): the group of the longest utterances gets fewer lanes, so that all engines finish together.
            fixed = args.group_fixed

            def cut(limit):
                out, i = [], 0
                while i < len(order):
                    n = max(1, min(NLE, len(order) - i, limit // nfr[order[i]] - fixed))
                    out.append(order[i:i + n])
                    i += n
                return out
            lo, hi = 1, nfr[order[0]] * (NLE + fixed)
            while lo < hi:
                mid = (lo + hi) // 2
                if len(cut(mid)) <= NE:
                    hi = mid
                else:
                    lo = mid + 1
            groups = cut(lo)
        else:
            gsz = min(NLE, max(args.min_group, -(-len(order) // NE)))
            groups = [order[i:i + gsz] for i in range(0, len(order), gsz)]
        per, load = [[] for _ in range(NE)], [0] * NE
        for g in groups:
            e = min(range(NE), key=lambda k: (load[k], k))
            per[e].append(g)
            load[e] += nfr[g[0]]
        return per

    def my_share(step):
        if args.scaling == "strong":
            return shard.shard_contiguous(U, rank, world), 0
        return list(range(U)), rank * U              # weak: a whole batch per rank (record ids offset by the rank)

    host_t = {}

    def run_step(step, recs=None, sched=None):
        ids, base = my_share(step)
        per = sched if sched is not None else schedule(ids)

        def one(e):         # engine e decodes its groups one after the other (the C calls release the GIL)
            lib.check(L.s3a_set_device(local_rank))     # (HIP's current device is per host thread)
            ms, out, t_dec, t_hyp = 0.0, [], 0.0, 0.0
            for g in per[e]:
                t0 = time.perf_counter()
                refill = args.refill and len(g) > NLE
                if refill:
                    ms += decs[e].ud.decode_queue_dev([fdev[k] for k in g], [nfr[k] for k in g], D4x4)
                else:
                    ms += decs[e].ud.decode_dev([fdev[k] for k in g], [nfr[k] for k in g], D4x4)
                t1 = time.perf_counter()
                if recs is not None:
                    get = decs[e].queue_hyp if refill else decs[e].hyp_var
                    out += [get(z, utts[k], base + k) for z, k in enumerate(g)]
                t_dec += t1 - t0
                t_hyp += time.perf_counter() - t1
            return ms, out, t_dec, t_hyp
        done = list(pool.map(one, range(NE))) if NE > 1 else [one(0)]     # (one engine: on the calling thread)
        if recs is not None:
            for d_ in done:
                recs.extend(d_[1])
        host_t["decode_call_s"] = max(d_[2] for d_ in done)
        host_t["hyp_s"] = max(d_[3] for d_ in done)
        return max(d_[0] for d_ in done)

    # ---------------- the reference (rank 0, untimed): CPU baseline + what the device's output is compared with ----------------
    cpu, ref = None, None
    if rank == 0:
        n_procs = max(1, args.cpu_procs)
        if args.no_cpu:
            one = cpu_batch(targs, ctl, d, min(2, U), 1, "gate_")
            ref = (one[2], one[3], min(2, U)) if one else None
        else:
            cpu, ref = cpu_baseline(targs, ctl, d, U, n_procs, args.check_all, args.cpu_physical)
        assert ref, "the reference decoder failed on the task"
    t_setup = time.perf_counter() - t_setup

    def sync_all():
        if dist is not None:
            dist.barrier()
        lib.check(L.s3a_dev_sync())
        if dist is not None and not rehearsal:
            import torch
            torch.cuda.synchronize()

    # ---------------- the timed region ----------------
    # ONE exchange, in C, for every N: s3a_gather_init (RCCL loaded at run time, ncclUniqueId through a file under the
    # run's own temporary directory) + s3a_gather_hyps (three ncclAllGathers).  torch.distributed only supplies the barrier
    # and the max over ranks.  S3A_BENCH_ONE_GPU=1 (several ranks on one GPU, where RCCL cannot run) keeps shard.gather_var.
    cgather = None
    if not rehearsal and not os.environ.get("S3A_BENCH_NO_RCCL"):
        import ctypes
        saved = os.dup(1)
        try:
            sys.stdout.flush()
            os.dup2(2, 1)                   # (RCCL prints a version banner to C stdout: keep stdout to the ONE JSON line)
            cgather = lib.Gather(rank, world, os.path.join(d, "rccl-id"), run_id=exchange_run_id(dist, rank, tdev))
        except lib.S3AError as e:
            if world > 1:
                raise SystemExit(f"bench.py: the C exchange could not start on rank {rank}: {e}")
            cgather = None
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
    n_total = U if args.scaling == "strong" else U * world
    for i in range(args.warmup):
        run_step(i)
    sync_all()
    dev_ms, allrec = 0.0, None
    t0 = time.perf_counter()
    for i in range(args.steps):
        recs = []
        dev_ms += run_step(i, recs)
        if cgather is not None:
            allrec = cgather.gather(recs, n_total)       # the C side's exchange (s3a_gather_hyps over RCCL), every N
        elif dist is not None:
            allrec = shard.gather_var(recs, n_total, dist, device=tdev)          # rehearsal on one GPU: torch.distributed (gloo)
        else:
            allrec = sorted(recs, key=lambda t: t[0].utt_index)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    weak = None
    if world > 1 and args.scaling == "strong":
        # the second figure: weak scaling, every rank a whole batch (one step)
        args.scaling = "weak"
        sync_all()
        t1 = time.perf_counter()
        wrecs = []
        run_step(0, wrecs)
        wall = cgather.gather(wrecs, U * world) if cgather is not None else shard.gather_var(wrecs, U * world, dist, device=tdev)
        sync_all()
        wdt = time.perf_counter() - t1
        tw = torch.tensor([wdt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak = {"value": round(sum(h.n_frames for h, _ in wall) / float(tw.item()), 1), "unit": "frames/s", "utterances": U * world,
                "note": "every rank decodes the whole 1024-utterance batch (one step)"}
        args.scaling = "strong"

    if rank == 0:
        assert len(allrec) == n_total and all(h.status == 0 for h, _ in allrec)
        frames_step = sum(h.n_frames for h, _ in allrec)
        value = frames_step * args.steps / dt
        # ---- the gate: the LAST timed step's hypotheses, every line against the unmodified reference decoder's ----
        lines = shard.write_outputs(allrec[:U], dec.format_var, os.path.join(d, "bench.match"), os.path.join(d, "bench.matchseg"))
        n_chk = ref[2]
        got_h, got_s = "".join(l[0] for l in lines[:n_chk]), "".join(l[1] for l in lines[:n_chk])
        hyp_ok, seg_ok = got_h == ref[0], got_s == ref[1]
        if not (hyp_ok and seg_ok):
            rl, rs = ref[0].splitlines(keepends=True), ref[1].splitlines(keepends=True)
            bad = [k for k in range(min(n_chk, len(rl), len(lines))) if lines[k][0] != rl[k] or lines[k][1] != rs[k]]
            where = {k: (e, gi, z) for e, gs in enumerate(schedule(my_share(0)[0])) for gi, g in enumerate(gs) for z, k in enumerate(g)}
            print(f"bench: {len(bad)} of {n_chk} checked utterances differ from the reference: "
                  + ", ".join(f"utt {k} (engine, group, lane) = {where.get(k)}" for k in bad[:12]), file=sys.stderr)
            for k in bad[:2]:
                print("  device:", lines[k][1].strip()[:600], "\n  refrnc:", rs[k].strip()[:600], file=sys.stderr)
        assert hyp_ok and (seg_ok or args.fast), "device hypotheses differ from the unmodified reference decoder's"
        if args.plain:
            # the rocprofv3-wrapped command: the timed regime and nothing else (no lock-step decode for statistics, no profiled step)
            pp = [dd.ud.last_parts() for dd in decs]
            ph, fr, busy = {}, 0, []
            if all(p_["n_frames"] > 0 for p_ in pp):
                for z in range(0, NLE, max(1, NLE // 16)):
                    t_, f_, _, _ = decs[0].ud.frame_ticks(z)
                    fr += f_
                    for k, v in t_.items():
                        ph[k] = ph.get(k, 0.0) + v
                # (how long each lane's workgroup was inside the launches of the last call: the launch ends with the last of them)
                busy = sorted(decs[0].ud.frame_ticks(z)[0]["in_launch"] / 1e3 for z in range(NLE))
            print(json.dumps({"metric": "decoded_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "plain": True,
                              "config": {"frames_per_step": frames_step, "lanes_per_gpu": NL},
                              "kernels": {"ku_frames": {"ms_per_step": round(max(p_["frames_ms"] for p_ in pp), 2), "relay_launches": decs[0].ud.last_relay()},
                                          "ku_score_window": {"ms_per_step": round(max(p_["score_ms"] for p_ in pp), 2)}},
                              "lanes_busy_ms": ({"min": round(busy[0], 1), "p10": round(busy[len(busy) // 10], 1), "median": round(busy[len(busy) // 2], 1),
                                                 "mean": round(sum(busy) / len(busy), 1), "p90": round(busy[(9 * len(busy)) // 10], 1),
                                                         "max": round(busy[-1], 1)} if busy else None),
                              "search": {"us_per_lane_frame": round(1e3 * max(p_["frames_ms"] for p_ in pp) * NL / max(frames_step, 1), 2),
                                         "phases_us_per_lane_frame": {k: round(v / max(fr, 1), 2) for k, v in sorted(ph.items())}},
                              "identical_to_reference": {"hyp": bool(hyp_ok), "hypseg": bool(seg_ok)}}))
            return

        # ---- where the step's device time went (round 5: the engine runs ku_frames) ----
        # The timed region is TWO kinds of launches per engine and queue part: ku_score_window (every frame of the part scored first)
        # and ku_frames (the lanes decode the part's utterances from their first to their last frame, taking them from the queue
        # themselves).  HIP events on the engine's own stream bracket both (s3a_uttdec_last_parts: of the LAST timed step); inside
        # ku_frames the steps of a frame are clocked by the kernel itself (s3a_uttdec_frame_ticks, workgroup 0 of a lane, summed over
        # the lane's last utterance).  An engine that kept the launch path (fewer than 8 busy lanes, --variant, S3A_UTT_PERSIST=-1)
        # reports no parts: the per-launch profile of rounds 2-4 is then taken instead.
        sched = schedule(my_share(0)[0])
        parts = [dd.ud.last_parts() for dd in decs]
        persistent = all(p_["n_frames"] > 0 for p_ in parts)
        phases, ph_frames = {}, 0
        if persistent:
            for dd in decs:
                for z in range(0, NLE, max(1, NLE // 16)):
                    t_, f_, _, _ = dd.ud.frame_ticks(z)
                    for k, v in t_.items():
                        phases[k] = phases.get(k, 0.0) + v
                    ph_frames += f_
        # frame statistics (active HMMs, senones scored, word exits per frame): one lock-step decode of the first lanes' utterances
        g0 = sched[0][0][:NLE]
        dec.ud.decode_dev([fdev[k] for k in g0], [nfr[k] for k in g0], D4x4)
        nl0 = len(g0)
        stat = [dec.ud.result(z)["frame_stat"] for z in range(nl0)]
        res0 = dec.ud.result(0)
        lanes_bench = float(np.mean([min(NLE, sum(len(g) for g in e_)) for e_ in sched if e_])) if any(sched) else float(nl0)
        lanes_hmm, lanes_sen, lanes_gau, lanes_exit = (float(np.mean([s[:, c].mean() for s in stat])) for c in (1, 2, 3, 7))
        b = dec.b
        S, Sci, D, Cc = b["n_sen"], b["n_ci_sen"], dec.veclen, dec.g.C
        K = max(1, dec.ud.window())
        pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else {}
        pmc_stale = bool(pmc) and pmc.get("csrc_hash") != csrc_hash()
        if pmc_stale:                   # (measured on other kernels than the ones that just ran: no traffic figure rather than a wrong one)
            pmc = {"source": f"profiles/pmc_traffic.json is stale (its csrc_hash {pmc.get('csrc_hash')} is not this tree's {csrc_hash()}): not used"}
        prof, prof_alone, kern, search, roof, roof_scoring = {}, {}, {}, None, None, None
        fr_rank = sum(nfr[k] for k in my_share(0)[0])           # (this rank's frames of a step: the parts are this rank's engines')
        if persistent:
            score_ms = max(p_["score_ms"] for p_ in parts)
            frames_ms = max(p_["frames_ms"] for p_ in parts)
            n_score = sum(p_["n_score"] for p_ in parts)
            n_kf = sum(p_["n_frames"] for p_ in parts)
            tot_ms = score_ms + frames_ms
            # ALGORITHMIC bytes.  Scoring: the Gaussians' parameters once per model pass (a launch scores up to 8192
            # frames against them) + per frame the vector in and every senone's score and best component out.  The search: ~84 B of
            # HMM state read + written per active HMM and frame (4 scores + 4 history ids + 3 senone gathers), the frame's selected
            # senone scores (4 B each), 40 B per history entry made and 16 B per (exit, predecessor) pair at the word level.
            alg_score = n_score * S * Cc * (2 * D + 2) * 4 + fr_rank * (D * 4 + S * 5)
            per_lane_frame = 84.0 * lanes_hmm + 4.0 * (lanes_sen + Sci) + 28.0 * lanes_exit
            alg_frames = fr_rank * per_lane_frame
            kf_gbs = alg_frames / (frames_ms * 1e-3) / 1e9
            tr = pmc.get("kernels", {}).get("ku_frames", {})
            # (a call of ku_frames is a chain of launches since round 6 -- the relay --: per-launch figures are averages over the chain's
            # launches, as rocprofv3's kernel statistics count them)
            n_disp = max(n_kf, 1) * (1 + decs[0].ud.last_relay())
            traffic = int(tr["hbm_bytes_per_lane_frame"] * fr_rank / n_disp) if tr.get("hbm_bytes_per_lane_frame") else None
            flops = 4.0 * S * Cc * D * fr_rank
            sc_tfl = flops / (score_ms * 1e-3) / 1e12
            roof = {"kernel": "ku_frames", "bound": "hbm", "achieved": round(kf_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kf_gbs / HBM_PEAK_GBS, 5), "avg_launch_us": round(1e3 * frames_ms / max(n_disp / len(decs), 1), 1),
                    "launches_timed": int(n_disp), "launches_per_call": 1 + decs[0].ud.last_relay(),
                    "traffic": traffic, "traffic_source": pmc.get("source") if (traffic is not None or pmc_stale) else None,
                    "algorithmic_bytes_per_launch": int(alg_frames / n_disp), "algorithmic_bytes_per_lane_frame": round(per_lane_frame, 1),
                    "lanes_in_launch": round(lanes_bench, 1), "workgroups_per_lane": parts[0]["cluster"],
                    "measured": "HIP events on the engine's stream around ku_frames, last timed step",
                    "note": "the dominant kernel by device time (`kernels`): ONE launch decodes a queue part -- every lane its utterances, frame "
                            "after frame, the twelve steps of a frame as phases of a persistent 512-thread workgroup (`phases_us_per_lane_frame`). "
                            "Its steps WAIT -- on dependent round trips to memory and on the CUs' address path (a wave issues an instruction "
                            "every ~55 cycles: DESIGN.md 4.1) --, they are not bound by bytes: nowhere near the bandwidth roof; the scoring is "
                            "float64-VALU-bound (bit-exact mode: no FMA)"}
            roof_scoring = {"kernel": "ku_score_window", "bound": "valu-f64", "achieved": round(sc_tfl, 2), "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(sc_tfl / FP64_VEC_PEAK_TFLOPS, 4), "hbm_GBs": round(alg_score / (score_ms * 1e-3) / 1e9, 1),
                            "hbm_frac": round(alg_score / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "avg_launch_us": round(1e3 * score_ms / max(n_score / len(decs), 1), 2),
                            "launches_timed": int(n_score), "frames_per_launch": round(fr_rank / max(n_score, 1), 1),
                                    "share_of_step": round(score_ms / tot_ms, 4),
                            "note": "every frame of the step scored BEFORE the search (rows [frame][senone] in HBM, 5 B per senone and frame); hub4 "
                                    "single-frame scoring (one frame per model pass, HBM-bound) is in `scoring.hub4.frame_sync`: north_star's >= 0.60 "
                                    "of HBM peak is NOT met there at B = 1"}
            kern = {"ku_frames": {"avg_launch_us": roof["avg_launch_us"], "launches_timed": int(n_disp), "ms_per_step": round(frames_ms, 2),
                    "share": round(frames_ms / tot_ms, 4)},
                    "ku_score_window": {"avg_launch_us": roof_scoring["avg_launch_us"], "launches_timed": int(n_score), "ms_per_step": round(score_ms, 2),
                                        "share": round(score_ms / tot_ms, 4)}}
            search = {"ms_per_step": round(frames_ms, 2), "us_per_lane_frame": round(1e3 * frames_ms * lanes_bench * len(decs) / fr_rank,
                    2) if fr_rank else None,
                      "share_of_step": round(frames_ms / tot_ms, 4), "algorithmic_bytes_per_step": int(fr_rank * lanes_hmm * 84.0),
                      "achieved_GBs": round(fr_rank * lanes_hmm * 84.0 / (frames_ms * 1e-3) / 1e9, 1),
                      "frac": round(fr_rank * lanes_hmm * 84.0 / (frames_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "bound": "latency of dependent scattered accesses (waits), address path of the CUs",
                      "active_hmm_updates_per_s": round(lanes_hmm * value, 0),
                      "active_hmm_updates_per_s_cpu_single_core": round(cpu["active_hmm_per_frame"] * cpu["single_core"], 0) if cpu else None,
                      "phases_us_per_lane_frame": {k: round(v / max(ph_frames, 1), 2) for k, v in sorted(phases.items()) if k != "in_launch"},
                      "lane_frames_clocked": int(ph_frames),
                      "note": "84 B x active HMMs x frames of the step / ku_frames' device time; the phases are what one lane's workgroup spends per frame "
                              "(two lanes share a CU, so the chip's rate is lanes / that)"}
        else:
            # the launch path (rounds 2-4): HIP events around every launch of every 8th frame, all engines running
            for dd in decs:
                dd.ud.set_profile(8)
            run_step(0, sched=sched)
            lib.check(L.s3a_dev_sync())
            for dd in decs:
                for k, (us, n) in dd.ud.profile().items():
                    a0, n0 = prof.get(k, (0.0, 0))
                    prof[k] = (a0 + us, n0 + n)
                dd.ud.set_profile(0)
            prof = {k: v for k, v in prof.items() if v[1] > 0}
            pf_b = {k: (us / n / (K if k == "ku_score_window" else 1)) for k, (us, n) in prof.items() if n > 0}
            tot = sum(pf_b.values())
            kern = {k: {"avg_launch_us": round(prof[k][0] / prof[k][1], 2), "us_per_frame": round(v, 2), "share": round(v / tot, 4),
                    "launches_timed": int(prof[k][1])}
                    for k, v in pf_b.items()}
            dom = max(pf_b, key=lambda k: pf_b[k])
            dom_us = prof[dom][0] / prof[dom][1]
            ab = lanes_bench * lanes_hmm * 84.0 if dom != "ku_score_window" else S * Cc * (2 * D + 2) * 4 + lanes_bench * K * (D * 4 + S * 5)
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(ab / (dom_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ab / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), "avg_launch_us": round(dom_us, 2), "traffic": None,
                    "algorithmic_bytes_per_launch": int(ab), "launches_timed": int(prof[dom][1]), "lanes_in_launch": round(lanes_bench, 1),
                    "measured": f"the launch path: all {NE} engines running, HIP events around every launch of every 8th frame"}
            srch_us = sum(v for k, v in pf_b.items() if k not in ("ku_score_window", "ku_gated_cd", "ku_gated_ci", "ku_comsen_max"))
            search = {"us_per_frame": round(srch_us, 2), "achieved_GBs": round(lanes_bench * lanes_hmm * 84.0 / (srch_us * 1e-6) / 1e9, 1),
                      "frac": round(lanes_bench * lanes_hmm * 84.0 / (srch_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                              "active_hmm_updates_per_s": round(lanes_hmm * value, 0)}

        # ---- what N = 8 gives a GPU: 128 of the 1024 utterances (the engine gives a lane a cluster of workgroups then) ----
        proj = None
        if world == 1 and U >= 8 and not args.plain:
            ids8 = shard.shard_contiguous(U, 0, 8)
            sch8 = schedule(ids8)
            run_step(0, sched=sch8)
            lib.check(L.s3a_dev_sync())
            t8 = time.perf_counter()
            recs8 = []
            run_step(0, recs=recs8, sched=sch8)
            lib.check(L.s3a_dev_sync())
            t8 = time.perf_counter() - t8
            f8 = sum(nfr[k] for k in ids8)
            # (the clusters' hypotheses against the reference's lines of the same utterances: the XCD-local barrier at scale)
            rl8, rs8 = ref[0].splitlines(keepends=True), ref[1].splitlines(keepends=True)
            cmp8 = [(h.utt_index, dec.format_var(h, w)) for h, w in recs8 if h.utt_index < n_chk]
            same8 = all(h.status == 0 for h, _ in recs8) and all(m == rl8[k] and sg == rs8[k] for k, (m, sg) in cmp8)
            assert same8, "the 128-utterance projection's hypotheses differ from the unmodified reference decoder's"
            proj = {"gpus": 8, "utterances_per_gpu": len(ids8), "groups": [len(g) for e in sch8 for g in e],
                    "frames_per_sec_per_gpu": round(f8 / t8, 1), "implied_strong_scaling_efficiency": round(f8 / t8 / value, 3),
                    "workgroups_per_lane": decs[0].ud.last_parts()["cluster"], "relay_launches": decs[0].ud.last_relay(),
                    "identical_to_reference": bool(same8), "utterances_checked_against_reference": len(cmp8),
                    "note": "one GPU decoding rank 0's share of an 8-rank run of the same batch (hypotheses included, no gather): "
                            "fewer utterances than workgroup slots, so a lane is a cluster of workgroups with a counter barrier between the frame's steps"}

        # ---- configs[2]: ONE utterance alone (an engine of one lane: the frame as launches -- below 8 lanes they beat ku_frames) ----
        single = None
        if world == 1 and not args.plain:
            one = bundle.Decoder(bpath, 1, precision=lib.GMM_FAST if args.fast else lib.GMM_EXACT, cand_cap=args.cand_cap, max_frames=max(nfr) + 8)
            k1 = max(range(U), key=lambda k: nfr[k])
            one.ud.decode_dev([fdev[k1]], [nfr[k1]], D4x4)
            t1 = time.perf_counter()
            ms1 = one.ud.decode_dev([fdev[k1]], [nfr[k1]], D4x4)
            t1 = time.perf_counter() - t1
            h1 = one.hyp_var(0, utts[k1], k1)
            same = None
            if k1 < n_chk:
                same = one.format_var(*h1)[0] == ref[0].splitlines(keepends=True)[k1]
            single = {"frames": nfr[k1], "frames_per_sec": round(nfr[k1] / t1, 1), "xRT": round(nfr[k1] / t1 / 100.0, 1),
                    "us_per_frame_device": round(1e3 * ms1 / nfr[k1], 2),
                      "path": "launches" if one.ud.last_parts()["n_frames"] == 0 else "ku_frames", "identical_to_reference": same,
                      "note": "configs[2]: the batch's longest utterance decoded alone by a one-lane engine, host call to hypothesis (wall clock)"}
            del one

        res = {
            "metric": "decoded_frames_per_sec (full mode-4 decode, hub4-shaped CD-GMM 6144x8x39 + 20k-word lextrees + trigram; xRT = value/100/n_gpus)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32-sub/f64-acc/int32-logadd (bit-exact), int32 Viterbi + word level" if not args.fast else "f32 scoring (+-2 logs3), int32 search",
            "data": f"synthetic (seeded hub4-shaped model, dictionary, ARPA trigram; {U} DISTINCT ~10 s utterances sampled from the model "
                    "along LM sentences; real hub4 parameters are not in the reference checkout); features resident in HBM before the "
                    "timed region (the CPU baseline reads its feature files)",
            "config": {"workload": f"configs[3]: batch of {U} synthetic 10 s utterances, hub4_cd_continuous shape, full decode (senone "
                                   f"scoring + lextree Viterbi + trigram word level on the device), sharded over {world} GPU(s); a step = "
                                   f"the whole batch once" + (" per rank" if args.scaling == "weak" else ""),
                       "utterances_per_step": n_total, "frames_per_step": frames_step, "lanes_per_gpu": NL, "engines_per_gpu": NE,
                               "lane_refill": bool(args.refill and len(my_share(0)[0]) > NL),
                       "groups_rank0": [len(g) for e in sched for g in e],
                       "beams": "-beam 1e-60 -wbeam 1e-35 -maxhmmpf 20000 -maxwpf 10 -lw 9.5 (the reference's hub4 settings)",
                       "parallelism": f"utterance-sharded x{world} ({args.scaling}), ONE exchange per batch in C (three ncclAllGathers: counts, "
                                      f"headers, words padded per rank; no word limit), no per-frame collective"},
            "xRT_per_gpu": round(value / world / 100.0, 1),
            "device_ms_per_step": round(dev_ms / args.steps, 3),
            "host_side_last_step": {k: round(v, 3) for k, v in host_t.items()},
            "identical_to_reference": {"hyp": hyp_ok, "hypseg": seg_ok}, "utterances_checked_against_reference": n_chk,
            "exchange": (f"s3a_gather_hyps (C, RCCL, {world} rank{'s' if world > 1 else ''})" if cgather is not None else
                         "torch.distributed all_gather x2 over gloo (S3A_BENCH_ONE_GPU rehearsal)" if world > 1 else "none (RCCL not loadable)"),
            "load_s": round(t_load, 2), "setup_s": round(t_setup, 1),
            "per_frame": {"active_hmm": round(lanes_hmm, 1), "cd_senones_scored": round(lanes_sen, 1), "cd_gaussians": round(lanes_gau, 1),
                          "word_exits": round(lanes_exit, 2), "max_active_hmm": int(max(s_[:, 1].max() for s_ in stat)),
                          "frames_with_histogram_pruning": int(sum(int(s_[:, 6].sum()) for s_ in stat)), "max_candidates": int(res0["max_cand"]),
                                  "tie_frames_lane0": int(res0["n_tie_frames"])},
            "kernels": kern,
            "rank_devices": [0 if rehearsal else r_ for r_ in range(world)],
        }
        # (what a reader of the line's TAIL must find -- the driver stores the end of stdout -- comes last: the extra legs first, then the
        # baseline, the search's and the scoring's roofline entries, the single utterance, the projection, the dominant kernel's roofline)
        if world == 1 and not args.no_scoring:
            res["scoring"] = scoring_legs(lib, args.fast)
        if world == 1 and not args.no_ps:
            res["ps_fwdtree"] = ps_fwdtree_leg(d, args.ps_lanes)
        if world == 1 and not args.no_wide_beam:
            res["wide_beam"] = wide_beam_leg(lib, d, args.wide_lanes, args.wide_frames, args.fast)
            if args.wide_lanes < 512:           # (what more lanes buy the launches of this task: the same leg with 512 utterances, 16 of them compared)
                w2 = wide_beam_leg(lib, os.path.join(d, "wide512"), 512, args.wide_frames, args.fast, n_check=16)
                keep = ("lanes", "frames", "device_ms", "frames_per_sec", "xRT", "identical_to_reference", "error")
                res["wide_beam"]["lanes_512"] = {k: w2[k] for k in keep if k in w2}
        if weak:
            res["weak_scaling"] = weak
        if cpu:
            res["cpu_baseline"] = cpu
        res["search"] = search
        res["roofline_scoring"] = roof_scoring
        res["single_utterance"] = single
        if proj:
            res["strong_scaling_projection"] = proj
        res["roofline"] = roof
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
