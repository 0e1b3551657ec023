"""A complete synthetic mode-4 DECODING task in the reference's own file formats.

BASELINE.json's full-decode configurations (hub4 CD-GMM, 6144 senones x 8 Gaussians, and the
WSJ-shaped 8000 x 32 stress case, each with a 20 k-word trigram LM) cannot be run on the real
models: their mdef / means / variances are absent from the reference checkout (SURVEY.md
"facts" 1).  This module writes a task of the same SHAPE with seeded synthetic values:

    mdef                    text model definition: CI phones (incl. SIL + noise fillers), the
                            triphones the dictionary needs, senones tied per (base phone, state)
    means variances mixture_weights transition_matrices      S3 binary files (s3io)
    dict fillerdict         pronunciation dictionary of n_words words, filler words
    lm.arpa                 ARPA trigram (the reference reads the text format directly, lm.c:614-635)
    feat/uttNNNN.mfc ctl    39-dim FEATURE vectors (decoded with `-feat 1s_c -ceplen 39 -cmn none
                            -agc none -varnorm no`, so the files ARE the feature stream) sampled from
                            the model along sentences drawn from the LM: the search sees speech-like
                            input -- a clear best path, competitors falling out of the beam
    args                    the decoder arguments shared by the reference and the drop-in

The unmodified reference decoder (built by the test infrastructure) and the device path decode the same files;
tests/test_gpu_dropin.py diffs their -hyp/-hypseg outputs.  Deterministic: numpy PCG64, fixed seed.
"""
from __future__ import annotations

import os

import numpy as np

from . import s3io, synth

# sep / noise: class separation and sampling noise, tuned so that the reference's search on the hub4
# shape behaves like broadcast-news decoding (thousands of active HMMs, ~85 % of the senones scored,
# several word exits per frame, a mostly correct 1-best) rather than a trivially sharp or a hopeless one
HUB4_TASK = dict(n_sen=6144, n_ciphone=48, n_comp=8, n_words=20000, seed=0x5EED0101, sep=0.25, noise=1.5)
WSJ_TASK = dict(n_sen=8000, n_ciphone=50, n_comp=32, n_words=20000, seed=0x5EED0102, sep=0.25, noise=1.5)
N_FILLER_PHONES = 3          # SIL +NOISE+ +BREATH+
N_EMIT = 3
VECLEN = 39
WPOS = "besi"                # mdef word-position codes: begin, end, single, internal


def _phone_names(n_ciphone, sorted_names=False):
    """sorted_names: filler phones named so that the CI phone list is in strcmp order, which pocketsphinx's
    mdef reader insists on (bin_mdef.c:126); the values of the task do not depend on the names"""
    reg = [f"P{i:02d}" for i in range(n_ciphone - N_FILLER_PHONES)]
    return reg + (["SIL", "TNOISE", "UBREATH"] if sorted_names else ["SIL", "+NOISE+", "+BREATH+"])


def make_task(dirpath, n_sen, n_ciphone, n_comp, n_words, seed, n_utt=4, n_frames=1000,
              sep=1.0, noise=1.0, ctx_keep=0.5, n_emit=3, sorted_names=False):
    """Write the task under dirpath; returns a dict describing it (paths, args, truth).  n_emit = emitting states per
    HMM: 3 (hub4 / WSJ / RM1 / tidigits) or 5 (Bakis topology with skip transitions: hmm_vit_eval_5st_lr)."""
    N_EMIT = n_emit
    rng = np.random.Generator(np.random.PCG64(seed))
    os.makedirs(os.path.join(dirpath, "feat"), exist_ok=True)
    names = _phone_names(n_ciphone, sorted_names)
    n_reg = n_ciphone - N_FILLER_PHONES
    sil = n_reg
    n_ci_sen = n_ciphone * N_EMIT

    # ---------------- dictionary ----------------
    lens = np.clip(rng.geometric(0.28, n_words) + 1, 2, 10)
    lens[rng.random(n_words) < 0.01] = 1                      # a few single-phone words
    prons, seen = [], set()
    for w in range(n_words):
        while True:
            p = tuple(int(x) for x in rng.integers(0, n_reg, lens[w]))
            if p not in seen:
                seen.add(p)
                prons.append(p)
                break
            lens[w] = min(lens[w] + 1, 12)
    words = [f"W{w:05d}" for w in range(n_words)]
    with open(os.path.join(dirpath, "dict"), "w") as f:
        for w, p in zip(words, prons):
            f.write(f"{w}\t{' '.join(names[x] for x in p)}\n")
    with open(os.path.join(dirpath, "fillerdict"), "w") as f:
        f.write(f"<s>\tSIL\n</s>\tSIL\n<sil>\tSIL\n++NOISE++\t{names[-2]}\n++BREATH++\t{names[-1]}\n")

    # ---------------- triphones + tying ----------------
    tri = set()
    for p in prons:
        m = len(p)
        if m == 1:
            for lc in rng.choice(n_reg + 1, max(int((n_reg + 1) * ctx_keep), 1), replace=False):
                for rc in rng.choice(n_reg + 1, 4, replace=False):
                    tri.add((p[0], int(lc), int(rc), 2))
            continue
        for j in range(1, m - 1):
            tri.add((p[j], p[j - 1], p[j + 1], 3))
        for lc in rng.choice(n_reg + 1, max(int((n_reg + 1) * ctx_keep), 1), replace=False):
            tri.add((p[0], int(lc), p[1], 0))                 # contexts: regular phones and SIL (= n_reg)
        for rc in rng.choice(n_reg + 1, max(int((n_reg + 1) * ctx_keep), 1), replace=False):
            tri.add((p[m - 1], p[m - 2], int(rc), 1))
    tri = sorted(tri)
    pools = np.array_split(np.arange(n_ci_sen, n_sen), n_reg * N_EMIT)
    r_lc, r_rc, r_wp = (rng.integers(0, 1 << 20, n) for n in (n_ciphone, n_ciphone, 4))
    tri_states = {}
    for (b, lc, rc, wp) in tri:
        st = []
        for s in range(N_EMIT):
            pool = pools[b * N_EMIT + s]
            st.append(int(pool[(r_lc[lc] * 31 + r_rc[rc] * 17 + r_wp[wp] * 7 + s * 3) % len(pool)]))
        tri_states[(b, lc, rc, wp)] = st
    with open(os.path.join(dirpath, "mdef"), "w") as f:
        f.write("0.3\n%d n_base\n%d n_tri\n%d n_state_map\n%d n_tied_state\n%d n_tied_ci_state\n%d n_tied_tmat\n"
                % (n_ciphone, len(tri), (n_ciphone + len(tri)) * (N_EMIT + 1), n_sen, n_ci_sen, n_ciphone))
        f.write("#\n# Columns definitions\n#base lft  rt p attrib tmat      ... state id's ...\n")
        for c in range(n_ciphone):
            att = "filler" if c >= n_reg else "n/a"
            f.write("%9s   -   - - %6s %4d %s N\n" % (names[c], att, c, " ".join("%6d" % (N_EMIT * c + k) for k in range(N_EMIT))))
        for (b, lc, rc, wp) in tri:
            st = tri_states[(b, lc, rc, wp)]
            f.write("%9s %9s %9s %s    n/a %4d %s N\n"
                    % (names[b], names[lc], names[rc], WPOS[wp], b, " ".join("%6d" % x for x in st)))

    # ---------------- acoustic model ----------------
    scale = synth._DIM_SCALE
    centre = rng.standard_normal((n_ciphone, VECLEN)).astype(np.float32) * scale * sep
    sen_phone = np.zeros(n_sen, np.int32)
    sen_state = np.zeros(n_sen, np.int32)
    sen_phone[:n_ci_sen] = np.arange(n_ci_sen) // N_EMIT
    sen_state[:n_ci_sen] = np.arange(n_ci_sen) % N_EMIT
    for k, pool in enumerate(pools):
        sen_phone[pool] = k // N_EMIT
        sen_state[pool] = k % N_EMIT
    state_off = rng.standard_normal((n_ciphone, N_EMIT, VECLEN)).astype(np.float32) * scale * (0.5 * sep)
    sen_off = rng.standard_normal((n_sen, VECLEN)).astype(np.float32) * scale * (0.3 * sep)
    sen_off[:n_ci_sen] = 0
    comp_off = rng.standard_normal((n_sen, n_comp, VECLEN)).astype(np.float32) * scale * 0.35
    mean = (centre[sen_phone] + state_off[sen_phone, sen_state] + sen_off)[:, None, :] + comp_off
    mean = mean.astype(np.float32)
    logv = rng.uniform(np.log(0.5), np.log(2.0), (n_sen, n_comp, VECLEN)).astype(np.float32)
    var = (np.exp(logv) * (0.45 * scale) ** 2).astype(np.float32)
    var[:n_ci_sen] *= 2.5                                        # CI models are broader, as trained ones are
    mixw = (rng.dirichlet(np.ones(n_comp) * 2.0, n_sen) * rng.uniform(50.0, 5000.0, (n_sen, 1))).astype(np.float32)
    tmat = np.zeros((n_ciphone, N_EMIT, N_EMIT + 1), np.float32)
    for i in range(N_EMIT):
        stay = rng.uniform(0.45, 0.75, n_ciphone).astype(np.float32)
        tmat[:, i, i] = stay
        tmat[:, i, i + 1] = 1.0 - stay
    tmat[sil, :, :] = 0
    for i in range(N_EMIT):
        tmat[sil, i, i] = 0.85
        tmat[sil, i, i + 1] = 0.15
    if N_EMIT == 5:                     # the 5-state topology's skips (i -> i + 2, the exit included)
        for i in range(N_EMIT - 1):
            skip = 0.2 * tmat[:, i, i + 1]
            tmat[:, i, i + 2] = skip
            tmat[:, i, i + 1] -= skip
    s3io.write_gau(os.path.join(dirpath, "means"), mean, False)
    s3io.write_gau(os.path.join(dirpath, "variances"), var, False)
    s3io.write_mixw(os.path.join(dirpath, "mixture_weights"), mixw, False)
    s3io.write_tmat(os.path.join(dirpath, "transition_matrices"), tmat, False)

    # ---------------- language model (ARPA trigram) ----------------
    n_succ, n_tg = 6, 3
    ug = np.log10(rng.dirichlet(np.ones(n_words) * 0.8))
    ug = np.maximum(ug, -6.5)
    succ = [np.sort(rng.choice(n_words, n_succ, replace=False)) for _ in range(n_words)]
    start_succ = np.sort(rng.choice(n_words, 200 if n_words > 400 else n_words // 2, replace=False))
    with open(os.path.join(dirpath, "lm.arpa"), "w") as f:
        n_bg = n_words * (n_succ + 1) + len(start_succ)
        n_tgs = n_words * n_succ * n_tg
        f.write("\\data\\\nngram 1=%d\nngram 2=%d\nngram 3=%d\n\n\\1-grams:\n" % (n_words + 2, n_bg, n_tgs))
        f.write("-1.5000 </s> -0.3000\n-99.0000 <s> -0.5000\n")
        for w in range(n_words):
            f.write("%.4f %s -0.4000\n" % (ug[w], words[w]))
        f.write("\n\\2-grams:\n")
        for v in start_succ:
            f.write("-2.3000 <s> %s -0.2000\n" % words[v])
        tg_lines = []
        for w in range(n_words):
            f.write("-1.2000 %s </s> -0.1000\n" % words[w])
            for v in succ[w]:
                f.write("%.4f %s %s -0.2500\n" % (-0.6 - 0.05 * (int(v) % 7), words[w], words[v]))
                for u in succ[v][:n_tg]:
                    tg_lines.append("%.4f %s %s %s\n" % (-0.3 - 0.04 * (int(u) % 5), words[w], words[v], words[u]))
        f.write("\n\\3-grams:\n")
        f.writelines(tg_lines)
        f.write("\n\\end\\\n")

    # ---------------- utterances ----------------
    def states_of(ph, lc, rc, wp):
        lc = sil if lc >= n_reg else lc
        rc = sil if rc >= n_reg else rc
        return tri_states.get((ph, lc, rc, wp), [N_EMIT * ph + k for k in range(N_EMIT)])

    cum_mixw = np.cumsum(mixw / mixw.sum(axis=1, keepdims=True), axis=1)
    sd = np.sqrt(var) * noise

    def emit(sen, tm, out):
        """one pass through the phone's emitting states: geometric durations, a mixture component per frame"""
        for s in range(N_EMIT):
            dur = int(rng.geometric(1.0 - tmat[tm, s, s]))
            k = np.minimum((rng.random(dur)[:, None] >= cum_mixw[sen[s]][None, :]).sum(axis=1), n_comp - 1)
            out.append(mean[sen[s], k] + rng.standard_normal((dur, VECLEN)).astype(np.float32) * sd[sen[s], k])

    truth, ctl = [], []
    for u in range(n_utt):
        frames, sent = [], []
        w = int(rng.choice(start_succ))
        phones = [(sil, 2, -1)] * 1                              # leading silence
        while True:
            sent.append(w)
            m = len(prons[w])
            for j, ph in enumerate(prons[w]):
                phones.append((ph, 2 if m == 1 else 0 if j == 0 else 1 if j == m - 1 else 3, w))
            est = 8.6 * len(phones) * (N_EMIT / 3)
            if est >= n_frames - 30:
                break
            if rng.random() < 0.08:
                phones.append((sil, 2, -1))                      # <sil> between words
            w = int(rng.choice(succ[w]))
        phones.append((sil, 2, -1))
        for j, (ph, wp, _) in enumerate(phones):
            if ph == sil:
                for _ in range(int(rng.integers(2, 5))):
                    emit([N_EMIT * sil + k for k in range(N_EMIT)], sil, frames)
                continue
            lc = phones[j - 1][0] if j > 0 else sil
            rc = phones[j + 1][0] if j + 1 < len(phones) else sil
            emit(states_of(ph, lc, rc, wp), ph, frames)
        x = np.concatenate(frames).astype(np.float32)
        name = f"utt{u:04d}"
        with open(os.path.join(dirpath, "feat", name + ".mfc"), "wb") as f:
            np.array([x.size], "<i4").tofile(f)
            x.astype("<f4").tofile(f)
        ctl.append(name)
        truth.append(" ".join(words[v] for v in sent))
    with open(os.path.join(dirpath, "ctl"), "w") as f:
        f.write("\n".join(ctl) + "\n")
    with open(os.path.join(dirpath, "truth"), "w") as f:
        f.write("\n".join(f"{t} ({n})" for t, n in zip(truth, ctl)) + "\n")
    return dict(dir=dirpath, n_tri=len(tri), utts=ctl, truth=truth, args=decoder_args(dirpath))


def ps_decoder_args(d):
    """the same task files through pocketsphinx (make_task(..., sorted_names=True)): its continuous scorer, its own
    default beams, first pass only"""
    return ["-mdef", f"{d}/mdef", "-mean", f"{d}/means", "-var", f"{d}/variances", "-mixw", f"{d}/mixture_weights",
            "-tmat", f"{d}/transition_matrices", "-senmgau", ".cont.", "-dict", f"{d}/dict", "-fdict", f"{d}/fillerdict",
            "-lm", f"{d}/lm.arpa", "-feat", "1s_c", "-ceplen", str(VECLEN), "-cmn", "none", "-agc", "none", "-varnorm", "no",
            "-cepdir", f"{d}/feat", "-cepext", ".mfc", "-ctl", f"{d}/ctl", "-fwdflat", "no", "-bestpath", "no"]


def decoder_args(d, beam="1e-60", wbeam="1e-35"):
    """hub4 settings of the reference's own performance suite (src/tests/performance/hub4/ARGS.hub4_base)."""
    return ["-mdef", f"{d}/mdef", "-mean", f"{d}/means", "-var", f"{d}/variances", "-mixw", f"{d}/mixture_weights",
            "-tmat", f"{d}/transition_matrices", "-dict", f"{d}/dict", "-fdict", f"{d}/fillerdict",
            "-lm", f"{d}/lm.arpa", "-feat", "1s_c", "-ceplen", str(VECLEN), "-cmn", "none", "-agc", "none",
            "-varnorm", "no", "-cepdir", f"{d}/feat", "-cepext", ".mfc", "-ctl", f"{d}/ctl",
            "-beam", beam, "-wbeam", wbeam, "-epl", "4", "-fillprob", "0.02", "-lw", "9.5", "-maxwpf", "10",
            "-wip", "0.2", "-op_mode", "4"]


if __name__ == "__main__":
    import sys
    kind, out = sys.argv[1], sys.argv[2]
    kw = dict(HUB4_TASK if kind == "hub4" else WSJ_TASK if kind == "wsj" else
              dict(n_sen=600, n_ciphone=13, n_comp=4, n_words=300, seed=7))
    for a in sys.argv[3:]:
        k, v = a.split("=")
        kw[k] = float(v) if "." in v else int(v)
    t = make_task(out, **kw)
    # TASK_BEAM / TASK_WBEAM: other beams than the hub4 settings (e.g. the wide-beam stress configuration)
    args = decoder_args(out, os.environ.get("TASK_BEAM", "1e-60"), os.environ.get("TASK_WBEAM", "1e-35"))
    print(t["n_tri"], "triphones;", " ".join(args))
