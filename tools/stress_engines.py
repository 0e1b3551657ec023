#!/usr/bin/env python3
"""Self-consistency of concurrent engines: every utterance of a synthetic hub4-shaped task decoded by ONE engine alone
(the truth for this check), then by E engines side by side (a host thread + stream each), R rounds; reports every
utterance whose -hyp / -hypseg line differs, with its engine, group and lane.  usage: stress_engines.py [E] [R] [U] [lanes]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from concurrent.futures import ThreadPoolExecutor
from cmusphinx_amd import bundle, lib, s3io, synth_task

E, R, U, NLE = (int(a) for a in (sys.argv[1:5] + ["6", "3", "512", "128"][len(sys.argv) - 1:]))
L = lib.load()
d = os.path.join(tempfile.gettempdir(), "s3a_stress")
os.makedirs(d, exist_ok=True)
synth_task.make_task(d, n_utt=U, n_frames=1000, **synth_task.HUB4_TASK)
bp = os.path.join(d, "b.bundle")
assert subprocess.run([bench.SHIM] + synth_task.decoder_args(d), env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp),
                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
feats = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")).reshape(-1, 39) for u in utts]
nfr = [len(f) for f in feats]
decs = [bundle.Decoder(bp, NLE, max_frames=max(nfr) + 8) for _ in range(E)]
fdev = []
for f in feats:
    pad = np.zeros((len(f), 40), np.float32); pad[:, :39] = f
    fdev.append(lib.DevBuf(pad.nbytes).upload(pad))
order = sorted(range(U), key=lambda k: (-nfr[k], k))
groups = [order[i:i + NLE] for i in range(0, U, NLE)]


def decode(dec, g):
    dec.ud.decode_dev([fdev[k] for k in g], [nfr[k] for k in g], 40)
    return {k: dec.format_var(*dec.hyp_var(z, utts[k], k)) for z, k in enumerate(g)}


# lane reuse, one engine, nothing concurrent: group 0 fresh, group 1, group 0 again
a0 = decode(decs[0], groups[0]); decode(decs[0], groups[min(1, len(groups) - 1)]); a1 = decode(decs[0], groups[0])
print("reuse check (one engine): utterances of group 0 that differ between their 1st and 3rd decode on the same lanes:",
      [(k, groups[0].index(k)) for k in groups[0] if a0[k] != a1[k]][:16], flush=True)
truth = {}
for g in groups:
    truth.update(decode(decs[0], g))
print("truth:", len(truth), "utterances by one engine alone", flush=True)
pool = ThreadPoolExecutor(E)
bad_total = 0
if os.environ.get("STRESS_QUEUE"):
    # lane refill under concurrency: every round every engine decodes a different random share of the utterances as ONE
    # queue in a different random order (utterances land in different lanes, begin at different engine frames, after
    # different predecessors), E engines side by side
    rng = np.random.default_rng(12345)
    for r in range(R):
        perm = rng.permutation(U)
        shares = [list(map(int, perm[e::E])) for e in range(E)]

        def one_q(e):
            lib.check(L.s3a_set_device(0))
            g = shares[e]
            decs[e].ud.decode_queue_dev([fdev[k] for k in g], [nfr[k] for k in g], 40)
            return [(k, e, q) for q, k in enumerate(g) if decs[e].format_var(*decs[e].queue_hyp(q, utts[k], k)) != truth[k]]
        bad = [b for o in pool.map(one_q, range(E)) for b in o]
        bad_total += len(bad)
        print(f"queue round {r}: {len(bad)} mismatches (utterance, engine, place in the queue)", bad[:10], flush=True)
    print("TOTAL mismatches:", bad_total)
    sys.exit(1 if bad_total else 0)
for r in range(R):
    per = [[] for _ in range(E)]
    for i, g in enumerate(groups * max(1, (2 * E) // max(1, len(groups)))):
        per[i % E].append(g)

    def one(e):
        lib.check(L.s3a_set_device(0))
        out = []
        for gi, g in enumerate(per[e]):
            res = decode(decs[e], g)
            if os.environ.get("STRESS_SELFCHECK"):
                dirty = [(z, decs[e].ud.selfcheck(z).tolist()) for z in range(len(g))]
                print(f"   (engine {e}: utterance lengths of the group: {sorted(nfr[k] for k in g)[::16]})", flush=True)
                dirty = [(z, c) for z, c in dirty if c[6] != 2147483647 or c[7]]
                wrong = [z for z, k in enumerate(g) if res[k] != truth[k]]
                print(f"   engine {e} group {gi}: lanes with wrong results {wrong}; lanes left dirty after it {[(z, c[0], c[5], c[7]) for z, c in dirty]}", flush=True)
            for z, k in enumerate(g):
                if res[k] != truth[k]:
                    out.append((k, e, gi, z))
                    if len(out) <= 2:
                        a, b = res[k][1].split(), truth[k][1].split()
                        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
                        print(f"   utt {k} engine {e} group {gi} lane {z}: {len(a)} vs {len(b)} tokens, first differences {diff[:6]}", flush=True)
        return out
    seq = os.environ.get("STRESS_SEQUENTIAL") == "1"
    bad = [b for o in (map(one, range(E)) if seq else pool.map(one, range(E))) for b in o]
    bad_total += len(bad)
    for (k, e, gi, z) in bad[:6]:
        print("   selfcheck engine", e, "lane", z, decs[e].ud.selfcheck(z), flush=True)
    print("   selfcheck of a good lane: engine 0 lane 0", decs[0].ud.selfcheck(0), flush=True)
    print(f"round {r}: {len(bad)} mismatches", bad[:10], flush=True)
print("TOTAL mismatches:", bad_total)
sys.exit(1 if bad_total else 0)
