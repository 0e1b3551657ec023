#!/usr/bin/env python3
"""Where a lane's frame goes inside ku_frames (the persistent frame kernel): the bench's hub4-shaped task, ONE engine of
--lanes lanes, the batch as one queue; prints frames/s and the steps' microseconds per frame (s3a_uttdec_frame_ticks,
workgroup 0 of the lane's cluster, averaged over the sampled lanes).  Environment: S3A_UTT_WIN (frames per window =
frames per launch), S3A_UTT_CLUSTER (workgroups per lane), S3A_UTT_PERSIST=-1 (the launch path, for the A/B).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHIM = os.path.join(ROOT, "oracle", "_ref", "ref_s3amd_tst_decode")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=512)
    ap.add_argument("--utts", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--engines", type=int, default=1)
    ap.add_argument("--sample", type=int, default=8, help="lanes whose clocks are averaged")
    ap.add_argument("--placement", action="store_true", help="lock-step decode of --lanes utterances; per lane when and where the launch at frame 512 ran")
    args = ap.parse_args()
    from cmusphinx_amd import bundle, lib, s3io, synth_task
    L = lib.load()
    d = os.path.join(tempfile.gettempdir(), f"s3a_kf_{args.utts}_{args.frames}")
    bpath = os.path.join(d, "decoder.bundle")
    if not os.path.exists(bpath):
        os.makedirs(d, exist_ok=True)
        synth_task.make_task(d, n_utt=args.utts, n_frames=args.frames, sorted_names=True, **synth_task.HUB4_TASK)
        r = subprocess.run([SHIM] + synth_task.decoder_args(d), env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bpath),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert r.returncode == 0 and os.path.exists(bpath), "bundle export failed"
    utts = [l.split()[0] for l in open(os.path.join(d, "ctl")) if l.strip()]
    hfeat = [s3io.read_mfc(os.path.join(d, "feat", u + ".mfc")) for u in utts]
    NE = args.engines
    NLE = args.lanes // NE
    decs = [bundle.Decoder(bpath, NLE, max_frames=max(len(f) for f in hfeat) // 39 + 8) for _ in range(NE)]
    D4x4 = 4 * ((decs[0].veclen + 3) // 4)
    fdev, nfr = [], []
    for f in hfeat:
        f = f.reshape(-1, decs[0].veclen)
        pad = np.zeros((len(f), D4x4), np.float32)
        pad[:, :decs[0].veclen] = f
        fdev.append(lib.DevBuf(pad.nbytes).upload(pad))
        nfr.append(len(f))
    order = sorted(range(len(nfr)), key=lambda k: (-nfr[k], k))
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(NE)

    def one(e):
        lib.check(L.s3a_set_device(0))
        g = order[e::NE]
        if len(g) > NLE:
            return decs[e].ud.decode_queue_dev([fdev[k] for k in g], [nfr[k] for k in g], D4x4)
        return decs[e].ud.decode_dev([fdev[k] for k in g], [nfr[k] for k in g], D4x4)

    if args.placement:
        g = order[:NLE]
        ms = decs[0].ud.decode_dev([fdev[k] for k in g], [nfr[k] for k in g], D4x4)
        rows = [decs[0].ud.frame_dbg(z) for z in range(NLE)]
        t0 = min(r[0] for r in rows)
        starts = np.array([(r[0] - t0) * 0.01 for r in rows]); ends = np.array([(r[1] - t0) * 0.01 for r in rows])
        cu = {}
        for z, r in enumerate(rows):
            hw, xcc = r[2], r[3] & 0xf
            key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
            cu.setdefault(key, []).append(z)
        per = sorted(len(v) for v in cu.values())
        print(json.dumps({"device_ms": round(ms, 1), "start_us": {"p50": float(np.median(starts)), "p90": float(np.percentile(starts, 90)), "max": float(starts.max())},
                          "dur_us": {"min": float((ends - starts).min()), "p50": float(np.median(ends - starts)), "p90": float(np.percentile(ends - starts, 90)), "max": float((ends - starts).max())},
                          "end_us_max": float(ends.max()), "distinct_cus": len(cu), "lanes_per_cu_hist": {str(k): per.count(k) for k in sorted(set(per))},
                          "late_starters": [(int(z), round(float(starts[z]), 1)) for z in np.argsort(-starts)[:8]]}))
        return
    best = None
    for step in range(args.steps + 1):
        lib.check(L.s3a_dev_sync())
        t0 = time.perf_counter()
        ms = list(pool.map(one, range(NE)))
        lib.check(L.s3a_dev_sync())
        dt = time.perf_counter() - t0
        if step > 0:
            best = dt if best is None else min(best, dt)
    total = sum(nfr)
    acc, nf, nl, C = {}, 0, 0, 0
    for z in range(0, NLE, max(1, NLE // args.sample)):
        t, f, l, C = decs[0].ud.frame_ticks(z)
        for k, v in t.items():
            acc[k] = acc.get(k, 0.0) + v
        nf += f
        nl += l
        dbg = decs[0].ud.frame_dbg(z)
        for k in range(4):
            acc["dbg%d" % k] = acc.get("dbg%d" % k, 0.0) + dbg[k]
    out = {"lanes": args.lanes, "engines": NE, "utts": len(nfr), "frames": total, "frames_per_s": round(total / best, 1), "s_per_step": round(best, 4),
           "device_ms": [round(m, 1) for m in ms], "window": decs[0].ud.window(), "cluster": C,
           "us_per_frame": {k: round(v / max(nf, 1), 2) for k, v in acc.items()}, "frames_sampled": nf, "launches_sampled": nl}
    if nf:
        out["us_per_frame"]["sum_steps"] = round(sum(v for k, v in acc.items() if k not in ("in_launch", "emit_only") and not k.startswith("dbg")) / nf, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
