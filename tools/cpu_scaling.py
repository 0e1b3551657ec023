#!/usr/bin/env python3
"""How the unmodified reference decoder scales over processes on this host (bench.py's cpu_baseline picks from this):
the same task as bench.py, N processes with one utterance each, N = 8 16 32 48 64 96 128; prints aggregate frames/s.
Also prints the exported bundle's shape (nodes, composite senones, root lists)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from cmusphinx_amd import bundle, synth_task

d = os.path.join(tempfile.gettempdir(), "s3a_cpu_scaling")
os.makedirs(d, exist_ok=True)
synth_task.make_task(d, n_utt=128, n_frames=1000, **synth_task.HUB4_TASK)
targs = synth_task.decoder_args(d)
out = {}
for n in [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64, 96, 128]:
    t = time.time()
    r = bench.cpu_batch(targs, os.path.join(d, "ctl"), d, n, n, f"sc{n}_")
    out[n] = {"frames_per_sec": round(r[0], 1), "wall_s": round(time.time() - t, 1),
              "xclk_mean": round(float(np.mean([s["tot_xclk"] for s in r[1]])), 3)}
    print(n, out[n], flush=True)
bp = os.path.join(d, "b.bundle")
subprocess.run([bench.SHIM] + targs, env=dict(os.environ, S3A_UTT="1", S3A_EXPORT=bp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
b = bundle.read(bp)
off = b["comstate_off"]
print({k: b[k] for k in ("n_tree", "n_sseq", "n_comsseq", "n_comstate", "n_sen", "n_ci_sen", "n_ci", "n_word")})
print("composite senones", len(off) - 1, "avg members", off[-1] / (len(off) - 1), "max", int(np.diff(off).max()))
for t in b["trees"]:
    print("tree", t["type"], t["n_node"], "roots", t["n_root"], "lc", t["n_lc"], "composite nodes", int(np.sum(t["composite"] != 0)),
          "word nodes", int(np.sum(t["wid"] >= 0)), "lcroot", len(t["lcroot"]))
print(json.dumps(out))
