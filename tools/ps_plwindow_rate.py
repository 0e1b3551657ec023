#!/usr/bin/env python3
"""pocketsphinx's first pass on the device with and without -pl_window on the bench's pocketsphinx task (1024 utterances as one queue over 512 lanes);
the first 8 utterances against the unmodified pocketsphinx with the same options.  usage (GPU box): python tools/ps_plwindow_rate.py [window ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmusphinx_amd import synth_task
t = "/tmp/ps_plw_task"
if not os.path.exists(os.path.join(t, "ctl")):
    synth_task.make_task(t, n_utt=1024, n_frames=1000, sorted_names=True, **synth_task.HUB4_TASK)
args = synth_task.ps_decoder_args(t)
ctl8 = os.path.join(t, "ctl8")
open(ctl8, "w").writelines(open(os.path.join(t, "ctl")).readlines()[:8])
for w in [int(x) for x in sys.argv[1:]] or [0, 3, 8]:
    extra = ["-pl_window", str(w)] if w else []
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_ps_amdfwd")] + args + extra + ["-fresh", "yes", "-batch", "512", "-queue", "yes", "-hyp", t + "/a.match",
                        "-hypseg", t + "/a.seg"], capture_output=True, text=True)
    m = re.search(r"batch of (\d+) utterances, (\d+) frames: ([0-9.]+) ms on the device", r.stdout + r.stderr)
    a8 = [a for a in args]
    a8[a8.index("-ctl") + 1] = ctl8
    q = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_ps_fwd")] + a8 + extra + ["-fresh", "yes", "-hyp", t + "/r.match", "-hypseg", t + "/r.seg"], capture_output=True, text=True)
    same = r.returncode == 0 and q.returncode == 0 and open(t + "/a.match").readlines()[:8] == open(t + "/r.match").readlines() \
        and open(t + "/a.seg").readlines()[:8] == open(t + "/r.seg").readlines()
    print("-pl_window %d: %s frames in %s ms = %.0f frames/s; first 8 utterances identical to the unmodified pocketsphinx: %s" %
          (w, m.group(2) if m else "?", m.group(3) if m else "?", (int(m.group(2)) / float(m.group(3)) * 1e3) if m else 0.0, same))
