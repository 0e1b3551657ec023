#!/usr/bin/env python3
"""PMC traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, a run each) of pocketsphinx's first pass on the device in the regime the bench line
reports: configs[3]'s batch (1024 utterances, hub4-shaped task) as ONE queue over 512 lanes (ref_ps_amdfwd -batch 512 -queue yes).
usage (GPU box): python tools/ps_pmc_regime.py [OUT.json] [n_utt] [lanes]      -> the kernels' KB and k_psf_queue's bytes per frame"""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmusphinx_amd import synth_task
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_ps.json")
n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 512
t = "/tmp/ps_pmc_task"
if not os.path.exists(os.path.join(t, "ctl")):
    synth_task.make_task(t, n_utt=n_utt, n_frames=1000, sorted_names=True, **synth_task.HUB4_TASK)
cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ps_amdfwd")] + synth_task.ps_decoder_args(t) + ["-fresh", "yes", "-batch", str(lanes), "-queue", "yes",
                                                                                                "-hyp", t + "/pmc.match", "-hypseg", t + "/pmc.seg"]
agg, frames, ms = {}, 0, 0.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"/tmp/prof_ps_{c}"
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + cmd, cwd="/tmp", capture_output=True, text=True,
                       env=dict(os.environ, TMPDIR="/tmp"))
    log = r.stdout + r.stderr
    m = re.search(r"batch of (\d+) utterances, (\d+) frames: ([0-9.]+) ms on the device", log)
    if r.returncode != 0 or not m:
        print("the pass failed:", r.returncode, log[-1500:])
        sys.exit(1)
    frames, ms = int(m.group(2)), float(m.group(3))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].replace("void ", "").split("(")[0].strip()
            a = agg.setdefault(k, {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
            a[c + "_KB"] += float(row["Counter_Value"])
            if c == "FETCH_SIZE":
                a["launches"] += 1
    shutil.rmtree(d, ignore_errors=True)
for k, a in agg.items():
    a["bytes_per_frame"] = round((a["FETCH_SIZE_KB"] + a["WRITE_SIZE_KB"]) * 1024 / max(frames, 1), 1)
res = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of ref_ps_amdfwd: {n_utt} utterances as one queue over {lanes} lanes, hub4-shaped task "
                 "(tools/ps_pmc_regime.py); KB at the L2's memory-side interface, taken as they are (scattered accesses: uncalibrated)",
       "frames": frames, "device_ms_under_the_counters": ms, "lanes": lanes, "utterances": n_utt, "kernels": agg}
json.dump(res, open(out, "w"), indent=1)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["bytes_per_frame"])[:6]:
    print("%-40s launches %5d fetch %12.0f KB write %12.0f KB  %10.1f B per frame" % (k[:40], a["launches"], a["FETCH_SIZE_KB"], a["WRITE_SIZE_KB"], a["bytes_per_frame"]))
