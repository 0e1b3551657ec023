#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection CSVs under <dir>/prof_pmc_fetch, prof_pmc_write, and
the scoring passes) -> profiles/pmc_traffic.json: HBM-side bytes per launch of every kernel class bench.py reports.
MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB at the L2's memory-side interface; on gfx950 FETCH_SIZE shows HALF the
bytes of a wide (16 B per lane) coalesced streaming read -- doubled here for the streaming scoring kernels only; other
access patterns and WRITE_SIZE are uncalibrated (taken as they are).  usage: pmc_traffic.py <dir> <out.json> <source> [lanes of the profiled engine]"""
import csv, glob, json, os, sys
from collections import defaultdict

d, out, source = sys.argv[1], sys.argv[2], sys.argv[3]
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
CLASS = {"ku_enter1": "ku_enter1", "ku_enter2": "ku_enter2", "ku_enter3_mark": "ku_enter3_mark", "ku_comsen_mark": "ku_gated_ci",
         "ku_dyn_ci_beam": "ku_gated_ci", "ku_select": "ku_gated_cd", "ku_comsen_max": "ku_comsen_max", "ku_hmm_eval": "ku_hmm_eval",
         "ku_hist_count": "ku_hist_count", "ku_hist_sort": "ku_hist_sort", "ku_weak": "ku_weak", "ku_resolve_lists": "ku_resolve", "ku_resolve_plist": "ku_resolve", "ku_weak_heur": "ku_weak",
         "ku_resolve": "ku_resolve", "ku_scan": "ku_scan", "ku_emit_word": "ku_emit_word", "ku_score_window": "ku_score_window",
         "ku_gated_cd_multi": "ku_gated_cd", "ku_gated": "ku_gated_ci", "k_score_frames": "k_score_frames", "k_score_frame_sync": "k_score_frame_sync",
         "ku_lanes_begin": "ku_lanes_begin", "ku_lanes_end": "ku_lanes_end", "k_dag_pass": "k_dag_pass"}
STREAMING = {"ku_score_window", "k_score_frames", "k_score_frame_sync"}


def base(name):
    n = name.replace("void ", "").split("(")[0].split("<")[0].strip()
    return n.split("::")[-1]


def collect(sub, counter):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                agg[CLASS.get(base(row["Kernel_Name"]), base(row["Kernel_Name"]))].append(float(row["Counter_Value"]))
    return agg


res = {"source": source, "lanes": lanes, "unit": "bytes per launch (mean over the launches of the pass)", "kernels": {}, "scoring_kernels": {}}
for key, subs in (("kernels", ("prof_pmc_fetch", "prof_pmc_write")), ("scoring_kernels", ("prof_scoring_pmc_fetch", "prof_scoring_pmc_write"))):
    fe, wr = collect(subs[0], "FETCH_SIZE"), collect(subs[1], "WRITE_SIZE")
    for k in sorted(set(fe) | set(wr)):
        f_kb = sum(fe[k]) / len(fe[k]) if fe.get(k) else 0.0
        w_kb = sum(wr[k]) / len(wr[k]) if wr.get(k) else 0.0
        corr = 2.0 if k in STREAMING else 1.0
        res[key][k] = {"launches": len(fe.get(k, [])) or len(wr.get(k, [])), "fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                       "fetch_correction": corr, "hbm_bytes_per_launch": int((corr * f_kb + w_kb) * 1024)}
json.dump(res, open(out, "w"), indent=1)
for key in ("kernels", "scoring_kernels"):
    for k, v in sorted(res[key].items(), key=lambda t: -t[1]["hbm_bytes_per_launch"])[:16]:
        print(key, k, v)
