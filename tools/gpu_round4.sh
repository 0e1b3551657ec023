#!/bin/bash
# Round-4 measurement on the GPU box: the default bench line, then rocprofv3 kernel statistics of the same command
# (extra legs off) for profiles/.
cd $(dirname $0)/..
R=$(pwd); O=$R/gpurun_out/r4${1:-a}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --plain > $O/prof_bench.json 2> $O/prof_bench.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
rm -rf $O/prof
python - <<PY
import csv, json
r = json.load(open("$O/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step", "xRT_per_gpu", "identical_to_reference", "exchange")})
print("roofline", {k: r["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "alone")})
print("scoring", {k: r["roofline_scoring"].get(k) for k in ("achieved", "frac", "avg_launch_us")})
print("search", {k: r["search"].get(k) for k in ("us_per_frame", "us_per_frame_alone", "frac", "active_hmm_updates_per_s")})
print("ps", {k: (r.get("ps_fwdtree") or {}).get(k) for k in ("lanes", "utterances", "frames_per_sec", "xRT", "identical_to_pocketsphinx", "cpu_pocketsphinx", "error")})
print("wide", {k: (r.get("wide_beam") or {}).get(k) for k in ("frames_per_sec", "xRT", "per_frame", "identical_to_reference", "cpu_reference", "error")})
rows = list(csv.reader(open("$O/bench_kernel_stats.csv")))[1:12]
for x in rows: print(x[0][:40].ljust(42), x[1], "avg_us %.1f" % (float(x[3]) / 1e3), x[4])
PY
