#!/bin/bash
# round 5: the clusters' XCD-local barrier against the general one (S3A_UTT_PERSIST=1 / 2), engines of few lanes
# usage: tools/barrier_ab.sh NAME   (lines of "lanes utts cluster" on stdin)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-bar}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
while read -r lanes utts cl p; do
  [ -z "$lanes" ] && continue
  for p in $p; do
    tag="l${lanes}_u${utts}_c${cl}_p${p}"
    S3A_UTT_PERSIST=$p S3A_UTT_CLUSTER=$cl timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu --plain --lanes $lanes --engines 1 --utts $utts > $OUT/$tag.json 2> $OUT/$tag.err
    python - $OUT/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("search", {})
    ph = s.get("phases_us_per_lane_frame", {})
    print(sys.argv[2], "value", d["value"], "identical", d.get("identical_to_reference", {}).get("hyp"), "wg/lane", s.get("workgroups_per_lane"),
          "phases_sum", round(sum(v for v in ph.values() if isinstance(v, (int, float))), 1) if ph else None)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
