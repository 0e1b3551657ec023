#!/bin/bash
# same-box A/B of several builds / settings of the library in the bench's plain regime, run alternately ROUNDS times.
# usage: tools/ab_multi.sh NAME ROUNDS "label|lib.so (relative to cmusphinx_amd/, '-' = the tree's)|bench arguments" ...
# every case prints: label, round, frames/s, identical-to-reference, ku_frames ms, scoring ms, us per lane-frame
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; ROUNDS=$2; shift; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_tree.so
for i in $(seq 1 $ROUNDS); do
  k=0
  for spec in "$@"; do
    k=$((k+1))
    label=${spec%%|*}; rest=${spec#*|}; lib=${rest%%|*}; bargs=${rest#*|}
    if [ "$lib" = "-" ]; then cp /tmp/lib_tree.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/$lib cmusphinx_amd/libcmusphinx_amd.so; fi
    timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --plain $bargs > $OUT/${label}_$i.json 2> $OUT/${label}_$i.err
    python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/${label}_$i.json').read().strip().splitlines()[-1])
    k = d.get('kernels', {}); s = d.get('search', {})
    print('%-14s %d  %9.1f frames/s  identical %s  ku_frames %.1f ms  scoring %.1f ms  us/lane-frame %s' % ('$label', $i, d['value'], d['identical_to_reference']['hyp'],
          k.get('ku_frames', {}).get('ms_per_step', 0), k.get('ku_score_window', {}).get('ms_per_step', 0), s.get('us_per_lane_frame')))
    ph = s.get('phases_us_per_lane_frame')
    if ph and $i == 1: print('   phases', {a: round(b, 1) for a, b in ph.items()})
except Exception as e:
    print('$label', $i, 'FAILED', e); print(open('$OUT/${label}_$i.err').read()[-600:])
PY
  done
done
cp /tmp/lib_tree.so cmusphinx_amd/libcmusphinx_amd.so
