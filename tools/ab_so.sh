#!/bin/bash
# same-box A/B of two builds of the library: cmusphinx_amd/libA.so (baseline) against the tree's libcmusphinx_amd.so (B), the bench's
# plain regime, alternating; usage: tools/ab_so.sh NAME [rounds] [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-ab}; ROUNDS=${2:-2}; shift; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/libB.so
for i in $(seq 1 $ROUNDS); do
  for v in A B; do
    if [ $v = A ]; then cp cmusphinx_amd/libA.so cmusphinx_amd/libcmusphinx_amd.so; else cp /tmp/libB.so cmusphinx_amd/libcmusphinx_amd.so; fi
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --plain "$@" > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/${v}_$i.json').read().strip().splitlines()[-1]); print('$v', $i, d['value'], d['identical_to_reference']['hyp'])"
  done
done
cp /tmp/libB.so cmusphinx_amd/libcmusphinx_amd.so
