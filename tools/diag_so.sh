#!/bin/bash
# tools/kf_phases.py with diagnostic builds of the library swapped in (cmusphinx_amd/libD<k>.so: -DKF_DIAG=k adds a per-frame count to
# the unused clock slot "weak"): usage tools/diag_so.sh "1 2 3" [kf_phases arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; make -s -C oracle oracle >/dev/null 2>&1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/libkeep.so
for k in $1; do
  cp cmusphinx_amd/libD$k.so cmusphinx_amd/libcmusphinx_amd.so
  echo "KF_DIAG=$k: $(timeout 300 python tools/kf_phases.py "${@:2}" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('weak-slot value per frame', d['us_per_frame']['weak'], 'frames/s', d['frames_per_s'])")"
done
cp /tmp/libkeep.so cmusphinx_amd/libcmusphinx_amd.so
