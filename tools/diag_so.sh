#!/bin/bash
# run tools/kf_phases.py with a diagnostic build of the library (cmusphinx_amd/libD.so) swapped in
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; make -s -C oracle oracle >/dev/null 2>&1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/libkeep.so; cp cmusphinx_amd/libD.so cmusphinx_amd/libcmusphinx_amd.so
timeout 300 python tools/kf_phases.py "$@" 2>&1 | tail -1
cp /tmp/libkeep.so cmusphinx_amd/libcmusphinx_amd.so
