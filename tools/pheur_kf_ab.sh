#!/bin/bash
# -pheurtype through ku_frames against the launch path on the hub4-shaped synthetic task (same box; both compared with the unmodified reference)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
NU=${1:-128}; NF=${2:-600}; L=${3:-128}; shift 3
HE="-pheurtype 1 -pl_window 3 -pl_beam 1e-20 $@"
echo "== launches (S3A_UTT_PERSIST=-1)"; S3A_UTT_PERSIST=-1 bash tools/utt_task.sh hub4 $NU $NF "$L" $HE 2>&1 | grep -v "^INFO" | cut -c1-260
echo "== ku_frames"; SKIP_REF=1 bash tools/utt_task.sh hub4 $NU $NF "$L" $HE 2>&1 | cut -c1-260
